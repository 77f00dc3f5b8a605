// K4 softmax cross-entropy, K6 squared-L2-norm, a15/a17 step control (grad scaler + clip factor on
// device) and K7 fused ZeRO AdamW for gfx950.  All HBM-bound streaming kernels.
//
// Algorithmic bytes: CE fwd reads vocab*sizeof(logit) per token; CE bwd reads + writes it;
// sumsq reads 2 B/param (bf16 grads); AdamW reads g(2)+p,m,v(12) and writes p,m,v(12)+bf16 p(2)
// = 28 B/param.
#include "ie_common.h"

#include <algorithm>
#include <cstdlib>

#include <math.h>

namespace {

__host__ __device__ inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------ CE fwd
// One 256-thread block per token row: per-thread online (max, sum-exp) over 16-byte chunks, then a
// block combine.  Mirrors flash-attn's xentropy forward / nn.CrossEntropyLoss on fp32 logits: the
// bf16 -> fp32 cast of NaiveAMPModel (internlm/core/naive_amp.py:157-158) is exact, so reading bf16
// logits directly gives the same numbers while halving the traffic.
// ARG = true additionally produces what the reference's metric pass (AccPerplex / LossWithTypeId,
// internlm/model/metrics.py:108-199,281-310) re-reads the logits three more times for: the row argmax (first index
// of the maximum) and the plain (un-smoothed) negative log-likelihood.
template <bool BF, bool ARG>
__global__ __launch_bounds__(256) void ce_fwd_k(const void* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                float* __restrict__ loss_rows, float* __restrict__ lse_out, int64_t vocab,
                                                int64_t ignore_index, float ls, int vec_ok, int32_t* __restrict__ argmax_rows,
                                                float* __restrict__ nll_rows) {
    __shared__ float scratch[8];
    __shared__ int iscratch[4];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f, sx = 0.f;
    float bm = -INFINITY;   // ARG: running maximum with the index of its first occurrence
    int bi = 0x7fffffff;
    auto chunk = [&](const float* v, const int n, const int64_t base) {
        float cm = v[0];
        for (int e = 1; e < n; ++e) cm = fmaxf(cm, v[e]);
        if (cm > m) { s *= __expf(m - cm); m = cm; }
        for (int e = 0; e < n; ++e) { s += __expf(v[e] - m); sx += v[e]; }
        if (ARG && cm > bm) {
            bm = cm;
            for (int e = n - 1; e >= 0; --e) if (v[e] == cm) bi = (int)(base + e);
        }
    };
    if (BF) {
        const bf16_t* x = (const bf16_t*)logits + row * ld;
        const int64_t n8 = vec_ok ? vocab / 8 : 0;
        for (int64_t i = tid; i < n8; i += 256) {
            float v[8];
            unpack8(ld16(x + i * 8), v);
            chunk(v, 8, i * 8);
        }
        for (int64_t i = n8 * 8 + tid; i < vocab; i += 256) {
            const float v = bf2f(x[i]);
            chunk(&v, 1, i);
        }
    } else {
        const float* x = (const float*)logits + row * ld;
        const int64_t n4 = vec_ok ? vocab / 4 : 0;
        for (int64_t i = tid; i < n4; i += 256) {
            const float4 q = *reinterpret_cast<const float4*>(x + i * 4);
            const float v[4] = {q.x, q.y, q.z, q.w};
            chunk(v, 4, i * 4);
        }
        for (int64_t i = n4 * 4 + tid; i < vocab; i += 256) {
            const float v = x[i];
            chunk(&v, 1, i);
        }
    }
    const float mb = block_max<4>(m, scratch);
    const float contrib = (m == -INFINITY) ? 0.f : s * __expf(m - mb);
    const float sb = block_sum<4>(contrib, scratch);
    const float sxb = block_sum<4>(sx, scratch);
    int arg = 0;
    if (ARG) {
        // first index holding the block maximum: min over the threads whose own maximum equals it
        int cand = (bm == mb) ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) iscratch[tid >> 6] = cand;
        __syncthreads();
        arg = min(min(iscratch[0], iscratch[1]), min(iscratch[2], iscratch[3]));
    }
    if (tid == 0) {
        const float lse = mb + logf(sb);
        lse_out[row] = lse;
        const int64_t lab = labels[row];
        float loss = 0.f, nll = 0.f;
        if (lab != ignore_index && lab >= 0 && lab < vocab) {
            const float xl = BF ? bf2f(((const bf16_t*)logits)[row * ld + lab]) : ((const float*)logits)[row * ld + lab];
            nll = lse - xl;
            loss = nll;
            if (ls > 0.f) loss = (1.f - ls) * loss + ls * (lse - sxb / (float)vocab);
        }
        loss_rows[row] = loss;
        if (ARG) { argmax_rows[row] = arg; nll_rows[row] = nll; }
    }
}

// Metric accumulation (AccPerplex.update + LossWithTypeId.update, metrics.py:108-199,281-310) from the per-row outputs of
// ce_fwd_k<.., true>: one block, fixed reduction order (deterministic).  facc = {right, total, total_log_probs, loss,
// token_num}; per type: ds_right / ds_tokens (int64), ds_loss / ds_token_num (fp32).
__global__ __launch_bounds__(1024) void metric_accumulate_k(const float* __restrict__ nll_rows, const int32_t* __restrict__ argmax_rows,
                                                            const int64_t* __restrict__ labels, const int64_t* __restrict__ type_ids,
                                                            int64_t rows, int64_t ignore_index, int ntypes, float* __restrict__ facc,
                                                            int64_t* __restrict__ ds_right, int64_t* __restrict__ ds_tokens,
                                                            float* __restrict__ ds_loss, float* __restrict__ ds_token_num) {
    __shared__ double sh[3][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // t = -1: totals over all rows; t >= 0: rows of dataset type t
    for (int t = -1; t < ntypes; ++t) {
        double right = 0.0, tokens = 0.0, loss = 0.0;
        for (int64_t i = threadIdx.x; i < rows; i += 1024) {
            if (t >= 0 && type_ids[i] != t) continue;
            const int64_t lab = labels[i];
            // corrects = (label == argmax) -- no ignore mask needed, -100 is never an argmax (metrics.py:139-141,163)
            if (lab == (int64_t)argmax_rows[i]) right += 1.0;
            if (lab != ignore_index) { tokens += 1.0; loss += (double)nll_rows[i]; }
        }
        right = wave_sum_d(right); tokens = wave_sum_d(tokens); loss = wave_sum_d(loss);
        __syncthreads();
        if (lane == 0) { sh[0][w] = right; sh[1][w] = tokens; sh[2][w] = loss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = 0.0, k = 0.0, l = 0.0;
            for (int i = 0; i < 16; ++i) { r += sh[0][i]; k += sh[1][i]; l += sh[2][i]; }
            if (t < 0) {
                facc[0] += (float)r; facc[1] += (float)k; facc[2] += (float)l; facc[3] += (float)l; facc[4] += (float)k;
            } else {
                ds_right[t] += (int64_t)r; ds_tokens[t] += (int64_t)k; ds_loss[t] += (float)l; ds_token_num[t] += (float)k;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void ce_mean_k(const float* __restrict__ loss_rows, const int64_t* __restrict__ labels, int64_t rows,
                                                  int64_t ignore_index, float* __restrict__ loss_out, float* __restrict__ count_out) {
    __shared__ double sd[16];
    __shared__ double sc[16];
    double acc = 0.0, cnt = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 1024) {
        if (labels[i] != ignore_index) { acc += (double)loss_rows[i]; cnt += 1.0; }
    }
    acc = wave_sum_d(acc);
    cnt = wave_sum_d(cnt);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sd[w] = acc; sc[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int i = 0; i < 16; ++i) { a += sd[i]; c += sc[i]; }
        loss_out[0] = (float)(a / c);
        count_out[0] = (float)c;
    }
}

// ------------------------------------------------------------------------------------------ CE bwd
template <bool BF>
__global__ __launch_bounds__(256) void ce_bwd_k(const void* __restrict__ logits, void* dlogits, int64_t ld,
                                                const int64_t* __restrict__ labels, const float* __restrict__ lse,
                                                const float* __restrict__ dloss, float dloss_mul, const float* __restrict__ count,
                                                int64_t vocab, int64_t ignore_index, float ls, int vec_ok) {
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t lab = labels[row];
    const bool valid = lab != ignore_index;
    // count == NULL: dloss holds one upstream gradient per row (reduction = 'none' consumers)
    const float g = valid ? (count ? dloss[0] * dloss_mul / count[0] : dloss[row] * dloss_mul) : 0.f;
    const float l = lse[row];
    const float smooth = ls / (float)vocab;
    const float hot = 1.f - ls;
    if (BF) {
        const bf16_t* x = (const bf16_t*)logits + row * ld;
        bf16_t* d = (bf16_t*)dlogits + row * ld;
        if (vec_ok) {
            const int64_t n8 = vocab / 8;
            for (int64_t i = tid; i < n8; i += 256) {
                float v[8];
                unpack8(ld16(x + i * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float p = __expf(v[e] - l) - smooth;
                    if (i * 8 + e == lab) p -= hot;
                    v[e] = valid ? p * g : 0.f;
                }
                st16(d + i * 8, pack8(v));
            }
            for (int64_t i = n8 * 8 + tid; i < vocab; i += 256) {
                float p = __expf(bf2f(x[i]) - l) - smooth;
                if (i == lab) p -= hot;
                d[i] = f2bf(valid ? p * g : 0.f);
            }
        } else {
            for (int64_t i = tid; i < vocab; i += 256) {
                float p = __expf(bf2f(x[i]) - l) - smooth;
                if (i == lab) p -= hot;
                d[i] = f2bf(valid ? p * g : 0.f);
            }
        }
    } else {
        const float* x = (const float*)logits + row * ld;
        float* d = (float*)dlogits + row * ld;
        for (int64_t i = tid; i < vocab; i += 256) {
            float p = __expf(x[i] - l) - smooth;
            if (i == lab) p -= hot;
            d[i] = valid ? p * g : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------ sumsq
constexpr int kSumsqMaxBlocks = 1024;

template <bool BF>
__global__ __launch_bounds__(256) void sumsq_partial_k(const void* __restrict__ x, int64_t n, float* __restrict__ partial) {
    __shared__ float scratch[8];
    float acc = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    if (BF) {
        const bf16_t* p = (const bf16_t*)x;
        const int64_t n8 = aligned16(x) ? n / 8 : 0;
        for (int64_t i = tid; i < n8; i += nthreads) {
            float v[8];
            unpack8(ld16(p + i * 8), v);
            float c = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) c += v[e] * v[e];
            acc += c;
        }
        for (int64_t i = n8 * 8 + tid; i < n; i += nthreads) { const float v = bf2f(p[i]); acc += v * v; }
    } else {
        const float* p = (const float*)x;
        const int64_t n4 = aligned16(x) ? n / 4 : 0;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const float4 q = *reinterpret_cast<const float4*>(p + i * 4);
            acc += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
        for (int64_t i = n4 * 4 + tid; i < n; i += nthreads) acc += p[i] * p[i];
    }
    const float t = block_sum<4>(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sumsq_finish_k(const float* __restrict__ partial, int64_t nparts, float* out, int accumulate) {
    __shared__ double sd[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < nparts; i += 256) acc += (double)partial[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sd[0] + sd[1] + sd[2] + sd[3];
        if (accumulate) t += (double)out[0];
        out[0] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------ step control
__global__ void step_state_init_k(IeStepState* s, float initial_scale) {
    s->loss_scale = initial_scale;
    s->growth_step = 0;
    s->hysteresis_step = 0;
    s->adam_step = 0;
    s->skip = 0;
    s->found_inf = 0;
    s->found_nan = 0;
    s->inv_scale = 1.f;
    s->grad_norm = 0.f;
    s->loss_scale_used = initial_scale;
    s->skipped_total = 0;
    s->_pad = 0;
}

// Mirrors HybridZeroOptimizer._step (hybrid_zero_optim.py:695-779), DynamicGradScaler.update
// (optimizer/utils.py:484-509) and _unscale_and_clip_grads (:863-876), in that order.
__global__ void step_control_k(IeStepState* s, const float* __restrict__ sumsq, IeScalerConfig cfg) {
    const float ss = sumsq[0];
    const bool found_inf = isinf(ss);   // compute_norm sentinel -1 (optimizer/utils.py:371-373)
    const bool found_nan = isnan(ss);   // sentinel -2 (:375-376)
    const float scale_backup = s->loss_scale;  // `loss_scale = float(self.loss_scale.item())  # backup`
    s->loss_scale_used = scale_backup;
    if (cfg.dynamic) {
        if (found_inf) {
            s->hysteresis_step += 1;
            s->growth_step = 0;
            if (s->hysteresis_step >= cfg.hysteresis) {
                float ns = s->loss_scale * cfg.backoff_factor;
                if (cfg.min_scale > 0.f) ns = fmaxf(ns, cfg.min_scale);
                s->loss_scale = ns;
            }
        } else {
            s->growth_step += 1;
            if (s->growth_step == cfg.growth_interval) {
                s->growth_step = 0;
                s->hysteresis_step = 0;
                float ns = s->loss_scale * cfg.growth_factor;
                if (cfg.max_scale > 0.f) ns = fminf(ns, cfg.max_scale);
                s->loss_scale = ns;
            }
        }
    }
    s->found_inf = found_inf ? 1 : 0;
    s->found_nan = found_nan ? 1 : 0;
    if (found_inf || found_nan) {
        s->skip = 1;
        s->skipped_total += 1;
        s->inv_scale = 0.f;
        s->grad_norm = found_inf ? -1.f : -2.f;
        return;
    }
    s->skip = 0;
    s->adam_step += 1;
    const double norm = sqrt((double)ss);  // global_norm_groups[...] = norm ** 0.5
    double combined = (double)scale_backup;
    if (cfg.dynamic && cfg.clip_grad_norm > 0.f) {
        const double clip = (norm / (double)scale_backup + 1e-6) / (double)cfg.clip_grad_norm;
        if (clip > 1.0) combined = clip * (double)scale_backup;
    }
    // fp32 models: the reference neither unscales nor clips (hybrid_zero_optim.py:773), scale is 1.
    s->inv_scale = cfg.dynamic ? (float)(1.0 / combined) : 1.f;
    s->grad_norm = (float)(norm / (double)scale_backup);
}

// Several optimizer parameter groups (the MoE models: default / fp32 gates / experts, train/utils.py:25-80): the overflow check and the
// scaler see all of them, but every group is unscaled and clipped by its OWN norm (hybrid_zero_optim.py:760-779,863-876).
__global__ void step_control_groups_k(IeStepState* s, const float* __restrict__ sumsq, int ng, IeScalerConfig cfg, float* __restrict__ inv_out,
                                      float* __restrict__ norm_out) {
    bool found_inf = false, found_nan = false;
    double total = 0.0;
    for (int g = 0; g < ng; ++g) {
        found_inf |= isinf(sumsq[g]);
        found_nan |= isnan(sumsq[g]);
        total += (double)sumsq[g];
    }
    const float scale_backup = s->loss_scale;
    s->loss_scale_used = scale_backup;
    if (cfg.dynamic) {
        if (found_inf) {
            s->hysteresis_step += 1;
            s->growth_step = 0;
            if (s->hysteresis_step >= cfg.hysteresis) {
                float ns = s->loss_scale * cfg.backoff_factor;
                if (cfg.min_scale > 0.f) ns = fmaxf(ns, cfg.min_scale);
                s->loss_scale = ns;
            }
        } else {
            s->growth_step += 1;
            if (s->growth_step == cfg.growth_interval) {
                s->growth_step = 0;
                s->hysteresis_step = 0;
                float ns = s->loss_scale * cfg.growth_factor;
                if (cfg.max_scale > 0.f) ns = fminf(ns, cfg.max_scale);
                s->loss_scale = ns;
            }
        }
    }
    s->found_inf = found_inf ? 1 : 0;
    s->found_nan = found_nan ? 1 : 0;
    if (found_inf || found_nan) {
        s->skip = 1;
        s->skipped_total += 1;
        s->inv_scale = 0.f;
        s->grad_norm = found_inf ? -1.f : -2.f;
        for (int g = 0; g < ng; ++g) { inv_out[g] = 0.f; norm_out[g] = s->grad_norm; }
        return;
    }
    s->skip = 0;
    s->adam_step += 1;
    for (int g = 0; g < ng; ++g) {
        const double norm = sqrt((double)sumsq[g]);
        double combined = (double)scale_backup;
        if (cfg.dynamic && cfg.clip_grad_norm > 0.f) {
            const double clip = (norm / (double)scale_backup + 1e-6) / (double)cfg.clip_grad_norm;
            if (clip > 1.0) combined = clip * (double)scale_backup;
        }
        inv_out[g] = cfg.dynamic ? (float)(1.0 / combined) : 1.f;
        norm_out[g] = (float)(norm / (double)scale_backup);
    }
    s->inv_scale = inv_out[0];
    s->grad_norm = (float)(sqrt(total) / (double)scale_backup);
}

// ------------------------------------------------------------------------------------------ AdamW
struct AdamConsts {
    float decay;      // 1 - lr*wd
    float one_m_b1;   // 1 - beta1
    float beta2;
    float one_m_b2;
    float step_size;  // lr / (1 - beta1^t)
    float bc2_sqrt;   // sqrt(1 - beta2^t)
    float eps;
    float inv_scale;
};

// Every rounding of the update is written out (no contraction left to the compiler: `v * beta2 + one_m_b2 * g * g` can be fused two ways, and round 6 found two
// kernels inlining this function one ulp apart in v): the whole-chip kernel, the few-CU kernel and the scalar tails give the same bits for the same element.
__device__ __forceinline__ void adam_one(float g, float& p, float& m, float& v, const AdamConsts& c) {
#pragma clang fp contract(off)
    g = g * c.inv_scale;
    p = p * c.decay;
    m = __builtin_fmaf(c.one_m_b1, g - m, m);
    v = __builtin_fmaf(c.one_m_b2 * g, g, v * c.beta2);
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = __builtin_fmaf(-c.step_size, m / denom, p);
}

// (Round 6 tried every access of the stream 16 bytes wide -- eight elements per thread and trip, the bf16 gradient and shadow as 16-byte accesses, two
// independent groups of p / m / v in flight: in the training step 1279-1284 us per bucket against this kernel's 1233-1237, A B A B on one box,
// profiles/r06_step_adamw_16byte_abab.log.  More bytes in flight per thread is not what this stream lacks; the kernel below stays.)
template <bool GBF>
__global__ __launch_bounds__(256) void adamw_k(const void* __restrict__ g, float* __restrict__ p32, float* __restrict__ m,
                                               float* __restrict__ v, bf16_t* __restrict__ p16, int64_t n,
                                               const IeStepState* __restrict__ state, double lr, double beta1, double beta2,
                                               float eps, double wd, int vec_ok, const float* __restrict__ inv_scale_group) {
    __shared__ AdamConsts sc;
    __shared__ int skip;
    if (threadIdx.x == 0) {
        skip = state->skip;
        const int step = state->adam_step;
        const double bc1 = 1.0 - pow(beta1, (double)step);
        const double bc2 = 1.0 - pow(beta2, (double)step);
        sc.decay = (float)(1.0 - lr * wd);
        sc.one_m_b1 = (float)(1.0 - beta1);
        sc.beta2 = (float)beta2;
        sc.one_m_b2 = (float)(1.0 - beta2);
        sc.step_size = (float)(lr / bc1);
        sc.bc2_sqrt = (float)sqrt(bc2);
        sc.eps = eps;
        sc.inv_scale = inv_scale_group ? inv_scale_group[0] : state->inv_scale;   // a parameter group clipped by its own norm
    }
    __syncthreads();
    if (skip) return;
    const AdamConsts c = sc;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    const int64_t n4 = vec_ok ? n / 4 : 0;
    for (int64_t i = tid; i < n4; i += nthreads) {
        float gv[4];
        if (GBF) {
            const uint2 q = ld8((const bf16_t*)g + i * 4);
            gv[0] = bflo(q.x); gv[1] = bfhi(q.x); gv[2] = bflo(q.y); gv[3] = bfhi(q.y);
        } else {
            const float4 q = *reinterpret_cast<const float4*>((const float*)g + i * 4);
            gv[0] = q.x; gv[1] = q.y; gv[2] = q.z; gv[3] = q.w;
        }
        float4 pp = *reinterpret_cast<float4*>(p32 + i * 4);
        float4 mm = *reinterpret_cast<float4*>(m + i * 4);
        float4 vv = *reinterpret_cast<float4*>(v + i * 4);
        adam_one(gv[0], pp.x, mm.x, vv.x, c);
        adam_one(gv[1], pp.y, mm.y, vv.y, c);
        adam_one(gv[2], pp.z, mm.z, vv.z, c);
        adam_one(gv[3], pp.w, mm.w, vv.w, c);
        *reinterpret_cast<float4*>(p32 + i * 4) = pp;
        *reinterpret_cast<float4*>(m + i * 4) = mm;
        *reinterpret_cast<float4*>(v + i * 4) = vv;
        if (p16) {
            uint2 o;
            o.x = pack2bf(pp.x, pp.y);
            o.y = pack2bf(pp.z, pp.w);
            st8(p16 + i * 4, o);
        }
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nthreads) {
        const float gg = GBF ? bf2f(((const bf16_t*)g)[i]) : ((const float*)g)[i];
        float pp = p32[i], mm = m[i], vv = v[i];
        adam_one(gg, pp, mm, vv, c);
        p32[i] = pp; m[i] = mm; v[i] = vv;
        if (p16) p16[i] = f2bf(pp);
    }
}

// AdamW ON A FEW CUs (round 6; ie_tune_adamw_cus).  The update overlaps the next step's first forward layers, and a GEMM workgroup of this library needs a CU to
// itself (one 512-register wave per SIMD, 144 KB of LDS): beside adamw_k's 16 384 grid-stride blocks, which fill every wave slot of every CU for a whole bucket,
// the forward kernels only get CUs in the gaps between two buckets' launches (profiles/r06_step_sequence_adamw_beside_forward.txt: the attention forward 9.2 ms
// instead of 0.6, w2 2.8 ms instead of 1.3 while the update runs).  This kernel is the same arithmetic element by element (adam_one: results bit-identical) as
// `blocks` workgroups, each pinned to a CU of its own by 96 KB of LDS nobody reads (the dispatcher spreads them over the XCDs): the update keeps to its CUs, the
// persistent GEMM frame's queues hand the tiles to the blocks that did get a CU.  A CU streams about 28 GB/s of this mix, so the update needs 96 or more CUs to
// stay ahead of the forward (profiles/r06_step_adamw_cus_sweep.log: 16 CUs 732 ms, 32 666, 64 656, 96 ... 192 650 ... 651, the whole chip 654 ... 657 ms per step);
// the shape measured best is 512 threads x four 16-byte groups of every stream in flight, p / m / v / g non-temporal (they are not read again this step; the
// forward's weight panels keep the L2s) -- 1024 threads x two groups with ordinary accesses: + 1.5 ms, non-temporal: + 0.5 ms.
template <bool GBF, int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void adamw_cus_k(const void* __restrict__ g, float* __restrict__ p32, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ p16, int64_t n, const IeStepState* __restrict__ state, double lr, double beta1,
                                                    double beta2, float eps, double wd, const float* __restrict__ inv_scale_group) {
    __shared__ volatile unsigned char pin[96 * 1024];
    __shared__ AdamConsts sc;
    __shared__ int skip;
    if (n < 0) pin[threadIdx.x] = 0;   // (keeps the allocation)
    if (threadIdx.x == 0) {
        skip = state->skip;
        const int step = state->adam_step;
        const double bc1 = 1.0 - pow(beta1, (double)step);
        const double bc2 = 1.0 - pow(beta2, (double)step);
        sc.decay = (float)(1.0 - lr * wd);
        sc.one_m_b1 = (float)(1.0 - beta1);
        sc.beta2 = (float)beta2;
        sc.one_m_b2 = (float)(1.0 - beta2);
        sc.step_size = (float)(lr / bc1);
        sc.bc2_sqrt = (float)sqrt(bc2);
        sc.eps = eps;
        sc.inv_scale = inv_scale_group ? inv_scale_group[0] : state->inv_scale;
    }
    __syncthreads();
    if (skip) return;
    const AdamConsts c = sc;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int64_t tid = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * THREADS;
    const int64_t n4 = n / 4;
    auto ld4 = [](const float* q) -> f32x4 {
        return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)) : *reinterpret_cast<const f32x4*>(q);
    };
    auto st4 = [](float* q, f32x4 x) {
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(q));
        else *reinterpret_cast<f32x4*>(q) = x;
    };
    for (int64_t i0 = tid; i0 < n4; i0 += U * nthreads) {
        f32x4 gg[U], pp[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // U 16-byte groups of every stream in flight per thread
            const int64_t i = i0 + u * nthreads;
            if (i < n4) {
                if (GBF) {
                    const bf16_t* q = (const bf16_t*)g + i * 4;
                    const u32x2 w = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(q)) : *reinterpret_cast<const u32x2*>(q);
                    gg[u] = f32x4{bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
                } else {
                    gg[u] = ld4((const float*)g + i * 4);
                }
                pp[u] = ld4(p32 + i * 4);
                mm[u] = ld4(m + i * 4);
                vv[u] = ld4(v + i * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * nthreads;
            if (i < n4) {
                float P[4] = {pp[u].x, pp[u].y, pp[u].z, pp[u].w}, M[4] = {mm[u].x, mm[u].y, mm[u].z, mm[u].w}, V[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
                const float G[4] = {gg[u].x, gg[u].y, gg[u].z, gg[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) adam_one(G[e], P[e], M[e], V[e], c);
                st4(p32 + i * 4, f32x4{P[0], P[1], P[2], P[3]});
                st4(m + i * 4, f32x4{M[0], M[1], M[2], M[3]});
                st4(v + i * 4, f32x4{V[0], V[1], V[2], V[3]});
                if (p16) {   // (the next forward reads the shadow: an ordinary store)
                    uint2 o;
                    o.x = pack2bf(P[0], P[1]);
                    o.y = pack2bf(P[2], P[3]);
                    st8(p16 + i * 4, o);
                }
            }
        }
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nthreads) {
        const float gg = GBF ? bf2f(((const bf16_t*)g)[i]) : ((const float*)g)[i];
        float pp = p32[i], mm = m[i], vv = v[i];
        adam_one(gg, pp, mm, vv, c);
        p32[i] = pp; m[i] = mm; v[i] = vv;
        if (p16) p16[i] = f2bf(pp);
    }
}

}  // namespace

static int ce_fwd_launch(const void* logits, int dtype, int64_t ld, const int64_t* labels, float* loss_rows, float* lse, int32_t* argmax_rows,
                         float* nll_rows, int64_t rows, int64_t vocab, int64_t ignore_index, float label_smoothing, void* stream,
                         const char* what) {
    IE_CHECK_ARG(logits && labels && loss_rows && lse, "ie_ce_fwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && vocab > 0 && ld >= vocab, "ie_ce_fwd: bad shape");
    IE_CHECK_ARG(vocab < 0x7fffffff, "ie_ce_fwd: vocab too large");
    IE_CHECK_ARG(dtype == IE_BF16 || dtype == IE_F32, "ie_ce_fwd: bad dtype");
    if (rows == 0) return IE_OK;
    const int esz = dtype == IE_BF16 ? 2 : 4;
    const int vec_ok = aligned16(logits) && ((ld * esz) % 16 == 0);
    const dim3 grid((unsigned)rows), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (argmax_rows) {
        if (dtype == IE_BF16)
            hipLaunchKernelGGL((ce_fwd_k<true, true>), grid, block, 0, st, logits, ld, labels, loss_rows, lse, vocab, ignore_index,
                               label_smoothing, vec_ok, argmax_rows, nll_rows);
        else
            hipLaunchKernelGGL((ce_fwd_k<false, true>), grid, block, 0, st, logits, ld, labels, loss_rows, lse, vocab, ignore_index,
                               label_smoothing, vec_ok, argmax_rows, nll_rows);
    } else {
        if (dtype == IE_BF16)
            hipLaunchKernelGGL((ce_fwd_k<true, false>), grid, block, 0, st, logits, ld, labels, loss_rows, lse, vocab, ignore_index,
                               label_smoothing, vec_ok, nullptr, nullptr);
        else
            hipLaunchKernelGGL((ce_fwd_k<false, false>), grid, block, 0, st, logits, ld, labels, loss_rows, lse, vocab, ignore_index,
                               label_smoothing, vec_ok, nullptr, nullptr);
    }
    return ie_launch_status(what);
}

extern "C" int ie_ce_fwd(const void* logits, int dtype, int64_t ld, const int64_t* labels, float* loss_rows, float* lse, int64_t rows,
                         int64_t vocab, int64_t ignore_index, float label_smoothing, void* stream) {
    return ce_fwd_launch(logits, dtype, ld, labels, loss_rows, lse, nullptr, nullptr, rows, vocab, ignore_index, label_smoothing, stream,
                         "ie_ce_fwd launch");
}

extern "C" int ie_ce_fwd_metric(const void* logits, int dtype, int64_t ld, const int64_t* labels, float* loss_rows, float* lse,
                                int32_t* argmax_rows, float* nll_rows, int64_t rows, int64_t vocab, int64_t ignore_index,
                                float label_smoothing, void* stream) {
    IE_CHECK_ARG(argmax_rows && nll_rows, "ie_ce_fwd_metric: null pointer");
    return ce_fwd_launch(logits, dtype, ld, labels, loss_rows, lse, argmax_rows, nll_rows, rows, vocab, ignore_index, label_smoothing,
                         stream, "ie_ce_fwd_metric launch");
}

extern "C" int ie_metric_accumulate(const float* nll_rows, const int32_t* argmax_rows, const int64_t* labels, const int64_t* type_ids,
                                    int64_t rows, int64_t ignore_index, int ntypes, float* facc, int64_t* ds_right, int64_t* ds_tokens,
                                    float* ds_loss, float* ds_token_num, void* stream) {
    IE_CHECK_ARG(nll_rows && argmax_rows && labels && facc && rows >= 0, "ie_metric_accumulate: bad argument");
    IE_CHECK_ARG(ntypes >= 0 && ntypes <= 4096, "ie_metric_accumulate: bad ntypes");
    IE_CHECK_ARG(ntypes == 0 || (type_ids && ds_right && ds_tokens && ds_loss && ds_token_num), "ie_metric_accumulate: per-type buffers missing");
    hipLaunchKernelGGL(metric_accumulate_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll_rows, argmax_rows, labels, type_ids, rows,
                       ignore_index, ntypes, facc, ds_right, ds_tokens, ds_loss, ds_token_num);
    return ie_launch_status("ie_metric_accumulate launch");
}

extern "C" int ie_ce_mean(const float* loss_rows, const int64_t* labels, int64_t rows, int64_t ignore_index, float* loss_out,
                          float* count_out, void* stream) {
    IE_CHECK_ARG(loss_rows && labels && loss_out && count_out && rows >= 0, "ie_ce_mean: bad argument");
    hipLaunchKernelGGL(ce_mean_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_rows, labels, rows, ignore_index, loss_out, count_out);
    return ie_launch_status("ie_ce_mean launch");
}

extern "C" int ie_ce_bwd(const void* logits, void* dlogits, int dtype, int64_t ld, const int64_t* labels, const float* lse,
                         const float* dloss, float dloss_mul, const float* count, int64_t rows, int64_t vocab, int64_t ignore_index,
                         float label_smoothing, void* stream) {
    IE_CHECK_ARG(logits && dlogits && labels && lse && dloss, "ie_ce_bwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && vocab > 0 && ld >= vocab, "ie_ce_bwd: bad shape");
    IE_CHECK_ARG(dtype == IE_BF16 || dtype == IE_F32, "ie_ce_bwd: bad dtype");
    if (rows == 0) return IE_OK;
    const int esz = dtype == IE_BF16 ? 2 : 4;
    const int vec_ok = aligned16(logits) && aligned16(dlogits) && ((ld * esz) % 16 == 0);
    if (dtype == IE_BF16)
        hipLaunchKernelGGL((ce_bwd_k<true>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, dlogits, ld, labels, lse,
                           dloss, dloss_mul, count, vocab, ignore_index, label_smoothing, vec_ok);
    else
        hipLaunchKernelGGL((ce_bwd_k<false>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, dlogits, ld, labels, lse,
                           dloss, dloss_mul, count, vocab, ignore_index, label_smoothing, vec_ok);
    return ie_launch_status("ie_ce_bwd launch");
}

extern "C" int64_t ie_sumsq_max_partials(void) { return kSumsqMaxBlocks; }

extern "C" int ie_sumsq_partial(const void* x, int dtype, int64_t n, float* partial, int64_t part_offset, int64_t* nparts_out,
                                void* stream) {
    IE_CHECK_ARG(x && partial && n >= 0 && part_offset >= 0, "ie_sumsq_partial: bad argument");
    IE_CHECK_ARG(dtype == IE_BF16 || dtype == IE_F32, "ie_sumsq_partial: bad dtype");
    int64_t blocks = (n + 256 * 32 - 1) / (256 * 32);
    if (blocks > kSumsqMaxBlocks) blocks = kSumsqMaxBlocks;
    if (blocks < 1) blocks = 1;
    if (nparts_out) *nparts_out = blocks;
    if (dtype == IE_BF16)
        hipLaunchKernelGGL((sumsq_partial_k<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, partial + part_offset);
    else
        hipLaunchKernelGGL((sumsq_partial_k<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, partial + part_offset);
    return ie_launch_status("ie_sumsq_partial launch");
}

extern "C" int ie_sumsq_finish(const float* partial, int64_t nparts, float* out, int accumulate, void* stream) {
    IE_CHECK_ARG(partial && out && nparts >= 0, "ie_sumsq_finish: bad argument");
    hipLaunchKernelGGL(sumsq_finish_k, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nparts, out, accumulate);
    return ie_launch_status("ie_sumsq_finish launch");
}

extern "C" int ie_step_state_init(IeStepState* state_dev, float initial_scale, void* stream) {
    IE_CHECK_ARG(state_dev, "ie_step_state_init: null pointer");
    hipLaunchKernelGGL(step_state_init_k, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, initial_scale);
    return ie_launch_status("ie_step_state_init launch");
}

extern "C" int ie_step_control(IeStepState* state_dev, const float* sumsq_dev, const IeScalerConfig* cfg_host, void* stream) {
    IE_CHECK_ARG(state_dev && sumsq_dev && cfg_host, "ie_step_control: null pointer");
    IE_CHECK_ARG(cfg_host->growth_factor > 1.f && cfg_host->backoff_factor > 0.f && cfg_host->backoff_factor < 1.f &&
                     cfg_host->hysteresis >= 0,
                 "ie_step_control: bad scaler config");
    hipLaunchKernelGGL(step_control_k, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, sumsq_dev, *cfg_host);
    return ie_launch_status("ie_step_control launch");
}

extern "C" int ie_step_control_groups(IeStepState* state_dev, const float* sumsq_dev, int ngroups, const IeScalerConfig* cfg_host,
                                      float* group_inv_scale_dev, float* group_norm_dev, void* stream) {
    IE_CHECK_ARG(state_dev && sumsq_dev && cfg_host && group_inv_scale_dev && group_norm_dev && ngroups >= 1 && ngroups <= 16,
                 "ie_step_control_groups: bad argument");
    IE_CHECK_ARG(cfg_host->growth_factor > 1.f && cfg_host->backoff_factor > 0.f && cfg_host->backoff_factor < 1.f && cfg_host->hysteresis >= 0,
                 "ie_step_control_groups: bad scaler config");
    hipLaunchKernelGGL(step_control_groups_k, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, sumsq_dev, ngroups, *cfg_host, group_inv_scale_dev,
                       group_norm_dev);
    return ie_launch_status("ie_step_control_groups launch");
}

static int adamw_launch(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n, const IeStepState* state_dev,
                        const float* inv_scale_group, double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

static int g_adamw_cus = 0;   // ie_tune_adamw_cus: 0 = adamw_k over the whole chip, n = adamw_cus_k on n CUs
extern "C" int ie_tune_adamw_cus(int cus) {
    if (cus < 0 || cus > 256) return IE_ERR_INVALID;
    g_adamw_cus = cus;
    return IE_OK;
}

extern "C" int ie_adamw_step_group(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n, const IeStepState* state_dev,
                                   const float* inv_scale_group_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                                   void* stream) {
    IE_CHECK_ARG(inv_scale_group_dev, "ie_adamw_step_group: null group scale");
    return adamw_launch(g, g_dtype, p32, m, v, p16, n, state_dev, inv_scale_group_dev, lr, beta1, beta2, eps, weight_decay, stream);
}

extern "C" int ie_adamw_step(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n,
                             const IeStepState* state_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                             void* stream) {
    return adamw_launch(g, g_dtype, p32, m, v, p16, n, state_dev, nullptr, lr, beta1, beta2, eps, weight_decay, stream);
}

static int adamw_launch(const void* g, int g_dtype, float* p32, float* m, float* v, void* p16, int64_t n, const IeStepState* state_dev,
                        const float* inv_scale_group, double lr, double beta1, double beta2, double eps, double weight_decay, void* stream) {
    IE_CHECK_ARG(g && p32 && m && v && state_dev && n >= 0, "ie_adamw_step: bad argument");
    IE_CHECK_ARG(g_dtype == IE_BF16 || g_dtype == IE_F32, "ie_adamw_step: bad dtype");
    if (n == 0) return IE_OK;
    const int vec_ok = aligned16(p32) && aligned16(m) && aligned16(v) && ((((uintptr_t)g) & (g_dtype == IE_BF16 ? 7u : 15u)) == 0) &&
                       (!p16 || (((uintptr_t)p16) & 7u) == 0);
    const int cus = g_adamw_cus;
    if (cus > 0 && vec_ok && n >= (int64_t)cus * 8192) {   // (a short vector is not worth pinning CUs for)
        if (g_dtype == IE_BF16)
            hipLaunchKernelGGL((adamw_cus_k<true, 4, true, 512>), dim3((unsigned)cus), dim3(512), 0, (hipStream_t)stream, g, p32, m, v, (bf16_t*)p16, n, state_dev, lr,
                               beta1, beta2, (float)eps, weight_decay, inv_scale_group);
        else
            hipLaunchKernelGGL((adamw_cus_k<false, 4, true, 512>), dim3((unsigned)cus), dim3(512), 0, (hipStream_t)stream, g, p32, m, v, (bf16_t*)p16, n, state_dev, lr,
                               beta1, beta2, (float)eps, weight_decay, inv_scale_group);
        return ie_launch_status("ie_adamw_step launch (few CUs)");
    }
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    if (g_dtype == IE_BF16)
        hipLaunchKernelGGL((adamw_k<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, p32, m, v, (bf16_t*)p16, n,
                           state_dev, lr, beta1, beta2, (float)eps, weight_decay, vec_ok, inv_scale_group);
    else
        hipLaunchKernelGGL((adamw_k<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, p32, m, v, (bf16_t*)p16, n,
                           state_dev, lr, beta1, beta2, (float)eps, weight_decay, vec_ok, inv_scale_group);
    return ie_launch_status("ie_adamw_step launch");
}
