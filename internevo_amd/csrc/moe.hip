// K6  GShard mixture-of-experts layer (top-2 gating) in INDEX form, for gfx950.
//
// The reference (internlm/model/moe/gshard_layer.py) routes with one-hot masks and dense einsums: combine_weights [S, E, C],
// `sec,sm->ecm` to dispatch and `sec,ecm->sm` to combine -- O(S*E*C*M) multiply-adds that are all zero except two per token.
// Here a token's two choices are two ROW INDICES into the expert buffers; every step is an HBM-bound gather / scatter / reduction:
//   moe_gate_k        x[S,M] (bf16, read once) . wg[E,M]^T (fp32) -> logits, softmax gates, first choice = argmax(gates), second
//                     choice = argmax(logits + noise) with the first masked out            gshard_layer.py:217-238
//   moe_route_k       ONE workgroup: slot of a token in its expert's capacity buffer = its rank in token order among the tokens
//                     that chose the expert (second choices queue behind all first choices), capacity drop, renormalised weights,
//                     auxiliary loss, per-expert counts, the inverse map row -> token     :240-285 (the cumsum / one-hot part)
//   moe_dispatch_k    expert_in[row] = x[token_of[row]] (zeros for empty slots)            :446-448
//   moe_combine_fwd_k out[s] = bf16( w1 * expert_out[row1] + w2 * expert_out[row2] )       :482-486
//   moe_combine_bwd_k per row: d expert_out = w * d out[token]; d weight = <d out, expert_out[row]>
//   moe_dispatch_bwd_k d x[s] = sum of the d expert_in rows of the token's dispatched choices
//   moe_gate_bwd_k    d weights (+ the auxiliary loss) -> d gates -> softmax backward -> d logits; d x += d logits . wg
//   moe_dwg_*         d wg = d logits^T . x (split over token ranges, summed in a fixed order: deterministic)
// The expert FFNs themselves are the library's bf16 GEMMs + SwiGLU kernels on the [E*C, M] buffers (one GEMM per expert).
// dtypes follow the reference's NaiveAMP arrangement: the gate runs in fp32 on the fp32-cast input, its outputs (weights, l_aux)
// reach the rest of the layer rounded to bf16 (naive_amp.py:160-206) -- restated and pinned in oracle/moe.py.
//
// Roofline: all kernels HBM-bound; algorithmic bytes per token: gate 2M (+4E..), dispatch / combine 2 * 2 * 2M each.
#include "ie_common.h"

#include <algorithm>

namespace {

constexpr int kMaxE = 16;

__device__ __forceinline__ float hash_uniform(uint32_t seed, uint64_t idx) {
    // counter-based: murmur3 finaliser over (seed, index) twice -> 24 random bits -> open interval (0, 1)
    uint32_t a = (uint32_t)idx * 0x9E3779B1u + seed, b = (uint32_t)(idx >> 32) ^ (seed * 0x85EBCA6Bu);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        a ^= b + 0x7F4A7C15u;
        a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
        b += a;
        b ^= b >> 15; b *= 0x2C1B3C6Du; b ^= b >> 12; b *= 0x297A2D39u; b ^= b >> 15;
    }
    return ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void moe_gumbel_k(float* __restrict__ out, int64_t n, uint32_t seed, uint64_t offset) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < n) out[i] = -logf(-logf(hash_uniform(seed, offset + (uint64_t)i)));  // Gumbel(0, 1) by inversion (gshard_layer.py:63-70)
}

// one wave per token
__global__ __launch_bounds__(256) void moe_gate_k(const bf16_t* __restrict__ x, int64_t x_ld, const float* __restrict__ wg, const float* __restrict__ noise,
                                                  int64_t S, int M, int E, float* __restrict__ logits, float* __restrict__ gates,
                                                  int32_t* __restrict__ expert) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= S) return;
    float acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) acc[e] = 0.f;
    const bf16_t* xr = x + s * x_ld;
    for (int m0 = lane * 8; m0 < M; m0 += 512) {
        float xv[8];
        unpack8(ld16(xr + m0), xv);
#pragma unroll
        for (int e = 0; e < kMaxE; ++e)
            if (e < E) {
                const float4 a = *reinterpret_cast<const float4*>(wg + (int64_t)e * M + m0), b = *reinterpret_cast<const float4*>(wg + (int64_t)e * M + m0 + 4);
                acc[e] += xv[0] * a.x + xv[1] * a.y + xv[2] * a.z + xv[3] * a.w + xv[4] * b.x + xv[5] * b.y + xv[6] * b.z + xv[7] * b.w;
            }
    }
#pragma unroll
    for (int e = 0; e < kMaxE; ++e)
        if (e < E) acc[e] = wave_sum(acc[e]);
    if (lane == 0) {
        float mx = -INFINITY;
        for (int e = 0; e < E; ++e) mx = fmaxf(mx, acc[e]);
        float den = 0.f, g[kMaxE];
        for (int e = 0; e < E; ++e) { g[e] = expf(acc[e] - mx); den += g[e]; }
        int e1 = 0;
        float best = -INFINITY;
        for (int e = 0; e < E; ++e) {
            g[e] = g[e] / den;
            logits[s * E + e] = acc[e];
            gates[s * E + e] = g[e];
            if (g[e] > best) { best = g[e]; e1 = e; }   // first maximum, as torch.argmax
        }
        int e2 = 0;
        best = -INFINITY;
        bool any = false;
        for (int e = 0; e < E; ++e) {
            const float v = (e == e1) ? -3.4028234663852886e38f : acc[e] + (noise ? noise[s * E + e] : 0.f);  // finfo(float32).min
            if (!any || v > best) { best = v; e2 = e; any = true; }
        }
        expert[s] = e1;
        expert[S + s] = e2;
    }
}

// ONE workgroup of 1024 threads; thread t owns tokens [t * ch, (t + 1) * ch)
__global__ __launch_bounds__(1024) void moe_route_k(const float* __restrict__ gates, const int32_t* __restrict__ expert, int S, int E, int C,
                                                    int32_t* __restrict__ row, float* __restrict__ weight, int32_t* __restrict__ token_of,
                                                    float* __restrict__ l_aux, int32_t* __restrict__ exp_counts) {
    __shared__ int cnt[2][kMaxE][16];       // per-wave totals
    __shared__ int total[2][kMaxE];
    __shared__ float gsum[kMaxE][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int ch = (S + 1023) / 1024;
    const int s0 = min(S, t * ch), s1 = min(S, s0 + ch);
    for (int r = t; r < E * C; r += 1024) token_of[r] = -1;
    // ---- counts of this thread's chunk, per expert, for first and second choices; column sums of the gates
    int base[2][kMaxE];
    for (int e = 0; e < E; ++e) {
        int c1 = 0, c2 = 0;
        float gs = 0.f;
        for (int s = s0; s < s1; ++s) {
            c1 += expert[s] == e;
            c2 += expert[S + s] == e;
            gs += gates[(int64_t)s * E + e];
        }
        // exclusive scan over the 1024 threads: inside the wave by shuffles, across waves through LDS
        int i1 = c1, i2 = c2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(i1, o, 64), b = __shfl_up(i2, o, 64);
            if (lane >= o) { i1 += a; i2 += b; }
        }
        gs = wave_sum(gs);
        if (lane == 63) { cnt[0][e][wv] = i1; cnt[1][e][wv] = i2; }
        if (lane == 0) gsum[e][wv] = gs;
        base[0][e] = i1 - c1;
        base[1][e] = i2 - c2;
    }
    __syncthreads();
    if (t < 2 * E) {
        const int k = t / E, e = t % E;
        int run = 0;
        for (int w = 0; w < 16; ++w) { const int v = cnt[k][e][w]; cnt[k][e][w] = run; run += v; }
        total[k][e] = run;
    }
    __syncthreads();
    // ---- slots, capacity drop, renormalised weights, inverse map
    int seen[2][kMaxE];
    for (int e = 0; e < E; ++e) {
        seen[0][e] = base[0][e] + cnt[0][e][wv];
        seen[1][e] = base[1][e] + cnt[1][e][wv] + total[0][e];   // second choices queue behind ALL first choices of the expert
    }
    for (int s = s0; s < s1; ++s) {
        const int e1 = expert[s], e2 = expert[S + s];
        int p1 = 0, p2 = 0;
        for (int e = 0; e < E; ++e) {   // (register arrays indexed by a runtime value would go to scratch: select instead)
            if (e == e1) p1 = seen[0][e]++;
            if (e == e2) p2 = seen[1][e]++;
        }
        const bool k1 = p1 < C, k2 = p2 < C;
        const float g1 = k1 ? gates[(int64_t)s * E + e1] : 0.f, g2 = k2 ? gates[(int64_t)s * E + e2] : 0.f;
        const float den = fmaxf(g1 + g2, 1.1920928955078125e-07f);   // clamp(min = finfo(float32).eps)
        const float w1 = g1 / den, w2 = g2 / den;
        weight[s] = w1;
        weight[S + s] = w2;
        const int r1 = k1 ? e1 * C + p1 : -1, r2 = k2 ? e2 * C + p2 : -1;
        row[s] = r1;
        row[S + s] = r2;
        if (k1 && w1 != 0.f) token_of[r1] = 2 * s;        // dispatch mask = combine_weights != 0 (gshard_layer.py:283)
        if (k2 && w2 != 0.f) token_of[r2] = 2 * s + 1;
    }
    if (t == 0) {
        float acc = 0.f;
        for (int e = 0; e < E; ++e) {
            float g = 0.f;
            for (int w = 0; w < 16; ++w) g += gsum[e][w];
            acc += (g / (float)S) * ((float)total[0][e] / (float)S);
            exp_counts[e] = total[0][e];
        }
        l_aux[0] = rbf(acc / (float)E * (float)E * (float)E);   // mean_e(me * ce) * E * E, handed on in the model dtype (bf16)
    }
}

// one wave per expert-buffer row
__global__ __launch_bounds__(256) void moe_dispatch_k(const bf16_t* __restrict__ x, int64_t x_ld, const int32_t* __restrict__ token_of, int64_t rows, int M,
                                                      bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int tk = token_of[r];
    bf16_t* o = out + r * M;
    if (tk < 0) {
        for (int m0 = lane * 8; m0 < M; m0 += 512) st16(o + m0, make_uint4(0, 0, 0, 0));
    } else {
        const bf16_t* xr = x + (int64_t)(tk >> 1) * x_ld;
        for (int m0 = lane * 8; m0 < M; m0 += 512) st16(o + m0, ld16(xr + m0));
    }
}

// one wave per token
__global__ __launch_bounds__(256) void moe_combine_fwd_k(const bf16_t* __restrict__ eo, const int32_t* __restrict__ row, const float* __restrict__ weight,
                                                         int64_t S, int M, bf16_t* __restrict__ out, int64_t out_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= S) return;
    const int r1 = row[s], r2 = row[S + s];
    const float w1 = r1 >= 0 ? rbf(weight[s]) : 0.f, w2 = r2 >= 0 ? rbf(weight[S + s]) : 0.f;   // the einsum sees bf16 weights
    for (int m0 = lane * 8; m0 < M; m0 += 512) {
        float a[8], b[8], o[8];
        if (r1 >= 0) unpack8(ld16(eo + (int64_t)r1 * M + m0), a);
        if (r2 >= 0) unpack8(ld16(eo + (int64_t)r2 * M + m0), b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (r1 >= 0 ? w1 * a[i] : 0.f) + (r2 >= 0 ? w2 * b[i] : 0.f);
        st16(out + s * out_ld + m0, pack8(o));
    }
}

// one wave per expert-buffer row
__global__ __launch_bounds__(256) void moe_combine_bwd_k(const bf16_t* __restrict__ dout, int64_t d_ld, const bf16_t* __restrict__ eo,
                                                         const int32_t* __restrict__ token_of, const float* __restrict__ weight, int64_t rows, int64_t S,
                                                         int M, bf16_t* __restrict__ d_eo, float* __restrict__ d_weight) {
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int tk = token_of[r];
    bf16_t* o = d_eo + r * M;
    if (tk < 0) {
        for (int m0 = lane * 8; m0 < M; m0 += 512) st16(o + m0, make_uint4(0, 0, 0, 0));
        return;
    }
    const int64_t s = tk >> 1;
    const int k = tk & 1;
    const float w = rbf(weight[k * S + s]);
    float dot = 0.f;
    for (int m0 = lane * 8; m0 < M; m0 += 512) {
        float g[8], y[8], v[8];
        unpack8(ld16(dout + s * d_ld + m0), g);
        unpack8(ld16(eo + r * M + m0), y);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = w * g[i]; dot += g[i] * y[i]; }
        st16(o + m0, pack8(v));
    }
    dot = wave_sum(dot);
    if (lane == 0) d_weight[k * S + s] = rbf(dot);   // the gradient of a bf16 tensor is a bf16 tensor
}

// one wave per token
__global__ __launch_bounds__(256) void moe_dispatch_bwd_k(const bf16_t* __restrict__ d_ein, const int32_t* __restrict__ row, const int32_t* __restrict__ token_of,
                                                          int64_t S, int M, bf16_t* __restrict__ dx, int64_t dx_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= S) return;
    int r1 = row[s], r2 = row[S + s];
    if (r1 >= 0 && token_of[r1] != 2 * s) r1 = -1;        // kept but not dispatched (renormalised weight exactly 0)
    if (r2 >= 0 && token_of[r2] != 2 * s + 1) r2 = -1;
    for (int m0 = lane * 8; m0 < M; m0 += 512) {
        float a[8], b[8], o[8];
        if (r1 >= 0) unpack8(ld16(d_ein + (int64_t)r1 * M + m0), a);
        if (r2 >= 0) unpack8(ld16(d_ein + (int64_t)r2 * M + m0), b);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (r1 >= 0 ? a[i] : 0.f) + (r2 >= 0 ? b[i] : 0.f);
        st16(dx + s * dx_ld + m0, pack8(o));
    }
}

// one wave per token: d logits, and dx += bf16(d logits . wg)
__global__ __launch_bounds__(256) void moe_gate_bwd_k(const float* __restrict__ wg, const float* __restrict__ gates, const int32_t* __restrict__ expert,
                                                      const int32_t* __restrict__ row, const float* __restrict__ d_weight,
                                                      const int32_t* __restrict__ exp_counts, const float* __restrict__ scale_dev, float aux_factor,
                                                      int64_t S, int M, int E, float* __restrict__ d_logits, bf16_t* __restrict__ dx, int64_t dx_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t s = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (s >= S) return;
    const int e1 = expert[s], e2 = expert[S + s];
    const bool k1 = row[s] >= 0, k2 = row[S + s] >= 0;
    const float g1 = k1 ? gates[s * E + e1] : 0.f, g2 = k2 ? gates[s * E + e2] : 0.f;
    const float dw1 = k1 ? d_weight[s] : 0.f, dw2 = k2 ? d_weight[S + s] : 0.f;
    const float sum = g1 + g2, eps = 1.1920928955078125e-07f;
    const float den = fmaxf(sum, eps);
    const float dden = sum > eps ? -(dw1 * g1 + dw2 * g2) / (den * den) : 0.f;   // clamp passes no gradient below its bound
    const float dg1 = dw1 / den + dden, dg2 = dw2 / den + dden;
    const float d_laux = aux_factor * (scale_dev ? scale_dev[0] : 1.f);            // d(loss) / d(l_aux)
    float dg[kMaxE], gt[kMaxE], inner = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxE; ++e)
        if (e < E) {
            gt[e] = gates[s * E + e];
            // l_aux = E * sum_e mean_s(gates[:, e]) * ce_e with ce_e = (first choices of e) / S (a constant of the backward)
            dg[e] = d_laux * (float)E * ((float)exp_counts[e] / (float)S) / (float)S + ((k1 && e == e1) ? dg1 : 0.f) + ((k2 && e == e2) ? dg2 : 0.f);
            inner += dg[e] * gt[e];
        }
    float dl[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e)
        if (e < E) {
            dl[e] = gt[e] * (dg[e] - inner);
            if (lane == 0) d_logits[s * E + e] = dl[e];
        }
    for (int m0 = lane * 8; m0 < M; m0 += 512) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, old[8];
#pragma unroll
        for (int e = 0; e < kMaxE; ++e)
            if (e < E) {
                const float4 a = *reinterpret_cast<const float4*>(wg + (int64_t)e * M + m0), b = *reinterpret_cast<const float4*>(wg + (int64_t)e * M + m0 + 4);
                acc[0] += dl[e] * a.x; acc[1] += dl[e] * a.y; acc[2] += dl[e] * a.z; acc[3] += dl[e] * a.w;
                acc[4] += dl[e] * b.x; acc[5] += dl[e] * b.y; acc[6] += dl[e] * b.z; acc[7] += dl[e] * b.w;
            }
        unpack8(ld16(dx + s * dx_ld + m0), old);
#pragma unroll
        for (int i = 0; i < 8; ++i) old[i] += rbf(acc[i]);   // the gate's input gradient arrives in the model dtype and is added to the dispatch path's
        st16(dx + s * dx_ld + m0, pack8(old));
    }
}

// d wg partials: block (m chunk of 256 columns, token range); thread = one column, E accumulators
__global__ __launch_bounds__(256) void moe_dwg_partial_k(const bf16_t* __restrict__ x, int64_t x_ld, const float* __restrict__ d_logits, int64_t S, int M, int E,
                                                         int nsplit, float* __restrict__ part) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int sp = blockIdx.y;
    const int64_t per = (S + nsplit - 1) / nsplit, s0 = sp * per, s1 = min(S, s0 + per);
    __shared__ float dl[64][kMaxE];
    float acc[kMaxE];
#pragma unroll
    for (int e = 0; e < kMaxE; ++e) acc[e] = 0.f;
    for (int64_t sb = s0; sb < s1; sb += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * E; i += 256) {
            const int64_t s = sb + i / E;
            dl[i / E][i % E] = s < s1 ? d_logits[s * E + i % E] : 0.f;
        }
        __syncthreads();
        if (m < M) {
            const int n = (int)min((int64_t)64, s1 - sb);
            for (int j = 0; j < n; ++j) {
                const float xv = bf2f(x[(sb + j) * x_ld + m]);
#pragma unroll
                for (int e = 0; e < kMaxE; ++e)
                    if (e < E) acc[e] += dl[j][e] * xv;
            }
        }
    }
    if (m < M)
        for (int e = 0; e < E; ++e) part[((int64_t)sp * E + e) * M + m] = acc[e];
}

__global__ __launch_bounds__(256) void moe_dwg_reduce_k(const float* __restrict__ part, int nsplit, int64_t n, float* __restrict__ d_wg, int accumulate) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n) return;
    float a = accumulate ? d_wg[i] : 0.f;
    for (int sp = 0; sp < nsplit; ++sp) a += part[(int64_t)sp * n + i];
    d_wg[i] = a;
}

constexpr int kDwgSplit = 64;

}  // namespace

extern "C" int64_t ie_moe_dwg_workspace(int M, int E) { return (M <= 0 || E <= 0) ? -1 : (int64_t)kDwgSplit * E * M; }

extern "C" int ie_moe_gumbel_noise(float* out, int64_t n, uint32_t seed, uint64_t offset, void* stream) {
    IE_CHECK_ARG(out && n >= 0, "ie_moe_gumbel_noise: bad arguments");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(moe_gumbel_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
    return ie_launch_status("ie_moe_gumbel_noise launch");
}

extern "C" int ie_moe_gate_fwd(const void* x, int64_t x_ld, const float* wg, const float* noise, int64_t S, int M, int E, float* logits, float* gates,
                               int32_t* expert, void* stream) {
    IE_CHECK_ARG(x && wg && logits && gates && expert && S >= 0, "ie_moe_gate_fwd: null pointer");
    IE_CHECK_SUPPORTED(E >= 2 && E <= kMaxE && M % 8 == 0 && x_ld % 8 == 0, "ie_moe_gate_fwd: 2 <= experts <= 16, hidden size a multiple of 8");
    if (S == 0) return IE_OK;
    hipLaunchKernelGGL(moe_gate_k, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_ld, wg, noise, S, M, E, logits, gates,
                       expert);
    return ie_launch_status("ie_moe_gate_fwd launch");
}

extern "C" int ie_moe_route(const float* gates, const int32_t* expert, int64_t S, int E, int capacity, int32_t* row, float* weight, int32_t* token_of,
                            float* l_aux, int32_t* exp_counts, void* stream) {
    IE_CHECK_ARG(gates && expert && row && weight && token_of && l_aux && exp_counts && S > 0 && capacity > 0, "ie_moe_route: bad arguments");
    IE_CHECK_SUPPORTED(E >= 2 && E <= kMaxE && S < (1 << 30), "ie_moe_route: 2 <= experts <= 16");
    hipLaunchKernelGGL(moe_route_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, gates, expert, (int)S, E, capacity, row, weight, token_of, l_aux, exp_counts);
    return ie_launch_status("ie_moe_route launch");
}

// Expert-buffer rows in CHUNK-major order (round 6: the expert exchange in pieces, every piece's all_to_all under the expert products of the piece before it --
// gshard_layer.py:465-498 is blocking).  Slot c of expert e, logical row e C + c, lives at row k E Cn + e Cn + (c mod Cn) with Cn = C / nchunk, k = c / Cn: piece k
// = the rows [k E Cn, (k + 1) E Cn) = [expert][Cn slots] -- contiguous, so it is one all_to_all_single of its own.  row[2 S] is rewritten in place, token_of
// copied into the new order; every later kernel of the layer only ever uses the two arrays as opaque indices.
__global__ __launch_bounds__(256) void moe_chunk_rows_k(int32_t* __restrict__ row, const int32_t* __restrict__ token_in, int32_t* __restrict__ token_out, int64_t S, int E,
                                                        int C, int nchunk) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    const int Cn = C / nchunk;
    auto phys = [&](int r) { const int e = r / C, c = r - e * C, k = c / Cn; return k * E * Cn + e * Cn + (c - k * Cn); };
    if (i < (int64_t)E * C) token_out[phys((int)i)] = token_in[i];
    if (i < 2 * S) {
        const int r = row[i];
        if (r >= 0) row[i] = phys(r);
    }
}

extern "C" int ie_moe_chunk_rows(int32_t* row, const int32_t* token_of_in, int32_t* token_of_out, int64_t S, int E, int capacity, int nchunk, void* stream) {
    IE_CHECK_ARG(row && token_of_in && token_of_out && token_of_in != token_of_out && S > 0 && E >= 1 && capacity > 0, "ie_moe_chunk_rows: bad arguments");
    IE_CHECK_ARG(nchunk >= 1 && capacity % nchunk == 0 && (int64_t)E * capacity < (1ll << 31), "ie_moe_chunk_rows: the chunk count must divide the capacity");
    const int64_t n = std::max<int64_t>((int64_t)E * capacity, 2 * S);
    hipLaunchKernelGGL(moe_chunk_rows_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row, token_of_in, token_of_out, S, E, capacity, nchunk);
    return ie_launch_status("ie_moe_chunk_rows launch");
}

extern "C" int ie_moe_dispatch(const void* x, int64_t x_ld, const int32_t* token_of, int64_t rows, int M, void* expert_in, void* stream) {
    IE_CHECK_ARG(x && token_of && expert_in && rows >= 0 && M % 8 == 0 && x_ld % 8 == 0, "ie_moe_dispatch: bad arguments");
    if (rows == 0) return IE_OK;
    hipLaunchKernelGGL(moe_dispatch_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_ld, token_of, rows, M,
                       (bf16_t*)expert_in);
    return ie_launch_status("ie_moe_dispatch launch");
}

extern "C" int ie_moe_combine_fwd(const void* expert_out, const int32_t* row, const float* weight, int64_t S, int M, void* out, int64_t out_ld, void* stream) {
    IE_CHECK_ARG(expert_out && row && weight && out && S >= 0 && M % 8 == 0 && out_ld % 8 == 0, "ie_moe_combine_fwd: bad arguments");
    if (S == 0) return IE_OK;
    hipLaunchKernelGGL(moe_combine_fwd_k, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)expert_out, row, weight, S, M,
                       (bf16_t*)out, out_ld);
    return ie_launch_status("ie_moe_combine_fwd launch");
}

extern "C" int ie_moe_combine_bwd(const void* dout, int64_t d_ld, const void* expert_out, const int32_t* token_of, const float* weight, int64_t rows, int64_t S,
                                  int M, void* d_expert_out, float* d_weight, void* stream) {
    IE_CHECK_ARG(dout && expert_out && token_of && weight && d_expert_out && d_weight && M % 8 == 0 && d_ld % 8 == 0, "ie_moe_combine_bwd: bad arguments");
    if (rows == 0 || S == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_weight, 0, 2 * S * sizeof(float), st) != hipSuccess) return ie_launch_status("ie_moe_combine_bwd memset");
    hipLaunchKernelGGL(moe_combine_bwd_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)dout, d_ld, (const bf16_t*)expert_out, token_of,
                       weight, rows, S, M, (bf16_t*)d_expert_out, d_weight);
    return ie_launch_status("ie_moe_combine_bwd launch");
}

extern "C" int ie_moe_dispatch_bwd(const void* d_expert_in, const int32_t* row, const int32_t* token_of, int64_t S, int M, void* dx, int64_t dx_ld, void* stream) {
    IE_CHECK_ARG(d_expert_in && row && token_of && dx && M % 8 == 0 && dx_ld % 8 == 0, "ie_moe_dispatch_bwd: bad arguments");
    if (S == 0) return IE_OK;
    hipLaunchKernelGGL(moe_dispatch_bwd_k, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_expert_in, row, token_of, S, M,
                       (bf16_t*)dx, dx_ld);
    return ie_launch_status("ie_moe_dispatch_bwd launch");
}

extern "C" int ie_moe_gate_bwd(const void* x, int64_t x_ld, const float* wg, const float* gates, const int32_t* expert, const int32_t* row,
                               const float* d_weight, const int32_t* exp_counts, const float* loss_scale_dev, float aux_factor, int64_t S, int M, int E,
                               float* d_logits, void* dx, int64_t dx_ld, float* d_wg, int accumulate_d_wg, float* workspace, void* stream) {
    IE_CHECK_ARG(x && wg && gates && expert && row && d_weight && exp_counts && d_logits && dx && d_wg && workspace, "ie_moe_gate_bwd: null pointer");
    IE_CHECK_SUPPORTED(E >= 2 && E <= kMaxE && M % 8 == 0 && x_ld % 8 == 0 && dx_ld % 8 == 0, "ie_moe_gate_bwd: 2 <= experts <= 16, hidden size a multiple of 8");
    if (S == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_gate_bwd_k, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, wg, gates, expert, row, d_weight, exp_counts, loss_scale_dev, aux_factor,
                       S, M, E, d_logits, (bf16_t*)dx, dx_ld);
    hipLaunchKernelGGL(moe_dwg_partial_k, dim3((unsigned)((M + 255) / 256), kDwgSplit), dim3(256), 0, st, (const bf16_t*)x, x_ld, d_logits, S, M, E, kDwgSplit,
                       workspace);
    const int64_t n = (int64_t)E * M;
    hipLaunchKernelGGL(moe_dwg_reduce_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, kDwgSplit, n, d_wg, accumulate_d_wg);
    return ie_launch_status("ie_moe_gate_bwd launch");
}
