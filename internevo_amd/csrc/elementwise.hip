// K2 rotary, K8 SwiGLU, embedding and small elementwise kernels for gfx950.  All HBM-bound:
// 16-byte accesses per lane, grid sized to the problem, no LDS.
#include "ie_common.h"

namespace {

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// Generic strided rotary (drop-in for rotary_emb.apply_rotary; reference restatement
// internlm/model/modules/embedding.py:63-86).
template <bool BF>
__global__ __launch_bounds__(256) void apply_rotary_k(const void* __restrict__ x1, const void* __restrict__ x2,
                                                      const void* __restrict__ cs, const void* __restrict__ sn,
                                                      void* out1, void* out2, int64_t batch, int64_t seq, int64_t heads,
                                                      int64_t half, int64_t xs_b, int64_t xs_s, int64_t xs_h, int64_t os_b,
                                                      int64_t os_s, int64_t os_h, int64_t cs_ld, int conj) {
    const int64_t total = batch * seq * heads * half;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % half;
        int64_t r = idx / half;
        const int64_t h = r % heads; r /= heads;
        const int64_t s = r % seq;
        const int64_t b = r / seq;
        const int64_t xo = b * xs_b + s * xs_s + h * xs_h + i;
        const int64_t oo = b * os_b + s * os_s + h * os_h + i;
        float a, c, co, si;
        if (BF) {
            a = bf2f(((const bf16_t*)x1)[xo]); c = bf2f(((const bf16_t*)x2)[xo]);
            co = bf2f(((const bf16_t*)cs)[s * cs_ld + i]); si = bf2f(((const bf16_t*)sn)[s * cs_ld + i]);
        } else {
            a = ((const float*)x1)[xo]; c = ((const float*)x2)[xo];
            co = ((const float*)cs)[s * cs_ld + i]; si = ((const float*)sn)[s * cs_ld + i];
        }
        float o1, o2;
        if (conj) { o1 = a * co + c * si; o2 = -a * si + c * co; }
        else      { o1 = a * co - c * si; o2 = a * si + c * co; }
        if (BF) { ((bf16_t*)out1)[oo] = f2bf(o1); ((bf16_t*)out2)[oo] = f2bf(o2); }
        else    { ((float*)out1)[oo] = o1; ((float*)out2)[oo] = o2; }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused wqkv-output split + de-interleave + cos/sin gather + rotate (fwd) and its adjoint (bwd).
// One thread = one 16-element span of one (token, kv-group, slot) head.  D = head dim.
template <int D, bool INTERLEAVED>
__global__ __launch_bounds__(256) void qkv_rotary_fwd_k(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cs,
                                                        const bf16_t* __restrict__ sn, const int64_t* __restrict__ pos,
                                                        bf16_t* __restrict__ q_out, bf16_t* __restrict__ kv_out, int64_t T,
                                                        int hkv, int qpk, float q_scale) {
    constexpr int CH = D / 16;  // 16-element spans per head
    constexpr int H2 = D / 2;
    const int gs = qpk + 2;
    const int64_t total = T * hkv * gs * CH;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % CH);
    int64_t r = idx / CH;
    const int s = (int)(r % gs); r /= gs;
    const int g = (int)(r % hkv);
    const int64_t t = r / hkv;
    const bf16_t* src = qkv + ((t * hkv + g) * gs + s) * D;
    bf16_t* dst;
    if (s < qpk) dst = q_out + (t * (int64_t)(hkv * qpk) + (g * qpk + s)) * D;
    else if (s == qpk) dst = kv_out + ((t * 2 + 0) * hkv + g) * D;
    else dst = kv_out + ((t * 2 + 1) * hkv + g) * D;
    if (s == qpk + 1) {  // v: plain copy
        st16(dst + ch * 16, ld16(src + ch * 16));
        st16(dst + ch * 16 + 8, ld16(src + ch * 16 + 8));
        return;
    }
    const int i0 = ch * 8;  // first rotary-pair index of this span
    float x1[8], x2[8];
    if (INTERLEAVED) {
        float lo[8], hi[8];
        unpack8(ld16(src + ch * 16), lo);
        unpack8(ld16(src + ch * 16 + 8), hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x1[e] = lo[2 * e]; x2[e] = lo[2 * e + 1];
            x1[4 + e] = hi[2 * e]; x2[4 + e] = hi[2 * e + 1];
        }
    } else {
        unpack8(ld16(src + i0), x1);
        unpack8(ld16(src + H2 + i0), x2);
    }
    const int64_t p = pos[t];
    float co[8], si[8], o1[8], o2[8];
    unpack8(ld16(cs + p * H2 + i0), co);
    unpack8(ld16(sn + p * H2 + i0), si);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o1[e] = x1[e] * co[e] - x2[e] * si[e];
        o2[e] = x1[e] * si[e] + x2[e] * co[e];
    }
    if (s < qpk && q_scale != 1.f) {   // the attention's softmax scale rides on q (fp32, before the one rounding to bf16)
#pragma unroll
        for (int e = 0; e < 8; ++e) o1[e] *= q_scale, o2[e] *= q_scale;
    }
    st16(dst + i0, pack8(o1));
    st16(dst + H2 + i0, pack8(o2));
}

template <int D, bool INTERLEAVED>
__global__ __launch_bounds__(256) void qkv_rotary_bwd_k(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dkv,
                                                        const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                                        const int64_t* __restrict__ pos, bf16_t* __restrict__ dqkv, int64_t T,
                                                        int hkv, int qpk, float dq_scale) {
    constexpr int CH = D / 16;
    constexpr int H2 = D / 2;
    const int gs = qpk + 2;
    const int64_t total = T * hkv * gs * CH;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % CH);
    int64_t r = idx / CH;
    const int s = (int)(r % gs); r /= gs;
    const int g = (int)(r % hkv);
    const int64_t t = r / hkv;
    bf16_t* dst = dqkv + ((t * hkv + g) * gs + s) * D;
    const bf16_t* src;
    if (s < qpk) src = dq + (t * (int64_t)(hkv * qpk) + (g * qpk + s)) * D;
    else if (s == qpk) src = dkv + ((t * 2 + 0) * hkv + g) * D;
    else src = dkv + ((t * 2 + 1) * hkv + g) * D;
    if (s == qpk + 1) {
        st16(dst + ch * 16, ld16(src + ch * 16));
        st16(dst + ch * 16 + 8, ld16(src + ch * 16 + 8));
        return;
    }
    const int i0 = ch * 8;
    float d1[8], d2[8], co[8], si[8], x1[8], x2[8];
    unpack8(ld16(src + i0), d1);
    unpack8(ld16(src + H2 + i0), d2);
    const int64_t p = pos[t];
    unpack8(ld16(cs + p * H2 + i0), co);
    unpack8(ld16(sn + p * H2 + i0), si);
#pragma unroll
    for (int e = 0; e < 8; ++e) rot_conj1(d1[e], d2[e], co[e], si[e], x1[e], x2[e]);
    if (s < qpk && dq_scale != 1.f) {   // chain rule of the forward's q_scale
#pragma unroll
        for (int e = 0; e < 8; ++e) x1[e] *= dq_scale, x2[e] *= dq_scale;
    }
    if (INTERLEAVED) {
        float lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[2 * e] = x1[e]; lo[2 * e + 1] = x2[e];
            hi[2 * e] = x1[4 + e]; hi[2 * e + 1] = x2[4 + e];
        }
        st16(dst + ch * 16, pack8(lo));
        st16(dst + ch * 16 + 8, pack8(hi));
    } else {
        st16(dst + i0, pack8(x1));
        st16(dst + H2 + i0, pack8(x2));
    }
}

// ------------------------------------------------------------------------------------------------
// SwiGLU.  Reference: Silu(w1_o, w3_o) = F.silu(w1_o) * w3_o on bf16 tensors (model/utils.py:684-688):
// silu evaluated in fp32, rounded to bf16, product rounded to bf16.
// (element arithmetic: swiglu_fwd1 / swiglu_bwd1 in ie_common.h, shared with the fused GEMM epilogues)
__global__ __launch_bounds__(256) void swiglu_fwd_k(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b,
                                                    int64_t ldb, bf16_t* __restrict__ out, int64_t ldo, int64_t rows,
                                                    int64_t cols8) {
    const int64_t total = rows * cols8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cols8, c = (idx % cols8) * 8;
        float av[8], bv[8], o[8];
        unpack8(ld16(a + r * lda + c), av);
        unpack8(ld16(b + r * ldb + c), bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd1(av[e], bv[e]);
        st16(out + r * ldo + c, pack8(o));
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_k(const bf16_t* __restrict__ dout, int64_t lddo, const bf16_t* __restrict__ a,
                                                    int64_t lda, const bf16_t* __restrict__ b, int64_t ldb, bf16_t* __restrict__ da,
                                                    int64_t ldda, bf16_t* __restrict__ db, int64_t lddb, bf16_t* __restrict__ act,
                                                    int64_t ldact, int64_t rows, int64_t cols8) {
    const int64_t total = rows * cols8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cols8, c = (idx % cols8) * 8;
        float av[8], bv[8], gv[8], oa[8], ob[8], oc[8];
        unpack8(ld16(a + r * lda + c), av);
        unpack8(ld16(b + r * ldb + c), bv);
        unpack8(ld16(dout + r * lddo + c), gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) swiglu_bwd1(gv[e], av[e], bv[e], oa[e], ob[e], oc[e]);
        st16(da + r * ldda + c, pack8(oa));
        st16(db + r * lddb + c, pack8(ob));
        if (act) st16(act + r * ldact + c, pack8(oc));
    }
}

// ------------------------------------------------------------------------------------------------
// ring attention (internevo_amd/seqpar.py): a block's partial result (out_p, lse_p) of n rows merged into the running fp32 result of the same rows:
//   lse' = log(e^lse + e^lse_p),  acc' = e^(lse - lse') acc + e^(lse_p - lse') out_p.   One thread per 8 elements of a (row, head); a partial without keys
// (lse_p = -inf) changes nothing.
__global__ __launch_bounds__(256) void attn_merge_k(float* __restrict__ acc, float* __restrict__ lse_acc, int64_t Ta, const bf16_t* __restrict__ out_p,
                                                    int64_t p_ts, const float* __restrict__ lse_p, int64_t Tp, int64_t n, int hq, int d) {
    const int tpp = d / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = gid / tpp;
    const int sub = (int)(gid % tpp);
    if (pair >= n * hq) return;
    const int64_t row = pair / hq;
    const int h = (int)(pair % hq);
    const float a = lse_acc[(int64_t)h * Ta + row], b = lse_p[(int64_t)h * Tp + row];
    if (b == -INFINITY) return;
    const float m = fmaxf(a, b);
    const float ea = (a == -INFINITY) ? 0.f : __expf(a - m), eb = __expf(b - m);
    const float inv = 1.f / (ea + eb);
    const float wa = ea * inv, wb = eb * inv;
    float* ap = acc + (row * hq + h) * (int64_t)d + sub * 8;
    float pv[8];
    unpack8(ld16(out_p + row * p_ts + (int64_t)h * d + sub * 8), pv);
    float4 x0 = *reinterpret_cast<float4*>(ap), x1 = *reinterpret_cast<float4*>(ap + 4);
    x0.x = wa * x0.x + wb * pv[0]; x0.y = wa * x0.y + wb * pv[1]; x0.z = wa * x0.z + wb * pv[2]; x0.w = wa * x0.w + wb * pv[3];
    x1.x = wa * x1.x + wb * pv[4]; x1.y = wa * x1.y + wb * pv[5]; x1.z = wa * x1.z + wb * pv[6]; x1.w = wa * x1.w + wb * pv[7];
    *reinterpret_cast<float4*>(ap) = x0;
    *reinterpret_cast<float4*>(ap + 4) = x1;
    // (the threads of a (row, head) are neighbouring lanes of one wave: all of them have read the old lse before this store is issued)
    if (sub == 0) lse_acc[(int64_t)h * Ta + row] = m + __logf(ea + eb);
}

// dst (fp32) += src (bf16): the fp32 sums ring attention keeps of the blocks' dQ / dK / dV shares
__global__ __launch_bounds__(256) void acc_bf16_k(float* __restrict__ dst, const bf16_t* __restrict__ src, int64_t n8) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n8) return;
    float v[8];
    unpack8(ld16(src + idx * 8), v);
    float4 x0 = *reinterpret_cast<float4*>(dst + idx * 8), x1 = *reinterpret_cast<float4*>(dst + idx * 8 + 4);
    x0.x += v[0]; x0.y += v[1]; x0.z += v[2]; x0.w += v[3];
    x1.x += v[4]; x1.y += v[5]; x1.z += v[6]; x1.w += v[7];
    *reinterpret_cast<float4*>(dst + idx * 8) = x0;
    *reinterpret_cast<float4*>(dst + idx * 8 + 4) = x1;
}

__global__ __launch_bounds__(256) void add_bf16_k(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o,
                                                  int64_t n8, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n8) {
        float av[8], bv[8];
        unpack8(ld16(a + idx * 8), av);
        unpack8(ld16(b + idx * 8), bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] += bv[e];
        st16(o + idx * 8, pack8(av));
    }
    if (idx == 0) {
        for (int64_t i = n8 * 8; i < n; ++i) o[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
    }
}

__global__ __launch_bounds__(256) void cast_k(const void* __restrict__ src, int sdt, void* __restrict__ dst, int ddt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = sdt == IE_BF16 ? bf2f(((const bf16_t*)src)[i]) : ((const float*)src)[i];
        if (ddt == IE_BF16) ((bf16_t*)dst)[i] = f2bf(v); else ((float*)dst)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Embedding gather: one wave per token row, 16 B per lane.
__global__ __launch_bounds__(256) void embedding_fwd_k(const bf16_t* __restrict__ w, const int64_t* __restrict__ ids,
                                                       bf16_t* __restrict__ out, int64_t T, int64_t vocab, int64_t dim) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) id = 0;  // reference F.embedding would raise; never hit on valid input
    const bf16_t* src = w + id * dim;
    bf16_t* dst = out + t * dim;
    if ((dim & 7) == 0) {
        for (int64_t c = lane * 8; c < dim; c += 512) st16(dst + c, ld16(src + c));
    } else {
        for (int64_t c = lane; c < dim; c += 64) dst[c] = src[c];
    }
}

// ws layout (int32): present[vocab] | count | list[T]
__global__ __launch_bounds__(256) void embedding_mark_k(const int64_t* __restrict__ ids, int* __restrict__ ws, int64_t T, int64_t vocab) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) {
        const int64_t id = ids[t];
        if (id >= 0 && id < vocab) {
            if (atomicExch(&ws[id], 1) == 0) {
                const int slot = atomicAdd(&ws[vocab], 1);
                ws[vocab + 1 + slot] = (int)id;
            }
        }
    }
}

// grid (max distinct rows, column chunks of 256).  Each block owns one referenced vocabulary row and
// 256 columns; tokens are visited in index order and summed in fp32, so the result is deterministic
// whatever order the rows were appended to the list in (torch's embedding backward also accumulates
// in fp32 per row).  Rows nobody referenced are left untouched (accumulate) or were zeroed (memset).
__global__ __launch_bounds__(256) void embedding_bwd_k(const bf16_t* __restrict__ dout, const int64_t* __restrict__ ids,
                                                       bf16_t* __restrict__ dw, const int* __restrict__ ws, int64_t T, int64_t vocab,
                                                       int64_t dim, int accumulate) {
    if ((int)blockIdx.x >= ws[vocab]) return;
    __shared__ unsigned long long masks[4];
    __shared__ unsigned short hits[256];
    const int64_t v = ws[vocab + 1 + blockIdx.x];
    const int64_t c = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    // 256 ids per pass: every thread tests one id, the 4 wave ballots go through LDS, the hits are listed IN TOKEN ORDER (a hit's place = the hits in
    // front of it), then all threads add their column of the listed rows in that order -- sixteen loads in flight before the sixteen adds (round 6: the
    // benchmark's data, dummy_dataset.py, has fewer than 30 distinct ids, some of them thousands of times; one load per add made the kernel a chain of HBM
    // latencies, 1.25 ms per step).  Same adds in the same order: the sum is unchanged bit for bit.
    for (int64_t base = 0; base < T; base += 256) {
        const int64_t t = base + threadIdx.x;
        const bool hit = t < T && ids[t] == v;
        const unsigned long long m = __ballot(hit);
        __syncthreads();   // (the list of the pass before has been read)
        if (lane == 0) masks[wave] = m;
        __syncthreads();
        int before = 0, nh = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int k = __popcll(masks[w]);
            if (w < wave) before += k;
            nh += k;
        }
        if (hit) hits[before + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)threadIdx.x;
        if (nh == 0) continue;   // (uniform: every thread read the same four masks)
        __syncthreads();
        if (c < dim) {
            for (int k0 = 0; k0 < nh; k0 += 16) {
                bf16_t x[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k0 + k < nh) x[k] = dout[(base + hits[k0 + k]) * dim + c];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k0 + k < nh) acc += bf2f(x[k]);
            }
        }
    }
    if (c < dim) {
        float r = rbf(acc);
        if (accumulate) r += bf2f(dw[v * dim + c]);
        dw[v * dim + c] = f2bf(r);
    }
}

inline unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 1 << 20) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------ a18 Ulysses exchange layout
// _SeqAllToAll (internlm/model/modules/multi_head_attention.py:27-53) splits the head dimension into `S` contiguous blocks,
// sends block r to rank r and concatenates what it receives along the sequence.  With all_to_all_single the send buffer must
// hold block r contiguously: [A tokens][B][S][C] -> [S][A][B][C] (q / ctx: B = 1, C = heads_per_rank * d; kv: B = 2).
// The receive buffer [S][A][B][C] IS the gathered-sequence tensor [S*A tokens][B][C], so only the send side needs a copy;
// the inverse direction (gathered heads back to [A][B][S][C]) is the same kernel with the index maps swapped.
__global__ __launch_bounds__(256) void seq_head_permute_k(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t A, int B, int S,
                                                          int64_t C8, int inverse, int64_t total) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i indexes the [A][B][S][C8] side
        const int64_t c = i % C8;
        int64_t r = i / C8;
        const int s = (int)(r % S);
        r /= S;
        const int b = (int)(r % B);
        const int64_t a = r / B;
        const int64_t j = (((int64_t)s * A + a) * B + b) * C8 + c;  // the [S][A][B][C8] side
        if (inverse) out[i] = in[j];
        else out[j] = in[i];
    }
}

__global__ __launch_bounds__(256) void scale_bf16_k(bf16_t* __restrict__ x, int64_t n, float f) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = f2bf(bf2f(x[i]) * f);
}


// ---- ScaleColumnParallelLinearWithNormHead (ops/linear.py:79-153) and the embedding's gradient scale (modeling_internlm2.py:970-973) ----
// x -> s x + (1 - s) x.detach(): the VALUE the reference computes in bf16 (three roundings: s x, (1 - s) x, their sum); the gradient factor s
// is applied by the backward kernels
__device__ __forceinline__ float grad_scale_mix(float x, float s) {
    return rbf(rbf(s * x) + rbf((1.f - s) * x));
}

__global__ __launch_bounds__(256) void grad_scale_mix_k(bf16_t* __restrict__ x, int64_t n, float s) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = f2bf(grad_scale_mix(bf2f(x[i]), s));
}

// one block per weight row: out = normalize(mix(w)) (each step optional); inv_norm[row] = 1 / max(bf16(||e||), 1e-12) for the backward
__global__ __launch_bounds__(256) void head_weight_fwd_k(const bf16_t* __restrict__ w, int64_t w_ld, bf16_t* __restrict__ out, int64_t o_ld,
                                                         float* __restrict__ inv_norm, int64_t cols, float s, int norm_head) {
    __shared__ float scratch[8];
    const int64_t row = blockIdx.x;
    const bf16_t* wr = w + row * w_ld;
    bf16_t* orow = out + row * o_ld;
    float ss = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        float e = bf2f(wr[c]);
        if (s != 1.f) e = grad_scale_mix(e, s);
        ss += e * e;
        if (!norm_head) orow[c] = f2bf(e);
    }
    if (!norm_head) return;
    ss = block_sum<4>(ss, scratch);
    const float n = fmaxf(rbf(sqrtf(ss)), 1e-12f);   // F.normalize: input.norm(2, dim, keepdim=True) (a bf16 tensor) .clamp_min(eps)
    if (threadIdx.x == 0) inv_norm[row] = 1.f / n;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        float e = bf2f(wr[c]);
        if (s != 1.f) e = grad_scale_mix(e, s);
        orow[c] = f2bf(e / n);
    }
}

// gradient of the above: dy = gradient w.r.t. the weight the GEMM used, y = that weight (normalised rows); dw (+)= s * (dy - y (y . dy)) / n
__global__ __launch_bounds__(256) void head_weight_bwd_k(const bf16_t* __restrict__ dy, int64_t dy_ld, const bf16_t* __restrict__ y, int64_t y_ld,
                                                         const float* __restrict__ inv_norm, bf16_t* __restrict__ dw, int64_t dw_ld, int64_t cols,
                                                         float s, int norm_head, int accumulate) {
    __shared__ float scratch[8];
    const int64_t row = blockIdx.x;
    const bf16_t* dr = dy + row * dy_ld;
    const bf16_t* yr = y + row * y_ld;
    bf16_t* wr = dw + row * dw_ld;
    float dot = 0.f, inv = 1.f;
    if (norm_head) {
        for (int64_t c = threadIdx.x; c < cols; c += 256) dot += bf2f(dr[c]) * bf2f(yr[c]);
        dot = block_sum<4>(dot, scratch);
        inv = inv_norm[row];
    }
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        float g = bf2f(dr[c]);
        if (norm_head) g = (g - bf2f(yr[c]) * dot) * inv;
        g *= s;
        wr[c] = f2bf(accumulate ? bf2f(wr[c]) + g : g);
    }
}

}  // namespace

extern "C" int ie_apply_rotary(const void* x1, const void* x2, const void* cos_, const void* sin_, void* out1, void* out2,
                               int dtype, int64_t batch, int64_t seq, int64_t heads, int64_t half, int64_t xs_b, int64_t xs_s,
                               int64_t xs_h, int64_t os_b, int64_t os_s, int64_t os_h, int64_t cs_ld, int conj, void* stream) {
    IE_CHECK_ARG(x1 && x2 && cos_ && sin_ && out1 && out2, "ie_apply_rotary: null pointer");
    IE_CHECK_ARG(batch >= 0 && seq >= 0 && heads >= 0 && half >= 0, "ie_apply_rotary: bad shape");
    IE_CHECK_ARG(dtype == IE_BF16 || dtype == IE_F32, "ie_apply_rotary: bad dtype");
    const int64_t total = batch * seq * heads * half;
    if (total == 0) return IE_OK;
    dim3 grid(grid_for(total, 256, 65536));
    if (dtype == IE_BF16)
        hipLaunchKernelGGL((apply_rotary_k<true>), grid, dim3(256), 0, (hipStream_t)stream, x1, x2, cos_, sin_, out1, out2, batch, seq,
                           heads, half, xs_b, xs_s, xs_h, os_b, os_s, os_h, cs_ld, conj);
    else
        hipLaunchKernelGGL((apply_rotary_k<false>), grid, dim3(256), 0, (hipStream_t)stream, x1, x2, cos_, sin_, out1, out2, batch, seq,
                           heads, half, xs_b, xs_s, xs_h, os_b, os_s, os_h, cs_ld, conj);
    return ie_launch_status("ie_apply_rotary launch");
}

extern "C" int ie_qkv_rotary_fwd_scaled(const void* qkv, const void* cos_, const void* sin_, const int64_t* pos, void* q_out, void* kv_out,
                                        int64_t T, int hkv, int q_per_kv, int d, int interleaved, float q_scale, void* stream) {
    IE_CHECK_ARG(qkv && cos_ && sin_ && pos && q_out && kv_out, "ie_qkv_rotary_fwd: null pointer");
    IE_CHECK_ARG(q_scale > 0.f, "ie_qkv_rotary_fwd_scaled: q_scale must be positive");
    IE_CHECK_ARG(T >= 0 && hkv > 0 && q_per_kv > 0, "ie_qkv_rotary_fwd: bad shape");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_qkv_rotary_fwd: head dim must be 64 or 128");
    IE_CHECK_ARG(aligned16(qkv) && aligned16(cos_) && aligned16(sin_) && aligned16(q_out) && aligned16(kv_out),
                 "ie_qkv_rotary_fwd: pointers must be 16-byte aligned");
    if (T == 0) return IE_OK;
    const int64_t total = T * hkv * (q_per_kv + 2) * (d / 16);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define IE_L(DD, IL)                                                                                                      \
    hipLaunchKernelGGL((qkv_rotary_fwd_k<DD, IL>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)cos_,         \
                       (const bf16_t*)sin_, pos, (bf16_t*)q_out, (bf16_t*)kv_out, T, hkv, q_per_kv, q_scale)
    if (d == 128) { if (interleaved) IE_L(128, true); else IE_L(128, false); }
    else          { if (interleaved) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
    return ie_launch_status("ie_qkv_rotary_fwd launch");
}

extern "C" int ie_qkv_rotary_fwd(const void* qkv, const void* cos_, const void* sin_, const int64_t* pos, void* q_out, void* kv_out,
                                 int64_t T, int hkv, int q_per_kv, int d, int interleaved, void* stream) {
    return ie_qkv_rotary_fwd_scaled(qkv, cos_, sin_, pos, q_out, kv_out, T, hkv, q_per_kv, d, interleaved, 1.f, stream);
}

extern "C" int ie_qkv_rotary_bwd_scaled(const void* dq, const void* dkv, const void* cos_, const void* sin_, const int64_t* pos, void* dqkv,
                                        int64_t T, int hkv, int q_per_kv, int d, int interleaved, float dq_scale, void* stream) {
    IE_CHECK_ARG(dq && dkv && cos_ && sin_ && pos && dqkv, "ie_qkv_rotary_bwd: null pointer");
    IE_CHECK_ARG(dq_scale > 0.f, "ie_qkv_rotary_bwd_scaled: dq_scale must be positive");
    IE_CHECK_ARG(T >= 0 && hkv > 0 && q_per_kv > 0, "ie_qkv_rotary_bwd: bad shape");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_qkv_rotary_bwd: head dim must be 64 or 128");
    IE_CHECK_ARG(aligned16(dq) && aligned16(dkv) && aligned16(cos_) && aligned16(sin_) && aligned16(dqkv),
                 "ie_qkv_rotary_bwd: pointers must be 16-byte aligned");
    if (T == 0) return IE_OK;
    const int64_t total = T * hkv * (q_per_kv + 2) * (d / 16);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define IE_L(DD, IL)                                                                                                   \
    hipLaunchKernelGGL((qkv_rotary_bwd_k<DD, IL>), grid, dim3(256), 0, st, (const bf16_t*)dq, (const bf16_t*)dkv,       \
                       (const bf16_t*)cos_, (const bf16_t*)sin_, pos, (bf16_t*)dqkv, T, hkv, q_per_kv, dq_scale)
    if (d == 128) { if (interleaved) IE_L(128, true); else IE_L(128, false); }
    else          { if (interleaved) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
    return ie_launch_status("ie_qkv_rotary_bwd launch");
}

extern "C" int ie_qkv_rotary_bwd(const void* dq, const void* dkv, const void* cos_, const void* sin_, const int64_t* pos, void* dqkv,
                                 int64_t T, int hkv, int q_per_kv, int d, int interleaved, void* stream) {
    return ie_qkv_rotary_bwd_scaled(dq, dkv, cos_, sin_, pos, dqkv, T, hkv, q_per_kv, d, interleaved, 1.f, stream);
}

extern "C" int ie_swiglu_fwd(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows,
                             int64_t cols, void* stream) {
    IE_CHECK_ARG(a && b && out, "ie_swiglu_fwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && cols >= 0, "ie_swiglu_fwd: bad shape");
    IE_CHECK_SUPPORTED(cols % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 && aligned16(a) && aligned16(b) && aligned16(out),
                       "ie_swiglu_fwd: cols/strides must be multiples of 8 and pointers 16-byte aligned");
    if (rows * cols == 0) return IE_OK;
    dim3 grid(grid_for(rows * (cols / 8), 256, 1 << 16));
    hipLaunchKernelGGL(swiglu_fwd_k, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                       (bf16_t*)out, ldo, rows, cols / 8);
    return ie_launch_status("ie_swiglu_fwd launch");
}

extern "C" int ie_swiglu_bwd(const void* dout, int64_t lddo, const void* a, int64_t lda, const void* b, int64_t ldb, void* da,
                             int64_t ldda, void* db, int64_t lddb, void* act_out, int64_t ldact, int64_t rows, int64_t cols,
                             void* stream) {
    IE_CHECK_ARG(dout && a && b && da && db, "ie_swiglu_bwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && cols >= 0, "ie_swiglu_bwd: bad shape");
    IE_CHECK_SUPPORTED(cols % 8 == 0 && lddo % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldda % 8 == 0 && lddb % 8 == 0 &&
                           (!act_out || ldact % 8 == 0) && aligned16(dout) && aligned16(a) && aligned16(b) && aligned16(da) &&
                           aligned16(db) && (!act_out || aligned16(act_out)),
                       "ie_swiglu_bwd: cols/strides must be multiples of 8 and pointers 16-byte aligned");
    if (rows * cols == 0) return IE_OK;
    dim3 grid(grid_for(rows * (cols / 8), 256, 1 << 16));
    hipLaunchKernelGGL(swiglu_bwd_k, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, lddo, (const bf16_t*)a, lda,
                       (const bf16_t*)b, ldb, (bf16_t*)da, ldda, (bf16_t*)db, lddb, (bf16_t*)act_out, ldact, rows, cols / 8);
    return ie_launch_status("ie_swiglu_bwd launch");
}

extern "C" int ie_seq_head_permute(const void* in, void* out, int64_t A, int B, int S, int64_t C, int inverse, void* stream) {
    IE_CHECK_ARG(in && out && A >= 0 && B > 0 && S > 0 && C > 0, "ie_seq_head_permute: bad argument");
    IE_CHECK_ARG(C % 8 == 0 && aligned16(in) && aligned16(out), "ie_seq_head_permute: C must be a multiple of 8 bf16 and pointers 16-byte aligned");
    const int64_t total = A * B * S * (C / 8);
    if (total == 0) return IE_OK;
    hipLaunchKernelGGL(seq_head_permute_k, dim3(grid_for(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, A,
                       B, S, C / 8, inverse, total);
    return ie_launch_status("ie_seq_head_permute launch");
}

extern "C" int ie_scale_bf16(void* x, int64_t n, float factor, void* stream) {
    IE_CHECK_ARG(x && n >= 0, "ie_scale_bf16: bad argument");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(scale_bf16_k, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n, factor);
    return ie_launch_status("ie_scale_bf16 launch");
}

extern "C" int ie_grad_scale_mix(void* x, int64_t n, float scale, void* stream) {
    IE_CHECK_ARG(x && n >= 0, "ie_grad_scale_mix: bad argument");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(grad_scale_mix_k, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n, scale);
    return ie_launch_status("ie_grad_scale_mix launch");
}

extern "C" int ie_head_weight_fwd(const void* w, int64_t w_ld, void* out, int64_t out_ld, float* inv_norm, int64_t rows, int64_t cols, float scale,
                                  int norm_head, void* stream) {
    IE_CHECK_ARG(w && out && rows >= 0 && cols > 0 && (inv_norm || !norm_head), "ie_head_weight_fwd: bad argument");
    if (rows == 0) return IE_OK;
    hipLaunchKernelGGL(head_weight_fwd_k, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, w_ld, (bf16_t*)out, out_ld, inv_norm,
                       cols, scale, norm_head);
    return ie_launch_status("ie_head_weight_fwd launch");
}

extern "C" int ie_head_weight_bwd(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, const float* inv_norm, void* dw, int64_t dw_ld, int64_t rows,
                                  int64_t cols, float scale, int norm_head, int accumulate, void* stream) {
    IE_CHECK_ARG(dy && y && dw && rows >= 0 && cols > 0 && (inv_norm || !norm_head), "ie_head_weight_bwd: bad argument");
    if (rows == 0) return IE_OK;
    hipLaunchKernelGGL(head_weight_bwd_k, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, dy_ld, (const bf16_t*)y, y_ld, inv_norm,
                       (bf16_t*)dw, dw_ld, cols, scale, norm_head, accumulate);
    return ie_launch_status("ie_head_weight_bwd launch");
}

extern "C" int ie_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    IE_CHECK_ARG(a && b && out && n >= 0, "ie_add_bf16: bad argument");
    IE_CHECK_ARG(aligned16(a) && aligned16(b) && aligned16(out), "ie_add_bf16: pointers must be 16-byte aligned");
    if (n == 0) return IE_OK;
    const int64_t n8 = n / 8;
    dim3 grid((unsigned)((n8 + 256) / 256));
    hipLaunchKernelGGL(add_bf16_k, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n8, n);
    return ie_launch_status("ie_add_bf16 launch");
}

extern "C" int ie_attn_merge(float* acc, float* lse_acc, int64_t Ta, const void* out_p, int64_t p_ts, const float* lse_p, int64_t Tp, int64_t n, int hq,
                             int d, void* stream) {
    IE_CHECK_ARG(acc && lse_acc && out_p && lse_p && n >= 0 && n <= Ta && n <= Tp && hq > 0 && d > 0 && d % 8 == 0 && p_ts % 8 == 0,
                 "ie_attn_merge: bad argument");
    IE_CHECK_ARG(aligned16(acc) && aligned16(out_p), "ie_attn_merge: pointers must be 16-byte aligned");
    if (n == 0) return IE_OK;
    const int64_t threads = n * hq * (d / 8);
    hipLaunchKernelGGL(attn_merge_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, lse_acc, Ta, (const bf16_t*)out_p, p_ts,
                       lse_p, Tp, n, hq, d);
    return ie_launch_status("ie_attn_merge launch");
}

extern "C" int ie_acc_bf16(float* dst, const void* src, int64_t n, void* stream) {
    IE_CHECK_ARG(dst && src && n >= 0 && n % 8 == 0 && aligned16(dst) && aligned16(src), "ie_acc_bf16: 16-byte aligned pointers and a multiple of 8 elements");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(acc_bf16_k, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, (const bf16_t*)src, n / 8);
    return ie_launch_status("ie_acc_bf16 launch");
}

extern "C" int ie_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    IE_CHECK_ARG(src && dst && n >= 0, "ie_cast: bad argument");
    IE_CHECK_ARG((src_dtype == IE_BF16 || src_dtype == IE_F32) && (dst_dtype == IE_BF16 || dst_dtype == IE_F32), "ie_cast: bad dtype");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(cast_k, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, dst, dst_dtype, n);
    return ie_launch_status("ie_cast launch");
}

extern "C" int ie_embedding_fwd(const void* weight, const int64_t* ids, void* out, int64_t T, int64_t vocab, int64_t dim, void* stream) {
    IE_CHECK_ARG(weight && ids && out && T >= 0 && vocab > 0 && dim > 0, "ie_embedding_fwd: bad argument");
    IE_CHECK_ARG(dim % 8 != 0 || (aligned16(weight) && aligned16(out)), "ie_embedding_fwd: pointers must be 16-byte aligned");
    if (T == 0) return IE_OK;
    hipLaunchKernelGGL(embedding_fwd_k, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)weight, ids,
                       (bf16_t*)out, T, vocab, dim);
    return ie_launch_status("ie_embedding_fwd launch");
}

extern "C" int ie_embedding_bwd(const void* dout, const int64_t* ids, void* dweight, int* ws, int64_t T, int64_t vocab,
                                int64_t dim, int accumulate, void* stream) {
    IE_CHECK_ARG(dout && ids && dweight && ws && T >= 0 && vocab > 0 && dim > 0, "ie_embedding_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(int) * (vocab + 1), st) != hipSuccess) {
        ie_set_error("ie_embedding_bwd: memset failed");
        return IE_ERR_LAUNCH;
    }
    if (!accumulate && hipMemsetAsync(dweight, 0, sizeof(bf16_t) * vocab * dim, st) != hipSuccess) {
        ie_set_error("ie_embedding_bwd: memset failed");
        return IE_ERR_LAUNCH;
    }
    if (T == 0) return IE_OK;
    hipLaunchKernelGGL(embedding_mark_k, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, ids, ws, T, vocab);
    const int64_t max_rows = T < vocab ? T : vocab;
    hipLaunchKernelGGL(embedding_bwd_k, dim3((unsigned)max_rows, (unsigned)((dim + 255) / 256)), dim3(256), 0, st, (const bf16_t*)dout, ids,
                       (bf16_t*)dweight, ws, T, vocab, dim, accumulate);
    return ie_launch_status("ie_embedding_bwd launch");
}

// ---- y[r, :] += bias  (bf16, in place): the bias of the InternLM-1 attention linears (multi_head_attention.py:371-408, model_type INTERNLM_MoE).
// HBM-bound: 4 bytes per element.
namespace {
__global__ __launch_bounds__(256) void bias_add_k(bf16_t* __restrict__ y, int64_t ld, const bf16_t* __restrict__ bias, int64_t rows, int cols) {
    const int c8 = cols / 8;
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= rows * c8) return;
    const int64_t r = i / c8;
    const int c = (int)(i % c8) * 8;
    float a[8], b[8];
    unpack8(ld16(y + r * ld + c), a);
    unpack8(ld16(bias + c), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    st16(y + r * ld + c, pack8(a));
}
}  // namespace

extern "C" int ie_bias_add_bf16(void* y, int64_t ld, const void* bias, int64_t rows, int64_t cols, void* stream) {
    IE_CHECK_ARG(y && bias && rows >= 0 && cols >= 0 && ld >= cols, "ie_bias_add_bf16: bad argument");
    IE_CHECK_SUPPORTED(cols % 8 == 0 && ld % 8 == 0 && (((uintptr_t)y) & 15u) == 0 && (((uintptr_t)bias) & 15u) == 0,
                       "ie_bias_add_bf16: columns / leading dimension multiples of 8, 16-byte aligned pointers");
    if (rows == 0 || cols == 0) return IE_OK;
    const int64_t n = rows * (cols / 8);
    hipLaunchKernelGGL(bias_add_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)y, ld, (const bf16_t*)bias, rows, (int)cols);
    return ie_launch_status("ie_bias_add_bf16 launch");
}
