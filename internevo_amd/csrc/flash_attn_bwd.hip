// K1 backward: varlen, causal / full, GQA flash attention for gfx950.  Deterministic (no atomics):
//   1. flash_delta_k : delta[h][t] = sum_d dO * O
//   2. flash_dq_k    : one block per 128 query rows, loops over K/V tiles, recomputes P, dQ += dS K
//   3. flash_dkdv_k  : one block per 128 keys (one wave per 32 keys), loops over the q heads of the
//                      GQA group and their query tiles, recomputes P, dV += P^T dO, dK += dS^T Q
// The recompute costs 7 instead of 5 tile products but keeps every output written by exactly one
// workgroup in a fixed order, so the result is bit-reproducible run to run.
//
// Layout tricks are the forward's (flash_common.h): in the dQ kernel scores are formed transposed
// (lane = query row), in the dK/dV kernel un-transposed (lane = key), so the accumulator registers
// of S / dS are directly the k-slots of the next MFMA; the operands that need their contraction
// index contiguous (K^T, Q^T, dO^T) come from transposed LDS images built at staging time.
#include "flash_common.h"

namespace {

using namespace fa;

// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void flash_delta_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ out,
                                                     int64_t o_ts, float* __restrict__ delta, int64_t T, int hq) {
    constexpr int TPP = D / 8;  // threads per (token, head) pair
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = gid / TPP;
    const int sub = (int)(gid % TPP);
    float acc = 0.f;
    const bool ok = pair < T * hq;
    if (ok) {
        const int64_t t = pair / hq;
        const int h = (int)(pair % hq);
        float a[8], b[8];
        unpack8(ld16(dout + t * do_ts + (int64_t)h * D + sub * 8), a);
        unpack8(ld16(out + t * o_ts + (int64_t)h * D + sub * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += a[e] * b[e];
    }
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (ok && sub == 0) {
        const int64_t t = pair / hq;
        const int h = (int)(pair % hq);
        delta[(int64_t)h * T + t] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void flash_dq_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                  int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                  int64_t kv_ts, const float* __restrict__ lse, const float* __restrict__ delta,
                                                  bf16_t* __restrict__ dq, int64_t dq_ts, const int32_t* __restrict__ cu, int64_t T,
                                                  int hq, int hkv, float scale) {
    using G = Geo<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 64 * G::ROW_PITCH + D * G::T_PITCH];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + 64 * G::ROW_PITCH;
    unsigned char* Kt = smem + 2 * 64 * G::ROW_PITCH;

    const int seq = blockIdx.z;
    const int h = blockIdx.y;
    const int qt = gridDim.x - 1 - blockIdx.x;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 128;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qw0 = q0 + wave * 32;
    const int my_q = qw0 + (lane & 31);
    const bool q_valid = my_q < len;

    s16x8 qf[G::KS], dof[G::KS];
    {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
        const bf16_t* dp = dout + (int64_t)(tok0 + my_q) * do_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } a, b;
            a.u = q_valid ? ld16(qp + ks * 16) : z4();
            b.u = q_valid ? ld16(dp + ks * 16) : z4();
            qf[ks] = a.s;
            dof[ks] = b.s;
        }
    }
    const float lse2 = q_valid ? lse[(int64_t)h * T + tok0 + my_q] * kLog2e : INFINITY;
    const float dlt = q_valid ? delta[(int64_t)h * T + tok0 + my_q] : 0.f;
    const float sc2 = scale * kLog2e;

    f32x16 dqacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) dqacc[db] = zero16();

    const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
    const int ntiles = (kv_end + 63) / 64;
    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;

    RowStager<D, 64, 256> k_st, v_st;
    TransStager<D, 256> kt_st;
    k_st.load(kbase, kv_ts, len);
    v_st.load(vbase, kv_ts, len);
    kt_st.load(kbase, kv_ts, len);
    k_st.store(Ks);
    v_st.store(Vs);
    kt_st.store(Kt);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * 64;
        const bool more = t + 1 < ntiles;
        if (more) {
            const int rem = len - (kv0 + 64);
            k_st.load(kbase + (int64_t)(kv0 + 64) * kv_ts, kv_ts, rem);
            v_st.load(vbase + (int64_t)(kv0 + 64) * kv_ts, kv_ts, rem);
            kt_st.load(kbase + (int64_t)(kv0 + 64) * kv_ts, kv_ts, rem);
        }
        const bool active = !CAUSAL || kv0 <= qw0 + 31;
        if (active) {
            const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > len);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, lane), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Vs, 32 * c, ks, lane), dof[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[r] * sc2 - lse2);
                    if (need_mask) {
                        const int key = kv0 + 32 * c + creg_row(r, lane);
                        if (key >= len || (CAUSAL && key > my_q)) p = 0.f;
                    }
                    s[r] = p * (dp[r] - dlt);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x8 dsf = pack_frag(s, s2);
#pragma unroll
                    for (int db = 0; db < G::DB; ++db)
                        dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag(Kt, 32 * db, 2 * c + s2, lane), dsf, dqacc[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            k_st.store(Ks);
            v_st.store(Vs);
            kt_st.store(Kt);
        }
        __syncthreads();
    }

    if (q_valid) {
        bf16_t* op = dq + (int64_t)(tok0 + my_q) * dq_ts + (int64_t)h * D;
#pragma unroll
        for (int db = 0; db < G::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(dqacc[db][4 * g + 0] * scale, dqacc[db][4 * g + 1] * scale);
                w.y = pack2bf(dqacc[db][4 * g + 2] * scale, dqacc[db][4 * g + 3] * scale);
                st8(op + 32 * db + 8 * g + 4 * (lane >> 5), w);
            }
    }
}

// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void flash_dkdv_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                    int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                    int64_t kv_ts, const float* __restrict__ lse, const float* __restrict__ delta,
                                                    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dkv_ts,
                                                    const int32_t* __restrict__ cu, int64_t T, int hq, int hkv, float scale) {
    using G = Geo<D>;
    constexpr int ROWIMG = 64 * G::ROW_PITCH;
    constexpr int TIMG = D * G::T_PITCH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ROWIMG + 2 * TIMG + 2 * 64 * 4];
    unsigned char* Qs = smem;
    unsigned char* dOs = smem + ROWIMG;
    unsigned char* Qt = smem + 2 * ROWIMG;
    unsigned char* dOt = smem + 2 * ROWIMG + TIMG;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * ROWIMG + 2 * TIMG);
    float* dlt_s = lse_s + 64;

    const int seq = blockIdx.z;
    const int hk = blockIdx.y;
    const int kb = blockIdx.x;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int k0 = kb * 128;
    if (k0 >= len) return;
    const int grp = hq / hkv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kw0 = k0 + wave * 32;
    const int my_k = kw0 + (lane & 31);
    const bool k_valid = my_k < len;

    // K, V fragments of this wave's 32 keys (B operands: lane = key)
    s16x8 kf[G::KS], vf[G::KS];
    {
        const bf16_t* kp = k + (int64_t)(tok0 + my_k) * kv_ts + (int64_t)hk * D + (lane >> 5) * 8;
        const bf16_t* vp = v + (int64_t)(tok0 + my_k) * kv_ts + (int64_t)hk * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } a, b;
            a.u = k_valid ? ld16(kp + ks * 16) : z4();
            b.u = k_valid ? ld16(vp + ks * 16) : z4();
            kf[ks] = a.s;
            vf[ks] = b.s;
        }
    }
    const float sc2 = scale * kLog2e;
    f32x16 dkacc[G::DB], dvacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { dkacc[db] = zero16(); dvacc[db] = zero16(); }

    const int nqt_all = (len + 63) / 64;
    const int qt_start = CAUSAL ? (k0 / 64) : 0;
    const int nqt = nqt_all - qt_start;  // >= 1 because k0 < len
    const int nit = grp * nqt;

    RowStager<D, 64, 256> q_st, do_st;
    TransStager<D, 256> qt_st, dot_st;
    float lse_r = 0.f, dlt_r = 0.f;  // threads 0..63 stage lse/delta of the tile

    auto issue = [&](int it) {
        const int h = hk * grp + it / nqt;
        const int q0 = (qt_start + it % nqt) * 64;
        const int rem = len - q0;
        const bf16_t* qb = q + (int64_t)(tok0 + q0) * q_ts + (int64_t)h * D;
        const bf16_t* dob = dout + (int64_t)(tok0 + q0) * do_ts + (int64_t)h * D;
        q_st.load(qb, q_ts, rem);
        do_st.load(dob, do_ts, rem);
        qt_st.load(qb, q_ts, rem);
        dot_st.load(dob, do_ts, rem);
        if (threadIdx.x < 64) {
            const bool ok = (int)threadIdx.x < rem;
            lse_r = ok ? lse[(int64_t)h * T + tok0 + q0 + threadIdx.x] * kLog2e : INFINITY;
            dlt_r = ok ? delta[(int64_t)h * T + tok0 + q0 + threadIdx.x] : 0.f;
        }
    };
    auto commit = [&]() {
        q_st.store(Qs);
        do_st.store(dOs);
        qt_st.store(Qt);
        dot_st.store(dOt);
        if (threadIdx.x < 64) { lse_s[threadIdx.x] = lse_r; dlt_s[threadIdx.x] = dlt_r; }
    };

    issue(0);
    commit();
    __syncthreads();

    for (int it = 0; it < nit; ++it) {
        const int q0 = (qt_start + it % nqt) * 64;
        const bool more = it + 1 < nit;
        if (more) issue(it + 1);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            const int qb0 = q0 + 32 * qs;  // first query row of this sub-block
            const bool active = qb0 < len && (!CAUSAL || qb0 + 31 >= kw0);
            if (!active) continue;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Qs, 32 * qs, ks, lane), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(dOs, 32 * qs, ks, lane), vf[ks], dp, 0, 0, 0);
            }
            const bool need_mask = (CAUSAL && kw0 + 31 > qb0) || (kw0 + 32 > len);
            f32x16 p;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = 32 * qs + 8 * g + 4 * (lane >> 5);  // tile-local row of regs 4g..4g+3
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + rl);
                const float4 d4 = *reinterpret_cast<const float4*>(dlt_s + rl);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
                const float dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float pv = __builtin_amdgcn_exp2f(s[r] * sc2 - lv[e]);  // lse2 = +inf for rows >= len -> 0
                    if (need_mask) {
                        const int qrow = q0 + rl + e;
                        if (my_k >= len || (CAUSAL && my_k > qrow)) pv = 0.f;
                    }
                    p[r] = pv;
                    s[r] = pv * (dp[r] - dv4[e]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const s16x8 pf = pack_frag(p, s2);
                const s16x8 dsf = pack_frag(s, s2);
#pragma unroll
                for (int db = 0; db < G::DB; ++db) {
                    dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag(dOt, 32 * db, 2 * qs + s2, lane), pf, dvacc[db], 0, 0, 0);
                    dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag(Qt, 32 * db, 2 * qs + s2, lane), dsf, dkacc[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) commit();
        __syncthreads();
    }

    if (k_valid) {
        bf16_t* dkp = dk + (int64_t)(tok0 + my_k) * dkv_ts + (int64_t)hk * D;
        bf16_t* dvp = dv + (int64_t)(tok0 + my_k) * dkv_ts + (int64_t)hk * D;
#pragma unroll
        for (int db = 0; db < G::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 a, b;
                a.x = pack2bf(dkacc[db][4 * g + 0] * scale, dkacc[db][4 * g + 1] * scale);
                a.y = pack2bf(dkacc[db][4 * g + 2] * scale, dkacc[db][4 * g + 3] * scale);
                b.x = pack2bf(dvacc[db][4 * g + 0], dvacc[db][4 * g + 1]);
                b.y = pack2bf(dvacc[db][4 * g + 2], dvacc[db][4 * g + 3]);
                st8(dkp + 32 * db + 8 * g + 4 * (lane >> 5), a);
                st8(dvp + 32 * db + 8 * g + 4 * (lane >> 5), b);
            }
    }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" int ie_flash_attn_bwd(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v,
                                 int64_t kv_ts, const void* out, int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_ts,
                                 void* dk, void* dv, int64_t dkv_ts, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen,
                                 int hq, int hkv, int d, float softmax_scale, int causal, void* stream) {
    IE_CHECK_ARG(dout && q && k && v && out && lse && delta && dq && dk && dv && cu_seqlens, "ie_flash_attn_bwd: null pointer");
    IE_CHECK_ARG(nseq >= 0 && T >= 0 && max_seqlen >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0, "ie_flash_attn_bwd: bad shape");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_flash_attn_bwd: head dim must be 64 or 128");
    IE_CHECK_SUPPORTED(aligned16(dout) && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && aligned16(dq) &&
                           aligned16(dk) && aligned16(dv) && do_ts % 8 == 0 && q_ts % 8 == 0 && kv_ts % 8 == 0 && o_ts % 8 == 0 &&
                           dq_ts % 4 == 0 && dkv_ts % 4 == 0,
                       "ie_flash_attn_bwd: pointers must be 16-byte aligned and token strides multiples of 8");
    if (nseq == 0 || T == 0 || max_seqlen == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        const int64_t threads = T * hq * (d / 8);
        dim3 grid((unsigned)((threads + 255) / 256));
        if (d == 128)
            hipLaunchKernelGGL((flash_delta_k<128>), grid, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)out, o_ts, delta, T, hq);
        else
            hipLaunchKernelGGL((flash_delta_k<64>), grid, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)out, o_ts, delta, T, hq);
    }
    const unsigned nt128 = (unsigned)((max_seqlen + 127) / 128);
    dim3 gq(nt128, (unsigned)hq, (unsigned)nseq);
    dim3 gk(nt128, (unsigned)hkv, (unsigned)nseq);
#define IE_L(DD, CA)                                                                                                               \
    do {                                                                                                                           \
        hipLaunchKernelGGL((flash_dq_k<DD, CA>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts,           \
                           (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv,        \
                           softmax_scale);                                                                                          \
        hipLaunchKernelGGL((flash_dkdv_k<DD, CA>), gk, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts,         \
                           (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dk, (bf16_t*)dv, dkv_ts, cu_seqlens, T,   \
                           hq, hkv, softmax_scale);                                                                                 \
    } while (0)
    if (d == 128) { if (causal) IE_L(128, true); else IE_L(128, false); }
    else          { if (causal) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
    return ie_launch_status("ie_flash_attn_bwd launch");
}
