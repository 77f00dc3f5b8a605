// K1 backward: varlen, causal / full, GQA flash attention for gfx950.  Deterministic (no atomics):
//   1. flash_delta_k : delta[h][t] = sum_d dO * O
//   2. flash_dq_k    : one block per 128 query rows, loops over K/V tiles, recomputes P, dQ += dS K
//   3. flash_dkdv_k  : one block per 64 keys (one wave per 32 keys), loops over the q heads of the
//                      GQA group and their query tiles, recomputes P, dV += P^T dO, dK += dS^T Q
// The recompute costs 7 instead of 5 tile products but keeps every output written by exactly one
// workgroup in a fixed order, so the result is bit-reproducible run to run.
//
// Tiles arrive by LDS-DMA into double-buffered natural-layout images (flash_common.h); one image serves
// both the row fragments (ds_read_b128) and the transposed fragments (ds_read_b64_tr_b16), so the dQ kernel
// stages only K and V, the dK/dV kernel only Q and dO.  In the dQ kernel scores are formed transposed
// (lane = query row), in the dK/dV kernel un-transposed (lane = key), so the accumulator registers of
// S / dS are directly the k-slots of the next MFMA.  Tile loops are unrolled by the two pipeline stages:
// every LDS address is a precomputed per-lane offset + an immediate.
#include "flash_common.h"

#include <type_traits>

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
template <int D, bool CAUSAL, int MINW>
__global__ __launch_bounds__(256, MINW) void flash_dq_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                        int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                        int64_t kv_ts, const float* __restrict__ lse, const float* __restrict__ delta,
                                                        bf16_t* __restrict__ dq, int64_t dq_ts, const int32_t* __restrict__ cu, int64_t T,
                                                        int hq, int hkv, float scale) {
    using G = Geo<D>;
    constexpr int STAGE = 2 * G::IMG_BYTES;  // K image, V image
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int seq = blockIdx.z;
    // dispatch order = x fastest: ALL heads of the heaviest (last) query tile first, then the next tile ... -- a global
    // longest-job-first order over the causal work (head-major order started the last heads' heavy tiles half-way through the
    // kernel and left most SIMD slots idle in the tail); neighbouring blocks are the q heads of one kv head -> shared K/V in L2
    // consecutive workgroups go to consecutive XCDs: x -> head so that the q heads of one kv head (x = hk, hk + hkv, ...) share an XCD's L2
    const int h = ((int)blockIdx.x % hkv) * (hq / hkv) + (int)blockIdx.x / hkv;
    const int qt = gridDim.y - 1 - blockIdx.y;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 128;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw0 = q0 + wave * 32;
    const int my_q = qw0 + (lane & 31);
    const bool q_valid = my_q < len;

    const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
    const int ntiles = (kv_end + 63) / 64;
    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    TileSrc<D, 4> ksrc, vsrc;
    ksrc.init(kbase, kv_ts, T - tok0, D, wave, lane);
    vsrc.init(vbase, kv_ts, T - tok0, D, wave, lane);
    ksrc.issue(smem, 0, 0, wave);
    vsrc.issue(smem + G::IMG_BYTES, 0, 0, wave);

    FragOffs<D> fo;
    fo.init(lane);

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

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto stage_c, int t) {
        constexpr int S = decltype(stage_c)::value;
        const int kv0 = t * 64;
        const unsigned char* Ks = smem + S * STAGE;
        const unsigned char* Vs = Ks + G::IMG_BYTES;
        if (t + 1 < ntiles) {
            unsigned char* nxt = smem + (1 - S) * STAGE;
            ksrc.issue(nxt, kv0 + 64, 0, wave);
            vsrc.issue(nxt + G::IMG_BYTES, kv0 + 64, 0, wave);
        }
        const bool active = !CAUSAL || kv0 <= qw0 + 31;
        if (active) {
            const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > len);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Vs, 32 * c, ks, fo), dof[ks], dp, 0, 0, 0);
                }
                if (need_mask) {  // wave-uniform; selects instead of per-element branches
                    const int lim = CAUSAL ? min(len - 1, my_q) : len - 1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 32 * c + creg_row(r, lane);
                        const float p = key > lim ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lse2));
                        s[r] = p * (dp[r] - dlt);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lse2)) * (dp[r] - dlt);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x8 dsf = pack_frag(s, s2);
#pragma unroll
                    for (int db = 0; db < G::DB; ++db)
                        dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag<D>(Ks, db, 2 * c + s2, fo), dsf, dqacc[db], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    for (int t = 0; t < ntiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
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
// Block = DKV_WAVES waves x 32 keys.  With causal masking the work of a key block falls linearly with its
// position, and there are only (len/64) x hkv blocks, so (a) blocks are small (64 keys) to get >= 2 per CU and
// (b) the block index is folded so that the two blocks dispatched far apart (the ones that share a CU under
// round-robin dispatch) are a heavy and a light one: kb(j) = j for the first half, mirrored for the second.

// HS > 1 (causal balance): the q heads of a kv head are split over HS blocks, each writing fp32 partial dK / dV
// to part[2][HS][T][hkv][D]; flash_dkdv_reduce_k sums them in a fixed order (deterministic, no atomics).  With all blocks
// resident at once the kernel takes as long as its heaviest block (key block 0 sees every query tile: 2x the mean); HS x more,
// HS x smaller blocks dispatched heavy-first let the light ones back-fill the tail.
template <int D, bool CAUSAL, int HS, int DKV_WAVES>
__global__ __launch_bounds__(64 * DKV_WAVES) void flash_dkdv_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                               int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               int64_t kv_ts, const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dkv_ts,
                                                               const int32_t* __restrict__ cu, int64_t T, int hq, int hkv, float scale,
                                                               float* __restrict__ part) {
    using G = Geo<D>;
    constexpr int STAGE = 2 * G::IMG_BYTES + 1024;  // Q image, dO image, lse2[64], delta[64] (+ pad to keep 1 KiB alignment)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int seq = blockIdx.z;
    // x = (key block, head split, kv head), kv head fastest: every (split, head) of the heaviest key block is dispatched first
    const int hk = blockIdx.x % hkv;
    const int hsi = (blockIdx.x / hkv) % HS;
    const int nkb = gridDim.x / (hkv * HS);
    const int j = blockIdx.x / (hkv * HS);
    const int first = (nkb + 1) / 2;
    const int kb = HS > 1 ? j : (j < first ? j : nkb - 1 - (j - first));  // HS > 1: plain order = heavy blocks first
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int k0 = kb * 32 * DKV_WAVES;
    if (k0 >= len) return;
    const int grp = hq / hkv / HS;          // q heads handled by this block
    const int h_first = hk * (hq / hkv) + hsi * grp;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kw0 = k0 + wave * 32;
    const int my_k = kw0 + (lane & 31);
    const bool k_valid = my_k < len;

    const int nqt_all = (len + 63) / 64;
    const int qt_start = CAUSAL ? (k0 / 64) : 0;
    const int nqt = nqt_all - qt_start;  // >= 1 because k0 < len
    const int nit = grp * nqt;

    TileSrc<D, DKV_WAVES> qsrc, dosrc;
    qsrc.init(q + (int64_t)tok0 * q_ts, q_ts, T - tok0, (int64_t)hq * D, wave, lane);
    dosrc.init(dout + (int64_t)tok0 * do_ts, do_ts, T - tok0, (int64_t)hq * D, wave, lane);
    // stage `it`: Q / dO images by DMA; lse (log2 domain) and delta through registers of threads 0..63
    auto issue = [&](int it, unsigned char* stage, float& lse_r, float& dlt_r) {
        const int h = h_first + it / nqt;
        const int q0 = (qt_start + it % nqt) * 64;
        const int rem = len - q0;
        qsrc.issue(stage, q0, h * D, wave);
        dosrc.issue(stage + G::IMG_BYTES, q0, h * D, wave);
        if (threadIdx.x < 64) {
            const bool ok = (int)threadIdx.x < rem;
            lse_r = ok ? lse[(int64_t)h * T + tok0 + q0 + threadIdx.x] * kLog2e : INFINITY;
            dlt_r = ok ? delta[(int64_t)h * T + tok0 + q0 + threadIdx.x] : 0.f;
        }
    };
    auto commit = [&](unsigned char* stage, float lse_r, float dlt_r) {
        if (threadIdx.x < 64) {
            reinterpret_cast<float*>(stage + 2 * G::IMG_BYTES)[threadIdx.x] = lse_r;
            reinterpret_cast<float*>(stage + 2 * G::IMG_BYTES + 256)[threadIdx.x] = dlt_r;
        }
    };

    float lse_r = 0.f, dlt_r = 0.f;
    issue(0, smem, lse_r, dlt_r);
    commit(smem, lse_r, dlt_r);

    FragOffs<D> fo;
    fo.init(lane);

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
    const int rl_lane = 4 * (lane >> 5);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // One query tile (64 rows = two 32-row sub-blocks) of one q head against this wave's 32 keys.  The serial chain
    //   S / dP MFMAs -> softmax -> packing -> dV / dK MFMAs
    // of the two sub-blocks is interleaved (ablations: the phases, not the LDS feed, cost the time, profiles/r02_flash_dkdv_ablation.jsonl):
    //   A0: S0 / dP0 MFMAs            | row fragments fetched two k-steps ahead
    //   A1: S1 / dP1 MFMAs            | softmax of sub-block 0: its 16 elements (fma, exp2, sub, mul, half a pack each) spread over the MFMA gaps
    //   C0: dV / dK MFMAs of block 0  | softmax of sub-block 1 likewise; transposed fragments fetched two MFMAs ahead
    //   C1: dV / dK MFMAs of block 1
    // MFMAs are asm with explicit register files (flash_common.h): dK / dV accumulators and the K / V fragments live in AGPRs, the scores in
    // arch VGPRs.  One wave hides 5-6 vector instructions under a 32-cycle MFMA (tools/probes/mfma_shadow.hip): one element per gap fits at
    // D = 128.  Every gap is pinned by sched_barrier.  A sub-block that is masked entirely or lies behind the sequence is computed like the
    // others and contributes zeros (the mask; lse = +inf behind the sequence).
    auto step = [&](auto stage_c, int it) {
        constexpr int S = decltype(stage_c)::value;
        const int q0 = (qt_start + it % nqt) * 64;
        // wave-uniform: does this tile need the mask (the diagonal tile, the last keys)?  Unmasked tiles compare against a first row that
        // no row is below, so the two extra vector instructions per element are the same in both cases (one code path: two copies of the
        // tile body -- masked / unmasked -- made hipcc spill 91 VGPRs)
        const bool need_mask = (CAUSAL && kw0 + 31 > q0) || (kw0 + 32 > len);
        const unsigned char* stage = smem + S * STAGE;
        unsigned char* nxt = smem + (1 - S) * STAGE;
        const bool more = it + 1 < nit;
        if (more) issue(it + 1, nxt, lse_r, dlt_r);
        const unsigned char* Qs = stage;
        const unsigned char* dOs = stage + G::IMG_BYTES;
        const float* lse_s = reinterpret_cast<const float*>(stage + 2 * G::IMG_BYTES);
        const float* dlt_s = lse_s + 64;
        const int first_q = (my_k >= len) ? 0x3fffffff : (CAUSAL ? my_k : 0);  // first query row that sees this lane's key
        const int first_rel = need_mask ? first_q - q0 - rl_lane : (int)0x80000000;   // the same in tile-local rows of this lane's registers 0 .. 3
        constexpr int KS = G::KS, NTR = 2 * G::DB;      // MFMAs per accumulator in an A phase; (dV, dK) MFMA pairs in a C phase
        constexpr int EA = 16 / (2 * KS) > 0 ? 16 / (2 * KS) : 1, EC = 16 / (2 * NTR) > 0 ? 16 / (2 * NTR) : 1;   // softmax elements per MFMA gap
        constexpr int R = 3;                            // row fragments in flight (k-steps ahead + 1)
        s16x8 rq[R], rdo[R];
        auto fetch_rows = [&](int u) {                  // u = 0 .. 2 KS - 1: k-step u % KS of sub-block u / KS
            rq[u % R] = row_frag<D>(Qs, 32 * (u / KS), u % KS, fo);
            rdo[u % R] = row_frag<D>(dOs, 32 * (u / KS), u % KS, fo);
        };
        f32x16 s0, dp0, s1, dp1;
        s16x8 pf0[2], dsf0[2], pf1[2], dsf1[2];
        float4 l4[2], d4[2];                           // lse / delta of register groups gi, gi + 1 (gi = 4 qs + r / 4): fetched one group ahead
        auto fetch_ld = [&](int gi) {
            const int rl = 32 * (gi / 4) + 8 * (gi % 4) + rl_lane;   // tile-local row of registers 4 (r / 4) .. + 3
            l4[gi % 2] = *reinterpret_cast<const float4*>(lse_s + rl);
            d4[gi % 2] = *reinterpret_cast<const float4*>(dlt_s + rl);
        };
        float e_prev = 0.f;
        ie_f32x2 t_prev = {0.f, 0.f};
        // one element of a sub-block's softmax (register r of S / dP); behind the odd element of a pair the bf16 pack of the pair
        auto elem = [&](const f32x16& sv, const f32x16& dpv, s16x8 (&pf)[2], s16x8 (&dsf)[2], int qs, int r) {
            const int gi = 4 * qs + r / 4;
            if (r % 4 == 0 && gi + 1 < 8) fetch_ld(gi + 1);
            const float4 lq = l4[gi % 2], dq = d4[gi % 2];
            const float lv = r % 4 == 0 ? lq.x : r % 4 == 1 ? lq.y : r % 4 == 2 ? lq.z : lq.w;
            float e = __builtin_amdgcn_exp2f(fms_pinned(sv[r], sc2, lv));   // lse2 = +inf for rows >= len -> 0
            e = (32 * qs + 8 * (r / 4) + r % 4) < first_rel ? 0.f : e;
            if (r % 2 == 0) {   // dP - delta of the pair (r, r + 1) as one packed subtract, held HERE (hipcc otherwise hoists it up to the lse / delta load and waits there)
                e_prev = e;
                ie_f32x2 dl;
                dl.x = r % 4 == 0 ? dq.x : dq.z;
                dl.y = r % 4 == 0 ? dq.y : dq.w;
                asm volatile("" : "+v"(dl));
                t_prev.x = dpv[r] - dl.x;
                t_prev.y = dpv[r + 1] - dl.y;
            } else {   // registers r - 1, r -> 32-bit word (r / 2) % 4 of fragment r / 8
                const float ds_prev = e_prev * t_prev.x, ds = e * t_prev.y;
                union { uint4 u; s16x8 x; } a, b;
                a.x = pf[r / 8];
                b.x = dsf[r / 8];
                const unsigned pp = pack2bf(e_prev, e), dd = pack2bf(ds_prev, ds);
                const int w = (r / 2) % 4;
                if (w == 0) { a.u.x = pp; b.u.x = dd; } else if (w == 1) { a.u.y = pp; b.u.y = dd; } else if (w == 2) { a.u.z = pp; b.u.z = dd; } else { a.u.w = pp; b.u.w = dd; }
                pf[r / 8] = a.x;
                dsf[r / 8] = b.x;
            }
        };
        s16x8 tfa[2], tfb[2];                           // transposed fragments in flight: [m % 2] for dV (dO^T) and dK (Q^T)
        auto fetch_tr = [&](int m, int qs) {            // m = 0 .. NTR - 1: (s2, db) = (m / DB, m % DB) of sub-block qs
            tfa[m % 2] = trans_frag<D>(dOs, m % G::DB, 2 * qs + m / G::DB, fo);
            tfb[m % 2] = trans_frag<D>(Qs, m % G::DB, 2 * qs + m / G::DB, fo);
        };
        fetch_rows(0);
        fetch_rows(1);
        fetch_ld(0);
        __builtin_amdgcn_sched_barrier(0);
        // A0 (the fragments of sub-block 1's first k-steps are requested under its last MFMAs)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            fetch_rows(ks + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) mfma_s_first(s0, rq[ks % R], kf[ks]);
            else mfma_s(s0, rq[ks % R], kf[ks]);
            if (ks == 0) mfma_s_first(dp0, rdo[ks % R], vf[ks]);
            else mfma_s(dp0, rdo[ks % R], vf[ks]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // A1 | softmax of sub-block 0
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = KS + ks;
            if (ks + 2 < KS) fetch_rows(u + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) mfma_s_first(s1, rq[u % R], kf[ks]);
            else mfma_s(s1, rq[u % R], kf[ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) asm volatile("s_nop 11" : "+v"(s0), "+v"(dp0));   // S0 / dP0: 12 wait states behind their last MFMA before the vector ALU reads them
#pragma unroll
            for (int x = 0; x < EA; ++x) elem(s0, dp0, pf0, dsf0, 0, 2 * EA * ks + x);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) mfma_s_first(dp1, rdo[u % R], vf[ks]);
            else mfma_s(dp1, rdo[u % R], vf[ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < EA; ++x) elem(s0, dp0, pf0, dsf0, 0, 2 * EA * ks + EA + x);
            if (ks == KS - 2) fetch_tr(0, 0);
            if (ks == KS - 1) fetch_tr(1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // C0 | softmax of sub-block 1
#pragma unroll
        for (int m = 0; m < NTR; ++m) {
            mfma_o(dvacc[m % G::DB], tfa[m % 2], pf0[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0) asm volatile("s_nop 11" : "+v"(s1), "+v"(dp1));
#pragma unroll
            for (int x = 0; x < EC; ++x) elem(s1, dp1, pf1, dsf1, 1, 2 * EC * m + x);
            __builtin_amdgcn_sched_barrier(0);
            mfma_o(dkacc[m % G::DB], tfb[m % 2], dsf0[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < EC; ++x) elem(s1, dp1, pf1, dsf1, 1, 2 * EC * m + EC + x);
            if (m + 2 < NTR) fetch_tr(m + 2, 0);
            else fetch_tr(m + 2 - NTR, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // C1
#pragma unroll
        for (int m = 0; m < NTR; ++m) {
            mfma_o(dvacc[m % G::DB], tfa[m % 2], pf1[m / G::DB]);
            mfma_o(dkacc[m % G::DB], tfb[m % 2], dsf1[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
            if (m + 2 < NTR) fetch_tr(m + 2, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) commit(nxt, lse_r, dlt_r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto pin = [&]() {
#pragma unroll
        for (int db = 0; db < G::DB; ++db) pin_acc(dkacc[db], dvacc[db]);
    };
    for (int it = 0; it < nit; it += 2) {
        pin();
        step(std::integral_constant<int, 0>{}, it);
        pin();
        if (it + 1 < nit) step(std::integral_constant<int, 1>{}, it + 1);
        pin();
    }
#pragma unroll
    for (int db = 0; db < G::DB; ++db) mfma_settle_acc(dkacc[db], dvacc[db]);

    if (k_valid && HS == 1) {
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
    if (k_valid && HS > 1) {
        const int64_t row = (((int64_t)hsi * T + tok0 + my_k) * hkv + hk) * D;
        float* pk = part + row;
        float* pv = part + (int64_t)HS * T * hkv * D + row;
#pragma unroll
        for (int db = 0; db < G::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * db + 8 * g + 4 * (lane >> 5);
                *reinterpret_cast<float4*>(pk + c) = make_float4(dkacc[db][4 * g + 0], dkacc[db][4 * g + 1], dkacc[db][4 * g + 2], dkacc[db][4 * g + 3]);
                *reinterpret_cast<float4*>(pv + c) = make_float4(dvacc[db][4 * g + 0], dvacc[db][4 * g + 1], dvacc[db][4 * g + 2], dvacc[db][4 * g + 3]);
            }
    }
}

// dk = bf16(scale * sum_hs part_k), dv = bf16(sum_hs part_v); one thread per 4 consecutive d of a (token, kv head)
template <int HS>
__global__ __launch_bounds__(256) void flash_dkdv_reduce_k(const float* __restrict__ part, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                           int64_t dkv_ts, int64_t T, int hkv, int D, float scale,
                                                           const int32_t* __restrict__ cu, int nseq) {
    const int64_t n4 = T * hkv * (D / 4);
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % (D / 4));
    const int64_t th = i / (D / 4);  // tok * hkv + hk
    const int hk = (int)(th % hkv);
    const int64_t tok = th / hkv;
    if (tok >= cu[nseq]) return;  // tokens behind the last sequence were written by no block (and are left alone, as without the split)
    const int64_t stride = T * hkv * (int64_t)D;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
    for (int hs = 0; hs < HS; ++hs) {
        const float4 x = *reinterpret_cast<const float4*>(part + hs * stride + th * D + c4 * 4);
        const float4 y = *reinterpret_cast<const float4*>(part + (HS + hs) * stride + th * D + c4 * 4);
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
    }
    uint2 ka, vb;
    ka.x = pack2bf(a.x * scale, a.y * scale);
    ka.y = pack2bf(a.z * scale, a.w * scale);
    vb.x = pack2bf(b.x, b.y);
    vb.y = pack2bf(b.z, b.w);
    st8(dk + tok * dkv_ts + (int64_t)hk * D + c4 * 4, ka);
    st8(dv + tok * dkv_ts + (int64_t)hk * D + c4 * 4, vb);
}


inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

int g_dkdv_split = 0;  // 0 = automatic (see dkdv_split), else the forced head split of the dK/dV kernel
int g_dkdv_waves = 2;  // waves (x 32 keys) per dK/dV block
int g_dq_minw = 2;  // waves/SIMD the dQ kernel is compiled for (2: 256 VGPRs with a small spill; 1: no spill, half the occupancy)

}  // namespace

// tuning hook (A/B benchmarking only): occupancy target of the dQ kernel, 1 or 2 waves per SIMD
extern "C" int ie_tune_flash_dq_occupancy(int waves_per_simd) {
    IE_CHECK_ARG(waves_per_simd == 1 || waves_per_simd == 2, "ie_tune_flash_dq_occupancy: 1 or 2");
    g_dq_minw = waves_per_simd;
    return IE_OK;
}

// head split of the dK/dV kernel: causal attention only (the imbalance it removes is the causal one); the smallest split that
// gives about four blocks per CU (256 CUs), as far as it divides the GQA group, at most 4
static int dkdv_split(int hq, int hkv, int causal, int64_t blocks_unsplit) {
    const int grp = hq / hkv;
    int hs = 1;
    if (g_dkdv_split > 0) {
        hs = g_dkdv_split;
        while (hs > 1 && grp % hs != 0) hs >>= 1;
        return hs;
    }
    if (!causal) return 1;
    while (hs < 4 && blocks_unsplit * hs < 1024 && grp % (hs * 2) == 0) hs *= 2;
    return hs;
}

extern "C" int ie_tune_flash_dkdv_split(int split) {
    IE_CHECK_ARG(split == 0 || split == 1 || split == 2 || split == 4, "ie_tune_flash_dkdv_split: 0 (auto), 1, 2 or 4");
    g_dkdv_split = split;
    return IE_OK;
}

extern "C" int64_t ie_flash_attn_bwd_workspace(int64_t T, int hq, int hkv, int d) {
    if (T < 0 || hq <= 0 || hkv <= 0 || d <= 0) return -1;
    return (int64_t)hq * T + 2ll * 4 * T * hkv * d;  // delta + the largest set of dK/dV partials
}

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
    dim3 gq((unsigned)hq, nt128, (unsigned)nseq);
    const int DKV_WAVES = g_dkdv_waves;
    const unsigned nkb = (unsigned)((max_seqlen + 32 * DKV_WAVES - 1) / (32 * DKV_WAVES));
    const int hs = dkdv_split(hq, hkv, causal, (int64_t)nkb * hkv * nseq);
    dim3 gk(nkb * (unsigned)hkv * (unsigned)hs, 1, (unsigned)nseq);
    float* part = delta + (int64_t)hq * T;  // workspace layout: delta[hq*T] | partial dK [hs][T][hkv][d] | partial dV [hs][T][hkv][d]
#define IE_DKDV_W(DD, CA, HS_, NW_)                                                                                                \
    hipLaunchKernelGGL((flash_dkdv_k<DD, CA, HS_, NW_>), gk, dim3(64 * NW_), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, \
                       (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dk, (bf16_t*)dv, dkv_ts, cu_seqlens, T, hq, hkv, \
                       softmax_scale, part)
#define IE_DKDV_HS(DD, CA, HS_)                                                                                                    \
    do {                                                                                                                           \
        if (DKV_WAVES == 4) IE_DKDV_W(DD, CA, HS_, 4);                                                                             \
        else IE_DKDV_W(DD, CA, HS_, 2);                                                                                            \
        if (HS_ > 1) {                                                                                                             \
            const int64_t n4 = T * hkv * (DD / 4);                                                                                 \
            hipLaunchKernelGGL((flash_dkdv_reduce_k<HS_>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, (bf16_t*)dk,   \
                               (bf16_t*)dv, dkv_ts, T, hkv, DD, softmax_scale, cu_seqlens, nseq);                                   \
        }                                                                                                                          \
    } while (0)
#define IE_DKDV(DD, CA)                                                                                                            \
    do {                                                                                                                           \
        if (hs == 4) IE_DKDV_HS(DD, CA, 4);                                                                                        \
        else if (hs == 2) IE_DKDV_HS(DD, CA, 2);                                                                                   \
        else IE_DKDV_HS(DD, CA, 1);                                                                                                \
    } while (0)
#define IE_L(DD, CA)                                                                                                               \
    do {                                                                                                                           \
        if (g_dq_minw == 2)                                                                                                        \
            hipLaunchKernelGGL((flash_dq_k<DD, CA, 2>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts,    \
                               (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv,    \
                               softmax_scale);                                                                                      \
        else                                                                                                                       \
            hipLaunchKernelGGL((flash_dq_k<DD, CA, 1>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts,    \
                               (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv,    \
                               softmax_scale);                                                                                      \
        IE_DKDV(DD, CA);                                                                                                           \
    } while (0)
    if (d == 128) { if (causal) IE_L(128, true); else IE_L(128, false); }
    else          { if (causal) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
#undef IE_DKDV
#undef IE_DKDV_HS
#undef IE_DKDV_W
    return ie_launch_status("ie_flash_attn_bwd launch");
}

// tuning hook (A/B benchmarking only): bits 0-1: 0 = 2 waves x 32 keys per dK/dV block, 1 = 4 waves
extern "C" int ie_tune_flash_bwd_variant(int variant) {
    IE_CHECK_ARG(variant >= 0 && variant <= 1, "ie_tune_flash_bwd_variant: 0 or 1");
    g_dkdv_waves = (variant & 1) ? 4 : 2;
    return IE_OK;
}
