// K1 forward: varlen, causal / full, GQA flash attention for gfx950.
//
// Grid (q tiles of 128 rows [heaviest first], q heads, sequences); block = 4 wave64, each wave owns
// 32 query rows.  K/V tiles of 64 keys arrive by LDS-DMA into double-buffered natural-layout images
// (flash_common.h): the next tile is in flight under the current tile's MFMAs, one barrier per tile,
// no staging registers, no ds_write.  Scores are formed as S^T = K Q^T so every lane owns ONE query
// row: the online-softmax max/sum are in-lane reductions plus a single lane^32 exchange, and the
// rescale of O^T (also lane-per-row) is a plain multiply, skipped when no row maximum moved.
// P feeds the PV MFMA straight from the score accumulators; V^T fragments come from the natural V image
// through ds_read_b64_tr_b16.  The tile loop is unrolled by the two pipeline stages so every LDS address
// is a precomputed per-lane offset + an immediate (no address arithmetic in the loop).
//
// Roofline: MFMA-bound; causal flop = 4*D*sum_s(len_s^2)/2 per head; algorithmic bytes = q+k+v+o.
#include "flash_common.h"

#include <type_traits>

namespace {

using namespace fa;

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void flash_fwd_k(const bf16_t* __restrict__ q, int64_t q_ts, const bf16_t* __restrict__ k,
                                                      const bf16_t* __restrict__ v, int64_t kv_ts, bf16_t* __restrict__ out,
                                                      int64_t o_ts, float* __restrict__ lse, const int32_t* __restrict__ cu, int64_t T,
                                                      int hq, int hkv, float scale, const int32_t* __restrict__ cu_k, int64_t Tk) {
    // cu_k (full attention only; ie_flash_attn_fwd_x): the keys of sequence s are rows cu_k[s] .. cu_k[s + 1] of a K / V tensor of Tk rows -- a
    // rectangle of scores per sequence (ring attention: a rank's queries against another rank's keys); NULL: the queries' own rows
    using G = Geo<D>;
    constexpr int STAGE = 2 * G::IMG_BYTES;                                       // K image, V image
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
    const int qw0 = q0 + wave * 32;            // first query row of this wave
    const int my_q = qw0 + (lane & 31);        // this lane's query row (within the sequence)
    const bool q_valid = my_q < len;

    const int tok0k = (!CAUSAL && cu_k) ? cu_k[seq] : tok0;
    const int lenk = (!CAUSAL && cu_k) ? cu_k[seq + 1] - tok0k : len;
    const int64_t Tkk = (!CAUSAL && cu_k) ? Tk : T;
    const int kv_end = CAUSAL ? min(len, q0 + 128) : lenk;
    const int ntiles = (kv_end + 63) / 64;
    const bf16_t* kbase = k + (int64_t)tok0k * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0k * kv_ts + (int64_t)hk * D;

    TileSrc<D, 4> ksrc, vsrc;
    ksrc.init(kbase, kv_ts, Tkk - tok0k, D, wave, lane);
    vsrc.init(vbase, kv_ts, Tkk - tok0k, D, wave, lane);
    if (ntiles > 0) {   // (a sequence without keys -- ie_flash_attn_fwd_x only -- requests nothing: out = 0, lse = -inf)
        ksrc.issue(smem, 0, 0, wave);
        vsrc.issue(smem + G::IMG_BYTES, 0, 0, wave);
    }

    FragOffs<D> fo;
    fo.init(lane);

    // Q fragments (B operand of S^T = K Q^T), loaded while the first tile is in flight
    s16x8 qf[G::KS];
    {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } cv;
            cv.u = q_valid ? ld16(qp + ks * 16) : z4();
            qf[ks] = cv.s;
        }
    }

    f32x16 oacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) oacc[db] = zero16();
    float m_run = -INFINITY, l_run = 0.f;     // running max of the RAW scores, running sum
    const float sc2 = scale * kLog2e;         // exp(scale * s) = exp2(sc2 * s)

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
        // wave-uniform: does this wave see any unmasked key in this tile?
        const bool active = !CAUSAL || kv0 <= qw0 + 31;
        if (active) {
            f32x16 sacc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sacc[c] = zero16();
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks)
                    sacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], sacc[c], 0, 0, 0);
            }
            const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > lenk);
            float mx = -INFINITY;
            if (need_mask) {  // wave-uniform: only the diagonal / last tiles pay for the masks (selects, no per-element branches)
                const int lim = CAUSAL ? min(len - 1, my_q) : lenk - 1;  // largest visible key of this lane's query
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 32 * c + creg_row(r, lane);
                        sacc[c][r] = key > lim ? -INFINITY : sacc[c][r];
                    }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[c][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * sc2);  // m_run = -inf -> 0
            const float moff = m_use * sc2;
            float rs = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[c][r], sc2, -moff));
                    sacc[c][r] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (!__all(alpha == 1.f)) {  // wave-uniform: some row's running max moved
#pragma unroll
                for (int db = 0; db < G::DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            }
            // O^T += V^T P^T
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x8 pf = pack_frag(sacc[c], s2);
#pragma unroll
                    for (int db = 0; db < G::DB; ++db)
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag<D>(Vs, db, 2 * c + s2, fo), pf, oacc[db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile landed (this wave's share)
        __syncthreads();
    };

    for (int t = 0; t < ntiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    }

    {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        store_row_block<D>(out + (int64_t)(tok0 + my_q) * o_ts + (int64_t)h * D, oacc, inv, lane, q_valid, (o_ts & 7) == 0);
        // lse of the SCALED scores in natural log: scale * m + ln(l)
        if (q_valid && lane < 32) lse[(int64_t)h * T + tok0 + my_q] = (l_run > 0.f) ? m_run * scale + logf(l_run) : -INFINITY;
    }
}



// acc *= f, element-wise, with the accumulator staying in the accumulation file (a vector-ALU use of it in HIP source would make
// hipcc keep it in arch VGPRs and copy all of it into the accumulation file in front of every MFMA statement)
__device__ __forceinline__ void acc_scale(f32x16& acc, float f) {
    asm volatile("s_nop 15" : "+a"(acc));  // the last MFMA into acc has retired
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float e = acc[r], t;
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(e), "=&v"(t) : "v"(f));
        acc[r] = e;
    }
    asm volatile("s_nop 3" : "+a"(acc));  // accumulation-file write -> MFMA source
}

// ------------------------------------------------------------------------------------------------
// K1 forward, one wave per SIMD: block = 4 wave64 = 256 query rows, each wave owns 64 rows (two 32-row blocks X = 0, 1)
// and the whole 512-entry register file of its SIMD.  Every K fragment read from LDS feeds TWO MFMAs (both row blocks)
// and so does every V^T fragment: half the LDS bytes per MFMA of the 32-rows-per-wave kernel above, which is what that
// kernel waits for (one wave per SIMD cannot hide an LDS round trip behind a partner wave).
// The tile loop is software-pipelined over two phases of 8*KS/2.. MFMAs each so that the matrix pipe and the vector ALU
// always have independent work in the same basic block:
//   phase 1 of tile t:  S(t+1) = K(t+1) Q^T  (MFMA)   beside   finish-softmax(t): p = exp2(S(t) - m), row sums, bf16 pack
//   phase 2 of tile t:  O += V(t)^T P(t)     (MFMA)   beside   start-softmax(t+1): row maxima of S(t+1), new m, alpha
// (the O rescale by alpha(t), when some row maximum moved by more than the deferral threshold, opens phase 1).
// S is double-buffered in registers (the loop is unrolled by two, which also makes every LDS stage address an immediate).
// K(t+2) and V(t+1) are requested by LDS-DMA at the top of tile t: each has a whole tile of MFMAs to land; one barrier per tile.
template <int D, bool CAUSAL, int THR>
__global__ __launch_bounds__(256, 1) void flash_fwd64_k(const bf16_t* __restrict__ q, int64_t q_ts, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, int64_t kv_ts, bf16_t* __restrict__ out,
                                                        int64_t o_ts, float* __restrict__ lse, const int32_t* __restrict__ cu, int64_t T,
                                                        int hq, int hkv, float scale) {
    using G = Geo<D>;
    constexpr int IMG = G::IMG_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * IMG];  // K stage 0, K stage 1, V stage 0, V stage 1

    const int seq = blockIdx.z;
    const int h = ((int)blockIdx.x % hkv) * (hq / hkv) + (int)blockIdx.x / hkv;  // the q heads of one kv head share an XCD's L2
    const int qt = gridDim.y - 1 - blockIdx.y;                                    // heaviest (last) query tiles first
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 256;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw0 = q0 + wave * 64;
    int my_q[2];
    bool q_valid[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        my_q[X] = qw0 + 32 * X + (lane & 31);
        q_valid[X] = my_q[X] < len;
    }
    const int kv_end = CAUSAL ? min(len, q0 + 256) : len;
    const int nt = (kv_end + 63) / 64;                                                    // tiles the block streams
    const int ntw = qw0 >= len ? 0 : (CAUSAL ? min(nt, qw0 / 64 + 1) : nt);               // tiles this wave computes on

    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    TileSrc<D, 4> ksrc, vsrc;
    ksrc.init(kbase, kv_ts, T - tok0, D, wave, lane);
    vsrc.init(vbase, kv_ts, T - tok0, D, wave, lane);
    unsigned char* Kst = smem;
    unsigned char* Vst = smem + 2 * IMG;
    ksrc.issue(Kst, 0, 0, wave);
    if (nt > 1) ksrc.issue(Kst + IMG, 64, 0, wave);
    vsrc.issue(Vst, 0, 0, wave);

    FragOffs<D> fo;
    fo.init(lane);

    s16x8 qf[2][G::KS];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q[X]) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } cv;
            cv.u = q_valid[X] ? ld16(qp + ks * 16) : z4();
            qf[X][ks] = cv.s;
        }
    }

    f32x16 oacc[2][G::DB];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int db = 0; db < G::DB; ++db) oacc[X][db] = zero16();
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f}, moff[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f};
    bool resc[2] = {false, false};
    const float sc2 = scale * kLog2e;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 pf[2][2][2];  // P fragments [X][c][s2]: 8 bf16 = keys 32c + 16*s2-step in C/D register order

    // other half of the row: lane ^ 32 (v_permlane32_swap: vector-ALU latency; __shfl_xor would be a ds_bpermute round trip that
    // a lone wave on its SIMD cannot hide)
    auto xhalf_max = [&](float v) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    auto xhalf_sum = [&](float v) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };
    // masks of tile t (wave-uniform decision: diagonal / last tiles only); selects, no per-element branches
    auto apply_mask = [&](f32x16 (&s)[2][2], int t) {
        const int kv0 = t * 64;
        if (!((CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > len))) return;
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int lim = CAUSAL ? min(len - 1, my_q[X]) : len - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * c + creg_row(r, lane);
                    s[X][c][r] = key > lim ? -INFINITY : s[X][c][r];
                }
        }
    };
    // start-softmax given the in-lane maxima of a tile's scores: row maxima, deferral decision, alpha, exponent offset
    auto sm_start = [&](float (&mx)[2]) {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const float m = xhalf_max(mx[X]);
            const float m_new = fmaxf(m_run[X], m);
            // deferred rescale: keep the old reference maximum while no row of the wave moved by more than THR (log2 units):
            // p <= 2^THR then, exact in fp32 / same relative precision in bf16; THR = 0 rescales whenever any maximum moved
            const bool keep = __all((m_new - m_run[X]) * sc2 <= (float)THR);
            const float m_sel = keep ? m_run[X] : m_new;
            const float m_use = (m_sel == -INFINITY) ? 0.f : m_sel;
            alpha[X] = keep ? 1.f : __builtin_amdgcn_exp2f((m_run[X] - m_use) * sc2);  // m_run = -inf -> 0
            resc[X] = !keep;
            moff[X] = m_use * sc2;
            m_run[X] = m_sel;
            l_run[X] *= alpha[X];
        }
    };

    // ---- prologue: S(0), start-softmax(0)
    f32x16 sA[2][2], sB[2][2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ntw > 0) {
        float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const s16x8 kfr = row_frag<D>(Kst, 32 * c, ks, fo);
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    if (ks == 0) mfma_s_first(sA[X][c], kfr, qf[X][ks]);
                    else mfma_s(sA[X][c], kfr, qf[X][ks]);
                }
            }
        mfma_settle(sA[0][0], sA[0][1], sA[1][0], sA[1][1]);
        apply_mask(sA, 0);
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[X] = fmaxf(mx[X], sA[X][c][r]);
        sm_start(mx);
    }

    constexpr int NM = 4 * G::KS;   // MFMAs of a phase (= 8 * DB)
    constexpr int EPC = 64 / NM;    // score elements per lane handled beside one MFMA (2 for D = 128, 4 for D = 64)
    static_assert(8 * G::DB == NM && EPC >= 2, "phase geometry");

    // tile t with its scores in `sc`; the scores of tile t+1 go to `sn`.  PAR = t & 1 selects the LDS stages.
    auto tile = [&](auto par_c, int t, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2]) {
        constexpr int PAR = decltype(par_c)::value;
        // every wave's DMA pieces of K(t+1) and V(t) have landed; every wave is done with K(t) and V(t-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) ksrc.issue(Kst + PAR * IMG, (t + 2) * 64, 0, wave);
        if (t + 1 < nt) vsrc.issue(Vst + (1 - PAR) * IMG, (t + 1) * 64, 0, wave);
        if (t >= ntw) return;
        const unsigned char* Kn = Kst + (1 - PAR) * IMG;
        const unsigned char* Vc = Vst + PAR * IMG;
        const bool has_next = t + 1 < ntw;   // this wave also works on tile t + 1

        // ---- phase 1
#pragma unroll
        for (int X = 0; X < 2; ++X)
            if (resc[X]) {
#pragma unroll
                for (int db = 0; db < G::DB; ++db) acc_scale(oacc[X][db], alpha[X]);
            }
        float rs[2] = {0.f, 0.f};
        const float nmoff[2] = {-moff[0], -moff[1]};
        auto fin_chunk = [&](int i) {  // elements [i*EPC, (i+1)*EPC) of the 64 scores a lane holds: X = e/32, c = (e/16)&1, r = e&15
#pragma unroll
            for (int u = 0; u < EPC; u += 2) {
                const int e = i * EPC + u, X = e >> 5, c = (e >> 4) & 1, r = e & 15;
                const float p0 = __builtin_amdgcn_exp2f(fma_pinned(sc[X][c][r], sc2, nmoff[X]));
                const float p1 = __builtin_amdgcn_exp2f(fma_pinned(sc[X][c][r + 1], sc2, nmoff[X]));
                rs[X] += p0;  // (two dependent adds: the second exp's result is not read by the instruction right behind it)
                rs[X] += p1;
                pf[X][c][r >> 3][(r & 7) >> 1] = pack2bf(p0, p1);
            }
        };
        // Fragments are requested PF MFMA pairs ahead of their use (a lone wave per SIMD sees the whole LDS round trip; the asm
        // statements pin the order of the LDS reads), across the phase boundary too: the first V^T fragments of phase 2 are
        // requested under the last pairs of phase 1.  The vector-ALU chunks run SK ahead of the MFMAs so that the first K
        // fragments' round trip (they cannot be requested before the barrier) has cover.
        constexpr int PF = 3, SK = 2, NK = 2 * G::KS, NV = 4 * G::DB;
        s16x8 kq[NK], vq[NV];
        auto load_v = [&](int j) { vq[j] = trans_frag<D>(Vc, j % G::DB, j / G::DB, fo); };
        if (has_next) {
#pragma unroll
            for (int j = 0; j < PF; ++j) kq[j] = row_frag<D>(Kn, 32 * (j / G::KS), j % G::KS, fo);
#pragma unroll
            for (int i = 0; i < SK; ++i) fin_chunk(i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const int c = j / G::KS, ks = j % G::KS;
                if (j + PF < NK) kq[j + PF] = row_frag<D>(Kn, 32 * ((j + PF) / G::KS), (j + PF) % G::KS, fo);
                else load_v(j + PF - NK);
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    if (ks == 0) mfma_s_first(sn[X][c], kq[j], qf[X][ks]);
                    else mfma_s(sn[X][c], kq[j], qf[X][ks]);
                    if (j * 2 + X + SK < NM) fin_chunk(j * 2 + X + SK);
                    __builtin_amdgcn_sched_barrier(0);  // this chunk's vector ALU work stays in the shadow of this MFMA
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < PF; ++j) load_v(j);
#pragma unroll
            for (int i = 0; i < NM; ++i) fin_chunk(i);
        }
#pragma unroll
        for (int X = 0; X < 2; ++X) l_run[X] += xhalf_sum(rs[X]);

        // ---- phase 2: the in-lane maxima of S(t+1) beside the first half of the MFMAs (4 scores per MFMA), the rest of
        // start-softmax(t+1) -- a short dependent chain -- in the middle, with MFMAs still queued behind it.  On a wave's last tile
        // the maxima are taken over stale registers and dropped: the per-MFMA chunks carry no branches.
        if (has_next) apply_mask(sn, t + 1);
        float mx[2] = {-INFINITY, -INFINITY};
        auto max_chunk = [&](int i) {  // 2 * EPC scores
#pragma unroll
            for (int u = 0; u < 2 * EPC; u += 2) {
                const int e = i * 2 * EPC + u, X = e >> 5, c = (e >> 4) & 1, r = e & 15;
                max3_pinned(mx[X], sn[X][c][r], sn[X][c][r + 1]);
            }
        };
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = j / (2 * G::DB), s2 = (j / G::DB) & 1, db = j % G::DB;
            if (j + PF < NV) load_v(j + PF);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                mfma_o(oacc[X][db], vq[j], __builtin_bit_cast(s16x8, pf[X][c][s2]));
                if (j * 2 + X < NM / 2) max_chunk(j * 2 + X);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j == NV / 2 - 1 && has_next) sm_start(mx);
        }
    };

    for (int t = 0; t < nt; t += 2) {
        tile(std::integral_constant<int, 0>{}, t, sA, sB);
        if (t + 1 < nt) tile(std::integral_constant<int, 1>{}, t + 1, sB, sA);
    }

#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int db = 0; db < G::DB; db += 2) mfma_settle_acc(oacc[X][db], oacc[X][db + 1]);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const float inv = l_run[X] > 0.f ? 1.f / l_run[X] : 0.f;
        store_row_block<D>(out + (int64_t)(tok0 + my_q[X]) * o_ts + (int64_t)h * D, oacc[X], inv, lane, q_valid[X], (o_ts & 7) == 0);
        if (q_valid[X] && lane < 32) lse[(int64_t)h * T + tok0 + my_q[X]] = (l_run[X] > 0.f) ? m_run[X] * scale + logf(l_run[X]) : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// K1 forward, one wave per SIMD, FOLDED softmax (round 3; variant 3).  Same block / wave / LDS geometry as flash_fwd64_k; what changes is
// the vector-ALU work per score, which is what bounds that kernel (7 VALU, 2 of them transcendental, + their wait states in every
// 32-cycle MFMA gap of its phase 1, where ~5 issues hide: profiles/r03_flash_fwd_isa_mix.md), and where in the tile it runs:
//   * Q is multiplied by softmax_scale * log2(e) ONCE, when its fragments are loaded (rounded to bf16: a relative perturbation of 2^-9 of
//     an input that is itself a bf16 GEMM output), so the MFMA accumulators hold scores in log2 units;
//   * the reference maximum is subtracted BY THE MATRIX PIPE: every score accumulator starts with one extra k-step whose K-side fragment
//     is the constant e_0 (1 in contraction slot 0, 0 elsewhere) and whose Q-side fragment carries -mhat[row] in slot 0, i.e.
//     S' = K Q~^T - mhat.  mhat is a bf16-representable reference (a truncated running maximum), moved only when some row's maximum
//     rises more than THR above it.  Softmax is invariant under the choice of the reference, so the result is exact; p <= 2^THR (+ the
//     truncation, 2^-7 |mhat|).  4 extra MFMAs per 64-key tile (68 instead of 64) replace 64 v_fma and their 32 wait states;
//   * what is left per score: half a v_max3, v_exp, v_add (row sum), half a v_cvt_pk.  Nothing in the common tile crosses lanes: row sums
//     are kept as per-lane partial sums (joined once in the epilogue) and the decision "no row rose more than THR above its reference"
//     is taken on the per-lane partial maxima.  A moved reference is the rare branch: O and the partial sums are rescaled by 2^-delta,
//     the scores of this tile and -- one tile later -- those of the next tile, already being formed against the old reference, get one
//     v_sub each;
//   * phase 1 (S'(t+1): 36 MFMAs) carries the exponentials / sums / packs of the first 32-key half of tile t and, once the first half
//     of S'(t+1) is complete, its in-lane maxima; phase 2 (O += V^T P: 32 MFMAs, first key half first) the second half of tile t, the
//     other maxima and the decision.  Exponentials are issued one MFMA gap ahead of the adds / packs that read them (no dependent
//     neighbours).  ~4 non-MFMA issues per gap in phase 1, ~5 in phase 2 (flash_fwd64_k: 9.3 and 4.6).
// ABL (timing ablations only, results wrong; kbench variants 10 + ABL): 1 no landing wait, 2 no barrier, 4 no K / V transfers in the loop,
// 8 no softmax vector work
template <int D, bool CAUSAL, int THR, int ABL = 0, int PFD = 3>
__global__ __launch_bounds__(256, 1) void flash_fwd64f_k(const bf16_t* __restrict__ q, int64_t q_ts, const bf16_t* __restrict__ k,
                                                         const bf16_t* __restrict__ v, int64_t kv_ts, bf16_t* __restrict__ out,
                                                         int64_t o_ts, float* __restrict__ lse, const int32_t* __restrict__ cu, int64_t T,
                                                         int hq, int hkv, float scale) {
    using G = Geo<D>;
    constexpr int IMG = G::IMG_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * IMG];  // K stage 0, K stage 1, V stage 0, V stage 1

    const int seq = blockIdx.z;
    const int h = ((int)blockIdx.x % hkv) * (hq / hkv) + (int)blockIdx.x / hkv;  // the q heads of one kv head share an XCD's L2
    const int qt = gridDim.y - 1 - blockIdx.y;                                    // heaviest (last) query tiles first
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 256;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw0 = q0 + wave * 64;
    int my_q[2];
    bool q_valid[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        my_q[X] = qw0 + 32 * X + (lane & 31);
        q_valid[X] = my_q[X] < len;
    }
    const int kv_end = CAUSAL ? min(len, q0 + 256) : len;
    const int nt = (kv_end + 63) / 64;                                                    // tiles the block streams
    const int ntw = qw0 >= len ? 0 : (CAUSAL ? min(nt, qw0 / 64 + 1) : nt);               // tiles this wave computes on

    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    TileSrc<D, 4> ksrc, vsrc;
    ksrc.init(kbase, kv_ts, T - tok0, D, wave, lane);
    vsrc.init(vbase, kv_ts, T - tok0, D, wave, lane);
    unsigned char* Kst = smem;
    unsigned char* Vst = smem + 2 * IMG;
    const uint32_t Kst_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem, Vst_lds = Kst_lds + 2 * IMG;
    ksrc.issue(Kst, 0, 0, wave);
    if (nt > 1) ksrc.issue(Kst + IMG, 64, 0, wave);
    vsrc.issue(Vst, 0, 0, wave);

    FragOffs<D> fo;
    fo.init(lane);

    const float sc2 = scale * kLog2e;
    s16x8 qf[2][G::KS];   // Q~ = bf16(q * scale * log2 e)
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q[X]) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } cv;
            cv.u = q_valid[X] ? ld16(qp + ks * 16) : z4();
            if (fabsf(sc2 - 1.f) > 1e-6f) {   // (a caller that stored q pre-scaled -- ie_qkv_rotary_fwd_scaled -- passes softmax_scale = ln 2)
                float f[8];
                unpack8(cv.u, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= sc2;
                cv.u = pack8(f);
            }
            qf[X][ks] = cv.s;
        }
    }
    // the extra k-step: K-side e_0 (contraction slot 0 lives in element 0 of lanes 0..31), Q-side -mhat[row] in the same slot
    s16x8 ones, mq[2];   // (mq lives in the accumulation file like the Q fragments)
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 0, mq[0][e] = 0, mq[1][e] = 0;
    if (lane < 32) ones[0] = (short)0x3F80;
    auto mfma_fold = [&](f32x16& d, int X) {   // d = e_0 (x) (-mhat): the start value of a score accumulator
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, 0" : "=&v"(d), "+a"(mq[X]) : "v"(ones));   // src0 = K side (e_0), src1 = Q side
    };

    f32x16 oacc[2][G::DB];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int db = 0; db < G::DB; ++db) oacc[X][db] = zero16();
    // l_part: PER-LANE partial row sums (a lane sums the 32 keys of every tile that its registers hold; the two lanes of a row are added
    // once, in the epilogue), two independent chains per 32-row block.
    float mhat[2] = {0.f, 0.f}, l_part[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, mm[2] = {0.f, 0.f};
    bool resc = false;   // the decision of the tile about to start: its rows rose more than THR above the reference (mm: in-lane maxima)
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

    auto xhalf_max = [&](float v) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    auto xhalf_sum = [&](float v) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };
    auto set_mq = [&](int X) {   // bf16(-mhat) into contraction slot 0 of the Q-side fragment (mhat is bf16-representable: exact)
        const unsigned bits = __float_as_uint(-mhat[X]) >> 16;
        mq[X][0] = (lane < 32) ? (short)bits : (short)0;
    };
    // masks of key half c of tile t (wave-uniform decision: diagonal / last tiles only); selects, no per-element branches
    auto apply_mask_c = [&](f32x16 (&s)[2][2], int t, int c) {
        const int kv0 = t * 64 + 32 * c;
        if (!((CAUSAL && kv0 + 31 > qw0) || (kv0 + 32 > len))) return;
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int lim = CAUSAL ? min(len - 1, my_q[X]) : len - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + creg_row(r, lane);
                s[X][c][r] = key > lim ? -INFINITY : s[X][c][r];
            }
        }
    };
    auto apply_mask = [&](f32x16 (&s)[2][2], int t) {
        apply_mask_c(s, t, 0);
        apply_mask_c(s, t, 1);
    };

    // ---- prologue: S'(0) against reference 0 (its tile moves the reference to the first maxima)
    f32x16 sA[2][2], sB[2][2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ntw > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const s16x8 kfr = row_frag<D>(Kst, 32 * c, ks, fo);
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    if (ks == 0) mfma_s_first(sA[X][c], kfr, qf[X][ks]);
                    else mfma_s(sA[X][c], kfr, qf[X][ks]);
                }
            }
        mfma_settle(sA[0][0], sA[0][1], sA[1][0], sA[1][1]);
        apply_mask(sA, 0);
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            float mx0 = -INFINITY;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx0 = fmaxf(mx0, sA[X][c][r]);
            const float m = xhalf_max(mx0);   // (key 0 is visible to every row: finite)
            mhat[X] = __uint_as_float(__float_as_uint(m == -INFINITY ? 0.f : m) & 0xffff0000u);
            set_mq(X);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sA[X][c][r] -= mhat[X];
        }
        asm volatile("s_nop 1" : "+a"(mq[0]), "+a"(mq[1]));   // accumulation-file write -> MFMA source
    }

    constexpr int NM1 = 4 * G::KS + 4;   // MFMAs of phase 1 (the four accumulators' extra k-steps included)
    constexpr int NM2 = 8 * G::DB;       // MFMAs of phase 2
    constexpr int U1 = 32, U2 = 32;      // scores per lane finished in phase 1 (first key half) / phase 2 (second half: P.V reads it last)
    constexpr int GF2 = NM2 * 3 / 4 - 1; // phase-2 gaps for the second half's scores (packed before the last quarter of the products)

    // tile t with its scores (against the reference mhat) in `sc`; the scores of tile t+1 go to `sn`.  PAR = t & 1: the LDS stages.
    auto tile = [&](auto par_c, int t, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2]) {
        constexpr int PAR = decltype(par_c)::value;
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
        // K(t+2) / V(t+1) go into the stages the barrier just freed, one 1-KiB piece per fourth MFMA gap of phase 1 (inline asm: hipcc neither
        // counts nor drains them; the wait at the top of the next tile is the explicit vmcnt(0) above).  Issued as one burst behind the
        // barrier they cost the whole burst in matrix-pipe idle time: 57 of 620 us on the 4 x 4096 call (profiles/r03_flash_fwd_ablation.md).
        constexpr int PERW = TileSrc<D, 4>::PERW;
        auto dma_piece = [&](int i) {
            if (ABL & 4) return;
            // (unconditional: a tile behind the block's last one is read and never used -- rows of the next sequence, or zeros behind the
            // tensor (num_records) -- which keeps the gaps free of branches; the epilogue waits for the stragglers)
            if (i < PERW) ksrc.issue_piece_asm(Kst_lds + PAR * IMG, (t + 2) * 64 * ksrc.ts2, wave, i);
            else if (i < 2 * PERW) vsrc.issue_piece_asm(Vst_lds + (1 - PAR) * IMG, (t + 1) * 64 * vsrc.ts2, wave, i - PERW);
        };
        if (t >= ntw) {   // this wave has no rows left that see tile t: it still moves its share of the tiles
#pragma unroll
            for (int i = 0; i < 2 * PERW; ++i) dma_piece(i);
            return;
        }
        const unsigned char* Kn = Kst + (1 - PAR) * IMG;
        const unsigned char* Vc = Vst + PAR * IMG;
        const bool has_next = t + 1 < ntw;

        if (resc) {   // (rare) the reference of both row blocks moves up to their rows' maxima; rows that did not rise keep theirs
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const float m = xhalf_max(mm[X]);                       // row maximum above the reference (-inf: a fully masked row)
                const float nm = __uint_as_float(__float_as_uint(mhat[X] + fmaxf(m, 0.f)) & 0xffff0000u);   // never down; truncated to bf16
                const float dlt = nm - mhat[X];
                const float al = __builtin_amdgcn_exp2f(-dlt);
                mhat[X] = nm;
#pragma unroll
                for (int db = 0; db < G::DB; ++db) acc_scale(oacc[X][db], al);
                l_part[X][0] *= al;
                l_part[X][1] *= al;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[X][c][r] -= dlt;   // formed against the old reference
                set_mq(X);
            }
            asm volatile("s_nop 1" : "+a"(mq[0]), "+a"(mq[1]));   // accumulation-file write -> MFMA source
            resc = false;
        }

        u32x4 pf[2][2][2];  // P fragments [X][c][s2]: 8 bf16 = keys 32c + 16*s2-step in C/D register order
        // score o of a lane's 64, in the order the P.V products consume them: o -> (c, 16-key step s2, X, r)
        auto E = [&](int o) {   // p = 2^S'
            const int c = o >> 5, s2 = (o >> 4) & 1, X = (o >> 3) & 1, r = 8 * s2 + (o & 7);
            if (!(ABL & 8)) asm volatile("v_exp_f32 %0, %0" : "+v"(sc[X][c][r]));
        };
        auto SP = [&](int o) {  // row sum; every second score: the bf16 pair
            const int c = o >> 5, s2 = (o >> 4) & 1, X = (o >> 3) & 1, r = 8 * s2 + (o & 7);
            if (!(ABL & 8)) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_part[X][o & 1]) : "v"(sc[X][c][r]));
            if (o & 1) pf[X][c][s2][(r & 7) >> 1] = (ABL & 8) ? __float_as_uint(sc[X][c][r]) : pack2bf(sc[X][c][r - 1], sc[X][c][r]);   // (native convert: the register tuple is hipcc's to place)
        };
        // gap g of Gn gaps works on scores [lo(g), lo(g + 1)) of U starting at o0: their exponentials now, sums / packs of the previous gap's
        auto fin_gap = [&](int g, int Gn, int U, int o0) {
            const int a0 = ((g - 1) * U / Gn) & ~1, a1 = (g * U / Gn) & ~1, a2 = g >= Gn ? a1 : (((g + 1) * U / Gn) & ~1);
#pragma unroll
            for (int o = a1; o < a2; ++o) E(o0 + o);
            if (g > 0) {
#pragma unroll
                for (int o = a0; o < a1; ++o) SP(o0 + o);
            }
        };

        constexpr int PF = PFD, NK = 2 * G::KS, NV = 4 * G::DB;
        s16x8 kq[NK], vq[NV];
        auto load_v = [&](int j) { if (!(ABL & 16)) vq[j] = trans_frag<D>(Vc, j % G::DB, j / G::DB, fo); };
        auto load_k = [&](int j) { if (!(ABL & 16)) kq[j] = row_frag<D>(Kn, 32 * (j / G::KS), j % G::KS, fo); };
        // ---- phase 1: S'(t+1) = [e_0 | K(t+1)] [-mhat | Q~]^T   beside   the first key half of softmax(t)
        if (has_next) {
#pragma unroll
            for (int j = 0; j < PF; ++j) load_k(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    mfma_fold(sn[X][c], X);
                    fin_gap(c * (NM1 / 2) + X, NM1, U1, 0);
                    if ((c * (NM1 / 2) + X) % 4 == 1) dma_piece((c * (NM1 / 2) + X) / 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const int j = c * G::KS + ks;
                    if (j + PF < NK) load_k(j + PF);
                    else load_v(j + PF - NK);
#pragma unroll
                    for (int X = 0; X < 2; ++X) {
                        mfma_s(sn[X][c], kq[j], qf[X][ks]);
                        fin_gap(c * (NM1 / 2) + 2 + 2 * ks + X, NM1, U1, 0);
                        if ((c * (NM1 / 2) + 2 + 2 * ks + X) % 4 == 1) dma_piece((c * (NM1 / 2) + 2 + 2 * ks + X) / 4);
                        __builtin_amdgcn_sched_barrier(0);  // this gap's vector ALU work stays in the shadow of this MFMA
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2 * PERW; ++i) dma_piece(i);
#pragma unroll
            for (int j = 0; j < PF; ++j) load_v(j);
#pragma unroll
            for (int g = 0; g < NM1; ++g) fin_gap(g, NM1, U1, 0);
        }
        fin_gap(NM1, NM1, U1, 0);   // sums / packs of the last gap's scores

        // ---- phase 2: O += V(t)^T P(t), first key half first   beside   the second key half of softmax(t) and the in-lane maxima of S'(t+1),
        // one v_max3 (4 scores) per MFMA; then the decision.  On a wave's last tile the maxima run over stale registers and are dropped: no
        // branches inside the phase.
        if (has_next) apply_mask(sn, t + 1);
        float mx[2][2];   // two chains per 32-row block (every chain starts from its first pair: no -inf moves)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = j / (2 * G::DB), s2 = (j / G::DB) & 1, db = j % G::DB;
            if (j + PF < NV) load_v(j + PF);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const int g = 2 * j + X;
                mfma_o(oacc[X][db], vq[j], __builtin_bit_cast(s16x8, pf[X][c][s2]));
                if (g <= GF2) fin_gap(g, GF2, U2, U1);
                {   // scores e = 2g, 2g + 1 of S'(t+1): row block Y, key half cc, registers rr, rr + 1
                    const int e0 = g * (64 / NM2);
#pragma unroll
                    for (int u = 0; u < 64 / NM2; u += 2) {
                        const int e = e0 + u, Y = e >> 5, cc = (e >> 4) & 1, rr = e & 15, n = (e & 31) >> 1, ch = n & 1;   // pair n of row block Y
                        if (ABL & 8) { if (n < 2) mx[Y][ch] = 0.f; }
                        else if (n < 2) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx[Y][ch]) : "v"(sn[Y][cc][rr]), "v"(sn[Y][cc][rr + 1]));
                        else max3_pinned(mx[Y][ch], sn[Y][cc][rr], sn[Y][cc][rr + 1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {   // the decision about tile t+1, on per-lane partial maxima: nothing crosses lanes
            float top;
#pragma unroll
            for (int Y = 0; Y < 2; ++Y) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mm[Y]) : "v"(mx[Y][0]), "v"(mx[Y][1]));
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(top) : "v"(mm[0]), "v"(mm[1]));
            resc = has_next & !__all(top <= (float)THR);
        }
    };

    for (int t = 0; t < nt; t += 2) {
        tile(std::integral_constant<int, 0>{}, t, sA, sB);
        if (t + 1 < nt) tile(std::integral_constant<int, 1>{}, t + 1, sB, sA);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tiles' look-ahead transfers have landed before the block gives its LDS back
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int db = 0; db < G::DB; db += 2) mfma_settle_acc(oacc[X][db], oacc[X][db + 1]);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const float l_row = xhalf_sum(l_part[X][0] + l_part[X][1]);
        const float inv = l_row > 0.f ? 1.f / l_row : 0.f;
        store_row_block<D>(out + (int64_t)(tok0 + my_q[X]) * o_ts + (int64_t)h * D, oacc[X], inv, lane, q_valid[X], (o_ts & 7) == 0);
        // natural-log LSE from log2-unit scores: (mhat + log2 l) ln 2
        if (q_valid[X] && lane < 32) lse[(int64_t)h * T + tok0 + my_q[X]] = (l_row > 0.f) ? (mhat[X] + __builtin_amdgcn_logf(l_row)) * kLn2 : -INFINITY;
    }
}


// ------------------------------------------------------------------------------------------------
// K1 forward, EIGHT waves per block in two groups that take turns on the matrix pipe (round 5; variant 4).  flash_fwd64_k gives every SIMD one wave and
// hides the softmax in the MFMA shadows of that same wave -- 9 issues per gap where 5 hide: the matrix pipe is 41 % busy
// (profiles/r05_pmc_flash_fwd_SQ_pass*.md).  Here a SIMD holds TWO waves of 32 query rows (256 rows per block as before; waves 0-3 = group A, 4-7 =
// group B: one of each per SIMD) and the block runs in HALF-STEPS closed by s_barrier: in a half-step one group is in its matrix phase
//     M(k):  S(k+1) = K(k+1) Q^T  and  O += V(k)^T P(k)       (32 MFMAs + their LDS fragment reads, s_setprio 1)
// while the other is in its vector phase
//     V(k):  softmax of S(k): row maxima, (deferred) rescale, p = exp2(..), row sums, bf16 pack;  group B also requests K(k+2) / V(k+1)
// and they swap in the next half-step: group A runs M(k) in half-step 2k and V(k) in 2k-1, group B one half-step later.  K and V tiles are
// double-buffered 64-key images as in the other kernels; a tile requested in half-step 2k (by the group then in its vector phase: it issues no LDS read
// in front of which hipcc would drain the request) is first read in half-step 2k+2.  Causal: a wave past its own diagonal tile keeps the barriers
// and skips the work.
template <int D, bool CAUSAL, int THR, int MODE>
__global__ __launch_bounds__(512, 2) void flash_fwd8_k(const bf16_t* __restrict__ q, int64_t q_ts, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, int64_t kv_ts, bf16_t* __restrict__ out, int64_t o_ts,
                                                       float* __restrict__ lse, const int32_t* __restrict__ cu, int64_t T, int hq, int hkv,
                                                       float scale) {
    using G = Geo<D>;
    constexpr int IMG = G::IMG_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * IMG];  // K stage 0, K stage 1, V stage 0, V stage 1

    const int seq = blockIdx.z;
    const int h = ((int)blockIdx.x % hkv) * (hq / hkv) + (int)blockIdx.x / hkv;  // the q heads of one kv head share an XCD's L2
    const int qt = gridDim.y - 1 - blockIdx.y;                                    // heaviest (last) query tiles first
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 256;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    // rows: group A takes the even 32-row slices, group B the odd ones, so that the two waves of a SIMD see the same diagonal tile
    const int qw0 = q0 + (2 * w4 + grp) * 32;
    const int my_q = qw0 + (lane & 31);
    const bool q_valid = my_q < len;
    const int kv_end = CAUSAL ? min(len, q0 + 256) : len;
    const int nt = (kv_end + 63) / 64;                                                    // tiles the block streams
    const int ntw = qw0 >= len ? 0 : (CAUSAL ? min(nt, (qw0 + 31) / 64 + 1) : nt);        // tiles this wave computes on

    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    TileSrc<D, 4> ksrc, vsrc;          // a tile is requested by the four waves of ONE group
    ksrc.init(kbase, kv_ts, T - tok0, D, w4, lane);
    vsrc.init(vbase, kv_ts, T - tok0, D, w4, lane);
    unsigned char* Kst = smem;
    unsigned char* Vst = smem + 2 * IMG;
    if (grp == 0) {                    // K(0), K(1), V(0) up front
        ksrc.issue(Kst, 0, 0, w4);
        if (nt > 1) ksrc.issue(Kst + IMG, 64, 0, w4);
        vsrc.issue(Vst, 0, 0, w4);
    }

    FragOffs<D> fo;
    fo.init(lane);
    s16x8 qf[G::KS];
    {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } cv;
            cv.u = q_valid ? ld16(qp + ks * 16) : z4();
            qf[ks] = cv.s;
        }
    }
    f32x16 oacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) oacc[db] = zero16();
    float m_ref = -INFINITY, l_part = 0.f;    // reference maximum of the row (raw score units); this lane's part of the row sum
    const float sc2 = scale * kLog2e;
    f32x16 sacc[2];                           // S^T of the tile whose softmax comes next: keys 0..31 / 32..63 of the tile x this lane's row
    s16x8 pf[2][2];                           // P fragments of the tile whose P.V comes next

#ifndef IE_FWD8_ABL   // timing ablations (A/B builds, results wrong): 1 no softmax arithmetic, 2 no MFMAs, 4 no transfers in the loop, 8 / 16 / 32 below
#define IE_FWD8_ABL 0
#endif
    auto xhalf_max = [&](float x) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    // scores of tile t
    auto scores = [&](int t) {
        const unsigned char* Ks = Kst + (t & 1) * IMG;
#ifndef IE_FWD8_ALT   // A/B: 1 = the two key halves' accumulation chains interleaved (consecutive MFMAs on different accumulators)
#define IE_FWD8_ALT 0
#endif
        if (IE_FWD8_ALT) {
            sacc[0] = zero16();
            sacc[1] = zero16();
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    sacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], sacc[c], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            sacc[c] = zero16();
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
                sacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((IE_FWD8_ABL & 32) ? qf[(ks + 1) % G::KS] : row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], sacc[c], 0,
                                                                  0, 0);
        }
    };
    // O^T += V(t)^T P(t)^T
    auto pv = [&](int t) {
        const unsigned char* Vs = Vst + (t & 1) * IMG;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < G::DB; ++db)
                {
                    const int idx = (2 * c + s2) * G::DB + db;   // (timing ablations: 8 = plain 16-byte reads in place of the transposing pairs, 16 = no reads)
                    const s16x8 vf = (IE_FWD8_ABL & 16)  ? qf[idx % G::KS]
                                     : (IE_FWD8_ABL & 8) ? row_frag<D>(Vs, 32 * (idx & 1), (idx >> 1) % G::KS, fo)
                                                         : trans_frag<D>(Vs, db, 2 * c + s2, fo);
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c][s2], oacc[db], 0, 0, 0);
                }
    };
    // softmax of tile t (scores in sacc) -> pf; the row's reference maximum moves only when some row of the wave rose more than THR log2 units
    // above its own (p <= 2^THR then; exactly compensated: O and l are rescaled whenever it moves)
    auto softmax = [&](int t) __attribute__((always_inline)) {
        const int kv0 = t * 64;
        const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > len);
        if (need_mask) {
            const int lim = CAUSAL ? min(len - 1, my_q) : len - 1;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * c + creg_row(r, lane);
                    sacc[c][r] = key > lim ? -INFINITY : sacc[c][r];
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[c][r]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_ref, mx);
        const bool keep = __all((m_new - m_ref) * sc2 <= (float)THR);   // (m_ref = -inf, the first tile: inf <= THR is false -> moves)
        if (!keep) {
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_ref - m_use) * sc2);   // m_ref = -inf -> 0
#pragma unroll
            for (int db = 0; db < G::DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            l_part *= alpha;
            m_ref = m_new;
        }
        const float moff = ((m_ref == -INFINITY) ? 0.f : m_ref) * sc2;
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[c][r], sc2, -moff));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[c][r + 1], sc2, -moff));
                rs0 += p0;
                rs1 += p1;
                sacc[c][r] = p0;
                sacc[c][r + 1] = p1;
            }
            pf[c][0] = pack_frag(sacc[c], 0);
            pf[c][1] = pack_frag(sacc[c], 1);
        }
        l_part += rs0 + rs1;
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Half-steps -2 .. 2 nt, each closed by a barrier.  Group A: M(-1) | V(0) M(0) | V(1) M(1) | ... | V(nt) M(nt); group B one half-step later:
    // idle | M(-1) | V(0) M(0) | ... | V(nt-1) M(nt-1) | V(nt).  Two straight-line loops (one per group: no phase branch inside).
    // (A hand-pipelined matrix phase -- fragments requested four MFMAs ahead, every MFMA pinned behind the request of its gap -- ran 688 us against
    // 614 us for hipcc's own interleave of the two loops below: 256 registers with spills.  profiles/r05_flash_fwd8.md)
    // MODE 2 (variant 6): the matrix phase as a prescribed order built on two measured rules (MI355X_MICROARCH.md, per-instruction constants; this
    // kernel's ablations in profiles/r05_flash_fwd8.md): (1) any instruction between two MFMAs on the SAME accumulator costs ~43 cycles (the
    // accumulator forwarding is lost) -- between MFMAs on different accumulators it costs its issue slot; (2) a lone wave pays the LDS round trips it
    // waits for as matrix-pipe idle time.  So: S(kk+1) as the two key halves' chains INTERLEAVED (neighbouring MFMAs on different accumulators), its
    // sixteen K fragments requested before the phase opens (end of the wave's vector phase: the tile is visible since the barrier that opened that
    // phase, the registers are free once the scores are packed), the two transposing reads of P.V fragment m issued right behind S MFMA m into the
    // ring slot that MFMA just read (sixteen MFMAs ahead of their use), and P.V as sixteen bare MFMAs, d-blocks rotating.  sched_barrier pins
    // the order; the counted lgkmcnt waits are hipcc's own (in-order counters: exact for a pinned straight-line order).
    constexpr bool PIPE = MODE == 2;
    constexpr int NR = 2 * G::KS;                      // ring slots: S MFMA m = (k-step m / 2, key half m % 2); P.V MFMA m = (key step m / DB, d-block m % DB)
    static_assert(4 * G::DB == 2 * G::KS, "one ring for the fragments of both products");
    s16x8 ring[NR];
    uint32_t va[2 * G::DB];                            // this phase's V fragment addresses (stage included; the key step is an immediate)
    // Fragment addresses from THREE lane constants instead of FragOffs' sixteen: the XOR swizzle commutes with the k-step / d-block bits,
    // row[ks] = row[0] ^ 32 ks and tr[db] = tr[0] ^ 64 db; 32-row blocks and key steps are immediates.
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t a_row = smem_lds + fo.row[0], a_tr0 = smem_lds + 2 * IMG + fo.tr0[0], a_tr1 = smem_lds + 2 * IMG + fo.tr1[0];
    auto kfrag = [&](uint32_t base, int half, int ks) __attribute__((always_inline)) {
        if (IE_FWD8_ABL & 32) return qf[(ks + 1) % G::KS];
        return *(const __attribute__((address_space(3))) s16x8*)((base ^ (32u * ks)) + half * 32 * (2 * D));
    };
    auto vfrag = [&](int db, int step) __attribute__((always_inline)) {
        if (IE_FWD8_ABL & 16) return qf[(db + 4 * step) % G::KS];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va[2 * db] + step * 16 * (2 * D)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va[2 * db + 1] + step * 16 * (2 * D)));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    };
#define IE_SB() __builtin_amdgcn_sched_barrier(0)
    auto load_k = [&](int t) __attribute__((always_inline)) {          // ring <- the sixteen K fragments of tile t
        uint32_t kb = a_row + (t & 1) * IMG;
        asm volatile("" : "+v"(kb));   // (opaque: the per-fragment XORs stay here instead of becoming eight loop-invariant registers)
#pragma unroll
        for (int m = 0; m < NR; ++m) ring[m] = kfrag(kb, m & 1, m >> 1);
    };
    auto set_va = [&](int t) __attribute__((always_inline)) {          // V fragment addresses of tile t
#pragma unroll
        for (int db = 0; db < G::DB; ++db) {
            va[2 * db] = (a_tr0 + (t & 1) * IMG) ^ (64u * db);
            va[2 * db + 1] = (a_tr1 + (t & 1) * IMG) ^ (64u * db);
            asm volatile("" : "+v"(va[2 * db]), "+v"(va[2 * db + 1]));
        }
    };
    // Straight-line pieces, no branch inside (where paths join hipcc's waits take the worse of the two histories; the loops below are peeled instead:
    // first half-step = S only, a wave's last tile = P.V only, both once per wave).
    auto pipe_s = [&](bool then_pv) __attribute__((always_inline)) {   // S from the ring; ring <- the P.V fragments if P.V follows (va set)
        IE_SB();
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            sacc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[m], qf[m >> 1], (m >> 1) ? sacc[m & 1] : zero16(), 0, 0, 0);
            IE_SB();
            if (then_pv) ring[m] = vfrag(m % G::DB, m / G::DB);
            IE_SB();
        }
    };
    auto pipe_pv = [&]() __attribute__((always_inline)) {              // O += V^T P from the ring
        IE_SB();
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            oacc[m % G::DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[m], pf[(m / G::DB) >> 1][(m / G::DB) & 1], oacc[m % G::DB], 0, 0, 0);
            IE_SB();
        }
    };
#undef IE_SB
#ifndef IE_FWD8_DEFER   // 1: group B requests its tiles from inline asm and waits for them at the end of its NEXT (matrix) phase, a half-step later
#define IE_FWD8_DEFER 0
#endif
    auto matrix_phase = [&](int kk, bool waits = false) __attribute__((always_inline)) {          // M(kk): S(kk + 1), P.V(kk)
        __builtin_amdgcn_s_setprio(1);
        const bool do_s = !(IE_FWD8_ABL & 2) && kk + 1 < ntw, do_pv = !(IE_FWD8_ABL & 2) && kk >= 0 && kk < ntw;
#ifndef IE_FWD8_SGB   // 1: the common matrix phase (both products) as ONE scheduling region with a prescribed interleave (A/B switch)
#define IE_FWD8_SGB 0
#endif
        if (IE_FWD8_SGB && do_s && do_pv) {
            // the wave is alone on the matrix pipe: every LDS round trip it has to wait for is pipe idle time.  hipcc's own order keeps the reads about
            // one MFMA ahead and drains them (lgkmcnt(0)) before most MFMAs of the second product; prescribed here: four fragments ahead throughout --
            // 4 reads | 12 x (MFMA, read) | 4 x (MFMA, 2 transposing reads) | 12 x (MFMA, 2 transposing reads) | 4 MFMAs
            scores(kk + 1);
            pv(kk);
            constexpr int NKF = 2 * G::KS, NVF = 4 * G::DB;   // fragments of the two products (one read / two reads each)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < NKF - 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int i = 0; i < NVF - 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
        } else {
            if (do_s) scores(kk + 1);
            if (do_pv) pv(kk);
        }
        __builtin_amdgcn_s_setprio(0);
        if (IE_FWD8_DEFER && waits) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(IE_FWD8_ABL & 64)) __builtin_amdgcn_s_barrier();   // (64: the loop without its barriers -- what keeping the groups in step costs)
    };
    auto vector_phase = [&](int kk, bool requests) __attribute__((always_inline)) {   // V(kk): softmax(kk); group B requests K(kk + 2) / V(kk + 1) into the stages tiles kk / kk - 1 left
        // (the fragment requests first: behind a transfer request hipcc would drain the transfer before any LDS read)
        if (requests && !(IE_FWD8_ABL & 4)) {
            if (IE_FWD8_DEFER) {
                const uint32_t sl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
                constexpr int PW = TileSrc<D, 4>::PERW;
                if (kk + 2 < nt) {
#pragma unroll
                    for (int pq = 0; pq < PW; ++pq) ksrc.issue_piece_asm(sl + (kk & 1) * IMG, (kk + 2) * 64 * ksrc.ts2, w4, pq);
                }
                if (kk + 1 < nt) {
#pragma unroll
                    for (int pq = 0; pq < PW; ++pq) vsrc.issue_piece_asm(sl + 2 * IMG + ((kk + 1) & 1) * IMG, (kk + 1) * 64 * vsrc.ts2, w4, pq);
                }
            } else {
                if (kk + 2 < nt) ksrc.issue(Kst + (kk & 1) * IMG, (kk + 2) * 64, 0, w4);
                if (kk + 1 < nt) vsrc.issue(Vst + ((kk + 1) & 1) * IMG, (kk + 1) * 64, 0, w4);
            }
        }
        if (!(IE_FWD8_ABL & 1) && kk < ntw) softmax(kk);
        if ((IE_FWD8_ABL & 1) && kk < ntw) {   // (the scores stay alive: packed as they are)
#pragma unroll
            for (int c = 0; c < 2; ++c) { pf[c][0] = pack_frag(sacc[c], 0); pf[c][1] = pack_frag(sacc[c], 1); }
            l_part += sacc[0][0];
        }
        if (requests && !IE_FWD8_DEFER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requested a whole half-step before anybody reads it
        if (!(IE_FWD8_ABL & 64)) __builtin_amdgcn_s_barrier();
    };
    if (MODE == 2) {
        // the same half-steps as MODE 0, the loops peeled by what a wave does in them: kk < ntw - 1 both products (the pipeline), kk = ntw - 1 its last
        // P.V, then nothing (a wave above the block's last diagonal tile; group A's extra half-steps) -- barriers and group B's requests in all of them
        auto requests = [&](int kk) __attribute__((always_inline)) {
            if (kk + 2 < nt) ksrc.issue(Kst + (kk & 1) * IMG, (kk + 2) * 64, 0, w4);
            if (kk + 1 < nt) vsrc.issue(Vst + ((kk + 1) & 1) * IMG, (kk + 1) * 64, 0, w4);
        };
        // (sched_barrier: hipcc otherwise sinks half of the softmax -- pure register arithmetic -- below the s_barrier into the wave's matrix phase)
        auto phase_end = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto run = [&](auto is_b) __attribute__((always_inline)) {
            constexpr bool B = decltype(is_b)::value;
            const int ni = B ? nt : nt + 1;
            if (B) phase_end();
            __builtin_amdgcn_s_setprio(1);                                   // M(-1)
            if (ntw > 0) { load_k(0); pipe_s(false); }
            __builtin_amdgcn_s_setprio(0);
            phase_end();
            int kk = 0;
            for (; kk < ntw - 1; ++kk) {
                if (B) requests(kk);
                if (!(IE_FWD8_ABL & 1)) softmax(kk);
                else {
#pragma unroll
                    for (int c = 0; c < 2; ++c) { pf[c][0] = pack_frag(sacc[c], 0); pf[c][1] = pack_frag(sacc[c], 1); }
                    l_part += sacc[0][0];
                }
                if (B) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(IE_FWD8_ABL & 128)) load_k(kk + 1);   // (behind the wait: hipcc drains every transfer it knows of in front of an LDS read anyway)
                set_va(kk);
                phase_end();
                __builtin_amdgcn_s_setprio(1);
                if (IE_FWD8_ABL & 128) load_k(kk + 1);
                pipe_s(true);
                pipe_pv();
                __builtin_amdgcn_s_setprio(0);
                phase_end();
            }
            if (kk < ntw) {
                if (B) requests(kk);
                softmax(kk);
                if (B) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                set_va(kk);
#pragma unroll
                for (int m = 0; m < NR; ++m) ring[m] = vfrag(m % G::DB, m / G::DB);
                phase_end();
                __builtin_amdgcn_s_setprio(1);
                pipe_pv();
                __builtin_amdgcn_s_setprio(0);
                phase_end();
                ++kk;
            }
            for (; kk < ni; ++kk) {
                if (B) { requests(kk); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                phase_end();
                phase_end();
            }
            if (B) phase_end();
        };
        if (grp == 0) run(std::false_type{}); else run(std::true_type{});
    } else if (MODE != 1) {
        if (grp == 0) {
            matrix_phase(-1);
            for (int kk = 0; kk <= nt; ++kk) {
                vector_phase(kk, false);
                matrix_phase(kk);
            }
        } else {
            __builtin_amdgcn_s_barrier();
            matrix_phase(-1);
            for (int kk = 0; kk < nt; ++kk) {
                vector_phase(kk, true);
                matrix_phase(kk, true);
            }
            __builtin_amdgcn_s_barrier();          // (V(nt): nothing left)
        }
    } else {
        // MODE 1: QUARTERS.  The matrix work of a tile takes a lone wave three times as long as its softmax, so with strictly alternating phases the
        // group in its vector phase waits two thirds of every half-step.  Here a wave's period is M1 (S(kk+1): 16 MFMAs) | M2 (P.V(kk), first key
        // half: 8) | M3 (second half: 8) | V (softmax(kk+1)), a barrier behind each, and group B runs TWO quarters behind group A: A M1 + B M3,
        // A M2 + B V, A M3 + B M1, A V + B M2 -- in every quarter at least one wave of the SIMD issues MFMAs, in two of four both do (the matrix pipe
        // does not care whose).  Requests: group A asks for K(kk + 3) in its vector quarter, group B for V(kk + 2) in its own (asm pieces; each group
        // waits for its requests at the end of its next M3, a barrier before anybody reads them).
        const uint32_t Kst_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem, Vst_lds = Kst_lds + 2 * IMG;
        constexpr int PERW = TileSrc<D, 4>::PERW;
        auto pvc = [&](int t, int c) __attribute__((always_inline)) {
            const unsigned char* Vs = Vst + (t & 1) * IMG;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < G::DB; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag<D>(Vs, db, 2 * c + s2, fo), pf[c][s2], oacc[db], 0, 0, 0);
        };
        auto period = [&](int kk) __attribute__((always_inline)) {
            const bool work = kk >= 0 && kk < ntw;
            __builtin_amdgcn_s_setprio(1);
            if (kk + 1 < ntw) scores(kk + 1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            if (work) pvc(kk, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            if (work) pvc(kk, 1);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this group's requests of a period ago have landed
            __builtin_amdgcn_s_barrier();
            if (grp == 0) {
                if (kk + 3 < nt) {
#pragma unroll
                    for (int pq = 0; pq < PERW; ++pq) ksrc.issue_piece_asm(Kst_lds + ((kk + 3) & 1) * IMG, (kk + 3) * 64 * ksrc.ts2, w4, pq);
                }
            } else if (kk + 2 < nt) {
#pragma unroll
                for (int pq = 0; pq < PERW; ++pq) vsrc.issue_piece_asm(Vst_lds + ((kk + 2) & 1) * IMG, (kk + 2) * 64 * vsrc.ts2, w4, pq);
            }
            if (kk + 1 < ntw) softmax(kk + 1);
            __builtin_amdgcn_s_barrier();
        };
        if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        for (int kk = -1; kk < nt; ++kk) period(kk);
        if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_part), __float_as_uint(l_part), false, false);
        const float l_row = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        const float inv = l_row > 0.f ? 1.f / l_row : 0.f;
        store_row_block<D>(out + (int64_t)(tok0 + my_q) * o_ts + (int64_t)h * D, oacc, inv, lane, q_valid, (o_ts & 7) == 0);
        if (q_valid && lane < 32) lse[(int64_t)h * T + tok0 + my_q] = (l_row > 0.f) ? m_ref * scale + logf(l_row) : -INFINITY;
    }
}

int g_fwd_variant = -1;  // -1: automatic; 0: 32 rows per wave (flash_fwd_k); 1: 64 rows per wave, THR = 0; 2: 64 rows per wave, THR = 4;
                         // 3: 64 rows per wave, folded softmax (flash_fwd64f_k), THR = 4; 4: eight waves in two groups alternating on the matrix pipe
                         // (flash_fwd8_k), THR = 4; 5: the same in quarter phases (slower: kept for the A/B)

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" int ie_flash_attn_fwd(const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts, void* out, int64_t o_ts,
                                 float* lse, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv, int d,
                                 float softmax_scale, int causal, void* stream) {
    IE_CHECK_ARG(q && k && v && out && lse && cu_seqlens, "ie_flash_attn_fwd: null pointer");
    IE_CHECK_ARG(nseq >= 0 && T >= 0 && max_seqlen >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0, "ie_flash_attn_fwd: bad shape");
    IE_CHECK_ARG(softmax_scale > 0.f, "ie_flash_attn_fwd: softmax_scale must be positive");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_flash_attn_fwd: head dim must be 64 or 128");
    IE_CHECK_SUPPORTED(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && q_ts % 8 == 0 && kv_ts % 8 == 0 && o_ts % 4 == 0,
                       "ie_flash_attn_fwd: pointers must be 16-byte aligned and token strides multiples of 8");
    if (nseq == 0 || T == 0 || max_seqlen == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    // automatic: 64 rows per wave (deferred rescale) for long sequences of head dim 128, where it measures 4-5 % faster in same-box
    // A/B runs (4 x 4096 causal: 679 vs 713 us, full attention 4 x 2048: 337 vs 351 us; profiles/r02_flash_attention.md); ragged packs
    // of short sequences leave too many of its 256-row blocks half empty (8 x <= 3000: 360 vs 336 us)
    // ... and the folded-softmax kernel (variant 3) where that one was picked AND the caller stored q pre-scaled (softmax_scale = ln 2, i.e.
    // scale * log2 e = 1: ie_qkv_rotary_fwd_scaled): scaling q inside the attention kernel costs one more bf16 rounding of q, which the
    // backward kernels (they recompute the scores from the q they are given) would not see
    const bool prescaled = fabsf(softmax_scale * kLog2e - 1.f) < 1e-6f;
    // round 5: ... and there the eight-wave kernel whose two wave groups take turns on the matrix pipe (variant 4) measures 7-8 % faster still, causal
    // and full, with the same results (profiles/r05_flash_fwd8.md)
    const int fwd_variant = g_fwd_variant >= 0 ? g_fwd_variant : ((d == 128 && T >= (int64_t)nseq * 2048) ? (prescaled ? 3 : 4) : 0);
#ifdef IE_ENABLE_ABLATIONS   // profiling builds only (hipcc -DIE_ENABLE_ABLATIONS): the shipped library holds no kernel with wrong results
    if (fwd_variant >= 10 && d == 128 && causal) {   // timing ablations of the folded kernel (results wrong)
        dim3 grid64((unsigned)hq, (unsigned)((max_seqlen + 255) / 256), (unsigned)nseq);
#define IE_LA(AB)                                                                                                                     \
    hipLaunchKernelGGL((flash_fwd64f_k<128, true, 4, AB>), grid64, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, \
                       kv_ts, (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale)
        switch (fwd_variant - 10) {
            case 1: IE_LA(1); break;
            case 3: IE_LA(3); break;
            case 7: IE_LA(7); break;
            case 8: IE_LA(8); break;
            case 15: IE_LA(15); break;
            case 16: IE_LA(16); break;
            case 31: IE_LA(31); break;
            case 40: hipLaunchKernelGGL((flash_fwd64f_k<128, true, 4, 0, 5>), grid64, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale); break;
            case 41: hipLaunchKernelGGL((flash_fwd64f_k<128, true, 4, 0, 7>), grid64, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale); break;
            default: IE_LA(0); break;
        }
#undef IE_LA
        return ie_launch_status("ie_flash_attn_fwd launch");
    }
#endif
    if (fwd_variant >= 4 && fwd_variant <= 6) {   // eight waves in two groups that share the matrix pipe: 4 = alternating half-steps, 5 = quarters, 6 = 4 with the pipelined matrix phase
        dim3 grid8((unsigned)hq, (unsigned)((max_seqlen + 255) / 256), (unsigned)nseq);
#define IE_L8M(DD, CA, MO)                                                                                                             \
    hipLaunchKernelGGL((flash_fwd8_k<DD, CA, 4, MO>), grid8, dim3(512), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, \
                       (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale)
#define IE_L8(DD, CA) do { if (fwd_variant == 4) IE_L8M(DD, CA, 0); else if (fwd_variant == 5) IE_L8M(DD, CA, 1); else IE_L8M(DD, CA, 2); } while (0)
        if (d == 128) { if (causal) IE_L8(128, true); else IE_L8(128, false); }
        else          { if (causal) IE_L8(64, true); else IE_L8(64, false); }
#undef IE_L8
#undef IE_L8M
        return ie_launch_status("ie_flash_attn_fwd launch");
    }
    if (fwd_variant >= 3) {
        dim3 grid64((unsigned)hq, (unsigned)((max_seqlen + 255) / 256), (unsigned)nseq);
#define IE_LF(DD, CA)                                                                                                                 \
    hipLaunchKernelGGL((flash_fwd64f_k<DD, CA, 4>), grid64, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, \
                       kv_ts, (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale)
        if (d == 128) { if (causal) IE_LF(128, true); else IE_LF(128, false); }
        else          { if (causal) IE_LF(64, true); else IE_LF(64, false); }
#undef IE_LF
        return ie_launch_status("ie_flash_attn_fwd launch");
    }
    if (fwd_variant > 0) {
        dim3 grid64((unsigned)hq, (unsigned)((max_seqlen + 255) / 256), (unsigned)nseq);
#define IE_L64(DD, CA, TH)                                                                                                           \
    hipLaunchKernelGGL((flash_fwd64_k<DD, CA, TH>), grid64, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, \
                       kv_ts, (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale)
#define IE_L64T(DD, CA) do { if (fwd_variant == 1) IE_L64(DD, CA, 0); else IE_L64(DD, CA, 4); } while (0)
        if (d == 128) { if (causal) IE_L64T(128, true); else IE_L64T(128, false); }
        else          { if (causal) IE_L64T(64, true); else IE_L64T(64, false); }
#undef IE_L64T
#undef IE_L64
        return ie_launch_status("ie_flash_attn_fwd launch");
    }
    dim3 grid((unsigned)hq, (unsigned)((max_seqlen + 127) / 128), (unsigned)nseq);
#define IE_L(DD, CA)                                                                                                              \
    hipLaunchKernelGGL((flash_fwd_k<DD, CA>), grid, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, \
                       (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale, (const int32_t*)nullptr, (int64_t)0)
    if (d == 128) { if (causal) IE_L(128, true); else IE_L(128, false); }
    else          { if (causal) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
    return ie_launch_status("ie_flash_attn_fwd launch");
}

// Full attention of a rectangle of scores per sequence: queries cu_q[s] .. cu_q[s + 1] (of Tq rows) against keys cu_k[s] .. cu_k[s + 1] (of Tk rows).
// A sequence without keys gives out = 0, lse = -inf.  The block of ring attention (internevo_amd/seqpar.py): a rank's queries of the sequence
// that spans a rank boundary against the part of that sequence another rank holds.
extern "C" int ie_flash_attn_fwd_x(const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts, void* out, int64_t o_ts, float* lse,
                                   const int32_t* cu_q, const int32_t* cu_k, int nseq, int64_t Tq, int64_t Tk, int max_seqlen_q, int hq, int hkv,
                                   int d, float softmax_scale, void* stream) {
    IE_CHECK_ARG(q && k && v && out && lse && cu_q && cu_k, "ie_flash_attn_fwd_x: null pointer");
    IE_CHECK_ARG(nseq >= 0 && Tq >= 0 && Tk >= 0 && max_seqlen_q >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0, "ie_flash_attn_fwd_x: bad shape");
    IE_CHECK_ARG(softmax_scale > 0.f, "ie_flash_attn_fwd_x: softmax_scale must be positive");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_flash_attn_fwd_x: head dim must be 64 or 128");
    IE_CHECK_SUPPORTED(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && q_ts % 8 == 0 && kv_ts % 8 == 0 && o_ts % 4 == 0,
                       "ie_flash_attn_fwd_x: pointers must be 16-byte aligned and token strides multiples of 8");
    if (nseq == 0 || Tq == 0 || max_seqlen_q == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)hq, (unsigned)((max_seqlen_q + 127) / 128), (unsigned)nseq);
#define IE_LX(DD)                                                                                                                      \
    hipLaunchKernelGGL((flash_fwd_k<DD, false>), grid, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, \
                       (bf16_t*)out, o_ts, lse, cu_q, Tq, hq, hkv, softmax_scale, cu_k, Tk)
    if (d == 128) IE_LX(128); else IE_LX(64);
#undef IE_LX
    return ie_launch_status("ie_flash_attn_fwd_x launch");
}

// tuning hook (A/B benchmarking only): kernel variant of the forward
extern "C" int ie_tune_flash_fwd_variant(int variant) {
#ifdef IE_ENABLE_ABLATIONS
    IE_CHECK_ARG(variant >= -1 && variant <= 60, "ie_tune_flash_fwd_variant: -1 (automatic), 0 .. 3 (10 + ABL: timing ablations of variant 3)");
#else
    // (forcing 3 at a softmax_scale other than ln 2 scales q inside the kernel: one more bf16 rounding of q than the backward kernels see -- correct to
    // bf16 rounding, tested in tests/test_kernels_gpu.py, but not what the automatic dispatch ever picks)
    IE_CHECK_ARG(variant >= -1 && variant <= 6, "ie_tune_flash_fwd_variant: -1 (automatic), 0 .. 6");
#endif
    g_fwd_variant = variant;
    return IE_OK;
}

