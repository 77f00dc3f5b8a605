// K1 backward: varlen, causal / full, GQA flash attention for gfx950.  Deterministic (no atomics):
//   1. flash_delta_k : delta[h][t] = sum_d dO * O; for the dK/dV kernel also -delta and -lse / scale
//   2. flash_dq_k    : one block per 128 query rows, loops over K/V tiles, recomputes P, dQ += dS K
//   3. flash_dkdv_k  : one block per 64 keys (one wave per 32 keys), loops over the q heads of the
//                      GQA group and their query tiles, recomputes P, dV += P^T dO, dK += dS^T Q
// The recompute costs 7 instead of 5 tile products but keeps every output written by exactly one
// workgroup in a fixed order, so the result is bit-reproducible run to run.
//
// Tiles arrive by LDS-DMA into natural-layout images (flash_common.h), two pipeline stages (four in the four-wave dK/dV block); one image
// serves both the row fragments (ds_read_b128) and the transposed fragments (ds_read_b64_tr_b16), so the dQ kernel
// stages only K and V, the dK/dV kernel only Q and dO.  In the dQ kernel scores are formed transposed
// (lane = query row), in the dK/dV kernel un-transposed (lane = key), so the accumulator registers of
// S / dS are directly the k-slots of the next MFMA.  Tile loops are unrolled by the pipeline stages:
// every LDS address is a precomputed per-lane offset + an immediate.  The dK/dV kernel is described at its definition; what was measured
// while it was rebuilt in round 2: profiles/r02_flash_dkdv_rework.md.

// Development switches of the dK/dV kernel (tools/kbench/mkvariant.sh builds a variant library with them; the product build defines none):
//   IE_DKDV_ROW_AHEAD / IE_DKDV_TR_AHEAD: LDS prefetch distances (k-steps / MFMA pairs)
#include "flash_common.h"

#include <type_traits>

//   IE_FLASH_ABLATE (results then wrong): 1 no transposed reads, 2 no softmax, 4 no row reads, 8 no S / dP MFMAs, 16 no dV / dK MFMAs,
//                   32 no tile wait + barrier, 64 no start-value reads
#ifndef IE_DKDV_ROW_AHEAD
#define IE_DKDV_ROW_AHEAD 2
#endif
#ifndef IE_DKDV_TR_AHEAD
#define IE_DKDV_TR_AHEAD 2
#endif
// IE_DKDV_TIMING (development only): s_memtime stamps at the phase boundaries of every tile, summed over all waves into g_dkdv_t and printed
// by the launcher: [issue + first fetches, A0, A1, C0, C1, end-of-tile wait + barrier, tiles]
#ifndef IE_DKDV_TIMING
#define IE_DKDV_TIMING 0
#endif
// IE_DKDV_REPEAT = 1 .. 4 (development only, results wrong): phase A0 / A1 / C0 / C1 of every tile runs twice (the transfers once), so the
// difference to the normal build is that phase's cost in place, with nothing perturbed
#ifndef IE_DKDV_REPEAT
#define IE_DKDV_REPEAT 0
#endif
#ifndef IE_FLASH_ABLATE
#define IE_FLASH_ABLATE 0
#endif
// cache policy of the dS^T spill stores (A/B builds): "" default, " nt" non-temporal, " sc0 sc1" write-through
#ifndef IE_SPILL_MOD
#define IE_SPILL_MOD ""
#endif
#if IE_DKDV_TIMING
__device__ unsigned long long g_dkdv_t[8];
// (the wait makes the stamp's registers valid before hipcc may copy or reuse them -- without it the late result lands in whatever the register
// holds by then -- but it also drains the LDS reads in flight: the phase behind a stamp loses part of its prefetch.  IE_DKDV_REPEAT measures
// without that.)
#define IE_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]) :: "memory")
#else
#define IE_STAMP(i)
#endif
namespace {

using namespace fa;

// one 8-element chunk of delta's dot product, every rounding written out: flash_delta_k and the dQ kernel's fused delta must give the same bits
__device__ __forceinline__ float dot8(const float* a, const float* b) {
#pragma clang fp contract(off)
    float acc = a[0] * b[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) acc = __builtin_fmaf(a[e], b[e], acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void flash_delta_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ out,
                                                     int64_t o_ts, const float* __restrict__ lse, float* __restrict__ delta,
                                                     float* __restrict__ nlse, float* __restrict__ ndelta, float inv_scale, int64_t T, int hq) {
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
        acc = dot8(a, b);
    }
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);   // (flash_dq_k's fused delta adds the sixteen chunk sums in exactly this tree)
    if (ok && sub == 0) {
        const int64_t t = pair / hq;
        const int h = (int)(pair % hq);
        delta[(int64_t)h * T + t] = acc;
        // the dK / dV kernel starts its S and dP accumulators from these: S' = q.k - lse / scale (so P = exp2(S' * scale * log2 e)), dP' = dP - delta
        nlse[(int64_t)h * T + t] = -lse[(int64_t)h * T + t] * inv_scale;
        ndelta[(int64_t)h * T + t] = -acc;
    }
}

// ------------------------------------------------------------------------------------------------
// FUSE_DELTA (round 6): the kernel computes delta = sum_d dO * O of its own query rows in its prologue -- a lane and its partner (lane ^ 32) hold the whole dO
// row as MFMA fragments already, the O row is read the same way -- and writes delta, -delta and -lse / scale for the dK / dV kernel behind it: flash_delta_k
// (54 us + a launch per call) is gone from the step.  The sixteen 8-element chunk sums are added in flash_delta_k's shuffle tree: the same bits.
// ROT (round 6; ie_flash_attn_bwd_qkv_rotary): dQ leaves through store_row_block_rot into the wqkv product's output-gradient layout (flash_common.h: FaRotOut).
template <int D, bool CAUSAL, int MINW, bool FUSE_DELTA, bool ROT = false>
__global__ __launch_bounds__(256, MINW) void flash_dq_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                        int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                        int64_t kv_ts, const float* __restrict__ lse, float* __restrict__ delta,
                                                        bf16_t* __restrict__ dq, int64_t dq_ts, const int32_t* __restrict__ cu, int64_t T,
                                                        int hq, int hkv, float scale, const int32_t* __restrict__ cu_k, int64_t Tk,
                                                        const bf16_t* __restrict__ out, int64_t o_ts, float* __restrict__ nlse,
                                                        float* __restrict__ ndelta, float inv_scale, FaRotOut ro) {
    // cu_k / Tk (full attention only; ie_flash_attn_bwd_x): the keys of sequence s are rows cu_k[s] .. cu_k[s + 1] of K / V tensors of Tk rows
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
    if (ntiles > 0) {   // (a sequence without keys: dQ = 0)
        ksrc.issue(smem, 0, 0, wave);
        vsrc.issue(smem + G::IMG_BYTES, 0, 0, wave);
    }

    FragOffs<D> fo;
    fo.init(lane);

    // Q, dO (and O for the fused delta) rows of this wave as fragments: whole-row pieces turned through the wave's quarter of LDS stage 1 (free until tile 0 requests
    // tile 1 into it; flash_common.h: load_row_frags_staged)
    constexpr int NROWT = FUSE_DELTA ? 3 : 2;
    s16x8 rowf[NROWT][G::KS];
    {
        static_assert(4 * 32 * 2 * D <= STAGE, "the four waves' row blocks in one stage");
        const bf16_t* qr = q + (int64_t)(tok0 + qw0) * q_ts + (int64_t)h * D;
        const bf16_t* dr = dout + (int64_t)(tok0 + qw0) * do_ts + (int64_t)h * D;
        const int nv = min(max(len - qw0, 0), 32);
        unsigned char* stg = smem + STAGE + wave * (32 * 2 * D);
        if constexpr (FUSE_DELTA) {
            const bf16_t* const row0[3] = {qr, dr, out + (int64_t)(tok0 + qw0) * o_ts + (int64_t)h * D};
            const int64_t tss[3] = {q_ts, do_ts, o_ts};
            load_row_frags_staged<D, 3>(row0, tss, nv, stg, lane, rowf);
        } else {
            const bf16_t* const row0[2] = {qr, dr};
            const int64_t tss[2] = {q_ts, do_ts};
            load_row_frags_staged<D, 2>(row0, tss, nv, stg, lane, rowf);
        }
    }
    s16x8 (&qf)[G::KS] = rowf[0];
    s16x8 (&dof)[G::KS] = rowf[1];
    const float lse_q = q_valid ? lse[(int64_t)h * T + tok0 + my_q] : INFINITY;
    const float lse2 = lse_q * kLog2e;
    float dlt;
    if constexpr (FUSE_DELTA) {
        static_assert(G::KS == 8 || G::KS == 4, "head dim 128 or 64");
        float pc[G::KS];   // chunk ks of this lane = chunk 2 ks + (lane >> 5) of the row
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            float a[8], b[8];
            union { s16x8 s; uint4 u; } d_, o_;
            d_.s = dof[ks];
            o_.s = rowf[NROWT - 1][ks];
            unpack8(d_.u, a);
            unpack8(o_.u, b);
            pc[ks] = dot8(a, b);
        }
        // flash_delta_k's tree over the row's chunks c (xor 8, 4, 2, 1 for D = 128; 4, 2, 1 for D = 64) with c = 2 ks + half: the last level is the partner lane
        float x;
        if constexpr (G::KS == 8) x = ((pc[0] + pc[4]) + (pc[2] + pc[6])) + ((pc[1] + pc[5]) + (pc[3] + pc[7]));
        else x = (pc[0] + pc[2]) + (pc[1] + pc[3]);
        dlt = x + __shfl_xor(x, 32, 64);
        if (q_valid && lane < 32) {
            const int64_t at = (int64_t)h * T + tok0 + my_q;
            delta[at] = dlt;
            nlse[at] = -lse_q * inv_scale;
            ndelta[at] = -dlt;
        }
        if (!q_valid) dlt = 0.f;
    } else {
        dlt = q_valid ? delta[(int64_t)h * T + tok0 + my_q] : 0.f;
    }
    const float sc2 = scale * kLog2e;
    // ROT: the row's position now (one register), so that the store's cos / sin reads are ONE round trip behind the last tile instead of two
    int64_t rot_p = 0;
    if constexpr (ROT) rot_p = q_valid ? ro.pos[tok0 + my_q] : 0;

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
            const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > lenk);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Vs, 32 * c, ks, fo), dof[ks], dp, 0, 0, 0);
                }
                if (need_mask) {  // wave-uniform; selects instead of per-element branches
                    const int lim = CAUSAL ? min(len - 1, my_q) : lenk - 1;
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

    if constexpr (ROT) {
        static_assert(4 * FA_ROT_STAGE_BYTES <= 2 * STAGE, "the cos / sin staging rows of four waves in the K / V stages");
        store_row_block_rot<D>(dq + (int64_t)(tok0 + qw0) * dq_ts + (int64_t)(h / ro.qpk) * ro.grp_stride + (int64_t)(h % ro.qpk) * D, dq_ts,
                               min(max(len - qw0, 0), 32), dqacc, scale, lane, ro.cs, ro.sn, (int)rot_p,
                               smem + wave * FA_ROT_STAGE_BYTES);   // (the last tile ended with a barrier: the stages are free)
    } else {
        store_row_block<D>(dq + (int64_t)(tok0 + my_q) * dq_ts + (int64_t)h * D, dqacc, scale, lane, q_valid, (dq_ts & 7) == 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Block = DKV_WAVES waves x 32 keys.  With causal masking the work of a key block falls linearly with its
// position, and there are only (len/64) x hkv blocks, so (a) blocks are small (64 keys) to get >= 2 per CU and
// (b) the block index is folded so that the two blocks dispatched far apart (the ones that share a CU under
// round-robin dispatch) are a heavy and a light one: kb(j) = j for the first half, mirrored for the second.  That holds while all blocks are
// resident at once; a grid of several rounds is dispatched heaviest-first instead (the folded order ended such a grid with medium blocks,
// a quarter of a CU's whole work each: ~8 % of the kernel at 4 x 4096 tokens).

// HS > 1 (causal balance): the q heads of a kv head are split over HS blocks, each writing fp32 partial dK / dV
// to part[2][HS][T][hkv][D]; flash_dkdv_reduce_k sums them in a fixed order (deterministic, no atomics).  With all blocks
// resident at once the kernel takes as long as its heaviest block (key block 0 sees every query tile: 2x the mean); HS x more,
// HS x smaller blocks dispatched heavy-first let the light ones back-fill the tail.
//
// SPILL (round 5): the kernel also writes dS^T = P^T o (dP^T - delta), the bf16 fragments it feeds its own dK product with, to `ds_ws`, and
// flash_dq_from_ds_k forms dQ = scale * dS K from them: five tile products for the whole backward instead of seven (flash_dq_k recomputes S and
// dP), paid for with 2 bytes per visible (query, key) pair written here and read there.  Layout: per (sequence, q head) a triangle (causal) or
// rectangle of 8-KiB IMAGES, one per (query tile of 64, key block of 64): image (qt, kb) at index qt (qt + 1) / 2 + kb (causal) or
// qt * nq_max + kb, `ds_head_stride` bytes per (sequence, head).  An image is copied into LDS linearly by the reader and is laid out for BOTH
// sides: (a) every store instruction of this kernel writes one contiguous KiB -- 8 whole cache lines; 32-byte pieces of 32 different lines, the
// natural [key][query] image, quadrupled the L2's write requests and cost the kernel 19 % (profiles/r05_flash_bwd_spill.md) -- and (b) the
// reader's transposing reads (ds_read_b64_tr_b16: every lane fetches 8 bytes = one key x four query columns) are bank-conflict free.  The 32
// keys of a wave own 4 KiB (image half (key >> 5) & 1) in four 1-KiB chunks, one per packed fragment dsf[qs][gp] (registers 8 gp .. 8 gp + 7 of
// sub-block qs: query rows 32 qs + 16 gp + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)); lane l's 16 bytes sit at position l ^ ((l >> 5) << 2) ^ (gp << 3)
// of the chunk (the XORs spread the 32 lanes of a read over all banks; flash_dq_from_ds_k::DsOffs is the inverse map).  Four 16-byte stores per
// wave and tile, from inline asm in the gaps of phase C1; the waits of the tile loop count them (vmcnt retires in order on gfx950: the
// transfers they wait for are older than the stores).
template <int D, bool CAUSAL, int HS, int DKV_WAVES, bool SPILL = false, bool ROT = false>
__global__ __launch_bounds__(64 * DKV_WAVES) void flash_dkdv_k(const bf16_t* __restrict__ dout, int64_t do_ts, const bf16_t* __restrict__ q,
                                                               int64_t q_ts, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               int64_t kv_ts, const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dkv_ts,
                                                               const int32_t* __restrict__ cu, int64_t T, int hq, int hkv, float scale,
                                                               float* __restrict__ part, unsigned char* __restrict__ ds_ws,
                                                               int64_t ds_head_stride, int nq_max, const int32_t* __restrict__ cu_k, int64_t Tk,
                                                               FaRotOut ro) {
    // ROT (round 6; HS == 1): dk / dv point at the k / v slots of kv group 0 inside the wqkv output-gradient rows, a kv head's slot is ro.grp_stride further;
    // dK leaves through store_row_block_rot (the key's position), dV as it is.
    // cu_k / Tk (full attention only; ie_flash_attn_bwd_x): the keys of sequence s are rows cu_k[s] .. cu_k[s + 1] of K / V / dK / dV tensors of Tk rows
    using G = Geo<D>;
    constexpr int STAGE = 2 * G::IMG_BYTES + 1024;  // Q image, dO image, -lse/scale [64], -delta [64] (+ pad to keep 1 KiB alignment)
    // Pipeline stages.  Four waves (one block per CU: the registers allow one wave per SIMD) keep FOUR and work EARLY: the transfers of tile
    // it + 2 are spread over all MFMA gaps of tile it (one 1-KiB piece per 8 MFMAs: the CU's vector-memory path takes 16 cycles per piece, and
    // four waves issuing a piece per 2 MFMAs at the same moment queued ~40 cycles per piece); the wait + barrier of a tile sit in front of its
    // last phase (C1), so the head of tile it + 1 (first row fragments, starting values) is fetched under C1's MFMAs instead of in front of an
    // idle matrix pipe.  That barrier proves every wave is through C0 of tile it, i.e. done with tile it - 1: the stage of tile it - 2, the
    // one tile it + 2 goes to, has been free for a whole tile (with three stages the target would be tile it - 1's, still read in its C1 by a
    // wave that is behind).  Two waves (two blocks per CU, 2 x 66 KB of LDS) keep two stages: the pieces of tile it + 1 go into the first half
    // of tile it, wait + barrier at its end.
    constexpr int NST = DKV_WAVES == 4 ? 4 : 2;
    constexpr bool EARLY = NST == 4;
    constexpr int PD = EARLY ? 2 : 1;   // tiles requested ahead
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int seq = blockIdx.z;
    // x = (key block, head split, kv head), kv head fastest: every (split, head) of the heaviest key block is dispatched first
    const int hk = blockIdx.x % hkv;
    const int hsi = (blockIdx.x / hkv) % HS;
    const int nkb = gridDim.x / (hkv * HS);
    const int j = blockIdx.x / (hkv * HS);
    // one round (every block resident at once: 256 CUs x 2 two-wave blocks or 1 four-wave block): folded order, see above; more rounds:
    // heaviest first (causal: the first keys see the most query tiles), so the blocks dispatched last are the light ones
    const bool one_round = gridDim.x * gridDim.z <= (DKV_WAVES == 4 ? 256u : 512u);
    const int first = (nkb + 1) / 2;
    const int kb = (HS > 1 || !one_round) ? j : (j < first ? j : nkb - 1 - (j - first));
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int tok0k = (!CAUSAL && cu_k) ? cu_k[seq] : tok0;
    const int lenk = (!CAUSAL && cu_k) ? cu_k[seq + 1] - tok0k : len;
    const int k0 = kb * 32 * DKV_WAVES;
    if (k0 >= lenk) return;
    const int grp = hq / hkv / HS;          // q heads handled by this block
    const int h_first = hk * (hq / hkv) + hsi * grp;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kw0 = k0 + wave * 32;
    const int my_k = kw0 + (lane & 31);
    const bool k_valid = my_k < lenk;

    const int nqt_all = (len + 63) / 64;
    const int qt_start = CAUSAL ? (k0 / 64) : 0;
    const int nqt = nqt_all - qt_start;  // >= 1 because k0 < len (0: a sequence without queries, ie_flash_attn_bwd_x only)
    const int nit = grp * nqt;
    if (nqt <= 0) {   // nobody looks at these keys: dK = dV = 0, and nothing is requested (the descriptors of an empty sequence describe nothing)
        if (HS == 1) {
            f32x16 z[G::DB];
#pragma unroll
            for (int db = 0; db < G::DB; ++db) z[db] = zero16();
            const int64_t hoff = ROT ? (int64_t)hk * ro.grp_stride : (int64_t)hk * D;
            store_row_block<D>(dk + (int64_t)(tok0k + my_k) * dkv_ts + hoff, z, 1.f, lane, k_valid, (dkv_ts & 7) == 0);
            store_row_block<D>(dv + (int64_t)(tok0k + my_k) * dkv_ts + hoff, z, 1.f, lane, k_valid, (dkv_ts & 7) == 0);
        }
        return;
    }

    // Q / dO descriptors end with this sequence: rows behind it arrive as zeros, and so do their lse2 / delta -- such a row gives S = 0,
    // P = exp2(0 - 0) = 1, dP = 0, dS = 1 * (0 - 0) = 0 and adds exact zeros to dV (P^T dO) and dK (dS^T Q), with no instruction spent on it
    TileSrc<D, DKV_WAVES> qsrc, dosrc;
    qsrc.init(q + (int64_t)tok0 * q_ts, q_ts, len, (int64_t)hq * D, wave, lane);
    dosrc.init(dout + (int64_t)tok0 * do_ts, do_ts, len, (int64_t)hq * D, wave, lane);
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // Every transfer of a stage is issued from inline asm (flash_common.h: issue_piece_asm); the only vmcnt waits of the tile loop are the
    // explicit ones at the end of a tile.  lse2 / delta of the 64 rows: one buffer_load_dword ... lds each (lane = row), wave 0 / wave 1.
    struct Next { int soff_q, soff_do; i32x4 rs_ld; int soff_ld; };
    // Tiles are visited query tile by query tile, from the LAST one down to the block's diagonal, the q heads of the group inside: every block of
    // a (sequence, kv head) -- they share an XCD's L2: x % hkv = kv head -- then asks for the same Q / dO tiles in the same order and, dispatched
    // together, at the same time; a block only stops earlier the later its keys are.  Head by head and upwards from the own diagonal, the order of
    // rounds 2 - 4, no two blocks were on the same tile: 34 % of the tile requests missed the L2 (profiles/r05_flash_bwd_spill.md).
    // (pl_h, pl_qt) = the next tile to request, advanced without a division
#ifndef IE_DKDV_ORDER   // A/B builds: 0 = the old order
#define IE_DKDV_ORDER 1
#endif
    constexpr bool QT_MAJOR = IE_DKDV_ORDER != 0;
    int pl_h = h_first, pl_qt = QT_MAJOR ? nqt_all - 1 : qt_start;
    auto plan = [&]() {
        Next n;
        const int h = min(pl_h, hq - 1);   // the requests behind the last tile may name a head this block does not own: any valid address will do
        const int q0 = max(pl_qt, qt_start) * 64;   // (likewise a query tile behind the block's last one)
        n.soff_q = q0 * qsrc.ts2 + h * D * 2;
        n.soff_do = q0 * dosrc.ts2 + h * D * 2;
        const float* lp = ((wave & 1) ? delta : lse) + (int64_t)h * T + tok0;   // even waves fetch -lse/scale, odd waves -delta (every wave one
        n.rs_ld[0] = (int)(uint32_t)(uintptr_t)lp;                               // piece: the counted wait at the end of a tile is the same for all)
        n.rs_ld[1] = (int)(((uintptr_t)lp >> 32) & 0xffff);
        n.rs_ld[2] = len * 4;
        n.rs_ld[3] = 0x00020000;
        n.soff_ld = q0 * 4;
        if (QT_MAJOR) { if (++pl_h == h_first + grp) { pl_h = h_first; --pl_qt; } }
        else if (++pl_qt == nqt_all) { pl_qt = qt_start; ++pl_h; }
        return n;
    };
    auto issue_ld = [&](const Next& n, uint32_t stage_lds) {   // one buffer_load_dword ... lds (lane = row) per wave
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                     :: "s"(stage_lds + 2 * G::IMG_BYTES + (wave & 1) * 256), "v"(lane * 4), "s"(n.rs_ld), "s"(n.soff_ld) : "memory");
    };
    constexpr int PERW = TileSrc<D, DKV_WAVES>::PERW;
    constexpr int NPIECE = 2 * PERW + 1;   // transfers per wave and tile
#pragma unroll
    for (int t = 0; t < PD; ++t) {    // tiles 0 .. PD - 1 up front
        const Next n = plan();
        issue_ld(n, smem_lds + t * STAGE);
#pragma unroll
        for (int pq = 0; pq < PERW; ++pq) {
            qsrc.issue_piece_asm(smem_lds + t * STAGE, n.soff_q, wave, pq);
            dosrc.issue_piece_asm(smem_lds + t * STAGE + G::IMG_BYTES, n.soff_do, wave, pq);
        }
    }

    FragOffs<D> fo;
    fo.init(lane);
    // LDS reads carry a 16-bit immediate offset: stages 2 and 3 (behind 64 KiB) are reached from a second set of per-lane offsets
    FragOffs<D> foh = fo;
    if (NST > 2) {
#pragma unroll
        for (int i = 0; i < G::KS; ++i) foh.row[i] += 2 * STAGE;
#pragma unroll
        for (int i = 0; i < G::DB; ++i) { foh.tr0[i] += 2 * STAGE; foh.tr1[i] += 2 * STAGE; }
    }

    // K, V fragments of this wave's 32 keys (B operands: lane = key)
    // (whole-row pieces turned through a stage the tiles requested up front do not use: flash_common.h: load_row_frags_staged)
    s16x8 kvf[2][G::KS];
    {
        static_assert(PD < NST && DKV_WAVES * 32 * 2 * D <= STAGE, "a free stage holds the waves' row blocks");
        const bf16_t* const row0[2] = {k + (int64_t)(tok0k + kw0) * kv_ts + (int64_t)hk * D, v + (int64_t)(tok0k + kw0) * kv_ts + (int64_t)hk * D};
        const int64_t tss[2] = {kv_ts, kv_ts};
        load_row_frags_staged<D, 2>(row0, tss, min(max(lenk - kw0, 0), 32), smem + (NST - 1) * STAGE + wave * (32 * 2 * D), lane, kvf);
    }
    s16x8 (&kf)[G::KS] = kvf[0];
    s16x8 (&vf)[G::KS] = kvf[1];
    const float sc2 = scale * kLog2e;
    f32x16 dkacc[G::DB], dvacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { dkacc[db] = zero16(); dvacc[db] = zero16(); }
    const int rl_lane = 4 * (lane >> 5);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // a use of the K / V fragments in front of the loop: hipcc does not read the asm wait above and would otherwise wait for their global
    // loads (vmcnt(0): everything, the asm transfers of the next tile included) at their first use INSIDE the tile loop, in every iteration
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+a"(kf[ks]), "+a"(vf[ks]));
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
#if IE_DKDV_TIMING
    unsigned tacc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int KS = G::KS, NTR = 2 * G::DB;      // MFMAs per accumulator in an A phase; (dV, dK) MFMA pairs in a C phase
    static_assert((2 * KS == 16 || 2 * KS == 8) && NTR == KS, "softmax_gap is written for 16 or 8 gaps per phase");
    constexpr int R = IE_DKDV_ROW_AHEAD + 1;        // row fragments in flight (k-steps ahead + 1)
    constexpr int LT = IE_DKDV_TR_AHEAD;            // transposed fragments in flight (MFMA pairs ahead)
    constexpr int SPAN = EARLY ? 2 * KS : KS;       // piece slots per image: EARLY Q over A0 + A1, dO over C0 + C1; else Q over A0, dO over A1
    // The head of a tile -- its first row fragments, the starting values of S0 / dP0 and their mask -- is state carried into step():
    // requested at the top of the tile (two stages) or under the previous tile's C1 MFMAs (EARLY).
    struct TileInfo { int q0; bool need_mask; int first_rel; const unsigned char* sp; };   // sp: SPILL, the image of this tile for this wave's keys
    int cur_qt = QT_MAJOR ? nqt_all - 1 : qt_start, cur_h = h_first;   // query tile / q head the next tile_info() describes
    const int first_q = (my_k >= lenk) ? 0x3fffffff : (CAUSAL ? my_k : 0);  // first query row that sees this lane's key
    // SPILL: the image of (head, query tile) for this wave's keys.  (A four-wave block's upper key half above the diagonal tile is all zeros and has no
    // image of its own: it goes to the spare image that ends every head's region -- the counted waits want the same number of stores from every wave
    // and tile.)  Recomputed once per query tile, advanced by one head's region in between: the scalar instructions of an address per store cost
    // the lone wave of a SIMD 8 % of the kernel (in-order issue: every one of them is a slot in front of the next MFMA).
    const int sp_kb = (k0 + 32 * wave) >> 6;
    auto sp_image = [&](int h, int qt) {
        const int64_t img_off = (CAUSAL && sp_kb > qt) ? ds_head_stride - 8192
                                                        : (CAUSAL ? (int64_t)qt * (qt + 1) / 2 + sp_kb : (int64_t)qt * nq_max + sp_kb) * 8192;
        return (const unsigned char*)ds_ws + ((int64_t)seq * hq + h) * ds_head_stride + img_off;
    };
    const unsigned char* sp_cur = SPILL ? sp_image(cur_h, cur_qt) : nullptr;
    auto tile_info = [&]() {
        TileInfo ti;
        ti.q0 = cur_qt * 64;
        ti.sp = sp_cur;
        if (QT_MAJOR) {
            if (++cur_h == h_first + grp) {
                cur_h = h_first;
                cur_qt = max(cur_qt - 1, qt_start);
                if (SPILL) sp_cur = sp_image(cur_h, cur_qt);
            } else if (SPILL) sp_cur += ds_head_stride;
        } else {
            if (++cur_qt == nqt_all) { cur_qt = qt_start; ++cur_h; }
            if (SPILL) sp_cur = sp_image(min(cur_h, hq - 1), cur_qt);
        }
        ti.need_mask = (CAUSAL && kw0 + 31 > ti.q0) || (kw0 + 32 > lenk);   // wave-uniform: the diagonal tile, the last keys
        ti.first_rel = first_q - ti.q0 - rl_lane;                           // first visible row, tile-local, relative to this lane's registers 0 .. 3
        return ti;
    };
    s16x8 rq[R], rdo[R];
    f32x16 s0, dp0;
    TileInfo ti;
    if (IE_FLASH_ABLATE & 4) for (int i = 0; i < R; ++i) { rq[i] = kf[i]; rdo[i] = vf[i]; }
    // (a stage is named by its index si: stage si lies at smem + si * STAGE; the reads address it as base(si) + offsets(si))
    auto base = [&](int si) { return smem + (si % 2) * STAGE; };
    auto offs = [&](int si) -> const FragOffs<D>& { return si >= 2 ? foh : fo; };
    const int bias_lane[2] = {rl_lane * 4, rl_lane * 4 + 2 * STAGE};
    // SPILL: this lane's four 16-byte slots inside an image, and the image of (head, query tile) for this wave's keys
    int sp_off[2][2];
    if (SPILL) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp)
                sp_off[qs][gp] = ((my_k >> 5) & 1) * 4096 + (2 * qs + gp) * 1024 + ((lane ^ ((lane >> 5) << 2) ^ (gp << 3)) << 4);
    }
    // store i = 0 .. 3 of a tile: fragment gp = i & 1 of sub-block qs = i >> 1 (one store per MFMA gap: a 1-KiB store takes the wave's issue for
    // about as long as a transfer piece does)
    auto spill_one = [&](const TileInfo& t, const s16x8& frag, int i) {
        if (!SPILL) return;
        asm volatile("global_store_dwordx4 %0, %1, %2" IE_SPILL_MOD :: "v"(sp_off[i >> 1][i & 1]), "v"(frag), "s"(t.sp) : "memory");
    };
#ifndef IE_SPILL_NFRAG   // A/B builds (results wrong below 4): how many of the four fragments of a tile are stored
#define IE_SPILL_NFRAG 4
#endif
    constexpr int NSPILL = (SPILL && !EARLY) ? IE_SPILL_NFRAG : 0;   // stores of this tile still counted at its wait (EARLY waits in front of C1, where they are issued)
    auto fetch_rows = [&](int si, int u) {   // u = 0 .. 2 KS - 1: k-step u % KS of sub-block u / KS
        if (IE_FLASH_ABLATE & 4) return;
        rq[u % R] = row_frag<D>(base(si), 32 * (u / KS), u % KS, offs(si));
        rdo[u % R] = row_frag<D>(base(si) + G::IMG_BYTES, 32 * (u / KS), u % KS, offs(si));
    };
    // S / dP accumulators start from -lse / scale and -delta of their rows (registers 4 g .. 4 g + 3 = rows 8 g + 4 (lane / 32) + 0 .. 3 of the
    // sub-block: one ds_read_b128 each), S in a tile that needs the mask from -1e30 where the key does not see the row: P = exp2(S' * sc2)
    // and dS = P * dP' then take mul, exp2, mul and half a pack per element
    auto load_bias = [&](int si, f32x16& sv, f32x16& dpv, int qs) {
        const unsigned char* lse_s = base(si) + 2 * G::IMG_BYTES + bias_lane[si >= 2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rb = (32 * qs + 8 * g) * 4;
            const float4 a = (IE_FLASH_ABLATE & 64) ? make_float4(-1.f, -2.f, -3.f, -4.f) : *reinterpret_cast<const float4*>(lse_s + rb);
            const float4 b = (IE_FLASH_ABLATE & 64) ? make_float4(.1f, .2f, .3f, .4f) : *reinterpret_cast<const float4*>(lse_s + 256 + rb);
            sv[4 * g + 0] = a.x; sv[4 * g + 1] = a.y; sv[4 * g + 2] = a.z; sv[4 * g + 3] = a.w;
            dpv[4 * g + 0] = b.x; dpv[4 * g + 1] = b.y; dpv[4 * g + 2] = b.z; dpv[4 * g + 3] = b.w;
        }
    };
    auto apply_mask = [&](const TileInfo& t, f32x16& sv, int qs) {
        if (t.need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = (32 * qs + 8 * (r / 4) + r % 4) < t.first_rel ? -1e30f : sv[r];
        }
    };
    // number of this tile's transfers issued before slot `upto`
    auto pieces_before = [](int upto) constexpr {
        int n = 1;
        for (int j = 0; j < upto && j < 2 * SPAN; ++j) n += ((j % SPAN) * PERW) % SPAN == 0;
        return n;
    };

    // One query tile (64 rows = two 32-row sub-blocks) of one q head against this wave's 32 keys.  The serial chain
    //   S / dP MFMAs -> softmax -> packing -> dV / dK MFMAs
    // of the two sub-blocks is interleaved:
    //   A0: S0 / dP0 MFMAs            | row fragments fetched two k-steps ahead; starting values of S1 / dP1
    //   A1: S1 / dP1 MFMAs            | softmax of sub-block 0, one register pair per two gaps
    //   C0: dV / dK MFMAs of block 0  | softmax of sub-block 1 likewise; transposed fragments fetched two MFMA pairs ahead
    //   C1: dV / dK MFMAs of block 1  | EARLY: the head of the next tile
    // and the transfers of a later tile go into the gaps one piece at a time.  MFMAs are asm with explicit register files (flash_common.h):
    // dK / dV accumulators and the K / V fragments live in AGPRs, the scores in arch VGPRs.  Every gap is pinned by sched_barrier.  A
    // sub-block that is masked entirely or lies behind the sequence is computed like the others and contributes zeros.
    auto step = [&](auto stage_c, int it) {
        constexpr int S = decltype(stage_c)::value;
#if IE_DKDV_TIMING
        unsigned long long ts[7];
#endif
        IE_STAMP(0);
        constexpr int NS = (S + 1) % NST;               // the next tile's stage
        if (!EARLY) {
            ti = tile_info();
#pragma unroll
            for (int u = 0; u < R - 1; ++u) fetch_rows(S, u);
            load_bias(S, s0, dp0, 0);
            apply_mask(ti, s0, 0);
        }
        const TileInfo tc = ti;                          // this tile (ti is overwritten with the next tile's under C1 when EARLY)
        const Next nx = plan();                          // tile it + PD goes into a stage no wave reads any more, its pieces into the MFMA gaps
        const uint32_t nxt_lds = smem_lds + ((S + PD) % NST) * STAGE;
        issue_ld(nx, nxt_lds);
        auto piece_slot = [&](int j) {                   // slot j = 0 .. 4 KS - 1 (A0, A1, C0, C1: one per k-step / MFMA pair)
            if (j >= 2 * SPAN) return;
            const int jj = j % SPAN;
            if ((jj * PERW) % SPAN != 0) return;
            if (j < SPAN) qsrc.issue_piece_asm(nxt_lds, nx.soff_q, wave, jj * PERW / SPAN);
            else dosrc.issue_piece_asm(nxt_lds + G::IMG_BYTES, nx.soff_do, wave, jj * PERW / SPAN);
        };
        const unsigned char* Qs = base(S);
        const unsigned char* dOs = Qs + G::IMG_BYTES;
        f32x16 s1, dp1;
        auto MS = [&](f32x16& d, const s16x8& a, s16x8& b_acc) { if (!(IE_FLASH_ABLATE & 8)) mfma_s(d, a, b_acc); };
        auto MO = [&](f32x16& d_acc, const s16x8& a, const s16x8& b) { if (!(IE_FLASH_ABLATE & 16)) mfma_o(d_acc, a, b); };
        s16x8 pf0[2], dsf0[2], pf1[2], dsf1[2];
        // The softmax of a sub-block, one register pair (r, r + 1) at a time, in two halves that sit in different MFMA gaps (asm: the
        // instructions stay where they are written, dependent ones never adjacent, no compiler-inserted wait states):
        //   half A: x = S' * sc2 (x2), P = exp2(x) (x2)        half B: dS = P * dP' (x2), bf16 pack of P and of dS
        // D = 128 has 16 gaps per phase for the 8 pairs (A and B alternate); D = 64 has 8 gaps: A and B of a pair share a gap.
        float pe0 = 0.f, pe1 = 0.f;
        const float sc2v = sc2;   // (a copy in this scope: clang does not capture the outer variable for an asm operand of a nested lambda)
        auto half_a = [&](const f32x16& sv, int p) {
            if (IE_FLASH_ABLATE & 2) return;
            asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1"
                         : "=&v"(pe0), "=&v"(pe1) : "v"(sv[2 * p]), "v"(sv[2 * p + 1]), "v"(sc2v));
        };
        auto half_b = [&](const f32x16& dpv, s16x8 (&pf)[2], s16x8 (&dsf)[2], int p) {
            if (IE_FLASH_ABLATE & 2) return;
            unsigned pp, dd;
            float t0, t1;
            asm volatile("v_mul_f32 %2, %4, %6\n\tv_mul_f32 %3, %5, %7\n\tv_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %2, %3"
                         : "=&v"(pp), "=&v"(dd), "=&v"(t0), "=&v"(t1) : "v"(pe0), "v"(pe1), "v"(dpv[2 * p]), "v"(dpv[2 * p + 1]));
            union { uint4 u; s16x8 x; } a, b;   // registers 2 p, 2 p + 1 -> 32-bit word p % 4 of fragment p / 4
            a.x = pf[p / 4];
            b.x = dsf[p / 4];
            const int w = p % 4;
            if (w == 0) { a.u.x = pp; b.u.x = dd; } else if (w == 1) { a.u.y = pp; b.u.y = dd; } else if (w == 2) { a.u.z = pp; b.u.z = dd; } else { a.u.w = pp; b.u.w = dd; }
            pf[p / 4] = a.x;
            dsf[p / 4] = b.x;
        };
        auto softmax_gap = [&](const f32x16& sv, const f32x16& dpv, s16x8 (&pf)[2], s16x8 (&dsf)[2], int j, int gaps) {   // gap j of a phase
            if (gaps == 16) {
                if (j % 2 == 0) half_a(sv, j / 2);
                else half_b(dpv, pf, dsf, j / 2);
            } else {   // 8 gaps
                half_a(sv, j);
                asm volatile("s_nop 0");   // a transcendental's result needs one wait state before a vector ALU instruction reads it
                half_b(dpv, pf, dsf, j);
            }
        };
        s16x8 tfa[LT], tfb[LT];                         // transposed fragments in flight: [m % LT] for dV (dO^T) and dK (Q^T)
        if (IE_FLASH_ABLATE & 1) for (int i = 0; i < LT; ++i) { tfa[i] = kf[i]; tfb[i] = vf[i]; }
        if (IE_FLASH_ABLATE & 2) { pf0[0] = pf0[1] = pf1[0] = pf1[1] = kf[2]; dsf0[0] = dsf0[1] = dsf1[0] = dsf1[1] = vf[2]; }
        auto fetch_tr = [&](int m, int qs) {            // m = 0 .. NTR - 1: (s2, db) = (m / DB, m % DB) of sub-block qs
            if (IE_FLASH_ABLATE & 1) return;
            tfa[m % LT] = trans_frag<D>(dOs, m % G::DB, 2 * qs + m / G::DB, offs(S));
            tfb[m % LT] = trans_frag<D>(Qs, m % G::DB, 2 * qs + m / G::DB, offs(S));
        };
        __builtin_amdgcn_sched_barrier(0);
        IE_STAMP(1);
        // A0 (the fragments of sub-block 1's first k-steps are requested under its last MFMAs)
#pragma unroll
        for (int rep = 0; rep < (IE_DKDV_REPEAT == 1 ? 2 : 1); ++rep)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            fetch_rows(S, ks + R - 1);
            if (rep == 0) piece_slot(ks);
            __builtin_amdgcn_sched_barrier(0);
            MS(s0, rq[ks % R], kf[ks]);
            MS(dp0, rdo[ks % R], vf[ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == KS / 2 - 1) load_bias(S, s1, dp1, 1);   // sub-block 1's starting values: requested half way, masked (diagonal tiles only) at the end
            if (ks == KS - 1) apply_mask(tc, s1, 1);
        }
        IE_STAMP(2);
        // A1 | softmax of sub-block 0
#pragma unroll
        for (int rep = 0; rep < (IE_DKDV_REPEAT == 2 ? 2 : 1); ++rep)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = KS + ks;
            if (ks + R - 1 < KS) fetch_rows(S, u + R - 1);
            if (rep == 0) piece_slot(KS + ks);
            __builtin_amdgcn_sched_barrier(0);
            MS(s1, rq[u % R], kf[ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) asm volatile("s_nop 11" : "+v"(s0), "+v"(dp0));   // S0 / dP0: 12 wait states behind their last MFMA before the vector ALU reads them
            softmax_gap(s0, dp0, pf0, dsf0, 2 * ks, 2 * KS);
            __builtin_amdgcn_sched_barrier(0);
            MS(dp1, rdo[u % R], vf[ks]);
            __builtin_amdgcn_sched_barrier(0);
            softmax_gap(s0, dp0, pf0, dsf0, 2 * ks + 1, 2 * KS);
            if (ks >= KS - LT) fetch_tr(ks - (KS - LT), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        IE_STAMP(3);
        // C0 | softmax of sub-block 1
#pragma unroll
        for (int rep = 0; rep < (IE_DKDV_REPEAT == 3 ? 2 : 1); ++rep)
#pragma unroll
        for (int m = 0; m < NTR; ++m) {
            if (rep == 0) piece_slot(2 * KS + m);
            __builtin_amdgcn_sched_barrier(0);
            MO(dvacc[m % G::DB], tfa[m % LT], pf0[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0) asm volatile("s_nop 11" : "+v"(s1), "+v"(dp1));
            softmax_gap(s1, dp1, pf1, dsf1, 2 * m, 2 * NTR);
            __builtin_amdgcn_sched_barrier(0);
            MO(dkacc[m % G::DB], tfb[m % LT], dsf0[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
            softmax_gap(s1, dp1, pf1, dsf1, 2 * m + 1, 2 * NTR);
            if (m + LT < NTR) fetch_tr(m + LT, 0);
            else fetch_tr(m + LT - NTR, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        IE_STAMP(4);
        if (EARLY && !(IE_FLASH_ABLATE & 32)) {
            // EARLY: the next tile's pieces (requested a tile ago) have landed -- only this tile's, all younger, may still be in flight -- and
            // every wave is done with the previous tile's stage.  The next tile's head goes under the MFMAs of C1.
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(pieces_before(3 * KS) + NSPILL) : "memory");
            __syncthreads();
        }
        if (EARLY) ti = tile_info();
        // C1
#pragma unroll
        for (int rep = 0; rep < (IE_DKDV_REPEAT == 4 ? 2 : 1); ++rep)
#pragma unroll
        for (int m = 0; m < NTR; ++m) {
            if (rep == 0) piece_slot(3 * KS + m);
            if (EARLY && rep == 0) {
                if (m < R - 1) fetch_rows(NS, m);
                if (m == R - 1) load_bias(NS, s0, dp0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            MO(dvacc[m % G::DB], tfa[m % LT], pf1[m / G::DB]);
            MO(dkacc[m % G::DB], tfb[m % LT], dsf1[m / G::DB]);
            __builtin_amdgcn_sched_barrier(0);
            if (SPILL && rep == 0 && m % (NTR / 4) == 0) {   // the four dS^T stores of this tile, spread over C1's gaps (they carry nothing else but fragment reads)
                const int i = m / (NTR / 4);
                if (i < IE_SPILL_NFRAG) spill_one(tc, i == 0 ? dsf0[0] : i == 1 ? dsf0[1] : i == 2 ? dsf1[0] : dsf1[1], i);
            }
            if (m + LT < NTR) fetch_tr(m + LT, 1);
            if (EARLY && rep == 0 && m == NTR - 1) apply_mask(ti, s0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        IE_STAMP(5);
        if (!EARLY && !(IE_FLASH_ABLATE & 32)) {
            // the next tile's pieces have landed (the spill stores behind them may still be on their way); every wave is done with this tile's stage
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSPILL) : "memory");
            __syncthreads();
        }
#if IE_DKDV_TIMING
        IE_STAMP(6);
#pragma unroll
        for (int i = 0; i < 6; ++i) tacc[i] += (unsigned)(ts[i + 1] - ts[i]);
        tacc[6] += 1;
#endif
    };
    auto pin = [&]() {
#pragma unroll
        for (int db = 0; db < G::DB; ++db) pin_acc(dkacc[db], dvacc[db]);
    };
    if (EARLY) {   // the head of tile 0
        ti = tile_info();
#pragma unroll
        for (int u = 0; u < R - 1; ++u) fetch_rows(0, u);
        load_bias(0, s0, dp0, 0);
        apply_mask(ti, s0, 0);
    }
    for (int it = 0; it < nit; it += NST) {
        pin();
        step(std::integral_constant<int, 0>{}, it);
        pin();
        if (it + 1 < nit) step(std::integral_constant<int, 1>{}, it + 1);
        pin();
        if (NST == 4) {
            if (it + 2 < nit) step(std::integral_constant<int, 2 % NST>{}, it + 2);
            pin();
            if (it + 3 < nit) step(std::integral_constant<int, 3 % NST>{}, it + 3);
            pin();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no transfer into this block's LDS may outlive the block
#pragma unroll
    for (int db = 0; db < G::DB; ++db) mfma_settle_acc(dkacc[db], dvacc[db]);
#if IE_DKDV_TIMING
    if (lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&g_dkdv_t[i], (unsigned long long)tacc[i]);
#endif

    if constexpr (ROT) {
        static_assert(HS == 1, "the rotating store writes final gradients");
        const int64_t hoff = (int64_t)hk * ro.grp_stride;
        const int64_t p = k_valid ? ro.pos[tok0k + my_k] : 0;
        static_assert(DKV_WAVES * FA_ROT_STAGE_BYTES <= NST * STAGE, "the cos / sin staging rows in the Q / dO stages");
        __syncthreads();   // (every wave is through its last tile: the stages are free)
        const int nvk = min(max(lenk - kw0, 0), 32);
        store_row_block_rot<D>(dk + (int64_t)(tok0k + kw0) * dkv_ts + hoff, dkv_ts, nvk, dkacc, scale, lane, ro.cs, ro.sn, (int)p, smem + wave * FA_ROT_STAGE_BYTES);
        store_rows_staged<D>(dv + (int64_t)(tok0k + kw0) * dkv_ts + hoff, dkv_ts, nvk, dvacc, 1.f, lane, smem + wave * FA_ROT_STAGE_BYTES);
    } else if (HS == 1) {
        store_row_block<D>(dk + (int64_t)(tok0k + my_k) * dkv_ts + (int64_t)hk * D, dkacc, scale, lane, k_valid, (dkv_ts & 7) == 0);
        store_row_block<D>(dv + (int64_t)(tok0k + my_k) * dkv_ts + (int64_t)hk * D, dvacc, 1.f, lane, k_valid, (dkv_ts & 7) == 0);
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

// ------------------------------------------------------------------------------------------------
// dQ from the spilled dS^T images (SPILL mode of flash_dkdv_k, which describes the layout): dQ^T[d][q] = scale * sum_keys K^T[d][key] dS^T[key][q].
// One block = 4 waves = the (up to) four q heads of a kv head for one query tile of 64 rows (fewer heads per kv head: the waves split into
// heads x consecutive query tiles), so one K tile per 64 keys, brought in by all four waves, serves four 8-KiB dS^T images, one per wave, which
// nobody else ever reads: the kernel streams its operand exactly once and is bound by that stream (2 bytes per visible (query, key) pair;
// its 8 MFMAs per wave, image and 16-key step are a quarter of what the matrix pipe could do in the time the bytes take).  Three LDS stages,
// two tiles in flight; every transfer is inline asm (hipcc would drain a transfer it knows about in front of the next LDS read) and the
// waits count them.  A operand = K^T fragment (lane = d), B operand = dS^T fragment (lane = query column), both by transposing reads with
// the same key order, so the accumulators are dQ^T blocks in the layout store_row_block() writes (lane = row).
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void flash_dq_from_ds_k(const unsigned char* __restrict__ ds_ws, int64_t ds_head_stride, int nq_max,
                                                             const bf16_t* __restrict__ k, int64_t kv_ts, bf16_t* __restrict__ dq, int64_t dq_ts,
                                                             const int32_t* __restrict__ cu, int64_t T, int hq, int hkv, float scale, int gw) {
    using G = Geo<D>;
    constexpr int NST = 3;
    constexpr int KIMG = G::IMG_BYTES, DIMG = 8192;
    constexpr int STAGE = KIMG + 4 * DIMG;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int seq = blockIdx.z;
    const int group = hq / hkv, qw = 4 / gw;
    const int hk = blockIdx.x % hkv, gc = blockIdx.x / hkv;
    const int qblk = gridDim.y - 1 - blockIdx.y;          // heaviest (last) query tiles first
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    if (qblk * qw * 64 >= len) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = hk * group + gc * gw + (wave % gw);
    const int qt = qblk * qw + wave / gw;
    const int nkb_seq = (len + 63) / 64;
    const bool wave_on = qt * 64 < len;
    const int nkb_w = !wave_on ? 0 : (CAUSAL ? qt + 1 : nkb_seq);                          // key blocks this wave works on
    const int nkb = CAUSAL ? min(qblk * qw + qw, nkb_seq) : nkb_seq;                        // key blocks the block streams

    // (K tiles through registers instead -- buffer_load, ds_write_b128 behind the tile's products -- to take them off the LDS-DMA path did not make
    // the kernel faster: 525 vs 520 us; profiles/r05_flash_bwd_spill.md)
    TileSrc<D, 4> ksrc;
    ksrc.init(k + (int64_t)tok0 * kv_ts + (int64_t)hk * D, kv_ts, T - tok0, D, wave, lane);
    constexpr int PERW = TileSrc<D, 4>::PERW;
    constexpr int NP = PERW + 8;                                                            // transfers per wave and tile
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    i32x4 rs;                                                                                // this wave's head: its images
    {
        const unsigned char* hb = ds_ws + ((int64_t)seq * hq + h) * ds_head_stride;
        rs[0] = (int)(uint32_t)(uintptr_t)hb;
        rs[1] = (int)(((uintptr_t)hb >> 32) & 0xffff);
        rs[2] = (int)(uint32_t)ds_head_stride;
        rs[3] = 0x00020000;
    }
    const int64_t img0 = CAUSAL ? (int64_t)qt * (qt + 1) / 2 : (int64_t)qt * nq_max;       // first image of this wave's query tile
    const int lane16 = lane * 16;
    auto issue_tile = [&](int kb, int st) {   // key block kb -> stage st: this wave's share of the K tile and, if it works on kb, its own dS^T image
        const uint32_t sb = smem_lds + st * STAGE;
        const int ksoff = kb * 64 * ksrc.ts2;
#pragma unroll
        for (int pq = 0; pq < PERW; ++pq) ksrc.issue_piece_asm(sb, ksoff, wave, pq);
        if (kb < nkb_w) {
            const int dsoff = (int)((img0 + kb) * DIMG);
#pragma unroll
            for (int pc = 0; pc < 8; ++pc)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(sb + KIMG + wave * DIMG + pc * 1024), "v"(lane16), "s"(rs), "s"(dsoff + pc * 1024) : "memory");
        }
    };
    issue_tile(0, 0);
    if (nkb > 1) issue_tile(1, 1);

    FragOffs<D> fk;
    fk.init(lane);
    // B fragment (lane = query column 32 X + (lane & 31), contraction over the keys of 16-key step st in the K^T fragment's order: slot e <-> key
    // 16 st + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)) from a dS^T image: lane p of a 16-lane group fetches the 8 bytes (key kq = p >> 2; columns
    // 4 u .. 4 u + 3 of the group's 16, u = p & 3) that writer lane kl + 32 hi stored as half g1 of its fragment gp = a -- see flash_dkdv_k
    int ds_b0, ds_b1;
    {
        const int gq = lane >> 4, hh = gq >> 1, a = gq & 1, pp = lane & 15, kq = pp >> 2, u = pp & 3, g1 = u >> 1, hi = u & 1;
        const int pos = kq | ((hh ^ hi) << 2) | (a << 3) | (hi << 5);
        ds_b0 = a * 1024 + 16 * pos + 8 * g1;
        ds_b1 = ds_b0 ^ 128;   // the second read's keys (+ 8): bit 3 of the position
    }
    auto ds_frag = [&](const unsigned char* img, int X, int st) {
        const unsigned char* base = img + X * 2048 + (st & 1) * 256 + (st >> 1) * 4096;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + ds_b0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + ds_b1));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    };
    f32x16 acc[2][G::DB];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int db = 0; db < G::DB; ++db) acc[X][db] = zero16();

    auto tile = [&](auto stage_c, int kb) {
        constexpr int S = decltype(stage_c)::value;
        // tile kb has landed (this wave's share); tile kb + 1, if there is one, may still be on its way: K pieces, and the image if the wave works on it
        if (kb + 1 >= nkb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kb + 1 < nkb_w) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PERW) : "memory");
        __syncthreads();                                               // ... everybody's share; and every wave is done with tile kb - 1's stage,
        if (kb + 2 < nkb) issue_tile(kb + 2, (S + 2) % NST);           // which tile kb + 2 goes to
        if (kb < nkb_w) {
            const unsigned char* Kimg = smem + S * STAGE;
            const unsigned char* Dimg = Kimg + KIMG + wave * DIMG;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const s16x8 b0 = ds_frag(Dimg, 0, st), b1 = ds_frag(Dimg, 1, st);
#pragma unroll
                for (int db = 0; db < G::DB; ++db) {
                    const s16x8 a = trans_frag<D>(Kimg, db, st, fk);
                    acc[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0][db], 0, 0, 0);
                    acc[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1][db], 0, 0, 0);
                }
            }
        }
    };
    for (int kb = 0; kb < nkb; kb += NST) {
        tile(std::integral_constant<int, 0>{}, kb);
        if (kb + 1 < nkb) tile(std::integral_constant<int, 1>{}, kb + 1);
        if (kb + 2 < nkb) tile(std::integral_constant<int, 2>{}, kb + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no transfer into this block's LDS may outlive the block

#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const int qrow = qt * 64 + 32 * X + (lane & 31);
        const bool valid = wave_on && qrow < len;
        store_row_block<D>(dq + (int64_t)(tok0 + (valid ? qrow : 0)) * dq_ts + (int64_t)h * D, acc[X], scale, lane, valid, (dq_ts & 7) == 0);
    }
}


inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

int g_dkdv_split = 0;  // 0 = automatic (see dkdv_split), else the forced head split of the dK/dV kernel
int g_dkdv_waves = 2;  // waves (x 32 keys) per dK/dV block
int g_dq_minw = 2;  // waves/SIMD the dQ kernel is compiled for (2: 256 VGPRs with a small spill; 1: no spill, half the occupancy)
int g_bwd_separate_delta = 0;  // 1: flash_delta_k in front of the dQ kernel instead of the delta in its prologue (ie_tune_flash_bwd_variant bit 2: A/B, tests)
int g_bwd_spill = 0;  // 1: five-product backward (dS^T spilled by the dK/dV kernel, dQ formed from it): ie_tune_flash_bwd_variant bit 1, opt-in
void* g_ds_ws = nullptr;     // the caller's spill buffer (ie_flash_attn_bwd_set_spill)
int64_t g_ds_ws_bytes = 0;

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
    return 3ll * hq * T + 2ll * 4 * T * hkv * d;  // delta, -lse / scale, -delta + the largest set of dK/dV partials
}

// cu_k == NULL: self-attention (ie_flash_attn_bwd); else the rectangle per sequence of ie_flash_attn_bwd_x (full attention; keys cu_k[s] .. cu_k[s + 1] of
// Tk-row K / V / dK / dV tensors, at most max_seqlen_k per sequence): the seven-product path without a head split
static int flash_bwd_impl(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts, const void* out,
                          int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_ts, void* dk, void* dv, int64_t dkv_ts,
                          const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv, int d, float softmax_scale, int causal,
                          void* stream, const int32_t* cu_k, int64_t Tk, int max_seqlen_k, const FaRotOut* rot = nullptr) {
    IE_CHECK_ARG(dout && q && k && v && out && lse && delta && dq && dk && dv && cu_seqlens, "ie_flash_attn_bwd: null pointer");
    IE_CHECK_ARG(nseq >= 0 && T >= 0 && max_seqlen >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0, "ie_flash_attn_bwd: bad shape");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_flash_attn_bwd: head dim must be 64 or 128");
    IE_CHECK_SUPPORTED(aligned16(dout) && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && aligned16(dq) &&
                           aligned16(dk) && aligned16(dv) && do_ts % 8 == 0 && q_ts % 8 == 0 && kv_ts % 8 == 0 && o_ts % 8 == 0 &&
                           dq_ts % 4 == 0 && dkv_ts % 4 == 0,
                       "ie_flash_attn_bwd: pointers must be 16-byte aligned and token strides multiples of 8");
    if (nseq == 0 || T == 0 || max_seqlen == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    // workspace layout: delta[hq*T] | -lse/scale [hq*T] | -delta [hq*T] | partial dK [hs][T][hkv][d] | partial dV [hs][T][hkv][d]
    float* nlse = delta + (int64_t)hq * T;
    float* ndelta = nlse + (int64_t)hq * T;
    const int nq_max = (max_seqlen + 63) / 64;
    const int64_t ds_head_stride = ((causal ? (int64_t)nq_max * (nq_max + 1) / 2 : (int64_t)nq_max * nq_max) + 1) * 8192;
    const unsigned nkb0 = (unsigned)(((cu_k ? max_seqlen_k : max_seqlen) + 32 * g_dkdv_waves - 1) / (32 * g_dkdv_waves));
    const int hs0 = cu_k ? 1 : dkdv_split(hq, hkv, causal, (int64_t)nkb0 * hkv * nseq);
    const bool spill = g_bwd_spill && !cu_k && hs0 == 1 && ds_head_stride < (1ll << 31) && g_ds_ws && (int64_t)nseq * hq * ds_head_stride <= g_ds_ws_bytes;
    const bool fuse_delta = !spill && !g_bwd_separate_delta;   // (the five-product path runs the dK / dV kernel FIRST: it keeps the delta kernel)
    const FaRotOut ro = rot ? *rot : FaRotOut{nullptr, nullptr, nullptr, 1, 0};
    if (rot) {   // (ie_flash_attn_bwd_qkv_rotary_is_fused says when)
        IE_CHECK_SUPPORTED(d == 128 && causal && !cu_k && hs0 == 1 && !spill && fuse_delta && dq_ts % 8 == 0 && dkv_ts % 8 == 0,
                           "ie_flash_attn_bwd_qkv_rotary: head dim 128, causal, no head split of the dK / dV kernel, the default backward path");
    }
    if (!fuse_delta) {
        const int64_t threads = T * hq * (d / 8);
        dim3 grid((unsigned)((threads + 255) / 256));
        if (d == 128)
            hipLaunchKernelGGL((flash_delta_k<128>), grid, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)out, o_ts, lse, delta, nlse, ndelta,
                               1.f / softmax_scale, T, hq);
        else
            hipLaunchKernelGGL((flash_delta_k<64>), grid, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)out, o_ts, lse, delta, nlse, ndelta,
                               1.f / softmax_scale, T, hq);
    }
    const unsigned nt128 = (unsigned)((max_seqlen + 127) / 128);
    dim3 gq((unsigned)hq, nt128, (unsigned)nseq);
    const int DKV_WAVES = g_dkdv_waves;
    const unsigned nkb = (unsigned)(((cu_k ? max_seqlen_k : max_seqlen) + 32 * DKV_WAVES - 1) / (32 * DKV_WAVES));
    const int hs = cu_k ? 1 : dkdv_split(hq, hkv, causal, (int64_t)nkb * hkv * nseq);
    dim3 gk(nkb * (unsigned)hkv * (unsigned)hs, 1, (unsigned)nseq);
    float* part = ndelta + (int64_t)hq * T;
#define IE_DKDV_W(DD, CA, HS_, NW_)                                                                                                \
    hipLaunchKernelGGL((flash_dkdv_k<DD, CA, HS_, NW_>), gk, dim3(64 * NW_), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, \
                       (const bf16_t*)k, (const bf16_t*)v, kv_ts, nlse, ndelta, (bf16_t*)dk, (bf16_t*)dv, dkv_ts, cu_seqlens, T, hq, hkv, \
                       softmax_scale, part, (unsigned char*)nullptr, (int64_t)0, 0, cu_k, Tk, ro)
#define IE_DKDV_ROT(NW_)                                                                                                           \
    hipLaunchKernelGGL((flash_dkdv_k<128, true, 1, NW_, false, true>), gk, dim3(64 * NW_), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, \
                       (const bf16_t*)k, (const bf16_t*)v, kv_ts, nlse, ndelta, (bf16_t*)dk, (bf16_t*)dv, dkv_ts, cu_seqlens, T, hq, hkv, \
                       softmax_scale, part, (unsigned char*)nullptr, (int64_t)0, 0, cu_k, Tk, ro)
#define IE_DKDV_SPILL(DD, CA, NW_)                                                                                                 \
    hipLaunchKernelGGL((flash_dkdv_k<DD, CA, 1, NW_, true>), gk, dim3(64 * NW_), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, \
                       (const bf16_t*)k, (const bf16_t*)v, kv_ts, nlse, ndelta, (bf16_t*)dk, (bf16_t*)dv, dkv_ts, cu_seqlens, T, hq, hkv, \
                       softmax_scale, part, (unsigned char*)ds_ws, ds_head_stride, nq_max, (const int32_t*)nullptr, (int64_t)0, ro)
#define IE_DQ_FROM_DS(DD, CA)                                                                                                      \
    hipLaunchKernelGGL((flash_dq_from_ds_k<DD, CA>), gds, dim3(256), 0, st, (const unsigned char*)ds_ws, ds_head_stride, nq_max,      \
                       (const bf16_t*)k, kv_ts, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv, softmax_scale, gw)
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
#define IE_DQ(DD, CA, MW, FD)                                                                                                      \
    hipLaunchKernelGGL((flash_dq_k<DD, CA, MW, FD>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts,         \
                       (const bf16_t*)k, (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv, softmax_scale, \
                       cu_k, Tk, (const bf16_t*)out, o_ts, nlse, ndelta, 1.f / softmax_scale, ro)
#define IE_L(DD, CA)                                                                                                               \
    do {                                                                                                                           \
        if (g_dq_minw == 2) { if (fuse_delta) IE_DQ(DD, CA, 2, true); else IE_DQ(DD, CA, 2, false); }                              \
        else                { if (fuse_delta) IE_DQ(DD, CA, 1, true); else IE_DQ(DD, CA, 1, false); }                              \
        IE_DKDV(DD, CA);                                                                                                           \
    } while (0)
    // five-product path: dK/dV kernel with the dS^T spill, dQ from the spill.  Not with a head split (small problems, where the split is what matters).
    if (spill) {
        void* ds_ws = g_ds_ws;
        const int group = hq / hkv, gw = group % 4 == 0 ? 4 : (group % 2 == 0 ? 2 : 1), qw = 4 / gw;
        dim3 gds((unsigned)(hkv * (group / gw)), (unsigned)((nq_max + qw - 1) / qw), (unsigned)nseq);
#define IE_SP(DD, CA)                                                                                                              \
    do {                                                                                                                           \
        if (DKV_WAVES == 4) IE_DKDV_SPILL(DD, CA, 4);                                                                              \
        else IE_DKDV_SPILL(DD, CA, 2);                                                                                             \
        IE_DQ_FROM_DS(DD, CA);                                                                                                     \
    } while (0)
        if (d == 128) { if (causal) IE_SP(128, true); else IE_SP(128, false); }
        else          { if (causal) IE_SP(64, true); else IE_SP(64, false); }
#undef IE_SP
    } else if (rot) {   // dQ and dK / dV straight into the wqkv output-gradient rows, rotated (d = 128, causal, hs = 1, delta in the dQ prologue: checked above)
        if (g_dq_minw == 2)
            hipLaunchKernelGGL((flash_dq_k<128, true, 2, true, true>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, (const bf16_t*)k,
                               (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv, softmax_scale, cu_k, Tk, (const bf16_t*)out, o_ts,
                               nlse, ndelta, 1.f / softmax_scale, ro);
        else
            hipLaunchKernelGGL((flash_dq_k<128, true, 1, true, true>), gq, dim3(256), 0, st, (const bf16_t*)dout, do_ts, (const bf16_t*)q, q_ts, (const bf16_t*)k,
                               (const bf16_t*)v, kv_ts, lse, delta, (bf16_t*)dq, dq_ts, cu_seqlens, T, hq, hkv, softmax_scale, cu_k, Tk, (const bf16_t*)out, o_ts,
                               nlse, ndelta, 1.f / softmax_scale, ro);
        if (DKV_WAVES == 4) IE_DKDV_ROT(4);
        else IE_DKDV_ROT(2);
    } else {
        if (d == 128) { if (causal) IE_L(128, true); else IE_L(128, false); }
        else          { if (causal) IE_L(64, true); else IE_L(64, false); }
    }
#undef IE_L
#undef IE_DQ
#undef IE_DKDV
#undef IE_DKDV_HS
#undef IE_DKDV_W
#undef IE_DKDV_ROT
#if IE_DKDV_TIMING
    {
        hipStreamSynchronize(st);
        unsigned long long h[8] = {0}, z[8] = {0};
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dkdv_t), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_dkdv_t), z, sizeof(z));
        const double n = h[6] ? (double)h[6] : 1.0;
        fprintf(stderr, "dkdv cycles per wave-tile: issue %.0f A0 %.0f A1 %.0f C0 %.0f C1 %.0f sync %.0f (tiles %llu)\n", h[0] / n, h[1] / n, h[2] / n,
                h[3] / n, h[4] / n, h[5] / n, h[6]);
    }
#endif
    return ie_launch_status("ie_flash_attn_bwd launch");
}

extern "C" int ie_flash_attn_bwd(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v,
                                 int64_t kv_ts, const void* out, int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_ts,
                                 void* dk, void* dv, int64_t dkv_ts, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen,
                                 int hq, int hkv, int d, float softmax_scale, int causal, void* stream) {
    return flash_bwd_impl(dout, do_ts, q, q_ts, k, v, kv_ts, out, o_ts, lse, delta, dq, dq_ts, dk, dv, dkv_ts, cu_seqlens, nseq, T, max_seqlen, hq, hkv, d,
                          softmax_scale, causal, stream, nullptr, 0, 0);
}

// The backward of MHA._packed_forward's attention block in one piece (round 6): flash_attn_varlen_kvpacked_func's backward, the rotary embedding's backward on
// dQ and dK and the backward of the GQA rearrange (modeling_internlm2.py:416-441 read upwards) -- the gradients leave the two attention kernels straight into
// dqkv [T][hkv][hq / hkv + 2][d], the output gradient of the wqkv product.  Bit-identical to ie_flash_attn_bwd + ie_qkv_rotary_bwd (interleaved = 0).
extern "C" int ie_flash_attn_bwd_qkv_rotary_is_fused(int nseq, int max_seqlen, int hq, int hkv, int d, int causal) {
    if (nseq <= 0 || max_seqlen <= 0 || hq <= 0 || hkv <= 0 || hq % hkv != 0 || d != 128 || !causal) return 0;
    if (g_bwd_separate_delta || g_bwd_spill) return 0;
    const unsigned nkb = (unsigned)((max_seqlen + 32 * g_dkdv_waves - 1) / (32 * g_dkdv_waves));
    return dkdv_split(hq, hkv, causal, (int64_t)nkb * hkv * nseq) == 1;
}
extern "C" int ie_flash_attn_bwd_qkv_rotary(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts,
                                            const void* out, int64_t o_ts, const float* lse, float* delta_ws, void* dqkv, const void* cos, const void* sin,
                                            const int64_t* positions, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv, int d,
                                            float softmax_scale, int causal, void* stream) {
    IE_CHECK_ARG(dqkv && cos && sin && positions && hq > 0 && hkv > 0 && hq % hkv == 0 && d > 0, "ie_flash_attn_bwd_qkv_rotary: bad argument");
    IE_CHECK_SUPPORTED(ie_flash_attn_bwd_qkv_rotary_is_fused(nseq, max_seqlen, hq, hkv, d, causal) || nseq == 0 || T == 0 || max_seqlen == 0,
                       "ie_flash_attn_bwd_qkv_rotary: not fused for this shape (ie_flash_attn_bwd_qkv_rotary_is_fused): call ie_flash_attn_bwd + ie_qkv_rotary_bwd");
    const int qpk = hq / hkv;
    const int64_t gs = (int64_t)(qpk + 2) * d, ts = (int64_t)hkv * gs;
    const FaRotOut ro{(const bf16_t*)cos, (const bf16_t*)sin, positions, qpk, gs};
    bf16_t* base = (bf16_t*)dqkv;
    return flash_bwd_impl(dout, do_ts, q, q_ts, k, v, kv_ts, out, o_ts, lse, delta_ws, base, ts, base + (int64_t)qpk * d, base + (int64_t)(qpk + 1) * d, ts,
                          cu_seqlens, nseq, T, max_seqlen, hq, hkv, d, softmax_scale, causal, stream, nullptr, 0, 0, &ro);
}

// Backward of ie_flash_attn_fwd_x (full attention of queries cu_q[s] .. cu_q[s + 1] against keys cu_k[s] .. cu_k[s + 1]).  `lse` and `out` are what
// the probabilities are normalised with and delta = rowsum(dout * out) is taken from -- for one block of ring attention the MERGED lse / out of the
// whole row, which makes this block's dq / dk / dv its additive share of the whole gradient.  dq [Tq], dk / dv [Tk] are overwritten (rows of empty
// sequences with zeros).  delta: workspace of ie_flash_attn_bwd_workspace(Tq, hq, hkv, d) floats.
extern "C" int ie_flash_attn_bwd_x(const void* dout, int64_t do_ts, const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts,
                                   const void* out, int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_ts, void* dk, void* dv,
                                   int64_t dkv_ts, const int32_t* cu_q, const int32_t* cu_k, int nseq, int64_t Tq, int64_t Tk, int max_seqlen_q,
                                   int max_seqlen_k, int hq, int hkv, int d, float softmax_scale, void* stream) {
    IE_CHECK_ARG(cu_k && Tk >= 0 && max_seqlen_k >= 0, "ie_flash_attn_bwd_x: cu_k, Tk, max_seqlen_k");
    if (nseq == 0 || Tk == 0 || max_seqlen_k == 0) return IE_OK;
    // (max_seqlen_q = 0 still runs the dK / dV kernel: keys nobody looks at get zeros)
    return flash_bwd_impl(dout, do_ts, q, q_ts, k, v, kv_ts, out, o_ts, lse, delta, dq, dq_ts, dk, dv, dkv_ts, cu_q, nseq, Tq, max_seqlen_q > 0 ? max_seqlen_q : 1,
                          hq, hkv, d, softmax_scale, 0, stream, cu_k, Tk, max_seqlen_k);
}

// Spill buffer of the five-product backward (ie_tune_flash_bwd_variant bit 1): bytes for a call with these shapes, and the hook that hands the
// library a buffer of the caller's (it stays the caller's; NULL / 0 takes it back).  Without a large enough buffer the seven-product path runs.
extern "C" int64_t ie_flash_attn_bwd_spill_bytes(int nseq, int max_seqlen, int hq, int causal) {
    if (nseq < 0 || max_seqlen < 0 || hq <= 0) return -1;
    const int64_t nq = (max_seqlen + 63) / 64;
    return (int64_t)nseq * hq * ((causal ? nq * (nq + 1) / 2 : nq * nq) + 1) * 8192;
}
extern "C" int ie_flash_attn_bwd_set_spill(void* buf, int64_t bytes) {
    IE_CHECK_ARG((buf == nullptr) == (bytes == 0) && bytes >= 0 && (((uintptr_t)buf) & 1023u) == 0, "ie_flash_attn_bwd_set_spill: a 1-KiB aligned buffer and its size, or NULL and 0");
    g_ds_ws = buf;
    g_ds_ws_bytes = bytes;
    return IE_OK;
}

// tuning hook (A/B benchmarking only): bit 0: 0 = 2 waves x 32 keys per dK/dV block, 1 = 4 waves; bit 1: the five-product backward (needs a spill buffer);
// bit 2: delta by its own kernel in front of the dQ kernel (the default since round 6 computes it in the dQ kernel's prologue: the same bits)
extern "C" int ie_tune_flash_bwd_variant(int variant) {
    IE_CHECK_ARG(variant >= 0 && variant <= 7, "ie_tune_flash_bwd_variant: 0 .. 7");
    g_dkdv_waves = (variant & 1) ? 4 : 2;
    g_bwd_spill = (variant >> 1) & 1;
    g_bwd_separate_delta = (variant >> 2) & 1;
    return IE_OK;
}
