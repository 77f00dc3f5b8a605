// K3/K3b  bf16 GEMM, LDS-DMA variant: no staging registers and no ds_write at all.
//
// Same products / operand conventions as gemm_bf16.hip (fwd NT, dgrad NN, wgrad TN), for K % 64 == 0.
// Every operand tile goes HBM -> LDS with `global_load_lds_dwordx4` (1 KiB per wave-instruction, the
// destination is wave-uniform base + lane*16, pinned on hardware by tools/probes/probe_lds.hip), double
// buffered, the next tile in flight under the current tile's MFMAs.  Because the DMA image is lane-linear,
// bank-conflict avoidance is done by permuting the per-lane SOURCE address and applying the same involution
// when fragments are read (guide section 5.4 rule 21):
//   * k-contiguous operand: image [R][64] bf16, 128-B rows, 16-B chunk c of row r lives at slot
//     c ^ ((r >> 1) & 7): a ds_read_b128 lane group (16 rows distinct mod 16, one k-chunk) touches 16
//     distinct 16-B slots of the 256-B bank row -> conflict-free without padding;
//   * k-major operand ([K][R] in memory, the wgrad / dgrad case): image [64][R] bf16 in its NATURAL layout,
//     64-B block b of k-row k lives at block b ^ (k & 3); fragments are read with the hardware transposing
//     load ds_read_b64_tr_b16 (lane i of a 16-lane group receives X[k0..k0+3][m0+i] when lane p addresses
//     row k0 + (p >> 2), columns m0 + 4*(p & 3)..+3 -- pinned by the same probe), two reads per fragment,
//     conflict-free because the four k-rows of a read sit in four different 64-B blocks.
// So the transposed products cost no VALU transpose, no LDS write bandwidth and no extra registers.
// Block tile 256x256x64, 8 wave64 as 2(M) x 4(N), 128x64 per wave (8 accumulator tiles, 0.75 LDS reads/MFMA).
#include "ie_common.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int BK = 64;

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct DCfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int NT = 64 * NW;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;  // both image kinds are R * 64 * 2 bytes
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int CPITCH = BN * 2 + 16;
    static constexpr int SMEM_BYTES = (2 * STAGE_BYTES > BM * CPITCH) ? 2 * STAGE_BYTES : BM * CPITCH;
};

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Per-lane DMA source pointers of one operand, computed once and advanced by one k-tile per issue.
//   k-contiguous operand (R rows x 64 k): one wave-instruction = 8 rows; lane -> (row, 16-B chunk) with the chunk
//     index XOR-permuted by ((row >> 1) & 7); rows past the edge re-read the last row (their outputs are never stored).
//   k-major operand stored [K][ncols] (64 k-rows x R columns): one wave-instruction = 1024 / (2R) k-rows; the 16-B slot
//     index is XOR-permuted by ((k & 3) << 2), i.e. 64-B block b of k-row k goes to block b ^ (k & 3).
template <bool KM, int R, int NW>
struct DmaSrc {
    static constexpr int LPR = R / 8;                      // KM: lanes (16-B slots) per k-row
    static constexpr int RPI = 64 / LPR;                   // KM: k-rows per instruction
    static constexpr int PERW = KM ? (64 / RPI) / NW : R / 8 / NW;
    const bf16_t* p[PERW];
    int64_t step;
    __device__ __forceinline__ void init(const bf16_t* __restrict__ P, int64_t ld, int r0, int nr, int /*K*/, int wave, int lane) {
#pragma unroll
        for (int q = 0; q < PERW; ++q) {
            const int g = wave + NW * q;
            if (KM) {
                const int kr = g * RPI + lane / LPR;
                const int cpos = (lane % LPR) ^ ((kr & 3) << 2);
                const int col = min(r0 + cpos * 8, nr - 8);
                p[q] = P + (int64_t)kr * ld + col;
            } else {
                const int row = 8 * g + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int gr = min(r0 + row, nr - 1);
                p[q] = P + (int64_t)gr * ld + c * 8;
            }
        }
        step = KM ? 64 * ld : 64;
    }
    __device__ __forceinline__ void issue(unsigned char* tile, int wave) {
#pragma unroll
        for (int q = 0; q < PERW; ++q) issue_one(q, tile, wave);
    }
    __device__ __forceinline__ void issue_one(int q, unsigned char* tile, int wave) {
        dma16(p[q], tile + (wave + NW * q) * 1024);
        p[q] += step;
    }
    // split form: the address VGPRs are not touched next to the load (the pointer bump of every slot is done in one place)
    __device__ __forceinline__ void issue_keep(int q, unsigned char* tile, int wave) { dma16(p[q], tile + (wave + NW * q) * 1024); }
    __device__ __forceinline__ void advance_all() {
#pragma unroll
        for (int q = 0; q < PERW; ++q) p[q] += step;
    }
};

// Buffer-addressed twin of DmaSrc (`buffer_load_dwordx4 ... offen lds`): ONE per-lane byte offset VGPR per operand, the
// per-piece part of the address in an SGPR (wave-uniform: piece g = wave + NW*q covers whole rows), the k advance folded
// into the descriptor base -- no per-piece address VGPRs to bump and no 64-bit per-lane addresses through the memory
// front end.  The swizzle term of a piece depends on g only through g & 1 == wave & 1 (NW is even), so it is per-wave
// constant.  Rows past the edge of the matrix fall outside num_records and read as zero (no clamping needed).
// HALF (k-major image read as 16-column fragments for v_mfma_f32_16x16x32_bf16): k-rows k and k + 8 -- lanes l and l + 16 of a transposing read, one 32-lane bank
// group -- additionally swap the 32-byte halves of their 64-byte blocks (slot ^ 2 for k & 8).  k & 8 of piece g = wave + NW q is q & 1 with two rows per piece and
// NW = 4, so a lane has two byte offsets, one for the even and one for the odd pieces.
template <bool KM, int R, int NW, bool HALF = false>
struct BufSrc {
    static constexpr int LPR = R / 8;
    static constexpr int RPI = 64 / LPR;
    static constexpr int PERW = KM ? (64 / RPI) / NW : R / 8 / NW;
    static_assert(NW % 2 == 0 && (!KM || RPI == 2 || RPI % 4 == 0), "swizzle must be per-wave constant");
    static_assert(!HALF || (KM && RPI == 2 && NW == 4), "the half swizzle is written for 256-column k-major images loaded by four waves");
    const bf16_t* base;   // tile origin at the current k-tile
    uint32_t bytes_left;  // bytes from `base` to the end of the matrix
    uint32_t step_bytes;
    int voff;
    int voff_odd;         // HALF: the offset of the odd pieces
    int soff[PERW];
    __device__ __forceinline__ void init(const bf16_t* __restrict__ P, int64_t ld, int r0, int nr, int K, int wave, int lane) {
        const int ldb = (int)ld * 2;
        if (KM) {  // matrix [K][ld], tile columns r0..r0+R
            const int kr_in = lane / LPR;                                   // k-row inside a piece
            const int kr3 = (RPI == 2 ? 2 * (wave & 1) + kr_in : kr_in) & 3;  // (g * RPI + kr_in) & 3
            const int cpos = (lane % LPR) ^ (kr3 << 2);
            voff = kr_in * ldb + cpos * 16;
            voff_odd = kr_in * ldb + (cpos ^ 2) * 16;
            base = P + r0;
            bytes_left = (uint32_t)((int64_t)K * ldb - (int64_t)r0 * 2);
            step_bytes = 64u * (uint32_t)ldb;
#pragma unroll
            for (int q = 0; q < PERW; ++q) soff[q] = (wave + NW * q) * RPI * ldb;
        } else {   // matrix [nr][ld], tile rows r0..r0+R
            const int row_in = lane >> 3;
            const int sw = (4 * (wave & 1) + (lane >> 4)) & 7;              // ((8 g + row_in) >> 1) & 7
            voff = row_in * ldb + (((lane & 7) ^ sw) * 16);
            base = P + (int64_t)r0 * ld;
            bytes_left = (uint32_t)(((int64_t)nr - r0) * ldb);
            step_bytes = 128u;
#pragma unroll
            for (int q = 0; q < PERW; ++q) soff[q] = 8 * (wave + NW * q) * ldb;
        }
    }
    __device__ __forceinline__ void issue_keep(int q, unsigned char* tile, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource type does not exist in the host pass
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_left, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + (wave + NW * q) * 1024), 16, (HALF && (q & 1)) ? voff_odd : voff, soff[q], 0, 0);
#endif
    }
    __device__ __forceinline__ void advance_all() {
        base = (const bf16_t*)((const unsigned char*)base + step_bytes);
        bytes_left -= step_bytes;
    }
    __device__ __forceinline__ void issue(unsigned char* tile, int wave) {
#pragma unroll
        for (int q = 0; q < PERW; ++q) issue_keep(q, tile, wave);
        advance_all();
    }
};

// the output tile's 16-byte row stores: non-temporal, so that a round's 32 MB of C (256 tiles of 128 KB, all CUs reach their epilogues together) do not push
// the A / B panels of the next round out of the L2s.  Same-box A/B (profiles/r04_gemm_mfma16_ab.md): +1 ... +2 % on the forward and input-gradient products,
// 0 ... +6 % on the weight-gradient products, +0.35 % on the training step (A B A B).  -DIE_GEMM_NT_STORE=0 builds the plain stores.
#ifndef IE_GEMM_NT_STORE
#define IE_GEMM_NT_STORE 1
#endif
__device__ __forceinline__ void st16_c(void* p, const uint4& v) {
#if IE_GEMM_NT_STORE
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
#else
    st16(p, v);
#endif
}

// fragment (rows rbase + (lane & 31), k-step ks) of a k-contiguous image
__device__ __forceinline__ s16x8 frag_kc(const unsigned char* tile, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31);
    const int c = (ks * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const s16x8*>(tile + row * 128 + c * 16);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
// frag_kc as four dwords (the fp8 schedule: one 16-byte chunk = sixteen e4m3 values, half of a v_mfma_f32_32x32x64_f8f6f4 operand)
__device__ __forceinline__ i32x4 frag_kc_q(const unsigned char* tile, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31);
    const int c = (ks * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const i32x4*>(tile + row * 128 + c * 16);
}

// the same image read for v_mfma_f32_16x16x32_bf16: rows rbase + (lane & 15), k = 32 * ks + 8 * (lane >> 4) .. + 7.  A 16-lane group reads one chunk
// column of sixteen rows: eight swizzle values x the two 128-byte halves of a bank line -> all 64 banks once.
__device__ __forceinline__ s16x8 frag_kc16(const unsigned char* tile, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 15);
    const int c = (ks * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const s16x8*>(tile + row * 128 + c * 16);
}

// fragment (columns mbase + (lane & 31), k-step ks) of a k-major image [64][R]
template <int R>
__device__ __forceinline__ s16x8 frag_km(const unsigned char* tile, int mbase, int ks, int lane) {
    const int p = lane & 15, gq = lane >> 4;
    const int mcol = mbase + 16 * (gq & 1) + 4 * (p & 3);
    const int swz = (p >> 2) << 6;  // (kb & 3) << 6 with kb & 3 == p >> 2
    const int kb0 = ks * 16 + 8 * (lane >> 5) + (p >> 2);
    const unsigned char* a0 = tile + kb0 * (2 * R) + ((mcol * 2) ^ swz);
    const unsigned char* a1 = a0 + 4 * (2 * R);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ---- k32 ring staging (SPREAD -21 / -22).  The LDS holds FOUR stages of 32 k each instead of two of 64: a stage is free as soon as
// every wave holds its fragments in registers, so the DMA of entry u+3 (8-wave phased schedule) / u+4 (one wave per SIMD) is
// issued two to three whole entries before its first read, and the landing wait is a COUNTED `s_waitcnt vmcnt(n)` that leaves the
// younger entries in flight.  With two 64-deep stages the last pieces of a tile were issued about one k-step (~0.3 us) before a
// `vmcnt(0)`, i.e. inside the L2 / Infinity-Cache latency: the wave parked at that wait for a fifth of its time (SQ_WAIT_ANY
// 22 % against 5 % for hipBLASLt's loop, which keeps two tiles in flight the same way -- profiles/r01_gemm_ablation.md).
//   k-contiguous operand: stage image [R][32] bf16 (64-B rows); 16-B chunk c of row r sits at slot c ^ ((r >> 2) & 3) (16 rows x
//     one chunk = 16 distinct 16-B slots of a 256-B bank row); a 1-KiB piece = 16 rows, 4 lanes per row;
//   k-major operand: stage image [32][R] in natural layout, 64-B block b of k-row k at block b ^ (k & 3) exactly as above.
template <bool KM, int R, int NW, bool HALF = false>   // HALF: as BufSrc (k-rows k and k + 8 swap the 32-byte halves of their 64-byte blocks; 16-column fragments)
struct RingSrc {
    static constexpr int LPR = R / 8;                      // KM: 16-B slots per k-row
    static constexpr int RPI = 64 / LPR;                   // KM: k-rows per piece
    static constexpr int PERW = R / 16 / NW;               // pieces per wave and stage (a stage image is R * 64 bytes)
    static_assert(PERW >= 1 && NW % 2 == 0 && (!KM || RPI == 2 || RPI % 4 == 0), "piece mapping");
    static_assert(!HALF || (KM && RPI == 2 && NW == 4), "the half swizzle is written for 256-column k-major images loaded by four waves");
    const bf16_t* base;
    uint32_t bytes_left, step_bytes;
    int voff;
    int voff_odd;                                          // HALF: the offset of the odd pieces (k & 8 of piece wave + 4 q is q & 1)
    int soff[PERW];
    __device__ __forceinline__ void init(const bf16_t* __restrict__ P, int64_t ld, int r0, int nr, int K, int wave, int lane) {
        const int ldb = (int)ld * 2;
        if (KM) {
            const int kr_in = lane / LPR;
            const int kr3 = (RPI == 2 ? 2 * (wave & 1) + kr_in : kr_in) & 3;
            voff = kr_in * ldb + (((lane % LPR) ^ (kr3 << 2)) * 16);
            voff_odd = kr_in * ldb + (((lane % LPR) ^ (kr3 << 2) ^ 2) * 16);
            base = P + r0;
            bytes_left = (uint32_t)((int64_t)K * ldb - (int64_t)r0 * 2);
            step_bytes = 32u * (uint32_t)ldb;
#pragma unroll
            for (int q = 0; q < PERW; ++q) soff[q] = (wave + NW * q) * RPI * ldb;
        } else {
            const int row_in = lane >> 2;
            voff = row_in * ldb + (((lane & 3) ^ ((row_in >> 2) & 3)) * 16);
            base = P + (int64_t)r0 * ld;
            bytes_left = (uint32_t)(((int64_t)nr - r0) * ldb);
            step_bytes = 64u;
#pragma unroll
            for (int q = 0; q < PERW; ++q) soff[q] = 16 * (wave + NW * q) * ldb;
        }
    }
    __device__ __forceinline__ void issue_keep(int q, unsigned char* img, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_left, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(img + (wave + NW * q) * 1024), 16, (HALF && (q & 1)) ? voff_odd : voff, soff[q], 0, 0);
#endif
    }
    __device__ __forceinline__ void advance() {
        base = (const bf16_t*)((const unsigned char*)base + step_bytes);
        bytes_left -= step_bytes;
    }
};

// fragment (rows rbase + (lane & 31), k-step ks of 2) of a k-contiguous k32 stage image
__device__ __forceinline__ s16x8 frag_kc32(const unsigned char* img, int rbase, int ks, int lane) {
    const int row = rbase + (lane & 31);
    const int c = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
    return *reinterpret_cast<const s16x8*>(img + row * 64 + c * 16);
}

// frag_km with the two transposing reads as inline asm.  hipcc treats the ds_read_tr builtin as a read of any memory and puts
// `s_waitcnt vmcnt(0)` in front of the first one after an LDS-DMA issue, which would drain the ring entries that are meant to
// stay in flight.  The asm form carries no memory dependence; the caller waits `lgkmcnt(0)` itself before the first use.
template <int R>
__device__ __forceinline__ s16x8 frag_km_nowait(const unsigned char* tile, int mbase, int ks, int lane) {
    const int p = lane & 15, gq = lane >> 4;
    const int mcol = mbase + 16 * (gq & 1) + 4 * (p & 3);
    const int swz = (p >> 2) << 6;
    const int kb0 = ks * 16 + 8 * (lane >> 5) + (p >> 2);
    const unsigned char* a0 = tile + kb0 * (2 * R) + ((mcol * 2) ^ swz);
    const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)a0;
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(l0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(l0), "n"(4 * 2 * R));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// the 16-column fragment of a k-major image [64][R] with the HALF swizzle (BufSrc), for v_mfma_f32_16x16x32_bf16: lane (p = lane & 15, gq = lane >> 4) gets column
// mbase + p at k = 32 ks + 8 gq .. + 7.  Lanes l and l + 16 read k-rows eight apart: the other 32-byte half of the same 64-byte block, so a 32-lane group covers the
// four block positions (k & 3) x both halves = all 64 banks.  Untracked asm reads as frag_km_nowait: the caller waits lgkmcnt itself.
template <int R>
__device__ __forceinline__ s16x8 frag_km16_nowait(const unsigned char* tile, int mbase, int ks, int lane) {
    const int p = lane & 15, gq = lane >> 4;
    const int mcol = mbase + 4 * (p & 3);
    const int swz = ((p >> 2) << 6) ^ ((gq & 1) << 5);
    const int kb0 = ks * 32 + 8 * gq + (p >> 2);
    const unsigned char* a0 = tile + kb0 * (2 * R) + ((mcol * 2) ^ swz);
    const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)a0;
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(l0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(l0), "n"(4 * 2 * R));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// SPREAD = 0: the next tile's DMA instructions are all issued at the top of the k-tile; SPREAD = n > 0: they are
// interleaved one by one with the MFMAs of the first n k-steps (a DMA issue costs ~60-180 cycles of issue time during
// which this wave cannot feed the matrix pipe; spreading them lets the previous MFMAs cover that time).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KM, bool B_KM, int SPREAD, int EPI = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_dma_k(const bf16_t* __restrict__ A0, int64_t lda, const bf16_t* __restrict__ B0,
                                                                     int64_t ldb, bf16_t* __restrict__ C0, int64_t ldc, int M, int N, int K,
                                                                     int accumulate, int tiles_m, int tiles_n, IeGemmBatch bt) {
    using G = DCfg<BM, BN, WAVES_M, WAVES_N>;
    constexpr int NT = G::NT, NW = G::NW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM_BYTES];

    // strided batch (the experts of a MoE layer): ONE launch covers bt.count equal products; the XCD-contiguous tile numbering runs over
    // the whole batch, so an XCD works through one product's tiles before the next and small products fill the chip together
    const int nblk1 = tiles_m * tiles_n, nblk = nblk1 * bt.count;
    int id;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int bz = id / nblk1;
    id -= bz * nblk1;
    const bf16_t* __restrict__ A = A0 + bz * bt.sa;
    const bf16_t* __restrict__ B = B0 + bz * bt.sb;
    bf16_t* __restrict__ C = C0 + bz * bt.sc;
    const int GM = (accumulate >> 8) ? (accumulate >> 8) : 4;  // tile rows per group (bits 8.. of the flag word: ie_tune_gemm_group)
    const int abl = (accumulate >> 4) & 15;                     // timing ablations (IE_GEMM_ABLATE, results are then wrong)
    accumulate &= 1;
    const int width = GM * tiles_n;
    const int group = id / width;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int pm = first_m + (id % width) % gsz;
    const int pn = (id % width) / gsz;
    // EPI 1 (w1 | w3 forward with the SwiGLU epilogue): tile column block pn = B rows pn*BN/2 .. +BN/2 of w1 AND the same rows of w3 (f rows further)
    static_assert(EPI == 0 || ((SPREAD == -4 || (SPREAD == -5 && EPI == 1)) && !A_KM && BM == 256 && BN == 256 && WAVES_M * WAVES_N == 4), "the fused epilogues ride on the refill schedules");
    static_assert(EPI != 1 || !B_KM, "EPI 1: the forward product");
    const int m0 = pm * BM, n0 = EPI == 1 ? pn * (BN / 2) : pn * BN;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr bool MFMA16 = SPREAD == -5 || SPREAD == -24;   // the schedules on v_mfma_f32_16x16x32_bf16
    f32x4 acc16[MFMA16 ? 2 * G::TM : 1][MFMA16 ? 2 * G::TN : 1];   // their accumulators (the other set is dead code there)
#pragma unroll
    for (auto& row : acc16)
#pragma unroll
        for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr bool RING = SPREAD == -21 || SPREAD == -22 || SPREAD == -23 || SPREAD == -24;
    const int nk = K / BK;
    constexpr bool USE_BUF = SPREAD == -2 || SPREAD == -3 || SPREAD == -4 || SPREAD == -5 || SPREAD == -6 || SPREAD == -11;
    std::conditional_t<USE_BUF, BufSrc<A_KM, BM, NW>, DmaSrc<A_KM, BM, NW>> sa;
    std::conditional_t<USE_BUF, BufSrc<B_KM, BN, NW, SPREAD == -5 && B_KM>, DmaSrc<B_KM, BN, NW>> sb;
    if constexpr (!RING) {
        sa.init(A, lda, m0, M, K, wave, lane);
        sb.init(B, ldb, n0, N, K, wave, lane);
        if constexpr (EPI == 1) {   // the pieces that carry tile rows BN/2.. come from w3: (f - BN/2) rows further down
#pragma unroll
            for (int q = 0; q < decltype(sb)::PERW; ++q)
                if (8 * NW * q >= BN / 2) sb.soff[q] += (int)((bt.f - BN / 2) * ldb * 2);
        }
        sa.issue(smem, wave);
        sb.issue(smem + G::A_BYTES, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    constexpr int NA = DmaSrc<A_KM, BM, NW>::PERW, NB = DmaSrc<B_KM, BN, NW>::PERW;
    constexpr int DMA_STRIDE = SPREAD > 0 ? (SPREAD * G::TM * G::TN) / (NA + NB) : 1;  // MFMAs between two DMA issues
    static_assert(SPREAD <= 0 || DMA_STRIDE >= 1, "more DMA slots than MFMAs in the spread window");
    if constexpr (RING) {
        using RA = RingSrc<A_KM, BM, NW, SPREAD == -24>;
        using RB = RingSrc<B_KM, BN, NW, SPREAD == -24>;
        constexpr int AB = BM * 64, ST = (BM + BN) * 64;     // bytes of A's stage image, of one stage
        static_assert(4 * ST <= G::SMEM_BYTES, "four k32 stages");
        constexpr int PPW = RA::PERW + RB::PERW;              // DMA pieces per wave and entry
        RA ra;
        RB rb;
        ra.init(A, lda, m0, M, K, wave, lane);
        rb.init(B, ldb, n0, N, K, wave, lane);
        const int nk = K / 32;
        auto issue_piece = [&](int sl, unsigned char* st) {
            if (sl < RA::PERW) ra.issue_keep(sl, st, wave);
            else rb.issue_keep(sl - RA::PERW, st + AB, wave);
        };
        auto issue_entry = [&](int e) {
            unsigned char* st = smem + (e & 3) * ST;
#pragma unroll
            for (int sl = 0; sl < PPW; ++sl) issue_piece(sl, st);
            ra.advance();
            rb.advance();
        };
        auto fragA = [&](const unsigned char* st, int i, int ks) {
            return A_KM ? frag_km_nowait<BM>(st, wm * G::WM + i * 32, ks, lane) : frag_kc32(st, wm * G::WM + i * 32, ks, lane);
        };
        auto fragB = [&](const unsigned char* st, int j, int ks) {
            return B_KM ? frag_km_nowait<BN>(st + AB, wn * G::WN + j * 32, ks, lane) : frag_kc32(st + AB, wn * G::WN + j * 32, ks, lane);
        };
        if constexpr (SPREAD == -24) {
            // ---- the one-wave-per-SIMD ring on v_mfma_f32_16x16x32_bf16, both operands k-major (the weight-gradient product).  An entry (32 k) is ONE k-step of
            // 64 MFMAs of 4 passes; a wave holds the 8 + 8 sixteen-column fragments of two entries (128 VGPRs; 64 accumulator tiles of 4 registers in AGPRs).
            // Entry u: its barrier first (this wave's reads of entry u, requested during entry u-1, are complete and its pieces of entry u+1 have landed: vmcnt
            // leaves entries u+2, u+3 in flight) -- so every wave holds entry u in registers, stage u is free and entry u+1 is complete; then 64 MFMAs on set
            // u & 1 with one companion per gap in the first half: the fragments of entry u+1 into the other set (even gaps, two transposing reads each) and
            // the pieces of entry u+4 into stage u (odd gaps 1, 5, ...).  One barrier per 64 MFMAs, as the 32x32x16 ring has.
            static_assert(A_KM && B_KM && G::TM == 4 && G::TN == 4 && PPW == 8 && EPI == 0, "written for 4 waves x 128x128, both operands k-major");
            auto mfma16 = [](f32x4& d, const s16x8& a, const s16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b)); };
            s16x8 af[2][8], bfr[2][8];
            auto rd = [&](const unsigned char* st, int set, int f) {
                if (f < 8) af[set][f] = frag_km16_nowait<BM>(st, wm * G::WM + f * 16, 0, lane);
                else bfr[set][f - 8] = frag_km16_nowait<BN>(st + AB, wn * G::WN + (f - 8) * 16, 0, lane);
            };
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nk) issue_entry(e);
            if (nk >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // (not __syncthreads: its fence would drain the three entries left in flight)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 16; ++f) rd(smem, 0, f);
            auto entry = [&](int u, auto set_, auto fast_) {
                constexpr int cur = decltype(set_)::value, nxt = cur ^ 1;
                constexpr bool FAST = decltype(fast_)::value;     // entries u+1 .. u+4 exist
                const bool more1 = FAST || u + 1 < nk, more4 = FAST || u + 4 < nk;
                unsigned char* st = smem + (u & 3) * ST;
                const unsigned char* nx = smem + ((u + 1) & 3) * ST;
                if (FAST || u + 3 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
                else if (u + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 64; ++m) {
                    const int i = m >> 3, j = m & 7;
                    mfma16(acc16[i][j], bfr[cur][j], af[cur][i]);  // D[n][m]
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 32 && !(m & 1)) {
                        if (more1) rd(nx, nxt, m >> 1);
                    } else if (m < 32 && (m & 3) == 1) {
                        if (more4) issue_piece(m >> 2, st);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more4) {
                    ra.advance();
                    rb.advance();
                }
            };
            int u = 0;
            for (; u + 5 < nk; u += 2) {
                entry(u, std::integral_constant<int, 0>{}, std::true_type{});
                entry(u + 1, std::integral_constant<int, 1>{}, std::true_type{});
            }
            for (; u < nk; u += 2) {
                entry(u, std::integral_constant<int, 0>{}, std::false_type{});
                if (u + 1 < nk) entry(u + 1, std::integral_constant<int, 1>{}, std::false_type{});
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (asm: no hazard bookkeeping by the compiler) before the epilogue reads them
            __syncthreads();
        } else if constexpr (SPREAD == -21) {
            // ---- 8 waves, role-split LOAD / COMPUTE phases as SPREAD -11; one phase pair = one ring entry (two k-steps).
            // LOAD(u) reads entry u's fragments, issues entry u+3 into the stage of entry u-1 (the lagging group, then in
            // COMPUTE(u-1), drained its reads of that stage before the barrier that closed its LOAD(u-1)) and waits until this
            // wave's pieces of entry u+1 have landed: everything but the 2 * PPW youngest pieces.
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (e < nk) issue_entry(e);
            if (nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool lag = wave >= NW / 2;
            if (lag) __builtin_amdgcn_s_barrier();
            auto entry = [&](int u, auto more_) {
                constexpr bool MORE = decltype(more_)::value;   // entry u+3 exists
                const unsigned char* st = smem + (u & 3) * ST;
                s16x8 af[2][G::TM], bfr[2][G::TN];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) af[ks][i] = fragA(st, i, ks);
#pragma unroll
                    for (int j = 0; j < G::TN; ++j) bfr[ks][j] = fragB(st, j, ks);
                }
                if constexpr (MORE) {
                    issue_entry(u + 3);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < G::TM; ++i)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);  // D[n][m]
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            int u = 0;
            for (; u + 3 < nk; ++u) entry(u, std::true_type{});
            for (; u < nk; ++u) entry(u, std::false_type{});
            if (!lag) __builtin_amdgcn_s_barrier();
            __syncthreads();
        } else {
            // ---- one wave per SIMD (4 waves, 128x128 per wave), fragments software-pipelined across entries.  Entry u: k-step 0
            // runs on fragment set 0 while set 1 (entry u, k-step 1) is requested one read per MFMA gap; then the entry's single
            // barrier (every wave now holds all of entry u in registers, and this wave's pieces of entry u+1 have landed:
            // vmcnt leaves entries u+2, u+3 in flight); k-step 1 runs on set 1 while set 0 of entry u+1 is requested (gaps
            // 0..7) and the pieces of entry u+4 are issued into the stage just freed (gaps 8..15).
            static_assert(G::TM * G::TN >= G::TM + G::TN + PPW, "not enough MFMA gaps for the DMA pieces");
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nk) issue_entry(e);
            if (nk >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // (not __syncthreads: its fence would drain the three entries left in flight)
            __builtin_amdgcn_sched_barrier(0);
            s16x8 af[2][G::TM], bfr[2][G::TN];
#pragma unroll
            for (int i = 0; i < G::TM; ++i) af[0][i] = fragA(smem, i, 0);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) bfr[0][j] = fragB(smem, j, 0);
            // SPLIT (-23): the pieces of an entry are issued half under k-step 1 of entry u (entry u+4, into the stage freed by
            // barrier(u)) and half under k-step 0 of entry u+1 (still entry u+4 = its u+3, before barrier(u+1)) instead of all eight
            // in consecutive gaps.  FAST = steady state (1 <= u, u+4 < nk): every condition is compile-time true; the first and the
            // last four entries run the same body with run-time flags.
            constexpr bool SPLIT = SPREAD == -23;
            constexpr int H = SPLIT ? PPW / 2 : PPW;       // pieces issued under k-step 1
            constexpr int G0 = G::TM + G::TN;              // first MFMA gap that carries a DMA piece
            auto entry = [&](int u, auto fast_) {
                constexpr bool FAST = decltype(fast_)::value;
                const bool more1 = FAST || u + 1 < nk, more4 = FAST || u + 4 < nk, more3b = FAST || (u >= 1 && u + 3 < nk);
                unsigned char* st = smem + (u & 3) * ST;
                unsigned char* s3 = smem + ((u + 3) & 3) * ST;
                const unsigned char* nx = smem + ((u + 1) & 3) * ST;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int cur = ks, nxt = ks ^ 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < G::TM; ++i)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j) {
                            const int m = i * G::TN + j;
                            // the MFMA first, then this gap's read / DMA piece: the compiler's wait for the operands of the MFMA
                            // must not cover the read issued in the same gap
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);  // D[n][m]
                            __builtin_amdgcn_sched_barrier(0);
                            if (m < G0 && (ks == 0 || more1)) {
                                const unsigned char* ft = ks == 0 ? st : nx;
                                if (m < G::TM) af[nxt][m] = fragA(ft, m, nxt);
                                else bfr[nxt][m - G::TM] = fragB(ft, m - G::TM, nxt);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (ks == 1 && m >= G0 && m - G0 < H && more4) {
                                issue_piece(m - G0, st);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (SPLIT && ks == 0 && m >= G0 && m - G0 < PPW - H && more3b) {
                                issue_piece(H + m - G0, s3);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) {
                        if (SPLIT && more3b) {
                            ra.advance();
                            rb.advance();
                        }
                        // this wave's pieces of entry u+1 have landed: everything but the entries after it that are in flight
                        if (FAST || u + 3 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
                        else if (u + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        if (!SPLIT && more4) {
                            ra.advance();
                            rb.advance();
                        }
                        if (A_KM || B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (frag_km_nowait: set 0 of entry u+1)
                    }
                }
            };
            int u = 0;
            entry(u++, std::false_type{});
            for (; u + 4 < nk; ++u) entry(u, std::true_type{});
            for (; u < nk; ++u) entry(u, std::false_type{});
            __syncthreads();
        }
    } else
    if constexpr (SPREAD == -1 || SPREAD == -11) {
        // ---- role-split schedule (the "two waves per SIMD alternate compute and load segments" regime of
        // MI355X_MICROARCH.md): every k-step is a LOAD phase (fragment ds_reads + a share of the next tile's DMA
        // issues) and a COMPUTE phase (TM*TN MFMAs), each closed by a raw s_barrier.  The second half of the waves
        // (which share SIMDs with the first half) runs ONE PHASE BEHIND (one extra barrier up front, one fewer at the
        // end), so in every barrier interval one wave of each SIMD issues MFMAs while its partner issues LDS reads /
        // DMAs: the matrix pipe no longer idles while both waves of a SIMD wait for the same LDS round trip.
        // Hazards (placement-independent, by barrier counting): a wave drains its own ds_reads (lgkmcnt(0)) before the
        // barrier that closes a LOAD phase, so when the leading group starts tile t (and issues DMA into the buffer of
        // tile t-1) the lagging group -- then in the COMPUTE phase of step 3 of tile t-1 -- has no read of that
        // buffer in flight; each wave waits for its own DMA of tile t+1 (vmcnt(0)) in step 3 before the barrier that
        // precedes the first read of tile t+1 by anyone.
        const bool lag = wave >= NW / 2;
        // KP k-steps per phase: KP = 1 -> 4 LOAD/COMPUTE pairs per k-tile; KP = 2 (SPREAD = -11) -> 2 pairs with twice the
        // fragments and MFMAs each: the fixed cost of a phase (LDS round trip, drain, two barriers) is paid half as often
        constexpr int KP = SPREAD == -11 ? 2 : 1;  // (4 k-steps per phase measured 15-20 % slower: the DMA wait lands inside the LOAD phase)
        constexpr int NPH = 4 / KP;
        constexpr int DMA_PHASES = KP >= 2 ? 1 : 2;                       // phases (from 0) that carry the next tile's DMA issues
        constexpr int PER_PHASE = (NA + NB + DMA_PHASES - 1) / DMA_PHASES;
        if (lag) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            unsigned char* nbuf = smem + ((kt + 1) & 1) * G::STAGE_BYTES;
            const unsigned char* At = smem + (kt & 1) * G::STAGE_BYTES;
            const unsigned char* Bt = At + G::A_BYTES;
#pragma unroll
            for (int kp = 0; kp < NPH; ++kp) {
                // LOAD phase
                s16x8 af[KP][G::TM], bfr[KP][G::TN];
#pragma unroll
                for (int u = 0; u < KP; ++u) {
                    const int ks = kp * KP + u;
#pragma unroll
                    for (int i = 0; i < G::TM; ++i)
                        af[u][i] = A_KM ? frag_km<BM>(At, wm * G::WM + i * 32, ks, lane) : frag_kc(At, wm * G::WM + i * 32, ks, lane);
#pragma unroll
                    for (int j = 0; j < G::TN; ++j)
                        bfr[u][j] = B_KM ? frag_km<BN>(Bt, wn * G::WN + j * 32, ks, lane) : frag_kc(Bt, wn * G::WN + j * 32, ks, lane);
                }
                if (kp < DMA_PHASES && more) {
#pragma unroll
                    for (int s = kp * PER_PHASE; s < (kp + 1) * PER_PHASE && s < NA + NB; ++s) {
                        if (s < NA) sa.issue_keep(s, nbuf, wave);
                        else sb.issue_keep(s - NA, nbuf + G::A_BYTES, wave);
                    }
                    if (kp == DMA_PHASES - 1) { sa.advance_all(); sb.advance_all(); }
                }
                if (kp == NPH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // COMPUTE phase
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int u = 0; u < KP; ++u)
#pragma unroll
                    for (int i = 0; i < G::TM; ++i)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[u][j], af[u][i], acc[i][j], 0, 0, 0);  // D[n][m]
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!lag) __builtin_amdgcn_s_barrier();
        __syncthreads();
    } else if constexpr (SPREAD == -4) {
        // ---- one wave per SIMD, OPERAND-WISE REFILL of the two 64-deep stages (the structure of hipBLASLt's 256x256x64 direct-to-LDS
        // loop, read from its code object; 128-byte rows as in SPREAD -2, but the DMA of tile t+2 starts ~1.5 tiles before its use).
        // A wave keeps the fragments of a WHOLE k-tile in registers (4 k-steps x (TM + TN) = 128 VGPRs; the accumulators sit in AGPRs),
        // read half a tile ahead: B's second half first, then A's.  So early in tile t every wave holds all of B(t) (barrier 1) and
        // then all of A(t) (barrier 2), and those LDS regions take B(t+2) / A(t+2) while tile t+1 lands in the other stage.  The
        // landing waits are counted: barrier 3 needs this wave's pieces of B(t+1) (issued a tile ago; A(t+1) and the 10 pieces of
        // tile t+2 issued so far stay in flight: vmcnt(18)), barrier 4 those of A(t+1) (vmcnt(15)).  Per tile: 64 MFMAs, 32 LDS reads,
        // 16 DMA pieces, 4 barriers; one instruction per MFMA gap.
        static_assert(G::TM == 4 && G::TN == 4 && NA == 8 && NB == 8, "written for 4 waves x 128x128");
#ifndef IE_REFILL_ABL   // profiling builds only (tools/kbench/mkvariant.sh ... -DIE_REFILL_ABL=n; results are then WRONG): 1 no barriers, 2 no landing waits, 4 no LDS waits; on the 16x16x32 schedule also 8 no DMA pieces, 16 no fragment reads
#define IE_REFILL_ABL 0
#endif
#define IE_RF_BARRIER() do { if (!(IE_REFILL_ABL & 1)) __builtin_amdgcn_s_barrier(); } while (0)
        s16x8 af[4][G::TM], bfr[4][G::TN];
        auto rdA = [&](const unsigned char* st, int ks, int i) {
            af[ks][i] = A_KM ? frag_km<BM>(st, wm * G::WM + i * 32, ks, lane) : frag_kc(st, wm * G::WM + i * 32, ks, lane);
        };
        // The dgrad layout (A k-contiguous, B k-major) reads B with the untracked asm form (frag_km_nowait: no vmcnt(0) drain; the waits
        // in front of barriers 1 and 2 cover the second half of a tile, the one at the end of a tile the first half of the next).
        // With BOTH operands k-major the compiler parks fragments in scratch and would copy an asm result before it has arrived, so
        // that layout keeps the tracked builtin (correct, drained; the dispatcher sends it to the k32 ring instead).
        constexpr bool NOWAIT_B = B_KM && !A_KM;
        auto rdB = [&](const unsigned char* st, int ks, int j) {
            if constexpr (NOWAIT_B) bfr[ks][j] = frag_km_nowait<BN>(st + G::A_BYTES, wn * G::WN + j * 32, ks, lane);
            else bfr[ks][j] = B_KM ? frag_km<BN>(st + G::A_BYTES, wn * G::WN + j * 32, ks, lane) : frag_kc(st + G::A_BYTES, wn * G::WN + j * 32, ks, lane);
        };
        if (nk > 1) {   // tile 1 into the second stage, B first (the order the counted waits assume)
#pragma unroll
            for (int q = 0; q < NB; ++q) sb.issue_keep(q, smem + G::STAGE_BYTES + G::A_BYTES, wave);
#pragma unroll
            for (int q = 0; q < NA; ++q) sa.issue_keep(q, smem + G::STAGE_BYTES, wave);
            sa.advance_all();
            sb.advance_all();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < G::TM; ++i) rdA(smem, ks, i);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) rdB(smem, ks, j);
        }
        auto tile = [&](int t, auto fast_) {
            constexpr bool FAST = decltype(fast_)::value;     // tiles t+1 and t+2 exist
            const bool more1 = FAST || t + 1 < nk, more2 = FAST || t + 2 < nk;
            unsigned char* cur = smem + (t & 1) * G::STAGE_BYTES;
            const unsigned char* nxt = smem + ((t + 1) & 1) * G::STAGE_BYTES;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                const int ks = m >> 4, i = (m >> 2) & 3, j = m & 3;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);  // D[n][m]
                __builtin_amdgcn_sched_barrier(0);
                if (m < 8) {
                    rdB(cur, 2 + (m >> 2), m & 3);                                   // B, k-steps 2 and 3
                } else if (m == 8) {
                    if (!(IE_REFILL_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    IE_RF_BARRIER();                                                 // 1: every wave holds all of B(t)
                } else if (m < 14) {
                    if (more2) sb.issue_keep(m - 9, cur + G::A_BYTES, wave);         // B(t+2) pieces 0..4
                } else if (m < 22) {
                    rdA(cur, 2 + ((m - 14) >> 2), (m - 14) & 3);                     // A, k-steps 2 and 3
                } else if (m == 22) {
                    if (!(IE_REFILL_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    IE_RF_BARRIER();                                                 // 2: every wave holds all of A(t)
                } else if (m < 26) {
                    if (more2) sb.issue_keep(5 + m - 23, cur + G::A_BYTES, wave);    // B(t+2) pieces 5..7
                } else if (m < 28) {
                    if (more2) sa.issue_keep(m - 26, cur, wave);                     // A(t+2) pieces 0..1
                } else if (m == 28) {
                    if (more1) {
                        if (IE_REFILL_ABL & 2) {
                        } else if (FAST || t + 2 < nk) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // B(t+1) landed (this wave's pieces)
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        IE_RF_BARRIER();                                             // 3: B(t+1) complete
                    }
                } else if (m < 37) {
                    if (more1) rdB(nxt, (m - 29) >> 2, (m - 29) & 3);                // B(t+1), k-steps 0 (registers free since MFMA 15) and 1 (since 31)
                } else if (m < 42) {
                    if (more2) sa.issue_keep(2 + m - 37, cur, wave);                 // A(t+2) pieces 2..6
                } else if (m == 42) {
                    if (more1) {
                        if (IE_REFILL_ABL & 2) {
                        } else if (FAST || t + 2 < nk) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // A(t+1) landed
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        IE_RF_BARRIER();                                             // 4: A(t+1) complete
                    }
                } else if (m < 51) {
                    if (more1) rdA(nxt, (m - 43) >> 2, (m - 43) & 3);                // A(t+1), k-steps 0 and 1
                } else if (m == 51) {
                    if (more2) sa.issue_keep(7, cur, wave);                          // A(t+2) piece 7
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more2) {
                sa.advance_all();
                sb.advance_all();
            }
            if (NOWAIT_B) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (NOWAIT_B) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int t = 0;
        for (; t + 2 < nk; ++t) tile(t, std::true_type{});
        for (; t < nk; ++t) tile(t, std::false_type{});
        __syncthreads();
    } else if constexpr (SPREAD == -5) {
        // ---- the refill schedule of SPREAD -4 on v_mfma_f32_16x16x32_bf16 (what hipBLASLt's 256x256x64 kernel issues): 128 MFMAs of 8 passes per k-tile
        // instead of 64 of 16, so the 52 companion instructions (LDS reads, DMA pieces, waits, barriers) sit behind every SECOND MFMA and each has a
        // whole MFMA of issue slack behind it.  Same LDS image, same DMA order, same counted waits; a fragment is 16 rows x 32 k, a wave holds
        // 2 k-steps x (8 + 8) fragments (128 VGPRs) and 8 x 8 accumulators of 4 registers (256 AGPRs).
        // B k-major (the input-gradient product): 16-column fragments by transposing reads of the HALF-swizzled image, untracked (as SPREAD -4 does).
        static_assert(!A_KM && G::TM == 4 && G::TN == 4 && NA == 8 && NB == 8 && EPI != 2, "written for 4 waves x 128x128, A k-contiguous");
        s16x8 af[2][8], bfr[2][8];
        // (asm form: the accumulators are pinned to AGPRs -- with the builtin the allocator parks some of the 64 four-register tiles in VGPRs and copies them
        // through a temporary around their MFMAs, 120 to 480 v_accvgpr moves per k-tile; the waits on the fragments are still the compiler's)
        auto mfma16 = [](f32x4& d, const s16x8& a, const s16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b)); };
        auto rdA = [&](const unsigned char* st, int ks, int i) { af[ks][i] = frag_kc16(st, wm * G::WM + i * 16, ks, lane); };
        auto rdB = [&](const unsigned char* st, int ks, int j) {
            if constexpr (B_KM) bfr[ks][j] = frag_km16_nowait<BN>(st + G::A_BYTES, wn * G::WN + j * 16, ks, lane);
            else bfr[ks][j] = frag_kc16(st + G::A_BYTES, wn * G::WN + j * 16, ks, lane);
        };
        if (nk > 1) {
#pragma unroll
            for (int q = 0; q < NB; ++q) sb.issue_keep(q, smem + G::STAGE_BYTES + G::A_BYTES, wave);
#pragma unroll
            for (int q = 0; q < NA; ++q) sa.issue_keep(q, smem + G::STAGE_BYTES, wave);
            sa.advance_all();
            sb.advance_all();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) rdA(smem, 0, i);
#pragma unroll
        for (int j = 0; j < 8; ++j) rdB(smem, 0, j);
#ifndef IE_RF5_DMA   // 1: the DMA pieces of tile t+2 one every third MFMA pair behind the pair's SECOND MFMA (B at pairs 8 .. 29, A at 32 .. 53); 0: in the bursts of SPREAD -4
#define IE_RF5_DMA 1
#endif
        constexpr bool EVEN_DMA = IE_RF5_DMA == 1;
        auto tile = [&](int t, auto fast_) {
            constexpr bool FAST = decltype(fast_)::value;     // tiles t+1 and t+2 exist
            const bool more1 = FAST || t + 1 < nk, more2 = FAST || t + 2 < nk;
            unsigned char* cur = smem + (t & 1) * G::STAGE_BYTES;
            const unsigned char* nxt = smem + ((t + 1) & 1) * G::STAGE_BYTES;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 64; ++m) {   // MFMAs 2m and 2m + 1 (k-step mm >> 6, A fragment (mm >> 3) & 7, B fragment mm & 7), the companion after the first
                const int ks = m >> 5, i = (m >> 2) & 7, j = (2 * m) & 7;
                mfma16(acc16[i][j], bfr[ks][j], af[ks][i]);  // D[n][m]
                __builtin_amdgcn_sched_barrier(0);
                if (m < 8) {
                    if (!(IE_REFILL_ABL & 16)) rdB(cur, 1, m);                                                  // B, second half of the tile
                } else if (m == 8) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    IE_RF_BARRIER();                                    // 1: every wave holds all of B(t)
                } else if (m < 14) {
                    if (more2 && !EVEN_DMA && !(IE_REFILL_ABL & 8)) sb.issue_keep(m - 9, cur + G::A_BYTES, wave);         // B(t+2) pieces 0..4
                } else if (m < 22) {
                    if (!(IE_REFILL_ABL & 16)) rdA(cur, 1, m - 14);                                             // A, second half
                } else if (m == 22) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    IE_RF_BARRIER();                                    // 2: every wave holds all of A(t)
                } else if (m < 26) {
                    if (more2 && !EVEN_DMA && !(IE_REFILL_ABL & 8)) sb.issue_keep(5 + m - 23, cur + G::A_BYTES, wave);    // B(t+2) pieces 5..7
                } else if (m < 28) {
                    if (more2 && !EVEN_DMA && !(IE_REFILL_ABL & 8)) sa.issue_keep(m - 26, cur, wave);                     // A(t+2) pieces 0..1
                } else if (m == 28) {
                    if (more1) {
                        if (FAST || t + 2 < nk) {   // B(t+1) landed (this wave's pieces): A(t+1) and the pieces of tile t+2 issued so far stay in flight
                            if constexpr (EVEN_DMA) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                        }
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        IE_RF_BARRIER();                                // 3: B(t+1) complete
                    }
                } else if (m < 37) {
                    if (more1 && !(IE_REFILL_ABL & 16)) rdB(nxt, 0, m - 29);                                  // B(t+1), first half (fragment j is free since MFMA 56 + j; this is MFMA 58 + 2j)
                } else if (m < 42) {
                    if (more2 && !EVEN_DMA && !(IE_REFILL_ABL & 8)) sa.issue_keep(2 + m - 37, cur, wave);                 // A(t+2) pieces 2..6
                } else if (m == 42) {
                    if (more1) {
                        if (FAST || t + 2 < nk) {   // A(t+1) landed
                            if constexpr (EVEN_DMA) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                        }
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        IE_RF_BARRIER();                                // 4: A(t+1) complete
                    }
                } else if (m < 51) {
                    if (more1 && !(IE_REFILL_ABL & 16)) rdA(nxt, 0, m - 43);                                  // A(t+1), first half
                } else if (m == 51) {
                    if (more2 && !EVEN_DMA && !(IE_REFILL_ABL & 8)) sa.issue_keep(7, cur, wave);                          // A(t+2) piece 7
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma16(acc16[i][j + 1], bfr[ks][j + 1], af[ks][i]);
                __builtin_amdgcn_sched_barrier(0);
                if (EVEN_DMA && m >= 8 && m <= 53 && (m - 8) % 3 == 0) {
                    if (more2 && !(IE_REFILL_ABL & 8)) {
                        if (m < 32) sb.issue_keep((m - 8) / 3, cur + G::A_BYTES, wave);   // B(t+2): behind barrier 1
                        else sa.issue_keep((m - 32) / 3, cur, wave);                       // A(t+2): behind barrier 2
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (more2) {
                sa.advance_all();
                sb.advance_all();
            }
            if (B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (untracked reads: the next tile's first half of B)
        };
        if (B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int t = 0;
        for (; t + 2 < nk; ++t) tile(t, std::true_type{});
        for (; t < nk; ++t) tile(t, std::false_type{});
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (asm: no hazard bookkeeping by the compiler) before the epilogue reads them
        __syncthreads();
    } else if constexpr (SPREAD == -6) {
        // ---- fp8 (OCP e4m3) operands on v_mfma_f32_32x32x64_f8f6f4: the refill schedule of SPREAD -4 with the matrices handed over as "bf16 matrices of half the
        // width" -- a 128-byte LDS row is 128 fp8 values of k, the DMA, the image, its swizzle and the counted waits are byte-for-byte those of the bf16 kernel, and a
        // k-tile is K = 128.  One MFMA consumes 64 k: a lane's 32-byte operand is TWO 16-byte chunks of its row, (chunk g, chunk 2 + g) of the tile half for lane
        // half g = lane >> 5 -- the same permutation of k on both operands, which is all the instruction needs.  32 MFMAs of 16 passes per k-tile (behind every
        // second companion slot), i.e. the matrix time of the bf16 tile for twice the k: same LDS reads and DMA pieces per MFMA cycle, twice the flops.
        // A chunk pair = the fragments (2s, 2s + 1) of SPREAD -4's k-steps, so its register timing carries over: the second half of a tile is read early in the
        // tile, the next tile's first half once MFMAs 0..15 have issued.
        static_assert(!A_KM && !B_KM && G::TM == 4 && G::TN == 4 && NA == 8 && NB == 8 && EPI == 0, "written for 4 waves x 128x128, both operands k-contiguous");
        i32x4 fa[4][4], fb[4][4];   // (chunk (ks, lane half) of a row: ks = 2 s, 2 s + 1 form the operand of MFMA step s)
        auto rdA = [&](const unsigned char* st, int ks, int i) { fa[ks][i] = frag_kc_q(st, wm * G::WM + i * 32, ks, lane); };
        auto rdB = [&](const unsigned char* st, int ks, int j) { fb[ks][j] = frag_kc_q(st + G::A_BYTES, wn * G::WN + j * 32, ks, lane); };
        auto pair = [](const i32x4& lo, const i32x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); };
        if (nk > 1) {   // tile 1 into the second stage, B first (the order the counted waits assume)
#pragma unroll
            for (int q = 0; q < NB; ++q) sb.issue_keep(q, smem + G::STAGE_BYTES + G::A_BYTES, wave);
#pragma unroll
            for (int q = 0; q < NA; ++q) sa.issue_keep(q, smem + G::STAGE_BYTES, wave);
            sa.advance_all();
            sb.advance_all();
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rdA(smem, ks, i);
#pragma unroll
            for (int j = 0; j < 4; ++j) rdB(smem, ks, j);
        }
        auto tile = [&](int t, auto fast_) {
            constexpr bool FAST = decltype(fast_)::value;     // tiles t+1 and t+2 exist
            const bool more1 = FAST || t + 1 < nk, more2 = FAST || t + 2 < nk;
            unsigned char* cur = smem + (t & 1) * G::STAGE_BYTES;
            const unsigned char* nxt = smem + ((t + 1) & 1) * G::STAGE_BYTES;
            __builtin_amdgcn_sched_barrier(0);
            auto companion = [&](int m) {   // the slot's LDS read / DMA piece / wait + barrier (the table of SPREAD -4)
                    if (m < 8) {
                        rdB(cur, 2 + (m >> 2), m & 3);                                   // B, second half of the tile
                    } else if (m == 8) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();                                    // 1: every wave holds all of B(t)
                    } else if (m < 14) {
                        if (more2) sb.issue_keep(m - 9, cur + G::A_BYTES, wave);         // B(t+2) pieces 0..4
                    } else if (m < 22) {
                        rdA(cur, 2 + ((m - 14) >> 2), (m - 14) & 3);                     // A, second half
                    } else if (m == 22) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();                                    // 2: every wave holds all of A(t)
                    } else if (m < 26) {
                        if (more2) sb.issue_keep(5 + m - 23, cur + G::A_BYTES, wave);    // B(t+2) pieces 5..7
                    } else if (m < 28) {
                        if (more2) sa.issue_keep(m - 26, cur, wave);                     // A(t+2) pieces 0..1
                    } else if (m == 28) {
                        if (more1) {
                            if (FAST || t + 2 < nk) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // B(t+1) landed (this wave's pieces)
                            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                            __builtin_amdgcn_s_barrier();                                // 3: B(t+1) complete
                        }
                    } else if (m < 37) {
                        if (more1) rdB(nxt, (m - 29) >> 2, (m - 29) & 3);                // B(t+1), first half (chunk pair j is free since the MFMA behind slot 24 + 2j)
                    } else if (m < 42) {
                        if (more2) sa.issue_keep(2 + m - 37, cur, wave);                 // A(t+2) pieces 2..6
                    } else if (m == 42) {
                        if (more1) {
                            if (FAST || t + 2 < nk) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // A(t+1) landed
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();                                // 4: A(t+1) complete
                        }
                    } else if (m < 51) {
                        if (more1) rdA(nxt, (m - 43) >> 2, (m - 43) & 3);                // A(t+1), first half
                    } else if (m == 51) {
                        if (more2) sa.issue_keep(7, cur, wave);                          // A(t+2) piece 7
                    }
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int q = 0; q < 32; ++q) {   // MFMA q (tile half q >> 4, A block (q >> 2) & 3, B block q & 3), then the companion slots 2q and 2q + 1
                const int sx = q >> 4, i = (q >> 2) & 3, j = q & 3;
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pair(fb[2 * sx][j], fb[2 * sx + 1][j]), pair(fa[2 * sx][i], fa[2 * sx + 1][i]), acc[i][j],
                                                                            0, 0, 0, 0, 0, 0);  // D[n][m]; e4m3 x e4m3, no block scales
                __builtin_amdgcn_sched_barrier(0);
                companion(2 * q);
                companion(2 * q + 1);
            }
            if (more2) {
                sa.advance_all();
                sb.advance_all();
            }
        };
        int t = 0;
        for (; t + 2 < nk; ++t) tile(t, std::true_type{});
        for (; t < nk; ++t) tile(t, std::false_type{});
        __syncthreads();
    } else if constexpr (SPREAD == -2 || SPREAD == -3) {
        // ---- one wave per SIMD (4 waves, 128x128 per wave: 0.5 LDS reads per MFMA), software-pipelined ACROSS k-tiles.
        // The single barrier of a k-tile sits between k-step 2 and k-step 3: by then every wave has requested all four
        // fragment sets of tile t, so after it (a) the first fragments of tile t+1 are read while the MFMAs of step 3 of
        // tile t run (no pipeline refill bubble at tile boundaries) and (b) the DMA of tile t+2 is issued into tile t's
        // buffer, one instruction per MFMA of step 3, which gives it three more k-steps to land before the next barrier.
        s16x8 af[2][G::TM], bfr[2][G::TN];
        auto load_frags = [&](const unsigned char* At, int ks, int set) {
            const unsigned char* Bt = At + G::A_BYTES;
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
                af[set][i] = A_KM ? frag_km<BM>(At, wm * G::WM + i * 32, ks, lane) : frag_kc(At, wm * G::WM + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
                bfr[set][j] = B_KM ? frag_km<BN>(Bt, wn * G::WN + j * 32, ks, lane) : frag_kc(Bt, wn * G::WN + j * 32, ks, lane);
        };
        // DMA placement: the next k-step's fragment reads go under the first TM+TN MFMAs of a k-step (one each), the DMA
        // pieces under the MFMAs after them (one each) -- never both in one MFMA gap: a gap hides ~5 single-issue
        // instructions, and a DMA piece with its M0 update next to a ds_read and its wait already overflows it.
        // Iteration t issues the first PER pieces of tile t+2 (into tile t's buffer, free after the barrier) under its
        // step 3 and the remaining pieces of tile t+1 under its steps 0 and 1 (they have step 2 to land).
        constexpr int NS = NA + NB, NM = G::TM * G::TN;
        auto issue_slot = [&](int slot, unsigned char* tile) {
            if (slot < NA) sa.issue_keep(slot, tile, wave);
            else sb.issue_keep(slot - NA, tile + G::A_BYTES, wave);
        };
        auto run = [&](auto) {
            constexpr int FIRST = G::TM + G::TN;        // MFMA positions before FIRST carry the fragment reads
            constexpr bool EARLY = SPREAD == -3;        // all pieces under steps 3 and 0: the last one has two k-steps to land, not one
            constexpr int PER = EARLY ? NS / 2 : (NS + 2) / 3;   // pieces under step 3; the rest split over steps 0 and 1
            static_assert(PER <= NM - FIRST && NS - PER <= 2 * (NM - FIRST), "not enough DMA positions");
            // piece issued at (window step kidx = 0 (k-step 3), 1 (k-step 0), 2 (k-step 1); MFMA position m), or -1
            auto slot_at = [](int kidx, int m) constexpr -> int {
                if (m < FIRST) return -1;
                const int pos = m - FIRST;
                if (kidx == 0) return pos < PER ? pos : -1;
                const int rest = NS - PER, first = EARLY ? rest : (rest + 1) / 2;  // step 0 takes `first`, step 1 the remainder
                if (kidx == 1) return pos < first ? PER + pos : -1;
                return pos < rest - first ? PER + first + pos : -1;
            };
            if (nk > 1) {
#pragma unroll
                for (int sl = 0; sl < PER; ++sl) issue_slot(sl, smem + G::STAGE_BYTES);
            }
            load_frags(smem, 0, 0);
            // one k-tile; D01: tile t+1 exists (rest of its DMA under steps 0-1, its first fragments under step 3); D3: tile t+2 exists
            auto tile = [&](int kt, auto d01, auto d3) {
                constexpr bool D01 = decltype(d01)::value, D3 = decltype(d3)::value;
                unsigned char* At = smem + (kt & 1) * G::STAGE_BYTES;
                unsigned char* Nt = smem + ((kt + 1) & 1) * G::STAGE_BYTES;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cur = ks & 1, nxt = cur ^ 1;
                    // the next k-step's fragments are requested one per MFMA (positions 0 .. TM+TN-1), not as a burst
                    const unsigned char* Ft = ks < 3 ? At : Nt;
                    const int fks = ks < 3 ? ks + 1 : 0;
                    const bool do_frags = ks < 3 || D01;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < G::TM; ++i)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j) {
                            const int m = i * G::TN + j;
                            // the MFMA first, then this gap's read / DMA piece: with the read first, the compiler's wait for the MFMA's
                            // operands at the loop head (lgkmcnt(0), it cannot count across the back edge) also covered the read just
                            // issued -- one exposed LDS round trip per k-tile
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);  // D[n][m]
                            __builtin_amdgcn_sched_barrier(0);
                            if (do_frags && m < G::TM + G::TN) {
                                if (m < G::TM) af[nxt][m] = A_KM ? frag_km<BM>(Ft, wm * G::WM + m * 32, fks, lane) : frag_kc(Ft, wm * G::WM + m * 32, fks, lane);
                                else bfr[nxt][m - G::TM] = B_KM ? frag_km<BN>(Ft + G::A_BYTES, wn * G::WN + (m - G::TM) * 32, fks, lane)
                                                                : frag_kc(Ft + G::A_BYTES, wn * G::WN + (m - G::TM) * 32, fks, lane);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (ks != 2) {
                                const int sl = slot_at(ks == 3 ? 0 : ks + 1, m);
                                if (sl >= 0 && (ks == 3 ? D3 : D01)) {
                                    issue_slot(sl, ks == 3 ? At : Nt);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 2) {
                        sa.advance_all();  // every slot moves on one k-tile here: slots >= PER were used for tile t+1 (steps 0-1),
                        sb.advance_all();  // slots < PER are next used for tile t+2 (step 3)
                        if (abl & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        if (!(abl & 1)) __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (abl & 4) {
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            using T_ = std::true_type;
            using F_ = std::false_type;
            for (int kt = 0; kt < nk - 2; ++kt) tile(kt, T_{}, T_{});
            if (nk >= 2) tile(nk - 2, T_{}, F_{});
            tile(nk - 1, F_{}, F_{});
        };
        run(0);
        __syncthreads();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        unsigned char* nbuf = smem + ((kt + 1) & 1) * G::STAGE_BYTES;
        if (SPREAD == 0 && more) {
            sa.issue(nbuf, wave);
            sb.issue(nbuf + G::A_BYTES, wave);
        }
        const unsigned char* At = smem + (kt & 1) * G::STAGE_BYTES;
        const unsigned char* Bt = At + G::A_BYTES;
        // Software-pipelined over the 4 k-steps: the fragments of step ks+1 are requested BEFORE the MFMAs of step ks
        // are issued (sched_barrier pins that order; hipcc otherwise reads, waits lgkmcnt(0), computes, reads ...),
        // so the LDS latency of all but the first step of a tile hides under 8 MFMAs.
        s16x8 af[2][G::TM], bfr[2][G::TN];
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
            af[0][i] = A_KM ? frag_km<BM>(At, wm * G::WM + i * 32, 0, lane) : frag_kc(At, wm * G::WM + i * 32, 0, lane);
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
            bfr[0][j] = B_KM ? frag_km<BN>(Bt, wn * G::WN + j * 32, 0, lane) : frag_kc(Bt, wn * G::WN + j * 32, 0, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
                    af[nxt][i] = A_KM ? frag_km<BM>(At, wm * G::WM + i * 32, ks + 1, lane) : frag_kc(At, wm * G::WM + i * 32, ks + 1, lane);
#pragma unroll
                for (int j = 0; j < G::TN; ++j)
                    bfr[nxt][j] = B_KM ? frag_km<BN>(Bt, wn * G::WN + j * 32, ks + 1, lane) : frag_kc(Bt, wn * G::WN + j * 32, ks + 1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
                    if (SPREAD > 0 && ks < SPREAD) {
                        const int mg = ks * G::TM * G::TN + i * G::TN + j;  // MFMA index inside the spread window
                        if (mg % DMA_STRIDE == 0 && mg / DMA_STRIDE < NA + NB && more) {
                            const int slot = mg / DMA_STRIDE;
                            if (slot < NA) sa.issue_one(slot, nbuf, wave);
                            else sb.issue_one(slot - NA, nbuf + G::A_BYTES, wave);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);  // D[n][m]
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's DMA has landed (this wave's part)
        __syncthreads();
    }

    // ---- epilogue (as gemm_bf16.hip): D^T accumulators -> LDS [m][n] bf16 -> 16-byte row stores
    if ((abl & 8) && !(abl & 4) && M != -12345) return;   // timing ablation: no epilogue at all
    auto stage = [&]() {
    if constexpr (MFMA16) {   // D[n][m] of a 16x16 block: lane = column m (lane & 15), registers = rows n 4 * (lane >> 4) .. + 3
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = wm * G::WM + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = wn * G::WN + j * 16 + 4 * (lane >> 4);
                uint2 v;
                v.x = pack2bf(acc16[i][j][0], acc16[i][j][1]);
                v.y = pack2bf(acc16[i][j][2], acc16[i][j][3]);
                st8(smem + m * G::CPITCH + n * 2, v);
            }
        }
        return;
    }
    float deq = 1.f;   // fp8 operands: the product of the two per-tensor dequantisation scales (device scalars: no host read between quantisation and product)
    if constexpr (SPREAD == -6) deq = reinterpret_cast<const float*>(bt.aux)[bz] * reinterpret_cast<const float*>(bt.act)[bz];
#pragma unroll
    for (int i = 0; i < G::TM; ++i) {
        const int m = wm * G::WM + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < G::TN; ++j) {
            if constexpr (SPREAD == -6) acc[i][j] *= deq;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * G::WN + j * 32 + 8 * g + 4 * (lane >> 5);
                uint2 v;
                v.x = pack2bf(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]);
                v.y = pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                st8(smem + m * G::CPITCH + n * 2, v);
            }
        }
    }
    };
    stage();
    __syncthreads();
    if constexpr (EPI == 1) {
        // [BM][BN/2 gate | BN/2 up] -> h13 (both halves, at columns n0.. and f + n0..) and act = silu(gate) * up
        constexpr int CH = BN / 16;   // 16-byte chunks per half row
        bf16_t* __restrict__ ACT = (bf16_t*)bt.act;
        const int F = (int)bt.f;
#pragma unroll
        for (int q = 0; q < BM * CH / NT; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c / CH, nc = c % CH;
            const int gm = m0 + row, gn = n0 + nc * 8;
            if (gm < M && gn < F) {
                const uint4 g = ld16(smem + row * G::CPITCH + nc * 16), u = ld16(smem + row * G::CPITCH + BN + nc * 16);
                float gv[8], uv[8], o[8];
                unpack8(g, gv);
                unpack8(u, uv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd1(gv[e], uv[e]);
                bf16_t* dst = C + (int64_t)gm * ldc + gn;
                st16_c(dst, g);
                st16_c(dst + F, u);
                st16_c(ACT + (int64_t)gm * bt.ld_act + gn, pack8(o));
            }
        }
        return;
    }
    if constexpr (EPI == 2) {
        // the tile is d(act) [BM][BN]: with gate | up from h13, d(gate) | d(up) -> C = dh13 [M, 2f].  Measured (profiles/r03_ffn_fuse_ab.jsonl): NOT
        // faster than the two-launch path -- all 256 CUs reach their epilogues together (equal tiles, one block per CU), so this epilogue's 512 KB
        // per tile move in chip-wide bursts at HBM speed with nothing to hide them (1826 vs 1798 us at 16 384 rows; requesting the h13 reads
        // before the accumulators are staged: 2016 us).  Off by default (ie_tune_ffn_fuse bit 1).
        constexpr int CH = BN / 8;
        const bf16_t* __restrict__ H13 = (const bf16_t*)bt.aux;
        const int F = (int)bt.f;
#pragma unroll 4
        for (int q = 0; q < BM * CH / NT; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c / CH, nc = c % CH;
            const int gm = m0 + row, gn = n0 + nc * 8;
            if (gm < M && gn < N) {
                const bf16_t* hp = H13 + (int64_t)gm * bt.ld_aux + gn;
                float gv[8], av[8], bv[8], oa[8], ob[8], oc;
                unpack8(ld16(hp), av);
                unpack8(ld16(hp + F), bv);
                unpack8(ld16(smem + row * G::CPITCH + nc * 16), gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) swiglu_bwd1(gv[e], av[e], bv[e], oa[e], ob[e], oc);
                bf16_t* dst = C + (int64_t)gm * ldc + gn;
                st16_c(dst, pack8(oa));
                st16_c(dst + F, pack8(ob));
            }
        }
        return;
    }
    constexpr int CPR = BN / 8;
    const bool no_store = (abl & 12) == 12;   // timing ablation: staging and LDS reads, but no global stores
#pragma unroll
    for (int q = 0; q < BM * CPR / NT; ++q) {
        const int c = threadIdx.x + NT * q;
        const int row = c / CPR, nc = c % CPR;
        const int gm = m0 + row, gn = n0 + nc * 8;
        if (gm < M && gn < N && !(no_store && gm != -12345)) {
            uint4 v = ld16(smem + row * G::CPITCH + nc * 16);
            bf16_t* dst = C + (int64_t)gm * ldc + gn;
            if (accumulate) {
                float o[8], n[8];
                unpack8(ld16(dst), o);
                unpack8(v, n);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += n[e];
                v = pack8(o);
            }
            st16_c(dst, v);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// PERSISTENT frame around the 16x16x32 refill schedule (round 5; schedule id -7 = -5 in a persistent frame; forward and input-gradient layouts).
// Measured on the plain kernel (profiles/r05_gemm_forward_k_sweep.log, r05_gemm_frame_ablation.log): a 16 384 x 4096 forward product costs 71 us + 0.0906 us x K,
// i.e. 17.8 us per tile ROUND that is not the k-loop -- 16 % of a K = 4096 product: a block's dispatch, its first k-tile's transfer with nothing to hide it
// (all 256 CUs fetch 16 MB together), the LDS-staged epilogue (4-5 us) -- and the library's persistent stream-K kernel on the same tile, MFMA and per-k-tile
// instruction mix keeps the matrix pipe 87.5 % busy in the training step where gemm_dma_k<..., -5, 0> reaches 75.4 %.  Here a block WALKS its tiles (block b:
// tiles b, b + grid, ... -- the order of the plain launch's rounds, so the XCD-aware numbering and the lockstep of a round are kept); the k-loop runs THROUGH the
// tile boundary: the last two k-tiles of an output tile request the first two k-tiles of the next one (the transfer sources are re-based, nothing else changes: the
// stages, their barriers and the counted waits continue), the last k-tile reads the next tile's first fragments; the accumulators go to memory FROM REGISTERS
// (8-byte row pieces: a lane of a 16x16 D^T block holds four consecutive columns of one row; the LDS image of C would need the LDS the next tile's k-tiles are
// landing in, and two barriers).  The stores share vmcnt with the counted landing waits: in the first k-tile behind an epilogue those waits also cover the
// stores issued in front of them (over-waiting is safe; the stores are acknowledged by the L2).  K / 64 must be even (the stage parity continues across tiles).
// EPI 1 (the w1 | w3 forward product with the SwiGLU gate): B = (w1 | w3) [2F, K]; an output tile is 128 gate columns + the SAME 128 up columns, laid out so
// that every wave holds both for its 64 columns: tile rows [0, 64) = w1 rows n0 .. + 64, [64, 128) = w3 rows n0 .. + 64, [128, 192) = w1 rows n0 + 64 .. + 128,
// [192, 256) = w3 rows n0 + 64 .. + 128 (a row offset on B's transfer pieces, as in gemm_dma_k's EPI 1) -- accumulator blocks j = 0 .. 3 are gate, 4 .. 7 up
// columns of one lane's rows.  The epilogue writes h13 (both halves) and act = silu(gate) * up, computed from the bf16-rounded gate / up exactly as the
// separate kernel and gemm_dma_k's fused epilogue do.  N = 2F; tiles_n = F / 128.
// EPI 2 (round 6; the w2 input-gradient product with the SwiGLU backward): the tile is d(act) [256 x 256] of dy @ w2; ACT / ld_act carry h13 [M, 2F] (gate | up,
// read), C = dh13 [M, 2F]: every 16-byte piece of a wave's 16-row turn is joined by the gate and up values of its eight columns and leaves as d(gate) and d(up)
// (swiglu_bwd1 on the bf16-rounded d(act): the arithmetic of swiglu_bwd_k and of gemm_dma_k's EPI 2, bit for bit).  The h13 pieces of a turn are requested
// before the turn's accumulators are packed.  N = F; tiles_n = F / 256.
// EPI 3 (round 6; the wqkv forward product of a d = 128 model): a wave's 128 columns are ONE head slot of the [kv group][qpk query heads, k, v][128] row layout
// (modeling_internlm2.py:416-420), so the wave-private turn holds 16 rows of one slot: q and k slots are de-interleaved (even | odd, :425-427), rotated with the
// cos / sin rows of the token's position (embedding.py:89-166) and stored into q [T, hkv qpk, 128] / kv [T, 2, hkv, 128]; v slots are copied.  The arithmetic is
// qkv_rotary_fwd_k's on the same bf16-rounded products: bit-identical to product + that kernel.  C is not written.  N = hkv (qpk + 2) 128.
template <bool B_KM, int EPI = 0>
__global__ __launch_bounds__(256) void gemm_p5_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, bf16_t* __restrict__ C,
                                                 int64_t ldc, int M, int N, int K, int accumulate, int tiles_m, int tiles_n, bf16_t* __restrict__ ACT,
                                                 int64_t ld_act, int F, unsigned* __restrict__ queue, IeRotaryEpi rot) {
    static_assert(EPI == 0 || ((EPI == 1 || EPI == 3) && !B_KM) || (EPI == 2 && B_KM), "EPI 1 / 3: forward products; EPI 2: the w2 input-gradient product");
    using G = DCfg<256, 256, 2, 2>;
    constexpr int NW = 4;
    constexpr int EP = 256 + 16;                      // pitch of a wave's private epilogue rows (16 rows x 128 bf16 columns)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G::STAGE_BYTES + NW * 16 * EP + 16];
    const int nblk = tiles_m * tiles_n;
    const int GM = (accumulate >> 8) ? (accumulate >> 8) : 4;
    const int abl = (accumulate >> 4) & 15;   // timing ablations (IE_GEMM_ABLATE; results then wrong): 8 = no stores
    accumulate &= 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / 2, wn = wave % 2;
    const int nk = K / 64;
    // tile b of the launch order -> (m0, n0): gemm_dma_k's XCD-contiguous numbering and its GM x tiles_n groups
    auto coords = [&](int b, int& m0, int& n0) {
        const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        const int width = GM * tiles_n, group = id / width, first_m = group * GM, gsz = min(tiles_m - first_m, GM);
        m0 = (first_m + (id % width) % gsz) * 256;
        n0 = ((id % width) / gsz) * (EPI == 1 ? 128 : 256);
    };
    auto init_b = [&](BufSrc<B_KM, 256, NW, B_KM>& sb_, int n0_) {
        sb_.init(B, ldb, n0_, N, K, wave, lane);
        if constexpr (EPI == 1) {   // piece p = wave + 4 q carries tile rows 8 p .. + 7: their source rows (see above)
            const int64_t rb = ldb * 2;
#pragma unroll
            for (int q = 0; q < 8; ++q) sb_.soff[q] += (int)(((q & 2) ? (int64_t)F : 0) * rb - (int64_t)(q < 2 ? 0 : q < 4 ? 64 : q < 6 ? 64 : 128) * rb);
        }
    };
    // Which tile next: a block does NOT own the tiles b, b + grid, ... -- if a collective's kernels hold some CUs, the blocks that cannot start until a running block
    // has finished would leave their whole share for the end (2 x for a 4-round product).  Tiles are handed out by eight counters, one per XCD (block b runs on
    // XCD b & 7 and takes the tiles x + 8 i of the launch order, i from the XCD's counter: on a free chip the same tiles in the same rounds as the static walk, the
    // XCD-aware numbering intact); the first tile too.  queue[0..7] the counters, queue[8] the blocks that are done: the last one zeroes them for the launch that
    // uses this slot next (gemm_bf16_dma.hip hands out 64 slots round-robin).  One lane asks; the answer travels through an LDS word behind the next barrier.
    unsigned* nextw = reinterpret_cast<unsigned*>(smem + 2 * G::STAGE_BYTES + NW * 16 * EP);
    const int xcd_ = blockIdx.x & 7;
    const int per_xcd = (nblk - xcd_ + 7) >> 3;             // tiles x, x + 8, ... below nblk
    auto grab = [&]() -> int {                              // (wave 0, lane 0)
        const unsigned i = atomicAdd(queue + xcd_, 1u);
        return (int)i < per_xcd ? xcd_ + 8 * (int)i : -1;
    };
    if (threadIdx.x == 0) nextw[0] = (unsigned)grab();
    __syncthreads();
    int b = (int)nextw[0];
    if (b < 0) {
        if (threadIdx.x == 0 && atomicAdd(queue + 8, 1u) == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) queue[i] = 0;
        }
        return;
    }
    __syncthreads();                                        // (everybody has read the word before the next answer is written)
    int m0, n0;
    coords(b, m0, n0);

    f32x4 acc16[8][8];
    BufSrc<false, 256, NW> sa;
    BufSrc<B_KM, 256, NW, B_KM> sb;
    sa.init(A, lda, m0, M, K, wave, lane);
    init_b(sb, n0);
    sa.issue(smem, wave);
    sb.issue(smem + G::A_BYTES, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) sb.issue_keep(q, smem + G::STAGE_BYTES + G::A_BYTES, wave);   // k-tile 1 into the second stage, B first (the order the counted waits assume)
#pragma unroll
    for (int q = 0; q < 8; ++q) sa.issue_keep(q, smem + G::STAGE_BYTES, wave);
    sa.advance_all();
    sb.advance_all();

    s16x8 af[2][8], bfr[2][8];
    auto mfma16 = [](f32x4& d, const s16x8& a, const s16x8& bb) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(bb)); };
    // (the first k-step of an output tile starts its accumulators from the literal 0: no zeroing pass, and the 256 accumulator registers are never anything
    // but asm outputs in AGPRs -- assigned from C++ they are parked in VGPRs around the tile loop and spill)
    auto mfma16z = [](f32x4& d, const s16x8& a, const s16x8& bb) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(d) : "v"(a), "v"(bb)); };
    auto rdA = [&](const unsigned char* st, int ks, int i) { af[ks][i] = frag_kc16(st, wm * G::WM + i * 16, ks, lane); };
    auto rdB = [&](const unsigned char* st, int ks, int j) {
        if constexpr (B_KM) bfr[ks][j] = frag_km16_nowait<256>(st + G::A_BYTES, wn * G::WN + j * 16, ks, lane);
        else bfr[ks][j] = frag_kc16(st + G::A_BYTES, wn * G::WN + j * 16, ks, lane);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) rdA(smem, 0, i);
#pragma unroll
    for (int j = 0; j < 8; ++j) rdB(smem, 0, j);
    if (B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // one k-tile (the table of SPREAD -5, its evenly spread transfer pieces).  more1 / more2: the k-tiles one / two ahead exist -- of this output tile or of the next
    constexpr int NST = EPI == 1 ? 48 : EPI == 2 ? 48 : 32;   // stores of a wave's epilogue that the first counted waits behind it may leave in flight (EPI 2 issues 64:
                                                              // the field ends at 63 = 15 + 48 -- a smaller count only waits for more)
    bool behind_stores = false;   // this k-tile follows an epilogue: 32 stores of this wave sit between the transfers the counted waits name and the younger ones
    // EPI 2's epilogue needs registers for the gate / up pieces of a turn: an output tile's LAST k-tile then leaves the next tile's first fragments in the LDS
    // (rd_next false; its waits and barriers still say that k-tile is complete) and they are read behind the epilogue, as the prologue reads the first tile's.
    auto tile = [&](int t, auto fast_, bool more1_, bool more2_, auto zero_, bool rd_next_ = true) {
        constexpr bool FAST = decltype(fast_)::value;
        constexpr bool ZERO = decltype(zero_)::value;   // the output tile's first k-tile
        const bool more1 = FAST || more1_, more2 = FAST || more2_;
        const bool rd_next = more1 && (FAST || rd_next_);
        unsigned char* cur = smem + (t & 1) * G::STAGE_BYTES;
        const unsigned char* nxt = smem + ((t + 1) & 1) * G::STAGE_BYTES;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int ks = m >> 5, i = (m >> 2) & 7, j = (2 * m) & 7;
            if (ZERO && ks == 0) mfma16z(acc16[i][j], bfr[ks][j], af[ks][i]);
            else mfma16(acc16[i][j], bfr[ks][j], af[ks][i]);  // D[n][m]
            __builtin_amdgcn_sched_barrier(0);
            if (m < 8) {
                rdB(cur, 1, m);
            } else if (m == 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                       // 1: every wave holds all of B(t)
            } else if (m >= 14 && m < 22) {
                rdA(cur, 1, m - 14);
            } else if (m == 22) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                       // 2: every wave holds all of A(t)
            } else if (m == 28) {
                if (more1) {
                    if (ZERO && behind_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(15 + NST) : "memory");   // (15 + the stores issued behind B(t+1))
                    else if (more2) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // B(t+1) landed: A(t+1) and the pieces of tile t+2 issued so far stay in flight
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                   // 3: B(t+1) complete
                }
            } else if (m >= 29 && m < 37) {
                if (rd_next) rdB(nxt, 0, m - 29);
            } else if (m == 42) {
                if (more1) {
                    if (ZERO && behind_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NST) : "memory");
                    else if (more2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // A(t+1) landed
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                   // 4: A(t+1) complete
                }
            } else if (m >= 43 && m < 51) {
                if (rd_next) rdA(nxt, 0, m - 43);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ZERO && ks == 0) mfma16z(acc16[i][j + 1], bfr[ks][j + 1], af[ks][i]);
            else mfma16(acc16[i][j + 1], bfr[ks][j + 1], af[ks][i]);
            __builtin_amdgcn_sched_barrier(0);
            if (m >= 8 && m <= 53 && (m - 8) % 3 == 0) {
                if (more2) {
                    if (m < 32) sb.issue_keep((m - 8) / 3, cur + G::A_BYTES, wave);   // B(t+2): behind barrier 1
                    else sa.issue_keep((m - 32) / 3, cur, wave);                       // A(t+2): behind barrier 2
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more2) {
            sa.advance_all();
            sb.advance_all();
        }
        if (B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    for (;;) {
        // the next tile: asked for now (the atomic's round trip runs under the first k-tile), published behind it, read behind the second k-tile's barriers
        int asked = -1;
        if (threadIdx.x == 0) asked = grab();
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        tile(0, yes, true, true, yes);
        if (threadIdx.x == 0) nextw[0] = (unsigned)asked;
        tile(1, yes, true, true, no);
        const int bn = (int)nextw[0];
        const bool has_next = bn >= 0;
        int m0n = 0, n0n = 0;
        if (has_next) coords(bn, m0n, n0n);
        int t = 2;
        for (; t + 2 < nk; ++t) tile(t, yes, true, true, no);
        if (has_next) {   // the transfers continue into the next output tile: its k-tiles 0 and 1 are "t + 2" of this tile's last two k-tiles
            sa.init(A, lda, m0n, M, K, wave, lane);
            init_b(sb, n0n);
        }
        // (ONE copy of the last two k-tiles with run-time flags: a second copy in another branch would merge 256 accumulator registers where the paths join)
        tile(t, no, true, has_next, no);
        tile(t + 1, no, has_next, has_next, no, EPI != 2);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (asm: no hazard bookkeeping by the compiler) before they are read
        // ---- this tile's accumulators to memory.  Straight from the registers a lane has 8-byte pieces of sixteen different rows per store (measured: 9 us per
        // tile round, twice the plain kernel's LDS-staged epilogue -- it ate what the continued transfers gain); so every WAVE turns one 16-row block at a time
        // through 4 KB of LDS of its own, beside the stages the next tile is landing in: eight 8-byte writes (its eight 16x16 blocks of the row block), then four
        // 16-byte pieces per lane read back ALONG the rows and stored -- 256 contiguous bytes per row.  No barrier: the region is the wave's own, LDS operations
        // of a wave complete in order.
        if (!((abl & 8) && M != -12345)) {
            unsigned char* ep = smem + 2 * G::STAGE_BYTES + wave * 16 * EP;
            unsigned char* wr = ep + (lane & 15) * EP + 8 * (lane >> 4);                 // row m = lane & 15, columns 4 (lane >> 4) .. + 3 of a 16-column block
            const unsigned char* rd = ep + (lane >> 4) * EP + (lane & 15) * 16;          // piece p = lane + 64 q: row p / 16, 16-byte column p % 16
            bf16_t* cdst = C + (int64_t)(m0 + wm * G::WM + (lane >> 4)) * ldc + n0 + wn * G::WN + (lane & 15) * 8;
            // EPI 2 addresses h13 and dh13 through buffer descriptors at the wave's corner of the tile (wave-uniform: m0, n0 per block, wm, wn from the
            // readfirstlane'd wave number): ONE per-lane byte offset each, the turn's / piece's row in the scalar offset -- no 64-bit per-lane addresses for the
            // sixteen loads and stores of a turn (the kernel is at the register cap)
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
            const int h_row = (int)ld_act * 2, c_row = (int)ldc * 2;
            const int h_voff = (lane >> 4) * h_row + (lane & 15) * 16, c_voff = (lane >> 4) * c_row + (lane & 15) * 16;
#if defined(__HIP_DEVICE_COMPILE__)
            const int64_t corner_h = (int64_t)(m0 + wm * G::WM) * ld_act + n0 + wn * G::WN, corner_c = (int64_t)(m0 + wm * G::WM) * ldc + n0 + wn * G::WN;
            auto h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 ? (bf16_t*)ACT + corner_h : C), 0, EPI == 2 ? 128 * h_row : 0, 0x00020000);
            auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (EPI == 2 ? corner_c : 0)), 0, EPI == 2 ? 128 * c_row : 0, 0x00020000);
#endif
            // EPI 3: the positions of this lane's two rows of every turn, and the cos / sin pieces of a turn, one turn ahead
            int64_t rps[8][2];
            uint4 rcs[2][2], rsn[2][2];
            if constexpr (EPI == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) rps[i][q] = rot.pos[m0 + wm * G::WM + i * 16 + ((lane + 64 * q) >> 3)];
            }
            auto rot_fetch = [&](int i) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i0 = ((lane + 64 * q) & 7) * 8;
                    rcs[i & 1][q] = ld16((const bf16_t*)rot.cos + rps[i][q] * 64 + i0);
                    rsn[i & 1][q] = ld16((const bf16_t*)rot.sin + rps[i][q] * 64 + i0);
                }
            };
            // EPI 0, accumulating: the old C (or ie_linear_fwd_add's addend) pieces of a turn, requested ONE TURN AHEAD -- read where they are used they cost a
            // round trip to memory per piece with nothing beside it (measured in the step: + 80 us on the wo / w2 forward products)
            uint4 old[2][4];
            const bf16_t* obase = (EPI == 0 && ACT) ? static_cast<const bf16_t*>(ACT) + (cdst - C) : static_cast<const bf16_t*>(cdst);
            auto old_fetch = [&](int i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) old[i & 1][q] = ld16(obase + (int64_t)(i * 16 + q * 4) * ldc);
            };
            if constexpr (EPI == 0) {
                if (accumulate) old_fetch(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (EPI == 0) {
                    if (accumulate && i + 1 < 8) old_fetch(i + 1);
                }
                u32x4_t hg[4], hu[4];   // EPI 2: gate / up of this turn's four pieces per lane, requested first
#if defined(__HIP_DEVICE_COMPILE__)
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        hg[q] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, h_voff, (i * 16 + q * 4) * h_row, 0);
                        hu[q] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, h_voff, (i * 16 + q * 4) * h_row + 2 * F, 0);
                    }
                }
#endif
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint2 v;
                    v.x = pack2bf(acc16[i][j][0], acc16[i][j][1]);
                    v.y = pack2bf(acc16[i][j][2], acc16[i][j][3]);
                    *reinterpret_cast<uint2*>(wr + j * 32) = v;
                }
                if constexpr (EPI == 3) {
                    const int slot = (n0 + wn * G::WN) >> 7, gs = rot.qpk + 2;
                    const int g = slot / gs, sl = slot - g * gs;              // kv group, slot inside it (wave-uniform)
                    // the table rows of turn i + 1 are requested before turn i is worked on (their positions were read in front of the first turn): a turn's
                    // two dependent round trips (position, then its cos / sin rows) would otherwise be exposed eight times per tile
                    if (sl <= rot.qpk) {
                        if (i == 0) rot_fetch(0);
                        if (i + 1 < 8) rot_fetch(i + 1);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int pc = lane + 64 * q, prow = pc >> 3, sp = pc & 7;          // 16-column span sp of row prow of the turn
                        const int64_t t = m0 + wm * G::WM + i * 16 + prow;
                        bf16_t* dst;
                        if (sl < rot.qpk) dst = (bf16_t*)rot.q + (t * (int64_t)(rot.hkv * rot.qpk) + (g * rot.qpk + sl)) * 128;
                        else dst = (bf16_t*)rot.kv + ((t * 2 + (sl - rot.qpk)) * rot.hkv + g) * 128;
                        if (sl == rot.qpk + 1) {   // v: plain copy
                            st16_c(dst + sp * 16, *reinterpret_cast<const uint4*>(ep + prow * EP + sp * 32));
                            st16_c(dst + sp * 16 + 8, *reinterpret_cast<const uint4*>(ep + prow * EP + sp * 32 + 16));
                            continue;
                        }
                        const int i0 = sp * 8;   // first rotary-pair index of this span
                        float x1[8], x2[8];
                        if (rot.interleaved) {
                            float lo[8], hi[8];
                            unpack8(*reinterpret_cast<const uint4*>(ep + prow * EP + sp * 32), lo);
                            unpack8(*reinterpret_cast<const uint4*>(ep + prow * EP + sp * 32 + 16), hi);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                x1[e] = lo[2 * e]; x2[e] = lo[2 * e + 1];
                                x1[4 + e] = hi[2 * e]; x2[4 + e] = hi[2 * e + 1];
                            }
                        } else {
                            unpack8(*reinterpret_cast<const uint4*>(ep + prow * EP + i0 * 2), x1);
                            unpack8(*reinterpret_cast<const uint4*>(ep + prow * EP + (64 + i0) * 2), x2);
                        }
                        float co[8], si[8], o1[8], o2[8];
                        unpack8(rcs[i & 1][q], co);
                        unpack8(rsn[i & 1][q], si);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            o1[e] = x1[e] * co[e] - x2[e] * si[e];
                            o2[e] = x1[e] * si[e] + x2[e] * co[e];
                        }
                        if (sl < rot.qpk && rot.q_scale != 1.f) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o1[e] *= rot.q_scale, o2[e] *= rot.q_scale;
                        }
                        st16_c(dst + i0, pack8(o1));
                        st16_c(dst + 64 + i0, pack8(o2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // (dword by dword: two elements' temporaries live at a time -- the kernel sits at the register cap)
                        const uint4 g4 = *reinterpret_cast<const uint4*>(rd + q * 4 * EP);
                        const unsigned gw[4] = {g4.x, g4.y, g4.z, g4.w};
                        u32x4_t da, db;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float a0, b0, a1, b1, oc;
                            swiglu_bwd1(bflo(gw[e]), bflo(hg[q][e]), bflo(hu[q][e]), a0, b0, oc);
                            swiglu_bwd1(bfhi(gw[e]), bfhi(hg[q][e]), bfhi(hu[q][e]), a1, b1, oc);
                            da[e] = pack2bf(a0, a1);
                            db[e] = pack2bf(b0, b1);
                        }
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_raw_buffer_store_b128(da, c_rsrc, c_voff, (i * 16 + q * 4) * c_row, 2);           // (aux 2 = nt, as st16_c)
                        __builtin_amdgcn_raw_buffer_store_b128(db, c_rsrc, c_voff, (i * 16 + q * 4) * c_row + 2 * F, 2);
#endif
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (as EPI 1: one turn's arithmetic at a time)
                    continue;
                }
                if constexpr (EPI == 1) {   // a row = [gate 64 | up 64]; pair p = lane + 64 q: row p / 8, 8-column piece p % 8
#pragma unroll 1
                    for (int q = 0; q < 2; ++q) {
                        const int prow = (lane >> 3) + 8 * q, pc = lane & 7;
                        const uint4 gv4 = *reinterpret_cast<const uint4*>(ep + prow * EP + pc * 16), uv4 = *reinterpret_cast<const uint4*>(ep + prow * EP + 128 + pc * 16);
                        float gv[8], uv[8], o[8];
                        unpack8(gv4, gv);
                        unpack8(uv4, uv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = swiglu_fwd1(gv[e], uv[e]);
                        const int64_t gm = m0 + wm * G::WM + i * 16 + prow;
                        const int gn = n0 + wn * 64 + pc * 8;
                        bf16_t* dst = C + gm * ldc + gn;
                        st16_c(dst, gv4);
                        st16_c(dst + F, uv4);
                        st16_c(ACT + gm * ld_act + gn, pack8(o));
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (keeps the accumulator reads of the next row block behind this one's arithmetic: 256 live registers otherwise)
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint4 v = *reinterpret_cast<const uint4*>(rd + q * 4 * EP);
                    bf16_t* dst = cdst + (int64_t)(i * 16 + q * 4) * ldc;
                    if (accumulate) {   // (EPI 0 with ACT: the addend is ANOTHER matrix of C's geometry -- C = bf16(bf16(A B) + ACT), ie_linear_fwd_add)
                        float o[8], n[8];
                        if constexpr (EPI == 0) unpack8(old[i & 1][q], o);
                        else unpack8(ld16(dst), o);
                        unpack8(v, n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += n[e];
                        v = pack8(o);
                    }
                    if (abl & 4) st16(dst, v);   // (A/B: plain instead of non-temporal stores)
                    else st16_c(dst, v);
                }
            }
        }
        if (!has_next) {
            if (threadIdx.x == 0 && atomicAdd(queue + 8, 1u) == gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) queue[i] = 0;
            }
            break;
        }
        if constexpr (EPI == 2) {   // the next tile's first fragments (its k-tile 0 is complete in stage 0 since the last k-tile's barriers; nk is even)
#pragma unroll
            for (int i = 0; i < 8; ++i) rdA(smem, 0, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) rdB(smem, 0, j);
            if (B_KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        behind_stores = !accumulate && !((abl & 8) && M != -12345);   // (accumulating epilogues read C: their waits have drained everything)
        b = bn;
        m0 = m0n;
        n0 = n0n;
    }
}

}  // namespace

// called from gemm_bf16.hip's dispatcher; arguments already validated there (K % 64 == 0, N % 8 == 0, ...)
// The kernel instantiation the last GEMM launch of this process went to, as rocprofv3 prints it (bench.py checks the committed HBM-traffic
// measurement against the kernels the run really launches: a measurement of other schedules must not be reported as this run's traffic).
static int g_last_k[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // kind (1 gemm_dma_k, 2 gemm_bf16_k), BM, BN, WM, WN, a k-major, b k-major, schedule, epilogue
extern "C" int ie_gemm_note_kernel(int kind, int bm, int bn, int wm, int wn, int akm, int bkm, int sp, int epi) {
    const int v[9] = {kind, bm, bn, wm, wn, akm, bkm, sp, epi};
    for (int i = 0; i < 9; ++i) g_last_k[i] = v[i];
    return IE_OK;
}
extern "C" int ie_gemm_last_kernel(char* buf, int n) {
    if (!buf || n <= 0) return IE_ERR_INVALID;
    const int* k = g_last_k;
    if (k[0] == 1 && k[7] == -7) snprintf(buf, (size_t)n, "gemm_p5_k<%s, %d>", k[6] ? "true" : "false", k[8]);   // (the persistent frame: its own kernel)
    else if (k[0] == 1) snprintf(buf, (size_t)n, "gemm_dma_k<%d, %d, %d, %d, %s, %s, %d, %d>", k[1], k[2], k[3], k[4], k[5] ? "true" : "false", k[6] ? "true" : "false", k[7], k[8]);
    else if (k[0] == 2) snprintf(buf, (size_t)n, "gemm_bf16_k<%d, %d, %d, %d, %s, %s>", k[1], k[2], k[3], k[4], k[5] ? "true" : "false", k[6] ? "true" : "false");
    else snprintf(buf, (size_t)n, "none");
    return IE_OK;
}

static int g_gemm_persistent_grid = 256;   // blocks of the persistent kernel (one per CU; a multiple of 8: the XCD-contiguous numbering)
// tile queues of the persistent kernel: 64 slots of 16 words in module-global device memory (no allocation, zero at module load), ONE SLOT PER STREAM: the
// launches of a stream run one after the other, and the LAST block of every launch zeroes the slot (gemm_p5_k), so the next launch of that stream finds it clean
// and no launch of another stream ever touches it.  (Rounds 5 / 6 handed the slots out round-robin over all launches: two products running at the same time on
// two streams shared a slot once in 64 pairs and would have skipped tiles silently.)  The first 63 streams that launch a persistent product get a slot each; any
// further stream shares the last one, zeroed in stream order in front of every launch (the only place the memset is still needed).  A launch that faults leaves
// the process without a usable context, so nothing can inherit its counters.  ie_tune_gemm_queue_memset(1): the memset in front of EVERY launch -- rocclr runs
// it as two fill kernels, 386 per benchmark step: 673.1 / 671.0 ms with them, 669.3 / 668.2 ms without (profiles/r06_step_queue_memset_abab.log) -- for whoever
// suspects a slot.
__device__ unsigned g_p5_queues[64 * 16];
static int g_p5_queue_memset = 0;   // (ie_tune_gemm_queue_memset)
static hipEvent_t g_p5_shared_ev[16] = {};   // (the event of the last launch on the shared slot, per device)
static std::mutex g_p5_mu;
static thread_local bool t_p5_shared = false;
static unsigned* p5_queue_slot(hipStream_t st) {
    static unsigned* base[16] = {};   // per device: a module-global has one address on every device of the process
    static hipStream_t owner[16][63];
    static int owners[16] = {};
    thread_local hipStream_t last_st = nullptr;
    thread_local int last_dev = -1, last_idx = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    int idx = (last_dev == dev && last_st == st && last_idx >= 0) ? last_idx : -1;
    if (idx < 0) {
        std::lock_guard<std::mutex> lk(g_p5_mu);
        if (!base[dev] && hipGetSymbolAddress((void**)&base[dev], HIP_SYMBOL(g_p5_queues)) != hipSuccess) return nullptr;
        for (int i = 0; i < owners[dev] && idx < 0; ++i)
            if (owner[dev][i] == st) idx = i;
        if (idx < 0 && owners[dev] < 63) {
            idx = owners[dev]++;
            owner[dev][idx] = st;
        }
        if (idx < 0) idx = 63;   // (the shared slot)
        last_dev = dev; last_st = st; last_idx = idx;
    }
    unsigned* slot = base[dev] + 16 * idx;
    t_p5_shared = idx == 63;
    if (idx == 63) {   // the streams that share the last slot take turns: each launch waits for the event the launch before it on that slot recorded (p5_queue_done)
        std::lock_guard<std::mutex> lk(g_p5_mu);
        if (g_p5_shared_ev[dev] && hipStreamWaitEvent(st, g_p5_shared_ev[dev], 0) != hipSuccess) return nullptr;
    }
    if ((g_p5_queue_memset || idx == 63) && hipMemsetAsync(slot, 0, 9 * sizeof(unsigned), st) != hipSuccess) return nullptr;
    return slot;
}
// behind every launch of the persistent kernel: the shared slot's next user (any stream) must not start before this launch is through
static void p5_queue_done(hipStream_t st, const unsigned* slot) {
    if (!t_p5_shared) return;   // (set by p5_queue_slot on this thread for the launch just made)
    (void)slot;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return;
    std::lock_guard<std::mutex> lk(g_p5_mu);
    if (!g_p5_shared_ev[dev] && hipEventCreateWithFlags(&g_p5_shared_ev[dev], hipEventDisableTiming) != hipSuccess) { g_p5_shared_ev[dev] = nullptr; return; }
    (void)hipEventRecord(g_p5_shared_ev[dev], st);
}
extern "C" int ie_tune_gemm_queue_memset(int on) {
    if (on != 0 && on != 1) return IE_ERR_INVALID;
    g_p5_queue_memset = on;
    return IE_OK;
}
static int g_gemm_persistent_on = 1;        // ie_tune_gemm_persistent's mode (the fused w1 | w3 product is launched from this file)
extern "C" int ie_gemm_dma_set_persistent_grid(int blocks) {   // (ie_tune_gemm_persistent: 0 = off, 8 .. 1024 = that many blocks (mode > 1), 256 + on for mode 1)
    if (blocks == 0) { g_gemm_persistent_on = 0; return IE_OK; }
    if (blocks < 8 || blocks % 8) return IE_ERR_INVALID;
    g_gemm_persistent_on = blocks == 256 ? 1 : blocks;
    g_gemm_persistent_grid = blocks;
    return IE_OK;
}
// would the persistent frame take an [M, N, K] product of the 16x16x32 refill schedule?  (whole tiles, an even number of k-tiles, the K rule of gemm_bf16.hip's
// dispatcher, more tiles than blocks)
extern "C" int ie_gemm_dma_persistent_takes(int64_t M, int64_t N, int64_t K) {
    return g_gemm_persistent_on && M % 256 == 0 && N % 256 == 0 && K >= 256 && (K / 64) % 2 == 0 && (K <= 4096 || K >= 12288 || g_gemm_persistent_on > 1) &&
           (M / 256) * (N / 256) > g_gemm_persistent_grid;
}
static int g_gemm_group = 0;  // 0 = the kernel's default (4 tile rows per group)
extern "C" int ie_tune_gemm_group(int gm) {
    if (gm < 0 || gm > 64) return IE_ERR_INVALID;
    g_gemm_group = gm;
    return IE_OK;
}

extern "C" int ie_gemm_dma_launch(int shape, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C,
                                  int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* stream, IeGemmBatch bt) {
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* b = (const bf16_t*)B;
    bf16_t* c = (bf16_t*)C;
    static const int ablate = getenv("IE_GEMM_ABLATE") ? atoi(getenv("IE_GEMM_ABLATE")) & 15 : 0;
    accumulate = (accumulate ? 1 : 0) | (ablate << 4) | (g_gemm_group << 8);
#define IE_SHAPE(BM_, BN_, WM_, WN_, SP_)                                                                                            \
    do {                                                                                                                          \
        const int tiles_m = (int)((M + BM_ - 1) / BM_), tiles_n = (int)((N + BN_ - 1) / BN_);                                       \
        dim3 grid((unsigned)(tiles_m * tiles_n * bt.count)), block(64 * WM_ * WN_);                                                           \
        ie_gemm_note_kernel(1, BM_, BN_, WM_, WN_, a_kmajor != 0, b_kmajor != 0, SP_, 0);                                           \
        if (a_kmajor) {                                                                                                           \
            if (b_kmajor) hipLaunchKernelGGL((gemm_dma_k<BM_, BN_, WM_, WN_, true, true, SP_>), grid, block, 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt); \
            else hipLaunchKernelGGL((gemm_dma_k<BM_, BN_, WM_, WN_, true, false, SP_>), grid, block, 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt); \
        } else {                                                                                                                  \
            if (b_kmajor) hipLaunchKernelGGL((gemm_dma_k<BM_, BN_, WM_, WN_, false, true, SP_>), grid, block, 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt); \
            else hipLaunchKernelGGL((gemm_dma_k<BM_, BN_, WM_, WN_, false, false, SP_>), grid, block, 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt); \
        }                                                                                                                         \
    } while (0)
    if (shape == 0) IE_SHAPE(256, 256, 2, 4, 0);
    else if (shape == 1) IE_SHAPE(128, 128, 2, 2, 0);
    else if (shape == 2) IE_SHAPE(256, 256, 2, 4, 2);
    else if (shape == 3) IE_SHAPE(256, 256, 2, 4, 4);
    else if (shape == 4) IE_SHAPE(128, 128, 2, 2, 2);
    else if (shape == 5) IE_SHAPE(256, 256, 2, 4, -1);
    else if (shape == 6) IE_SHAPE(128, 128, 2, 2, -1);
    else if (shape == 7) IE_SHAPE(256, 256, 2, 2, -2);
    else if (shape == 8) IE_SHAPE(128, 256, 2, 4, -1);
    else if (shape == 10) IE_SHAPE(128, 256, 2, 4, -11);
    else if (shape == 11) IE_SHAPE(256, 256, 2, 4, -21);
    else if (shape == 12) IE_SHAPE(256, 256, 2, 2, -22);
    else if (shape == 13) IE_SHAPE(256, 256, 2, 2, -23);
    else if (shape == 14) IE_SHAPE(256, 256, 2, 2, -3);
    else if (shape == 15) IE_SHAPE(256, 256, 2, 2, -4);
    else if (shape == 16) {   // the refill schedule on 16x16x32 MFMAs: A k-contiguous (forward and input-gradient products)
        if (a_kmajor) return IE_ERR_UNSUPPORTED;
        ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, b_kmajor != 0, -5, 0);
        const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
        if (b_kmajor) hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, false, true, -5>), dim3((unsigned)(tiles_m * tiles_n * bt.count)), dim3(256), 0, st, a, lda, b, ldb, c, ldc,
                                         (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt);
        else hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, false, false, -5>), dim3((unsigned)(tiles_m * tiles_n * bt.count)), dim3(256), 0, st, a, lda, b, ldb, c, ldc,
                                (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt);
    }
    else if (shape == 17) {   // the one-wave-per-SIMD ring on 16x16x32 MFMAs: both operands k-major (the weight-gradient product)
        if (!a_kmajor || !b_kmajor) return IE_ERR_UNSUPPORTED;
        ie_gemm_note_kernel(1, 256, 256, 2, 2, 1, 1, -24, 0);
        const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
        hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, true, true, -24>), dim3((unsigned)(tiles_m * tiles_n * bt.count)), dim3(256), 0, st, a, lda, b, ldb, c, ldc,
                           (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n, bt);
    }
    else if (shape == 18) {   // the same schedule in the persistent frame (gemm_p5_k): whole 256x256 tiles, an even number of k-tiles, one product
        if (a_kmajor || bt.count != 1 || M % 256 || N % 256 || (K / 64) % 2 || K < 256) return IE_ERR_UNSUPPORTED;
        ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, b_kmajor != 0, -7, 0);
        const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
        const unsigned grid = (unsigned)std::min(tiles_m * tiles_n, g_gemm_persistent_grid);
        unsigned* qslot = p5_queue_slot(st);
        if (!qslot) return IE_ERR_LAUNCH;
        if (b_kmajor) hipLaunchKernelGGL((gemm_p5_k<true, 0>), dim3(grid), dim3(256), 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n,
                                         (bf16_t*)nullptr, (int64_t)0, 0, qslot, IeRotaryEpi{});
        else hipLaunchKernelGGL((gemm_p5_k<false, 0>), dim3(grid), dim3(256), 0, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n,
                                (bf16_t*)nullptr, (int64_t)0, 0, qslot, IeRotaryEpi{});
        p5_queue_done(st, qslot);
    }
    else IE_SHAPE(256, 256, 2, 4, -11);
#undef IE_SHAPE
    return ie_launch_status("ie_gemm_bf16 (dma) launch");
}

// C[M, N] (bf16) = (A[M, K] e4m3) (B[N, K] e4m3)^T * scale_a * scale_b on v_mfma_f32_32x32x64_f8f6f4 (schedule -6).  Called from fp8.hip, which has validated the
// arguments: K % 128 == 0, lda / ldb even multiples of 16 bytes, operands below 4 GiB, M, N >= 8.  The matrices are handed to the kernel as bf16 matrices of half the width.
extern "C" int ie_gemm_fp8_dma_launch(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb, void* C, int64_t ldc, int64_t sc, int64_t count,
                                      int64_t M, int64_t N, int64_t K, const float* scale_a, const float* scale_b, int accumulate, void* stream) {
    IeGemmBatch bt{(int)count, sa / 2, sb / 2, sc, 0, scale_a, 0, (void*)scale_b, 0};   // (a strided batch: product z reads scale_a[z], scale_b[z])
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + 255) / 256);
    ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, 0, -6, 0);
    hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, false, false, -6>), dim3((unsigned)(tiles_m * tiles_n * count)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A,
                       lda / 2, (const bf16_t*)B, ldb / 2, (bf16_t*)C, ldc, (int)M, (int)N, (int)(K / 2), (accumulate ? 1 : 0) | (g_gemm_group << 8), tiles_m, tiles_n, bt);
    return ie_launch_status("ie_gemm_fp8 launch");
}

// out = bf16(bf16(x w^T) + addend) in the persistent frame (round 6; ie_linear_fwd_add): the residual add of the transformer block in the row-parallel
// product's epilogue -- `r = dropout(h) + residual` (modeling_internlm2.py:707-717,728-737) behind wo and behind w2 -- so that the norm behind it reads one matrix
// instead of two.  The accumulating epilogue's arithmetic (the product rounded to bf16, added in fp32, rounded) with the addend read from another matrix of the
// output's geometry.  Only where the frame takes the product (ie_gemm_dma_persistent_takes); elsewhere IE_ERR_UNSUPPORTED and the caller adds in the norm kernel.
extern "C" int ie_linear_fwd_add(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* addend, void* out, int64_t ld_out, int64_t M, int64_t N,
                                 int64_t K, void* stream) {
    IE_CHECK_ARG(x && w && addend && out && addend != out && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ld_out >= N, "ie_linear_fwd_add: bad argument");
    IE_CHECK_SUPPORTED(ie_gemm_dma_persistent_takes(M, N, K) && ldx % 8 == 0 && ldw % 8 == 0 && ld_out % 8 == 0 && (((uintptr_t)x) & 15u) == 0 && (((uintptr_t)w) & 15u) == 0 &&
                           (((uintptr_t)addend) & 15u) == 0 && (((uintptr_t)out) & 15u) == 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31),
                       "ie_linear_fwd_add: only where the persistent GEMM frame takes the product (ie_gemm_dma_persistent_takes), 16-byte aligned rows");
    hipStream_t st = (hipStream_t)stream;
    ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, 0, -7, 0);
    const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
    unsigned* qslot = p5_queue_slot(st);
    if (!qslot) return IE_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_p5_k<false, 0>), dim3((unsigned)std::min(tiles_m * tiles_n, g_gemm_persistent_grid)), dim3(256), 0, st, (const bf16_t*)x, ldx,
                       (const bf16_t*)w, ldw, (bf16_t*)out, ld_out, (int)M, (int)N, (int)K, 1 | (g_gemm_group << 8), tiles_m, tiles_n,
                       const_cast<bf16_t*>((const bf16_t*)addend), (int64_t)0, 0, qslot, IeRotaryEpi{});
    p5_queue_done(st, qslot);
    return ie_launch_status("ie_linear_fwd_add launch");
}

// The wqkv forward product with the GQA split + rotary embedding in the persistent frame's epilogue (gemm_p5_k<false, 3>).  The caller (gemm_bf16.hip) has checked
// that the frame takes the product and that the head dimension is 128.
extern "C" int ie_gemm_qkv_rotary_dma_launch(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, IeRotaryEpi rot, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
    ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, 0, -7, 3);
    unsigned* qslot = p5_queue_slot(st);
    if (!qslot) return IE_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_p5_k<false, 3>), dim3((unsigned)std::min(tiles_m * tiles_n, g_gemm_persistent_grid)), dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                       (bf16_t*)nullptr, (int64_t)N, (int)M, (int)N, (int)K, (g_gemm_group << 8), tiles_m, tiles_n, (bf16_t*)nullptr, (int64_t)0, 0, qslot, rot);
    p5_queue_done(st, qslot);
    return ie_launch_status("ie_gemm_qkv_rotary (persistent) launch");
}

// The two FFN products with the SwiGLU arithmetic in their epilogues (refill schedule, 256x256 tiles; the caller has checked K % 64 == 0,
// F % 128 == 0 (fwd) and that the operands fit 32-bit buffer offsets).
//   fwd: h13[M, 2F] = x[M, K] @ (w1 | w3)[2F, K]^T and act[M, F] = silu(h13[:, :F]) * h13[:, F:]
//   bwd: dh13[M, 2F] = SwiGLU backward of (dy[M, K] @ w2[K, F]) at h13
extern "C" int ie_gemm_swiglu_dma_launch(int bwd, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* h13,
                                         int64_t ld_h13, void* act, int64_t ld_act, int64_t M, int64_t F, int64_t K, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    IeGemmBatch bt{1, 0, 0, 0, F, h13, ld_h13, act, ld_act};
    const int tiles_m = (int)((M + 255) / 256);
    const int flags = (g_gemm_group << 8);
    ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, bwd ? 1 : 0, bwd ? -4 : -5, bwd ? 2 : 1);
    if (!bwd) {
        const int tiles_n = (int)(F / 128);
        // the same product in the persistent frame (gemm_p5_k<false, 1>): whole tiles, an even number of k-tiles, more tiles than blocks, K as in gemm_bf16.hip's rule
        if (g_gemm_persistent_on && M % 256 == 0 && K >= 256 && (K / 64) % 2 == 0 && (K <= 4096 || K >= 12288 || g_gemm_persistent_on > 1) &&
            tiles_m * tiles_n > g_gemm_persistent_grid) {
            ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, 0, -7, 1);
            unsigned* qslot = p5_queue_slot(st);
            if (!qslot) return IE_ERR_LAUNCH;
            hipLaunchKernelGGL((gemm_p5_k<false, 1>), dim3((unsigned)g_gemm_persistent_grid), dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc,
                               (int)M, (int)(2 * F), (int)K, flags, tiles_m, tiles_n, (bf16_t*)act, ld_act, (int)F, qslot, IeRotaryEpi{});
            p5_queue_done(st, qslot);
            return ie_launch_status("ie_gemm_swiglu (persistent) launch");
        }
        hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, false, false, -5, 1>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (bf16_t*)C, ldc, (int)M, (int)(2 * F), (int)K, flags, tiles_m, tiles_n, bt);
    } else {
        const int tiles_n = (int)((F + 255) / 256);
        // the persistent frame with the SwiGLU backward in its wave-private epilogue (gemm_p5_k<true, 2>), where gemm_bf16.hip's rule sends the plain product there
        if (ie_gemm_dma_persistent_takes(M, F, K) && M * ld_h13 * 2 < (1ll << 31) && M * ldc * 2 < (1ll << 31)) {
            ie_gemm_note_kernel(1, 256, 256, 2, 2, 0, 1, -7, 2);
            unsigned* qslot = p5_queue_slot(st);
            if (!qslot) return IE_ERR_LAUNCH;
            hipLaunchKernelGGL((gemm_p5_k<true, 2>), dim3((unsigned)g_gemm_persistent_grid), dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc,
                               (int)M, (int)F, (int)K, flags, tiles_m, tiles_n, (bf16_t*)const_cast<void*>(h13), ld_h13, (int)F, qslot, IeRotaryEpi{});
            p5_queue_done(st, qslot);
            return ie_launch_status("ie_gemm_swiglu bwd (persistent) launch");
        }
        hipLaunchKernelGGL((gemm_dma_k<256, 256, 2, 2, false, true, -4, 2>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, (bf16_t*)C, ldc, (int)M, (int)F, (int)K, flags, tiles_m, tiles_n, bt);
    }
    return ie_launch_status("ie_gemm_swiglu (dma) launch");
}
