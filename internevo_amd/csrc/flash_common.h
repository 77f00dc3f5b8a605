// Shared pieces of the flash-attention kernels (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
// Fragment conventions (pinned on hardware by ie_mfma_probe):
//   A operand: lane l holds A[i = l & 31][k = 8*(l >> 5) + 0..7]     (8 bf16, k contiguous)
//   B operand: lane l holds B[k = 8*(l >> 5) + 0..7][j = l & 31]
//   C/D      : lane l, reg r holds D[i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][j = l & 31]
//
// Every tile (K, V, Q, dO: 64 rows x D) is brought HBM -> LDS by the DMA path (global_load_lds_dwordx4,
// 1 KiB per wave-instruction, lane-linear destination) in its NATURAL [row][D] layout; ONE image serves
// both kinds of MFMA operand:
//   * "row" fragments (the tile's rows are the MFMA i/j index, D is contracted): ds_read_b128,
//   * "transposed" fragments (the tile's rows are CONTRACTED, e.g. V in P.V, K in dS.K, Q in dS^T.Q,
//     dO in P^T.dO): two ds_read_b64_tr_b16 (hardware transposing load; lane i of a 16-lane group receives
//     X[r0..r0+3][c0 + i] when lane p addresses row r0 + (p >> 2), columns c0 + 4*(p & 3)..+3 -- pinned by
//     tools/probes/probe_lds.hip).
// The 16-byte slot s of row r is stored at slot s ^ swz(r) (the DMA permutes the per-lane SOURCE address,
// the reads apply the same XOR).  swz is a permutation of row bits chosen so that BOTH read kinds are
// bank-conflict free: the 16 rows of a ds_read_b128 lane group land in 16 distinct slots of the 256-byte
// bank row, and the 4 rows of a transposing read land in 4 distinct 64-byte blocks.
// Because the contraction order inside an MFMA is free, the rows fetched by a transposed fragment are
// chosen to equal the C/D register order of the producing MFMA:
//   slot e (0..7) of 16-row step s  <->  row 16*s + 8*(e >> 2) + 4*(l >> 5) + (e & 3)
// so score / probability accumulators feed the next MFMA without any cross-lane movement.
#pragma once
#include "ie_common.h"

namespace fa {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

template <int D> struct Geo {
    static constexpr int ROW_BYTES = 2 * D;           // 256 (D=128) or 128 (D=64)
    static constexpr int IMG_BYTES = 64 * ROW_BYTES;  // one 64-row tile image
    static constexpr int KS = D / 16;                 // k-steps over the head dim
    static constexpr int DB = D / 32;                 // 32-wide head-dim blocks
    static constexpr int SLOTS = D / 8;               // 16-byte slots per row
    static constexpr int ROWS_PER_DMA = 64 / SLOTS;   // rows written by one wave-instruction (4 or 8)
};

// slot permutation of row r (see header): D=128 (16 slots/row): ((r & 3) << 2) | ((r >> 2) & 3);
//                                         D=64  ( 8 slots/row): (((r >> 1) & 1) << 2) | ((r >> 2) & 3)
template <int D> __device__ __forceinline__ int swz(int r) {
    return D == 128 ? (((r & 3) << 2) | ((r >> 2) & 3)) : ((((r >> 1) & 1) << 2) | ((r >> 2) & 3));
}

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// DMA one 64-row tile (rows row0.., token stride ts elements) into `img`.  4 waves share the tile.
// Rows >= valid_rows re-read the last valid row (finite data; such rows are masked by the callers).
template <int D, int NW>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ src, int64_t ts, int valid_rows, unsigned char* img, int wave, int lane) {
    using G = Geo<D>;
    constexpr int INSTRS = 64 / G::ROWS_PER_DMA;
#pragma unroll
    for (int q = 0; q < INSTRS / NW; ++q) {
        const int g = wave + NW * q;
        const int row = g * G::ROWS_PER_DMA + lane / G::SLOTS;
        const int c = (lane % G::SLOTS) ^ swz<D>(row);
        const int rr = min(row, valid_rows - 1);
        dma16(src + (int64_t)rr * ts + c * 8, img + g * 1024);
    }
}

// Buffer-addressed twin of dma_tile for the kernels' inner loops: the per-lane part of a piece's source address is a byte
// offset VGPR computed ONCE (voff[q]), the tile / head part goes into the scalar offset of `buffer_load_dwordx4 .. lds`,
// so issuing a tile costs no vector ALU at all (dma_tile recomputes a clamped 64-bit address per piece per tile: ~6 VALU
// incl. 64-bit multiplies).  No row clamping: rows behind the sequence are rows of the next sequence (finite, masked by the
// callers like the clamped rows were) and rows behind the tensor read as zero (num_records).
template <int D, int NW>
struct TileSrc {
    static constexpr int PERW = (64 / Geo<D>::ROWS_PER_DMA) / NW;
    const bf16_t* base;
    uint32_t bytes;
    int ts2;          // token stride in bytes
    int voff[PERW];
    // `base` = first row any tile may start at; rows_to_end = rows from there to the end of the tensor, tail_elems = elements
    // of the last row that belong to the tensor (head_dim for a single head, heads * head_dim when the head goes into `extra`)
    __device__ __forceinline__ void init(const bf16_t* b, int64_t ts, int64_t rows_to_end, int64_t tail_elems, int wave, int lane) {
        using G = Geo<D>;
        base = b;
        ts2 = (int)(ts * 2);
        bytes = (uint32_t)(((rows_to_end - 1) * ts + tail_elems) * 2);
#pragma unroll
        for (int q = 0; q < PERW; ++q) {
            const int g = wave + NW * q;
            const int row = g * G::ROWS_PER_DMA + lane / G::SLOTS;
            const int c = (lane % G::SLOTS) ^ swz<D>(row);
            voff[q] = row * ts2 + c * 16;
        }
    }
    // (Measured in round 2: hipcc puts `s_waitcnt vmcnt(0)` in front of the first `ds_read_b64_tr_b16` after a DMA issue it knows about -- the
    // transposing read carries no address for its alias analysis -- so the tile requested for the next iteration is drained before the
    // current tile's P.V / dS.K products.  Issuing the transfers from inline asm removes that wait, and changes nothing: the tiles are
    // L2-resident and land within the tile's first product anyway, while the asm form costs 3-4 % on the two-waves-per-SIMD kernels
    // (M0 save / restore, no scheduling around it).  Kept as the builtin; profiles/r02_flash_attention.md.)
    // tile starting `row0` rows (+ `extra` elements, e.g. a head offset) behind base -> img
    __device__ __forceinline__ void issue(unsigned char* img, int row0, int extra, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
        const int soff = row0 * ts2 + extra * 2;
#pragma unroll
        for (int q = 0; q < PERW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(img + (wave + NW * q) * 1024), 16, voff[q], soff, 0, 0);
#endif
    }
    // Piece q (0 .. PERW - 1) of the same transfer from inline asm, for kernels with ONE wave per SIMD that place the pieces in MFMA gaps.
    // hipcc knows nothing about it: it neither drains it before the next LDS read (an LDS-DMA it knows about is a possible writer of every
    // LDS address: `s_waitcnt vmcnt(0)` in front of the next ds_read, i.e. the prefetch of the next tile waited for at once -- 1260 of the
    // 4460 cycles per tile of the dK / dV kernel) nor counts it in the vmcnt waits it inserts.  The caller waits with an explicit
    // `s_waitcnt vmcnt(n)`.  soff = row0 * ts2 + extra * 2 (bytes).
    __device__ __forceinline__ void issue_piece_asm(uint32_t img_lds, int soff, int wave, int q) const {
#if defined(__HIP_DEVICE_COMPILE__)
        i32x4 rs;
        rs[0] = (int)(uint32_t)(uintptr_t)base;
        rs[1] = (int)(((uintptr_t)base >> 32) & 0xffff);
        rs[2] = (int)bytes;
        rs[3] = 0x00020000;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(img_lds + (uint32_t)((wave + NW * q) * 1024)), "v"(voff[q]), "s"(rs), "s"(soff) : "memory");
#endif
    }
};

// Per-lane byte offsets of the fragments inside ANY tile image, computed once per kernel: the XOR swizzle only
// involves the low row bits, which do not depend on the 32-row block / 16-row step / image / pipeline stage, so
// inside the tile loop every fragment read is `ds_read base_register offset:immediate` with zero address VALU.
template <int D>
struct FragOffs {
    int row[Geo<D>::KS];    // "row" fragment of k-step ks (rows (l & 31) of a 32-row block)
    int tr0[Geo<D>::DB];    // "transposed" fragment, first  ds_read_b64_tr_b16 (rows 4h + 0..3 of a 16-row step), d-block db
    int tr1[Geo<D>::DB];    // second read (rows 8 + 4h + 0..3)
    __device__ __forceinline__ void init(int lane) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < Geo<D>::KS; ++ks) row[ks] = r * (2 * D) + (((ks * 2 + h) ^ swz<D>(r)) << 4);
        const int p = lane & 15, gq = lane >> 4;
        const int r0 = 4 * h + (p >> 2), r1 = r0 + 8;
#pragma unroll
        for (int db = 0; db < Geo<D>::DB; ++db) {
            const int col = 32 * db + 16 * (gq & 1) + 4 * (p & 3);
            tr0[db] = r0 * (2 * D) + ((((col >> 3) ^ swz<D>(r0)) << 4) | ((col & 7) << 1));
            tr1[db] = r1 * (2 * D) + ((((col >> 3) ^ swz<D>(r1)) << 4) | ((col & 7) << 1));
        }
    }
};

// "row" fragment: rows row0 + (l & 31) (row0 a multiple of 32), d = 16*ks + 8*(l >> 5) .. +7
template <int D>
__device__ __forceinline__ s16x8 row_frag(const unsigned char* img, int row0, int ks, const FragOffs<D>& fo) {
    return *reinterpret_cast<const s16x8*>(img + row0 * (2 * D) + fo.row[ks]);
}

// "transposed" fragment: d = 32*db + (l & 31); contracted rows of 16-row step `step` in C/D register order
template <int D>
__device__ __forceinline__ s16x8 trans_frag(const unsigned char* img, int db, int step, const FragOffs<D>& fo) {
    const unsigned char* base = img + step * 16 * (2 * D);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + fo.tr0[db]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + fo.tr1[db]));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// pack accumulator regs [8*half .. 8*half+7] of a 32x32 C/D tile into a bf16 fragment
__device__ __forceinline__ s16x8 pack_frag(const f32x16& acc, int half) {
    union { uint4 u; s16x8 s; } cv;
    cv.u.x = pack2bf(acc[8 * half + 0], acc[8 * half + 1]);
    cv.u.y = pack2bf(acc[8 * half + 2], acc[8 * half + 3]);
    cv.u.z = pack2bf(acc[8 * half + 4], acc[8 * half + 5]);
    cv.u.w = pack2bf(acc[8 * half + 6], acc[8 * half + 7]);
    return cv.s;
}

__device__ __forceinline__ int creg_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

__device__ __forceinline__ uint4 z4() { return make_uint4(0, 0, 0, 0); }

// Epilogue store of a transposed 32-row accumulator block set (O^T, dQ^T: lane = row l & 31; dK, dV: lane = key), D / 32 accumulators of 16 registers:
// registers 4 g .. 4 g + 3 of block db are columns 32 db + 8 g + 4 (l >> 5) + 0 .. 3 of the lane's row, so lanes l and l + 32 hold the two 8-byte halves
// of every 16-byte group.  Stored as 8 bytes per lane that is D / 8 store instructions per row block and lane, and the store tail of a block is
// bound by the NUMBER of store instructions (the waves of a block reach it together; cdna_hip_programming.md T21).  Exchanging the halves of two
// neighbouring groups (g, g + 1) with v_permlane32_swap -- lanes 32 .. 63 of the first operand against lanes 0 .. 31 of the second -- leaves 16
// contiguous bytes in every lane: lanes < 32 the whole group g, lanes >= 32 the whole group g + 1: half the store instructions, same bytes, same
// addresses.  `row` = this lane's row (16-byte aligned when `wide`), `mul` = the factor applied before rounding; `valid` must agree between lanes
// l and l + 32 (it does: they hold the same row); every lane of the wave must reach the call (the exchange runs unpredicated).
template <int D>
__device__ __forceinline__ void store_row_block(bf16_t* row, const f32x16 (&acc)[Geo<D>::DB], float mul, int lane, bool valid, bool wide) {
    if (wide) {
#pragma unroll
        for (int db = 0; db < Geo<D>::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const unsigned a0 = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul), a1 = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
                const unsigned b0 = pack2bf(acc[db][4 * g + 4] * mul, acc[db][4 * g + 5] * mul), b1 = pack2bf(acc[db][4 * g + 6] * mul, acc[db][4 * g + 7] * mul);
                const auto rx = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                if (valid) *reinterpret_cast<uint4*>(row + 32 * db + 8 * g + 8 * (lane >> 5)) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
    } else if (valid) {
#pragma unroll
        for (int db = 0; db < Geo<D>::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
                w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
                st8(row + 32 * db + 8 * g + 4 * (lane >> 5), w);
            }
    }
}

// ---- Round 6: the 32 rows of a wave (row = lane & 31) as MFMA operand fragments -- frag[ks] = the 8 elements ks * 16 + (lane >> 5) * 8 .. of the lane's row --
// fetched in 16-byte pieces that cover WHOLE rows (piece lane + 64 j: D / 8 pieces per row, 4 or 8 rows per instruction) and turned through a wave-private LDS
// block, instead of one 32-byte segment of every row per instruction: those gathers keep a CU's address path busy for ~25 us of a 2048-block kernel per tensor
// (measured on the dQ kernel: + 25 us for the O rows of the fused delta, + 30 us for the cos / sin gathers of the rotating store).  All NT tensors' pieces are
// requested before the first is turned (one round trip).  stage: 32 * 2 D bytes of this wave's own, 16-byte slots XOR-swizzled by the row (the fragment reads of
// 16 consecutive rows fall into 16 different slots); rows >= nvalid are zero.  Results: exactly the values the direct loads returned.
template <int D, int NT>
__device__ __forceinline__ void load_row_frags_staged(const bf16_t* const (&row0)[NT], const int64_t (&ts)[NT], int nvalid, unsigned char* stage, int lane,
                                                      s16x8 (&out)[NT][Geo<D>::KS]) {
    using G = Geo<D>;
    constexpr int PPR = D / 8;   // 16-byte pieces per row
    uint4 pc[NT][G::KS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < G::KS; ++j) {
            const int id = lane + 64 * j, r = id / PPR, sub = id % PPR;
            pc[t][j] = r < nvalid ? ld16(row0[t] + (int64_t)r * ts[t] + sub * 8) : z4();
        }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < G::KS; ++j) {
            const int id = lane + 64 * j, r = id / PPR, sub = id % PPR;
            *reinterpret_cast<uint4*>(stage + r * (2 * D) + ((sub ^ (r & (PPR - 1))) * 16)) = pc[t][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int r = lane & 31, half = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } x;
            x.u = *reinterpret_cast<const uint4*>(stage + r * (2 * D) + (((2 * ks + half) ^ (r & (PPR - 1))) * 16));
            out[t][ks] = x.s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next tensor overwrites the block)
    }
}

// ---- Round 6: the attention backward writing straight into the wqkv product's output-gradient layout [T][hkv][q heads per kv head + 2][D] with the rotary
// embedding's backward applied to dQ and dK on the way out (ie_flash_attn_bwd_qkv_rotary): qkv_rotary_bwd_k and its launch are gone from the step.
struct FaRotOut {
    const bf16_t* cs;        // cos / sin tables [positions][D / 2] (the forward's)
    const bf16_t* sn;
    const int64_t* pos;      // position of every token row
    int qpk;                 // q heads per kv head
    int64_t grp_stride;      // elements from one kv group to the next inside an output row ((qpk + 2) * D)
};

// store_row_block with the conjugate rotation: the lane holds d = 32 db + 8 g + 4 (lane >> 5) + i of its row for db = 0 .. 3, so a pair (d, d + 64) = (db, db + 2)
// sits in ONE lane.  The two-kernel path rounds the gradient to bf16 (store_row_block), reads it back, rotates in fp32 (rot_conj1) and rounds again: the same here.
// The cos / sin rows of the wave's 32 rows come through LDS (`stage`: FA_ROT_STAGE_BYTES of this wave's own, free at the end of the kernel): 16-byte pieces,
// eight per lane, instead of sixteen 8-byte gathers per lane with 64 different cache lines each -- the gathers kept the CU's address path busy for 40 us of the
// dQ kernel (measured: 810 -> 840 us with them).  pos_lane: the position of THIS lane's row (lane & 31); the others' come by shuffle.
// The 32 rows of a wave out of their accumulators as WHOLE-ROW 16-byte pieces (4 rows per store instruction) through a wave-private, XOR-swizzled LDS block of
// 32 * 2 D bytes, instead of store_row_block's one 32-byte segment of every row per instruction (the store-side twin of load_row_frags_staged).  Rows >= nvalid
// are not written.  Same bytes in memory.
template <int D>
__device__ __forceinline__ void store_rows_staged(bf16_t* row0, int64_t ts, int nvalid, const f32x16 (&acc)[Geo<D>::DB], float mul, int lane, unsigned char* stage) {
    using G = Geo<D>;
    constexpr int PPR = D / 8;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
            w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(stage + r * (2 * D) + (((4 * db + g) ^ (r & (PPR - 1))) * 16) + hh * 8) = w;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < G::KS; ++j) {
        const int id = lane + 64 * j, rr = id / PPR, sub = id % PPR;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + rr * (2 * D) + ((sub ^ (rr & (PPR - 1))) * 16));
        if (rr < nvalid) st16(row0 + (int64_t)rr * ts + sub * 8, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

constexpr int FA_ROT_ROW = 272;                       // bytes per staged row: cos [64] at 0, sin [64] at 128, padded so that the 32 rows spread over the banks
constexpr int FA_ROT_STAGE_BYTES = 32 * FA_ROT_ROW;   // per wave
// (row0 / ts / nvalid: the wave's 32 rows in memory -- the rotated rows leave through store_rows_staged, in the same LDS block)
template <int D>
__device__ __forceinline__ void store_row_block_rot(bf16_t* row0, int64_t ts, int nvalid, f32x16 (&acc)[Geo<D>::DB], float mul, int lane, const bf16_t* cs,
                                                    const bf16_t* sn, int pos_lane, unsigned char* stage) {
    static_assert(D == 128, "pairs (d, d + 64) of a 128-wide head");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int id = lane + 64 * j, r = id >> 4, sub = id & 15;          // piece `sub` (0 .. 7 cos, 8 .. 15 sin) of row r
        const int pr = __shfl(pos_lane, r, 64);
        const bf16_t* src = (sub < 8 ? cs : sn) + (int64_t)pr * (D / 2) + (sub & 7) * 8;
        *reinterpret_cast<uint4*>(stage + r * FA_ROT_ROW + sub * 16) = ld16(src);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's own rows: no barrier)
    const int hh = lane >> 5;
    const unsigned char* mine = stage + (lane & 31) * FA_ROT_ROW;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 c2 = *reinterpret_cast<const uint2*>(mine + (32 * db + 8 * g + 4 * hh) * 2);
            const uint2 s2 = *reinterpret_cast<const uint2*>(mine + 128 + (32 * db + 8 * g + 4 * hh) * 2);
            const float co[4] = {bflo(c2.x), bfhi(c2.x), bflo(c2.y), bfhi(c2.y)}, si[4] = {bflo(s2.x), bfhi(s2.x), bflo(s2.y), bfhi(s2.y)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d1 = rbf(acc[db][4 * g + i] * mul), d2 = rbf(acc[db + 2][4 * g + i] * mul);
                float x1, x2;
                rot_conj1(d1, d2, co[i], si[i], x1, x2);
                acc[db][4 * g + i] = x1;
                acc[db + 2][4 * g + i] = x2;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the cos / sin rows have been read: the block is free)
    store_rows_staged<D>(row0, ts, nvalid, acc, 1.f, lane, stage);
}

// ---- MFMAs with an explicit register file for every operand (kernels with one wave per SIMD: 256 arch VGPRs + 256 accumulation VGPRs).
// hipcc allocates every MFMA accumulator of a > 256-register kernel in the accumulation file and then copies whatever the vector
// ALU touches back and forth (v_accvgpr_read / _write: > 1000 of them in this kernel); with the operand files spelled out the
// scores S (vector-ALU food) stay in arch VGPRs, O and the Q fragments (MFMA-only) live in the accumulation file.  The asm
// statements are opaque to the hazard recognizer, so the wait states the ISA asks for are written here:
//   * VALU-written VGPR -> MFMA A/B operand: 2 states.  The only VALU-written operands are the packed P fragments, written in
//     phase 1 and read in phase 2 with at least 12 MFMAs in between (K / V fragments come from LDS, Q from the accumulation file);
//   * MFMA result -> any non-accumulate reader: 12 states for an 8-pass MFMA: mfma_settle() before the vector ALU (or the epilogue)
//     looks at accumulators right behind their last MFMA; inside the tile loop >= 16 other MFMAs separate writer and reader.
// (b_acc is a read-write operand although the MFMA only reads it: that pins the Q fragment in ONE accumulation register tuple
// for the whole kernel; as a plain input hipcc copies it into a fresh tuple in front of every statement)
__device__ __forceinline__ void mfma_s_first(f32x16& d, const s16x8& a, s16x8& b_acc) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, 0" : "=&v"(d), "+a"(b_acc) : "v"(a));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const s16x8& a, s16x8& b_acc) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(d), "+a"(b_acc) : "v"(a));
}
__device__ __forceinline__ void mfma_o(f32x16& d_acc, const s16x8& a, const s16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d_acc) : "v"(a), "v"(b));
}
// vector-ALU instructions that must stay where they are written (a volatile asm statement keeps its place among the MFMA
// statements; hipcc would hoist all 64 scale-and-subtracts of a phase to its top, outside any MFMA shadow) and that hipcc would
// otherwise wrap in canonicalising v_max (fmaxf on values it cannot prove quiet).  Plain VALU -> VALU dependencies are interlocked.
__device__ __forceinline__ float fma_pinned(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float mul_pinned(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * b - c
__device__ __forceinline__ float fms_pinned(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void max3_pinned(float& m, float a, float b) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_settle(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile("s_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void mfma_settle_acc(f32x16& a, f32x16& b) { asm volatile("s_nop 15" : "+a"(a), "+a"(b)); }
// keeps loop-carried accumulators in the accumulator file across a loop edge (hipcc otherwise carries them in arch VGPRs and
// copies all of them in and out -- 2 x 128 v_accvgpr moves per tile in the dK / dV kernel)
__device__ __forceinline__ void pin_acc(f32x16& a, f32x16& b) { asm volatile("" : "+a"(a), "+a"(b)); }

}  // namespace fa
