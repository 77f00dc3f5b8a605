// Shared pieces of the flash-attention kernels (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
// Fragment conventions used everywhere (pinned on hardware by ie_mfma_probe):
//   A operand: lane l holds A[i = l & 31][k = 8*(l >> 5) + 0..7]     (8 bf16, k contiguous)
//   B operand: lane l holds B[k = 8*(l >> 5) + 0..7][j = l & 31]
//   C/D      : lane l, reg r holds D[i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][j = l & 31]
//
// "Row" LDS images are [row][D] with a (2*D + 16)-byte pitch (conflict-free ds_read_b128 of 16 rows).
// "Transposed" LDS images are [D][R] (R = 64 rows of the tile) with a (2*R + 8)-byte pitch; they are
// built by an in-register 4x8 transpose at staging time and consumed with two ds_read_b64 per
// fragment.  Because the contraction order inside an MFMA is free, the k-slot <-> row mapping of a
// transposed fragment is chosen to equal the C/D register order of the producing MFMA:
//   slot e (0..7) of k-step s  <->  row 16*s + 8*(e >> 2) + 4*(l >> 5) + (e & 3)
// so score/probability accumulators feed the next MFMA without any cross-lane movement.
#pragma once
#include "ie_common.h"

namespace fa {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

template <int D> struct Geo {
    static constexpr int ROW_PITCH = 2 * D + 16;   // bytes
    static constexpr int T_PITCH = 2 * 64 + 8;     // bytes, 64-row transposed tile
    static constexpr int KS = D / 16;              // k-steps over the head dim
    static constexpr int DB = D / 32;              // 32-wide head-dim blocks
    static constexpr int CHUNKS_PER_ROW = D / 8;   // 16-byte chunks per row
};

__device__ __forceinline__ uint4 z4() { return make_uint4(0, 0, 0, 0); }

// ---- stage ROWS x D rows (row-major, token stride ts elements) into a "row" LDS image ------------
// src points at (first row, head, d = 0).  Rows >= valid_rows are zero-filled.
template <int D, int ROWS, int NT>
struct RowStager {
    static constexpr int CPR = D / 8;
    static constexpr int TOTAL = ROWS * CPR;
    static constexpr int PER = (TOTAL + NT - 1) / NT;
    uint4 r[PER];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, int64_t ts, int valid_rows) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c / CPR, ch = c % CPR;
            r[q] = (c < TOTAL && row < valid_rows) ? ld16(src + (int64_t)row * ts + ch * 8) : z4();
        }
    }
    __device__ __forceinline__ void store(unsigned char* lds) const {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c / CPR, ch = c % CPR;
            if (c < TOTAL) st16(lds + row * (2 * D + 16) + ch * 16, r[q]);
        }
    }
};

// ---- stage 64 rows x D into a transposed LDS image [D][64] -----------------------------------------
// task = (rg = group of 4 rows, 0..15 ; dc = 8-column chunk); rg is the fast index across lanes so a
// 16-lane ds_write_b64 group writes one contiguous 128-byte image row.
template <int D, int NT>
struct TransStager {
    static constexpr int TASKS = 16 * (D / 8);
    static constexpr int PER = (TASKS + NT - 1) / NT;
    uint4 r[PER][4];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, int64_t ts, int valid_rows) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x + NT * q;
            const int rg = t & 15, dc = t >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rg * 4 + j;
                r[q][j] = (t < TASKS && row < valid_rows) ? ld16(src + (int64_t)row * ts + dc * 8) : z4();
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* lds) const {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x + NT * q;
            const int rg = t & 15, dc = t >> 4;
            if (t >= TASKS) continue;
            unsigned char* base = lds + (dc * 8) * (2 * 64 + 8) + rg * 8;
            const unsigned w0[4] = {r[q][0].x, r[q][0].y, r[q][0].z, r[q][0].w};
            const unsigned w1[4] = {r[q][1].x, r[q][1].y, r[q][1].z, r[q][1].w};
            const unsigned w2[4] = {r[q][2].x, r[q][2].y, r[q][2].z, r[q][2].w};
            const unsigned w3[4] = {r[q][3].x, r[q][3].y, r[q][3].z, r[q][3].w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint2 lo, hi;
                lo.x = (w0[p] & 0xffffu) | (w1[p] << 16);
                lo.y = (w2[p] & 0xffffu) | (w3[p] << 16);
                hi.x = (w0[p] >> 16) | (w1[p] & 0xffff0000u);
                hi.y = (w2[p] >> 16) | (w3[p] & 0xffff0000u);
                st8(base + (2 * p) * (2 * 64 + 8), lo);
                st8(base + (2 * p + 1) * (2 * 64 + 8), hi);
            }
        }
    }
};

// fragment of a "row" image: rows row0 + (l & 31), k-step ks
template <int D>
__device__ __forceinline__ s16x8 row_frag(const unsigned char* img, int row0, int ks, int lane) {
    return *reinterpret_cast<const s16x8*>(img + (row0 + (lane & 31)) * (2 * D + 16) + ks * 32 + (lane >> 5) * 16);
}

// fragment of a transposed image [D][64]: d = d0 + (l & 31); rows (k-slots) of 16-row step `step`
__device__ __forceinline__ s16x8 trans_frag(const unsigned char* img, int d0, int step, int lane) {
    const unsigned char* p = img + (d0 + (lane & 31)) * (2 * 64 + 8) + (step * 16 + 4 * (lane >> 5)) * 2;
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    const uint2 b = *reinterpret_cast<const uint2*>(p + 16);
    union { uint4 u; s16x8 s; } cv;
    cv.u = make_uint4(a.x, a.y, b.x, b.y);
    return cv.s;
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

}  // namespace fa
