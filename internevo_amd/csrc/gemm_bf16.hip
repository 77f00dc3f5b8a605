// K3/K3b  bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate, bf16 out.
//
// One kernel template covers the three products of a linear layer's step
//   fwd   y  = x W^T   (A [M][K], B [N][K])            a_kmajor=0 b_kmajor=0
//   dgrad dx = dy W    (A [M][K], B stored [K][N])     a_kmajor=0 b_kmajor=1
//   wgrad dW = dy^T x  (A stored [K][M], B [K][N])     a_kmajor=1 b_kmajor=1
// Block tile 128x128x64, 4 wave64 (2x2), each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).
// Staging is register-staged and double-buffered in LDS: the global loads of tile t+1 are issued
// before the MFMAs of tile t and written to the other LDS buffer after them (one barrier per
// k-tile).  The LDS image is always [row][k] with k contiguous and a 144-byte row pitch
// (conflict-free ds_read_b128 for 16 consecutive rows: 144/4 = 36 banks apart).  Operands whose
// contraction index is the slow one in memory (k-major) are transposed IN REGISTERS while they are
// staged (4 k-rows x 8 columns per thread -> eight 8-byte LDS writes), so global loads stay
// 16-byte coalesced in every variant and the MFMA side never changes.
// The product is formed as D^T (B rows as the MFMA "A" operand) so every lane ends up with 4
// consecutive output columns per accumulator quad; the epilogue goes through LDS and leaves as
// 16-byte row-contiguous stores, optionally fused with the bf16 `grad += new` of autograd.
//
// Roofline: MFMA-bound; 2*M*N*K flop, algorithmic bytes 2*(M*K + N*K + M*N).
#include "ie_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int PITCH = 144;                 // bytes per LDS row (64 bf16 + 16 B pad)
constexpr int TILE_BYTES = BM * PITCH;     // 18432
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int CPITCH = 272;                // epilogue tile pitch in bytes (128 bf16 + 16 B pad)
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // 73728 >= 128*272

__device__ __forceinline__ uint4 zero4() { return make_uint4(0, 0, 0, 0); }

// ---- operand with k contiguous: tile = 128 rows x 64 k --------------------------------------------
__device__ __forceinline__ void gload_kcontig(const bf16_t* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int K,
                                              uint4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = threadIdx.x + 256 * q;
        const int row = c >> 3, kc = c & 7;
        const int gr = row0 + row, gk = k0 + kc * 8;
        r[q] = (gr < nrows && gk < K) ? ld16(P + (int64_t)gr * ld + gk) : zero4();
    }
}
__device__ __forceinline__ void swrite_kcontig(unsigned char* tile, const uint4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = threadIdx.x + 256 * q;
        const int row = c >> 3, kc = c & 7;
        st16(tile + row * PITCH + kc * 16, r[q]);
    }
}

// ---- operand stored [K][R] (k-major): tile = 64 k-rows x 128 columns -------------------------------
// thread -> (h = 4-row half of an 8-row k group, kg = k group, mc = 8-column chunk)
__device__ __forceinline__ void gload_kmajor(const bf16_t* __restrict__ P, int64_t ld, int col0, int ncols, int k0, int K,
                                             uint4 (&r)[4]) {
    const int h = threadIdx.x & 1, kg = (threadIdx.x >> 1) & 7, mc = threadIdx.x >> 4;
    const int col = col0 + mc * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + kg * 8 + 4 * h + j;
        r[j] = (k < K && col < ncols) ? ld16(P + (int64_t)k * ld + col) : zero4();
    }
}
__device__ __forceinline__ void swrite_kmajor(unsigned char* tile, const uint4 (&r)[4]) {
    const int h = threadIdx.x & 1, kg = (threadIdx.x >> 1) & 7, mc = threadIdx.x >> 4;
    unsigned char* base = tile + (mc * 8) * PITCH + kg * 16 + h * 8;
    const unsigned w0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
    const unsigned w1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
    const unsigned w2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
    const unsigned w3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint2 lo, hi;
        lo.x = (w0[q] & 0xffffu) | (w1[q] << 16);
        lo.y = (w2[q] & 0xffffu) | (w3[q] << 16);
        hi.x = (w0[q] >> 16) | (w1[q] & 0xffff0000u);
        hi.y = (w2[q] >> 16) | (w3[q] & 0xffff0000u);
        st8(base + (2 * q) * PITCH, lo);
        st8(base + (2 * q + 1) * PITCH, hi);
    }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_k(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                      int64_t ldb, bf16_t* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                      int accumulate, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    // ---- block -> output tile: XCD-aware (block b runs on XCD b % 8; give each XCD a contiguous
    // run of tiles so neighbours share A/B panels in that XCD's L2), then grouped along M.
    const int nblk = tiles_m * tiles_n;
    int id;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    constexpr int GM = 8;
    const int width = GM * tiles_n;
    const int group = id / width;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int pm = first_m + (id % width) % gsz;
    const int pn = (id % width) / gsz;
    const int m0 = pm * BM, n0 = pn * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    const int nk = (K + BK - 1) / BK;

    if (A_KM) gload_kmajor(A, lda, m0, M, 0, K, ra); else gload_kcontig(A, lda, m0, M, 0, K, ra);
    if (B_KM) gload_kmajor(B, ldb, n0, N, 0, K, rb); else gload_kcontig(B, ldb, n0, N, 0, K, rb);
    if (A_KM) swrite_kmajor(smem, ra); else swrite_kcontig(smem, ra);
    if (B_KM) swrite_kmajor(smem + TILE_BYTES, rb); else swrite_kcontig(smem + TILE_BYTES, rb);
    __syncthreads();

    const int frag_off = (lane & 31) * PITCH + (lane >> 5) * 16;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = (kt + 1) * BK;
            if (A_KM) gload_kmajor(A, lda, m0, M, k0, K, ra); else gload_kcontig(A, lda, m0, M, k0, K, ra);
            if (B_KM) gload_kmajor(B, ldb, n0, N, k0, K, rb); else gload_kcontig(B, ldb, n0, N, k0, K, rb);
        }
        const unsigned char* At = smem + (kt & 1) * STAGE_BYTES + (wm * 64) * PITCH + frag_off;
        const unsigned char* Bt = smem + (kt & 1) * STAGE_BYTES + TILE_BYTES + (wn * 64) * PITCH + frag_off;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const s16x8*>(At + i * 32 * PITCH + ks * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const s16x8*>(Bt + j * 32 * PITCH + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]
        }
        if (more) {
            unsigned char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (A_KM) swrite_kmajor(nxt, ra); else swrite_kcontig(nxt, ra);
            if (B_KM) swrite_kmajor(nxt + TILE_BYTES, rb); else swrite_kcontig(nxt + TILE_BYTES, rb);
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators (D^T layout: lane -> m, regs -> n) -> LDS [m][n] bf16 -> 16-byte stores
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                uint2 v;
                v.x = pack2bf(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]);
                v.y = pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                st8(smem + m * CPITCH + n * 2, v);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = threadIdx.x + 256 * q;
        const int row = c >> 4, nc = c & 15;
        const int gm = m0 + row, gn = n0 + nc * 8;
        if (gm < M && gn < N) {
            uint4 v = ld16(smem + row * CPITCH + nc * 16);
            bf16_t* dst = C + (int64_t)gm * ldc + gn;
            if (accumulate) {
                float o[8], n[8];
                unpack8(ld16(dst), o);
                unpack8(v, n);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += n[e];
                v = pack8(o);
            }
            st16(dst, v);
        }
    }
}

__global__ __launch_bounds__(256) void colsum_bf16_k(const bf16_t* __restrict__ x, int64_t ld, bf16_t* __restrict__ out, int64_t rows,
                                                     int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    for (int64_t r = 0; r < rows; ++r) acc += bf2f(x[r * ld + c]);
    out[c] = f2bf(acc);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" int ie_gemm_bf16(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, int accumulate, void* stream) {
    IE_CHECK_ARG(A && B && C, "ie_gemm_bf16: null pointer");
    IE_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "ie_gemm_bf16: negative size");
    IE_CHECK_ARG(M < (1ll << 30) && N < (1ll << 30) && K < (1ll << 30), "ie_gemm_bf16: size too large");
    IE_CHECK_ARG(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "ie_gemm_bf16: leading dimension too small");
    IE_CHECK_SUPPORTED(aligned16(A) && aligned16(B) && aligned16(C) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                       "ie_gemm_bf16: pointers must be 16-byte aligned and leading dimensions multiples of 8");
    IE_CHECK_SUPPORTED(N % 8 == 0 && (a_kmajor ? M % 8 == 0 : K % 8 == 0) && (b_kmajor ? true : K % 8 == 0),
                       "ie_gemm_bf16: N (and the contiguous extent of each operand) must be a multiple of 8");
    if (M == 0 || N == 0) return IE_OK;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
    hipStream_t st = (hipStream_t)stream;
#define IE_L(AK, BKM)                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_k<AK, BKM>), grid, dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, \
                       (int)M, (int)N, (int)K, accumulate, tiles_m, tiles_n)
    if (a_kmajor) { if (b_kmajor) IE_L(true, true); else IE_L(true, false); }
    else          { if (b_kmajor) IE_L(false, true); else IE_L(false, false); }
#undef IE_L
    return ie_launch_status("ie_gemm_bf16 launch");
}

extern "C" int ie_colsum_bf16(const void* x, int64_t ld, void* out, int64_t rows, int64_t cols, void* stream) {
    IE_CHECK_ARG(x && out && rows >= 0 && cols >= 0 && ld >= cols, "ie_colsum_bf16: bad argument");
    if (cols == 0) return IE_OK;
    hipLaunchKernelGGL(colsum_bf16_k, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld,
                       (bf16_t*)out, rows, cols);
    return ie_launch_status("ie_colsum_bf16 launch");
}
