// K3/K3b  bf16 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate, bf16 out.
//
// One kernel template covers the three products of a linear layer's step
//   fwd   y  = x W^T   (A [M][K], B [N][K])            a_kmajor=0 b_kmajor=0
//   dgrad dx = dy W    (A [M][K], B stored [K][N])     a_kmajor=0 b_kmajor=1
//   wgrad dW = dy^T x  (A stored [K][M], B [K][N])     a_kmajor=1 b_kmajor=1
// Block tile BM x BN x 64 with WAVES_M x WAVES_N wave64; each wave owns (BM/WAVES_M) x (BN/WAVES_N)
// as 32x32 MFMA tiles.  Staging is register-staged and double-buffered in LDS: the global loads of
// tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer in the MIDDLE
// of them (so the ds_writes overlap the second half of the MFMAs; one barrier per k-tile).  The LDS
// image is always [row][k] with k contiguous and a 144-byte row pitch (conflict-free ds_read_b128
// for 16 consecutive rows: 144/4 = 36 banks apart).  Operands whose contraction index is the slow
// one in memory (k-major) are transposed IN REGISTERS while they are staged (4 k-rows x 8 columns
// per thread -> eight 8-byte LDS writes), so global loads stay 16-byte coalesced in every variant
// and the MFMA side never changes.  The product is formed as D^T (B rows as the MFMA "A" operand) so
// every lane ends up with 4 consecutive output columns per accumulator quad; the epilogue goes
// through LDS and leaves as 16-byte row-contiguous stores, optionally fused with the bf16
// `grad += new` of autograd.
//
// Tile shapes: 256x256 (8 waves, 128x64 per wave: 0.75 ds_read_b128 per MFMA, 147 KB LDS, 1 block/CU),
// 256x128 / 128x256 (8 waves, 64x64 per wave) and 128x128 (4 waves, 2 blocks/CU); ie_gemm_bf16 picks
// the shape that wastes the fewest CU-rounds for the problem (wave quantisation on 256 CUs).
//
// Roofline: MFMA-bound; 2*M*N*K flop, algorithmic bytes 2*(M*K + N*K + M*N).
#include "ie_common.h"

extern "C" int ie_gemm_dma_launch(int shape, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C,
                                  int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* stream, IeGemmBatch bt);

extern "C" int ie_gemm_note_kernel(int kind, int bm, int bn, int wm, int wn, int akm, int bkm, int sp, int epi);
extern "C" int ie_gemm_swiglu_dma_launch(int bwd, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* h13,
                                         int64_t ld_h13, void* act, int64_t ld_act, int64_t M, int64_t F, int64_t K, void* stream);
extern "C" int ie_swiglu_fwd(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols, void* stream);
extern "C" int ie_swiglu_bwd(const void* dout, int64_t lddo, const void* a, int64_t lda, const void* b, int64_t ldb, void* da, int64_t ldda, void* db,
                             int64_t lddb, void* act, int64_t ldact, int64_t rows, int64_t cols, void* stream);

namespace {

constexpr int BK = 64;
constexpr int PITCH = 144;  // bytes per LDS row (64 bf16 + 16 B pad)

__device__ __forceinline__ uint4 zero4() { return make_uint4(0, 0, 0, 0); }

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct Cfg {
    static constexpr int NT = 64 * WAVES_M * WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_BYTES = BM * PITCH, B_BYTES = BN * PITCH;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int CPITCH = BN * 2 + 16;
    static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
    static_assert(BM * CPITCH <= SMEM_BYTES, "epilogue tile must fit in the staging buffers");
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "staging must divide evenly");
};

// ---- operand with k contiguous: tile = R rows x 64 k ------------------------------------------------
template <int R, int NT>
struct StageKC {
    static constexpr int PER = R * 8 / NT;
    uint4 r[PER];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ P, int64_t ld, int row0, int nrows, int k0, int K) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c >> 3, kc = c & 7;
            const int gr = row0 + row, gk = k0 + kc * 8;
            r[q] = (gr < nrows && gk < K) ? ld16(P + (int64_t)gr * ld + gk) : zero4();
        }
    }
    __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int c = threadIdx.x + NT * q;
            const int row = c >> 3, kc = c & 7;
            st16(tile + row * PITCH + kc * 16, r[q]);
        }
    }
};

// ---- operand stored [K][R] (k-major): tile = 64 k-rows x R columns ----------------------------------
// task -> (h = 4-row half of an 8-row k group, kg = k group, mc = 8-column chunk); 2R tasks of 4 loads
template <int R, int NT>
struct StageKM {
    static constexpr int TASKS = 2 * R;
    static constexpr int PER = (TASKS + NT - 1) / NT;
    uint4 r[PER][4];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ P, int64_t ld, int col0, int ncols, int k0, int K) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x + NT * q;
            const int h = t & 1, kg = (t >> 1) & 7, mc = t >> 4;
            const int col = col0 + mc * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kg * 8 + 4 * h + j;
                r[q][j] = (t < TASKS && k < K && col < ncols) ? ld16(P + (int64_t)k * ld + col) : zero4();
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x + NT * q;
            if (t >= TASKS) continue;
            const int h = t & 1, kg = (t >> 1) & 7, mc = t >> 4;
            unsigned char* base = tile + (mc * 8) * PITCH + kg * 16 + h * 8;
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
                st8(base + (2 * p) * PITCH, lo);
                st8(base + (2 * p + 1) * PITCH, hi);
            }
        }
    }
};

template <bool KM, int R, int NT> struct StagerSel { using type = StageKC<R, NT>; };
template <int R, int NT> struct StagerSel<true, R, NT> { using type = StageKM<R, NT>; };

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KM, bool B_KM>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_k(const bf16_t* __restrict__ A, int64_t lda,
                                                                      const bf16_t* __restrict__ B, int64_t ldb,
                                                                      bf16_t* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                                      int accumulate, int tiles_m, int tiles_n) {
    using G = Cfg<BM, BN, WAVES_M, WAVES_N>;
    constexpr int NT = G::NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM_BYTES];

    // ---- block -> output tile: XCD-aware (block b runs on XCD b % 8; give each XCD a contiguous
    // run of tiles so neighbours share A/B panels in that XCD's L2), then grouped along M.
    const int nblk = tiles_m * tiles_n;
    int id;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    constexpr int GM = (BM >= 256) ? 4 : 8;
    const int width = GM * tiles_n;
    const int group = id / width;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int pm = first_m + (id % width) % gsz;
    const int pn = (id % width) / gsz;
    const int m0 = pm * BM, n0 = pn * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename StagerSel<A_KM, BM, NT>::type sa;
    typename StagerSel<B_KM, BN, NT>::type sb;
    const int nk = (K + BK - 1) / BK;

    sa.load(A, lda, m0, M, 0, K);
    sb.load(B, ldb, n0, N, 0, K);
    sa.store(smem);
    sb.store(smem + G::A_BYTES);
    __syncthreads();

    const int frag_off = (lane & 31) * PITCH + (lane >> 5) * 16;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = (kt + 1) * BK;
            sa.load(A, lda, m0, M, k0, K);
            sb.load(B, ldb, n0, N, k0, K);
        }
        const unsigned char* At = smem + (kt & 1) * G::STAGE_BYTES + (wm * G::WM) * PITCH + frag_off;
        const unsigned char* Bt = smem + (kt & 1) * G::STAGE_BYTES + G::A_BYTES + (wn * G::WN) * PITCH + frag_off;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks == 2 && more) {  // stage the next tile while the second half of this tile's MFMAs run
                unsigned char* nxt = smem + ((kt + 1) & 1) * G::STAGE_BYTES;
                sa.store(nxt);
                sb.store(nxt + G::A_BYTES);
            }
            s16x8 af[G::TM], bfr[G::TN];
#pragma unroll
            for (int i = 0; i < G::TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(At + i * 32 * PITCH + ks * 32);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) bfr[j] = *reinterpret_cast<const s16x8*>(Bt + j * 32 * PITCH + ks * 32);
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int j = 0; j < G::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // D[n][m]
        }
        __syncthreads();
    }

    // ---- epilogue: accumulators (D^T layout: lane -> m, regs -> n) -> LDS [m][n] bf16 -> 16-byte stores
#pragma unroll
    for (int i = 0; i < G::TM; ++i) {
        const int m = wm * G::WM + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < G::TN; ++j) {
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
    __syncthreads();
    constexpr int CPR = BN / 8;  // 16-byte chunks per output row
#pragma unroll
    for (int q = 0; q < BM * CPR / NT; ++q) {
        const int c = threadIdx.x + NT * q;
        const int row = c / CPR, nc = c % CPR;
        const int gm = m0 + row, gn = n0 + nc * 8;
        if (gm < M && gn < N) {
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
            st16(dst, v);
        }
    }
}

// Column sums of a [rows, cols] bf16 matrix (bias gradient).  HBM-bound: 2 * rows * cols bytes.  A block owns 256 columns (32 lanes x 8
// columns, 16-byte loads) and splits the rows over its 8 row-lanes, 4 rows in flight per lane; the 8 partial sums of a column meet in
// LDS in a fixed order (deterministic, no atomics).  (The first version -- one thread per column walking the rows one dependent load
// at a time -- took 1.04 ms for the [4096, 12288] Wqkv bias gradient, 23 % of the INTERNLM_MoE step; profiles/r02_moe_kernel_stats.md.)
__global__ __launch_bounds__(256) void colsum_bf16_vec_k(const bf16_t* __restrict__ x, int64_t ld, bf16_t* __restrict__ out, int64_t rows,
                                                         int64_t cols) {
    __shared__ float part[8][256 + 8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * 256 + cx * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < cols) {
        int64_t r = ry;
        for (; r + 24 < rows; r += 32) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) unpack8(ld16(x + (r + 8 * u) * ld + c0), v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
        }
        for (; r < rows; r += 8) {
            float v[8];
            unpack8(ld16(x + r * ld + c0), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[ry][cx * 8 + e] = acc[e];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < cols) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += part[q][threadIdx.x];
        out[c] = f2bf(s);
    }
}

// any cols / alignment: one thread per column
__global__ __launch_bounds__(256) void colsum_bf16_k(const bf16_t* __restrict__ x, int64_t ld, bf16_t* __restrict__ out, int64_t rows,
                                                     int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    for (int64_t r = 0; r < rows; ++r) acc += bf2f(x[r * ld + c]);
    out[c] = f2bf(acc);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

template <int BM, int BN, int WM_, int WN_>
void launch_shape(int a_km, int b_km, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, bf16_t* C, int64_t ldc,
                  int M, int N, int K, int accumulate) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid((unsigned)(tiles_m * tiles_n)), block(64 * WM_ * WN_);
    ie_gemm_note_kernel(2, BM, BN, WM_, WN_, a_km != 0, b_km != 0, 0, 0);
#define IE_L(AK, BKM) \
    hipLaunchKernelGGL((gemm_bf16_k<BM, BN, WM_, WN_, AK, BKM>), grid, block, 0, st, A, lda, B, ldb, C, ldc, M, N, K, accumulate, tiles_m, tiles_n)
    if (a_km) { if (b_km) IE_L(true, true); else IE_L(true, false); }
    else      { if (b_km) IE_L(false, true); else IE_L(false, false); }
#undef IE_L
}

// CU-rounds a tile shape needs on 256 CUs (blocks_per_cu co-resident), weighted by a per-shape efficiency
// prior (bigger tiles do more MFMA work per staged byte).  Smaller is better.
inline double shape_cost(int64_t M, int64_t N, int bm, int bn, int blocks_per_cu, double eff, int batch = 1) {
    const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
    const int64_t slots = 256 * blocks_per_cu;
    const int64_t rounds = (tiles + slots - 1) / slots;
    return (double)rounds * slots * bm * bn / eff;  // ~ time: rounds x work per round / efficiency
}

int g_dgrad_refill_all = 0;   // (A/B hook ie_tune_gemm_dgrad_refill_all: every input-gradient product on the refill schedule, not only the long / wide ones)
int pick_variant(int64_t M, int64_t N, int64_t K, bool a_kmajor, bool b_kmajor, int batch = 1) {
    const bool any_kmajor = a_kmajor || b_kmajor;
    // priors from the round-1 micro-benchmarks (profiles/r01_gemm_tile_tuning.json): relative MFMA efficiency per tile shape
    if (K > 0 && K % 64 == 0 && M >= 8 && N >= 8) {  // LDS-DMA kernels: 256x256 unless wave quantisation on 256 CUs favours 128x128
        const double d256 = shape_cost(M, N, 256, 256, 1, 1.36, batch);  // measured: variants 11 / 13 vs the 128x128 kernels on 1.5-round shapes (wqkv)
        const double d128 = shape_cost(M, N, 128, 128, 2, 1.00, batch);
        // 256x256: k-contiguous operands -> one wave per SIMD (128x128 per wave), buffer-addressed DMA, fragments pipelined
        // across k-tiles; a k-major operand (two transposing reads per fragment) -> 8 waves, role-split load/compute phases of
        // two k-steps each, buffer-addressed DMA (+12 % dgrad, +7 % wgrad over one k-step per phase, profiles/)
        // a k-major operand (dgrad, wgrad) -> the k32 ring variants (four 32-deep LDS stages, counted vmcnt, asm transposing reads:
        // two to three entries of DMA stay in flight, where the two-stage kernels drained the DMA before every LOAD phase):
        // round-2 same-box A/B at 16 384 tokens (profiles/r02_gemm_ring_ab.jsonl): dgrad +4 ... +11 % with the 8-wave phased ring (15),
        // wgrad +7 ... +10 % with the one-wave-per-SIMD ring (16), +2 ... +4 % more with its DMA pieces split over both k-steps (17).  The forward product (both operands k-contiguous) stays on 11: its
        // ring version reads 64-byte row segments instead of whole 128-byte lines and measured 1 ... 12 % slower.
        // Operand-wise refill (19: the whole k-tile's fragments in registers, the two 64-deep stages refilled operand by operand ~1.5
        // tiles ahead behind counted waits, four barriers per tile): the forward product +3 ... +10 % over 11 on every 7B shape; dgrad
        // level with the phased ring except on long / wide products (w1|w3 dgrad K = 28672: +8 %) -- profiles/r02_gemm_refill_ab.jsonl
        if (d256 <= d128) {
            if (!any_kmajor) return 20;   // round 4: the same schedule on v_mfma_f32_16x16x32_bf16, +4 ... +5 % on every 7B forward shape (profiles/r04_gemm_mfma16_ab.md)
            if (a_kmajor && b_kmajor) return 17;
#ifndef IE_DGRAD_REFILL   // (A/B builds: -DIE_DGRAD_REFILL=19 keeps the long input-gradient products on the 32x32x16 refill schedule)
#define IE_DGRAD_REFILL 20
#endif
            return (!a_kmajor && (K >= 6144 || N >= 8192 || g_dgrad_refill_all)) ? IE_DGRAD_REFILL : 15;   // (20 on a k-major B: +1 ... +2 % over 19, profiles/r04_gemm_mfma16_ab.md)
        }
        return any_kmajor ? 5 : 8;    // 128x128: spreading helps the k-contiguous product only
    }
    const double c0 = shape_cost(M, N, 128, 128, 2, 1.00);
    const double c1 = shape_cost(M, N, 256, 256, 1, 1.30);
    const double c2 = shape_cost(M, N, 256, 128, 1, 1.12);
    const double c3 = shape_cost(M, N, 128, 256, 1, 1.12);
    int v = 0;
    double best = c0;
    if (c1 < best) { best = c1; v = 1; }
    if (c2 < best) { best = c2; v = 2; }
    if (c3 < best) { best = c3; v = 3; }
    return v;
}

// Tail split.  On 256 CUs a 256x256 tiling whose last round is at most half full (w2 dgrad / wgrad: 896 tiles = 3.5 rounds, wqkv
// fwd / wgrad: 384 = 1.5 rounds) leaves half the chip idle for a whole tile time.  The output is cut along one axis into a part
// that fills whole rounds and a remainder of <= 128 tiles, which is run as <= 256 tiles of 128x256 by a second launch: the
// last round then costs about 0.6 of a 256x256 round instead of 1.
// Measured (tools/gemm_tail_probe.py, bit-identical results): back-to-back launches of one shape gain +14 % (wqkv fwd), +4 % (wqkv
// wgrad), +3 % (w2 dgrad), +2 % (w2 wgrad); inside the training step the GEMM average improves 0.7 % and tokens/s does not move
// (same-box A/B, 2 x 2 runs: 19.68 / 19.70 k off, 19.73 / 19.67 k on) -- the idle half round was already giving the busy CUs a
// higher clock under the power limit, and the extra launch has its own ramp.  Hence OFF by default; kept as a tuning hook.
// 1 (default since round 5): eligible products of the 16x16x32 refill schedule run in the persistent frame (gemm_p5_k) on 256 blocks; n >= 8, a multiple of 8:
// on n blocks (tests: small products then walk several tiles per block); 0: the plain launch.  In the training step (same box, A B A B under rocprofv3,
// profiles/r05_step_gemm_persistent_abab.log): the plain forward products 800 -> 758 us per launch (-5.3 %), the step 676.8 -> 673.8 ms, the same loss bit for bit.
int g_gemm_persistent = 1;
int64_t g_gemm_persistent_skip_n = 0;   // (A/B hook: products with this many output columns stay on the plain launch)
int g_tail_split = 0;  // 0 = off, 1 = remainder by variant 14 if an operand is k-major else 12 (as measured), 2 = always 14, 3 = always 12
struct TailSplit {
    bool on, along_n;
    int64_t cut;
};
inline TailSplit tail_split(int64_t M, int64_t N) {
    TailSplit s{false, false, 0};
    if (M % 256 || N % 256) return s;
    const int64_t tm = M / 256, tn = N / 256, tiles = tm * tn, rem = tiles % 256, full = tiles - rem;
    if (full == 0 || rem == 0 || rem > 128) return s;
    if (full % tm == 0) s = TailSplit{true, true, full / tm * 256};
    else if (full % tn == 0) s = TailSplit{true, false, full / tn * 256};
    return s;
}

// ---- tail k-split of the weight-gradient products (round 6).  A 256x256 tiling whose last round is at most half full (wqkv: 384 tiles = 1.5 rounds) leaves half the
// chip to the other half for a whole tile time -- and a weight-gradient tile is 256 k-tiles long.  With a workspace registered (ie_gemm_set_wgrad_ksplit_workspace) the
// remainder region (<= 128 tiles, a rectangle: tail_split) is computed as a strided BATCH OF TWO products over the two halves of the contraction -- 256 blocks, one
// round of half length, in ONE launch of the same kernel (both operands are k-major: a half is a pointer offset) -- into bf16 partial tiles, and a small kernel adds
// the halves in fp32 in fixed order (half 0 + half 1, then the old gradient if the product accumulates) and rounds once more.  Deterministic; differs from the
// unsplit product by that one extra bf16 rounding of the two partial sums (the reference itself adds one bf16 partial per micro-batch: hybrid_zero_optim.py's
// AccumulateGrad on bf16 .grad).  Emulated first on the GPU (profiles/r06_wgrad_round_emulation.log): wqkv 734 -> 658 us, w2 1453 -> 1432 us.
void* g_ksplit_ws = nullptr;
int64_t g_ksplit_ws_bytes = 0;
__global__ __launch_bounds__(256) void ksplit_fixup_k(const bf16_t* __restrict__ p0, const bf16_t* __restrict__ p1, bf16_t* __restrict__ C, int64_t ldc, int64_t rows,
                                                      int64_t cols8, int accumulate) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= rows * cols8) return;
    const int64_t r = i / cols8, c = (i - r * cols8) * 8;
    float a[8], b2[8], o[8];
    unpack8(ld16(p0 + r * cols8 * 8 + c), a);
    unpack8(ld16(p1 + r * cols8 * 8 + c), b2);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] + b2[e];
    bf16_t* dst = C + r * ldc + c;
    if (accumulate) {   // bf16(old + bf16(sum)): the accumulate epilogue's arithmetic on the rounded sum
        float old[8];
        unpack8(ld16(dst), old);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = old[e] + rbf(o[e]);
    }
    st16(dst, pack8(o));
}

int gemm_dispatch(int variant, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc,
                  int64_t M, int64_t N, int64_t K, int accumulate, void* stream, IeGemmBatch bt = IeGemmBatch{1, 0, 0, 0}) {
    IE_CHECK_ARG(bt.count >= 1 && bt.count <= 4096, "ie_gemm_bf16_batched: batch count out of range");
    IE_CHECK_SUPPORTED(bt.count == 1 || (bt.sa % 8 == 0 && bt.sb % 8 == 0 && bt.sc % 8 == 0), "ie_gemm_bf16_batched: strides must be multiples of 8 elements");
    IE_CHECK_ARG(A && B && C, "ie_gemm_bf16: null pointer");
    IE_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "ie_gemm_bf16: negative size");
    IE_CHECK_ARG(M < (1ll << 30) && N < (1ll << 30) && K < (1ll << 30), "ie_gemm_bf16: size too large");
    IE_CHECK_ARG(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "ie_gemm_bf16: leading dimension too small");
    IE_CHECK_SUPPORTED(aligned16(A) && aligned16(B) && aligned16(C) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
                       "ie_gemm_bf16: pointers must be 16-byte aligned and leading dimensions multiples of 8");
    IE_CHECK_SUPPORTED(N % 8 == 0 && (a_kmajor ? M % 8 == 0 : K % 8 == 0) && (b_kmajor ? true : K % 8 == 0),
                       "ie_gemm_bf16: N (and the contiguous extent of each operand) must be a multiple of 8");
    IE_CHECK_ARG(variant >= -1 && variant <= 22, "ie_gemm_bf16: unknown tile variant");
    if (M == 0 || N == 0) return IE_OK;
    // variant 11 addresses its operands through 32-bit buffer offsets: each operand must span < 4 GiB
    const bool fits32 = /* also needed by variants 13, 14 */ (a_kmajor ? K : M) * lda * 2 < (1ll << 32) && (b_kmajor ? K : N) * ldb * 2 < (1ll << 32);
    if (variant < 0) {
        variant = pick_variant(M, N, K, a_kmajor != 0, b_kmajor != 0, bt.count);
        if ((variant == 11 || variant == 13 || variant >= 15) && !fits32) variant = 9;
        // the 16x16x32 refill schedule in its persistent frame (variant 22, gemm_p5_k) where it applies: whole 256x256 tiles, an even number of k-tiles, one
        // product, more tiles than one round (ie_tune_gemm_persistent; same results bit for bit)
        // (measured at 16 384 rows, profiles/r05_gemm_persistent_kbench.log, r05_gemm_persistent_k_rule.log: K = 2048 / 4096 -2 ... -2.4 %, 14 336 -1.2 %, 28 672 level,
        // but K = 6144 +1 % and 8192 level on both layouts: the frame is taken where it was seen to pay)
        if (g_gemm_persistent && variant == 20 && bt.count == 1 && M % 256 == 0 && N % 256 == 0 && K >= 256 && (K / 64) % 2 == 0 &&
            (g_gemm_persistent > 1 || K <= 4096 || K >= 12288) && (M / 256) * (N / 256) > (g_gemm_persistent == 1 ? 256 : g_gemm_persistent) &&
            N != g_gemm_persistent_skip_n)
            variant = 22;
        if (g_ksplit_ws && variant == 17 && bt.count == 1 && K % 128 == 0 && K >= 1024 && fits32) {   // the weight-gradient ring, both operands k-major
            const TailSplit ks = tail_split(M, N);
            const int64_t rm = ks.on ? (ks.along_n ? M : M - ks.cut) : 0, rn = ks.on ? (ks.along_n ? N - ks.cut : N) : 0;
            if (ks.on && 2 * rm * rn * 2 <= g_ksplit_ws_bytes) {
                const char* a = (const char*)A;
                const char* b = (const char*)B;
                char* c = (char*)C;
                // whole rounds: the plain product on the rectangle in front of the cut
                int rc = ks.along_n ? gemm_dispatch(variant, A, lda, 1, B, ldb, 1, C, ldc, M, ks.cut, K, accumulate, stream)
                                    : gemm_dispatch(variant, A, lda, 1, B, ldb, 1, C, ldc, ks.cut, N, K, accumulate, stream);
                if (rc != IE_OK) return rc;
                // the remainder: two half-k products as one strided batch into the workspace, then the fix-up
                const void* ar = ks.along_n ? A : (const void*)(a + 2 * ks.cut);          // (k-major: a column offset)
                const void* br = ks.along_n ? (const void*)(b + 2 * ks.cut) : B;
                bf16_t* ws = (bf16_t*)g_ksplit_ws;
                IeGemmBatch two{2, (K / 2) * lda, (K / 2) * ldb, rm * rn, 0, nullptr, 0, nullptr, 0};
                rc = ie_gemm_dma_launch(variant - 4, ar, lda, 1, br, ldb, 1, ws, rn, rm, rn, K / 2, 0, stream, two);
                if (rc != IE_OK) return rc;
                bf16_t* cr = (bf16_t*)(ks.along_n ? c + 2 * ks.cut : c + 2 * ks.cut * ldc);
                const int64_t n8 = rm * (rn / 8);
                hipLaunchKernelGGL(ksplit_fixup_k, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, ws + rm * rn, cr, ldc, rm, rn / 8, accumulate ? 1 : 0);
                return ie_launch_status("ie_gemm_bf16 (k-split fix-up) launch");
            }
        }
        const TailSplit ts = (g_tail_split && bt.count == 1 && (variant == 11 || variant == 13 || variant >= 15)) ? tail_split(M, N) : TailSplit{false, false, 0};
        if (ts.on) {
            const int tail_variant = g_tail_split == 2 ? 14 : g_tail_split == 3 ? 12 : ((a_kmajor || b_kmajor) ? 14 : 12);
            const char* a = (const char*)A;
            const char* b = (const char*)B;
            char* c = (char*)C;
            if (ts.along_n) {
                const int rc = gemm_dispatch(variant, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, ts.cut, K, accumulate, stream);
                if (rc != IE_OK) return rc;
                return gemm_dispatch(tail_variant, A, lda, a_kmajor, b + 2 * (b_kmajor ? ts.cut : ts.cut * ldb), ldb, b_kmajor, c + 2 * ts.cut, ldc, M,
                                     N - ts.cut, K, accumulate, stream);
            }
            const int rc = gemm_dispatch(variant, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, ts.cut, N, K, accumulate, stream);
            if (rc != IE_OK) return rc;
            return gemm_dispatch(tail_variant, a + 2 * (a_kmajor ? ts.cut : ts.cut * lda), lda, a_kmajor, B, ldb, b_kmajor, c + 2 * ts.cut * ldc, ldc,
                                 M - ts.cut, N, K, accumulate, stream);
        }
    }
    IE_CHECK_SUPPORTED((variant != 11 && variant < 13) || fits32, "ie_gemm_bf16: tile variants 11 and 13..17 need operands smaller than 4 GiB");
    if (variant >= 4) {  // LDS-DMA kernels (gemm_bf16_dma.hip): need whole 64-wide k-tiles and >= 8 valid rows/cols to clamp to
        IE_CHECK_SUPPORTED(K > 0 && K % 64 == 0 && M >= 8 && N >= 8, "ie_gemm_bf16: the LDS-DMA variants need K % 64 == 0");
        return ie_gemm_dma_launch(variant - 4, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, accumulate, stream, bt);
    }
    if (bt.count > 1) {   // the register-staged kernels (ragged K) take one product per launch
        for (int z = 0; z < bt.count; ++z) {
            const int rc = gemm_dispatch(variant, (const bf16_t*)A + z * bt.sa, lda, a_kmajor, (const bf16_t*)B + z * bt.sb, ldb, b_kmajor,
                                         (bf16_t*)C + z * bt.sc, ldc, M, N, K, accumulate, stream);
            if (rc != IE_OK) return rc;
        }
        return IE_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* b = (const bf16_t*)B;
    bf16_t* c = (bf16_t*)C;
    switch (variant) {
        case 0: launch_shape<128, 128, 2, 2>(a_kmajor, b_kmajor, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate); break;
        case 1: launch_shape<256, 256, 2, 4>(a_kmajor, b_kmajor, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate); break;
        case 2: launch_shape<256, 128, 4, 2>(a_kmajor, b_kmajor, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate); break;
        default: launch_shape<128, 256, 2, 4>(a_kmajor, b_kmajor, st, a, lda, b, ldb, c, ldc, (int)M, (int)N, (int)K, accumulate); break;
    }
    return ie_launch_status("ie_gemm_bf16 launch");
}

}  // namespace

extern "C" int ie_gemm_bf16(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, int accumulate, void* stream) {
    return gemm_dispatch(-1, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, accumulate, stream);
}

// ---- the FFN products with the SwiGLU arithmetic fused into their epilogues (a7) ---------------------------------------------------
// Fused (one launch, the refill-schedule kernel with EPI 1 / 2) when the plain dispatcher would have picked that schedule for the shape;
// otherwise the same two launches as before (product, then the elementwise kernel).  Results are bit-identical either way.
// bit 0: the forward product (on: -134 us per 16 384-row layer call, profiles/r03_ffn_fuse_ab.jsonl); bit 1: the w2 input-gradient product on EVERY shape of the
// refill schedule (off: on the plain launch measured level to slower -- the epilogues of all CUs coincide, gemm_bf16_dma.hip EPI 2); bit 2 (round 6, on): the w2
// input-gradient product where the PERSISTENT frame takes it (gemm_p5_k<true, 2>): in the training step (A B A B on one box under rocprofv3,
// profiles/r06_step_ffn_fuse_bwd_abab.log) the product goes 1326 -> 1624 us per launch and the 410-us swiglu_bwd_k launch behind it disappears:
// 672.3 / 672.6 -> 668.5 / 667.1 ms per step (-0.7 %), the same loss bit for bit.  (The epilogue's 134 MB per tile round move at HBM speed: it cannot get shorter, only hidden.)
static int g_ffn_fuse = 5;
extern "C" int ie_gemm_dma_persistent_takes(int64_t M, int64_t N, int64_t K);   // (gemm_bf16_dma.hip: would the persistent frame take this [M, N, K] product?)
extern "C" int ie_tune_ffn_fuse(int mode) {
    if (mode < 0 || mode > 7) return IE_ERR_INVALID;
    g_ffn_fuse = mode;
    return IE_OK;
}

// 1 when the shape takes the one-launch path (contiguous, 16-byte aligned operands assumed), 0 when it takes two launches
extern "C" int ie_gemm_swiglu_is_fused(int bwd, int64_t M, int64_t F, int64_t K) {
    if (!(g_ffn_fuse & (bwd ? 6 : 1)) || K <= 0 || K % 64 != 0 || M < 8 || F < 8) return 0;
    if (bwd) return pick_variant(M, F, K, false, true) == IE_DGRAD_REFILL && !(g_tail_split && tail_split(M, F).on) &&
                    ((g_ffn_fuse & 2) || ie_gemm_dma_persistent_takes(M, F, K));
    return F % 128 == 0 && pick_variant(M, 2 * F, K, false, false) == 20 && !(g_tail_split && tail_split(M, 2 * F).on);
}

extern "C" int ie_gemm_swiglu_fwd(const void* x, int64_t ldx, const void* w13, int64_t ldw, void* h13, int64_t ldh, void* act, int64_t ld_act, int64_t M,
                                  int64_t F, int64_t K, void* stream) {
    IE_CHECK_ARG(x && w13 && h13 && act, "ie_gemm_swiglu_fwd: null pointer");
    IE_CHECK_ARG(M >= 0 && F >= 0 && K >= 0 && M < (1ll << 30) && F < (1ll << 29) && K < (1ll << 30), "ie_gemm_swiglu_fwd: bad size");
    IE_CHECK_ARG(ldx >= K && ldw >= K && ldh >= 2 * F && ld_act >= F, "ie_gemm_swiglu_fwd: leading dimension too small");
    IE_CHECK_SUPPORTED(aligned16(act) && ld_act % 8 == 0 && F % 8 == 0, "ie_gemm_swiglu_fwd: act must be 16-byte aligned, F and its leading dimension multiples of 8");
    if (M == 0 || F == 0) return IE_OK;
    const bool fits32 = M * ldx * 2 < (1ll << 32) && 2 * F * ldw * 2 < (1ll << 32);
    const bool fuse = (g_ffn_fuse & 1) && K > 0 && K % 64 == 0 && F % 128 == 0 && M >= 8 && fits32 && aligned16(x) && aligned16(w13) && aligned16(h13) && ldx % 8 == 0 &&
                      ldw % 8 == 0 && ldh % 8 == 0 && pick_variant(M, 2 * F, K, false, false) == 20 && !(g_tail_split && tail_split(M, 2 * F).on);
    if (fuse) return ie_gemm_swiglu_dma_launch(0, x, ldx, w13, ldw, h13, ldh, nullptr, 0, act, ld_act, M, F, K, stream);
    const int rc = gemm_dispatch(-1, x, ldx, 0, w13, ldw, 0, h13, ldh, M, 2 * F, K, 0, stream);
    if (rc != IE_OK) return rc;
    return ie_swiglu_fwd(h13, ldh, (const bf16_t*)h13 + F, ldh, act, ld_act, M, F, stream);
}

extern "C" int ie_gemm_swiglu_bwd(const void* dy, int64_t ldy, const void* w2, int64_t ldw, const void* h13, int64_t ldh, void* dh13, int64_t ldd,
                                  void* dact_scratch, int64_t ld_scratch, int64_t M, int64_t F, int64_t K, void* stream) {
    IE_CHECK_ARG(dy && w2 && h13 && dh13 && dact_scratch, "ie_gemm_swiglu_bwd: null pointer");
    IE_CHECK_ARG(M >= 0 && F >= 0 && K >= 0 && M < (1ll << 30) && F < (1ll << 29) && K < (1ll << 30), "ie_gemm_swiglu_bwd: bad size");
    IE_CHECK_ARG(ldy >= K && ldw >= F && ldh >= 2 * F && ldd >= 2 * F && ld_scratch >= F, "ie_gemm_swiglu_bwd: leading dimension too small");
    IE_CHECK_SUPPORTED(aligned16(h13) && aligned16(dh13) && ldh % 8 == 0 && ldd % 8 == 0 && F % 8 == 0,
                       "ie_gemm_swiglu_bwd: h13 / dh13 must be 16-byte aligned, F and the leading dimensions multiples of 8");
    if (M == 0 || F == 0) return IE_OK;
    const bool fits32 = M * ldy * 2 < (1ll << 32) && K * ldw * 2 < (1ll << 32);
    const bool fits_h = M * ldh * 2 < (1ll << 31) && M * ldd * 2 < (1ll << 31);   // (the persistent epilogue addresses a wave's corner of h13 / dh13 by 32-bit offsets)
    const bool fuse = (g_ffn_fuse & 6) && K > 0 && K % 64 == 0 && M >= 8 && F >= 8 && fits32 && aligned16(dy) && aligned16(w2) && ldy % 8 == 0 && ldw % 8 == 0 &&
                      pick_variant(M, F, K, false, true) == IE_DGRAD_REFILL && !(g_tail_split && tail_split(M, F).on) &&
                      ((g_ffn_fuse & 2) || (fits_h && ie_gemm_dma_persistent_takes(M, F, K)));
    if (fuse) return ie_gemm_swiglu_dma_launch(1, dy, ldy, w2, ldw, dh13, ldd, h13, ldh, nullptr, 0, M, F, K, stream);
    const int rc = gemm_dispatch(-1, dy, ldy, 0, w2, ldw, 1, dact_scratch, ld_scratch, M, F, K, 0, stream);
    if (rc != IE_OK) return rc;
    return ie_swiglu_bwd(dact_scratch, ld_scratch, h13, ldh, (const bf16_t*)h13 + F, ldh, dh13, ldd, (bf16_t*)dh13 + F, ldd, nullptr, 0, M, F, stream);
}

// ---- the wqkv forward product with the GQA split + rotary embedding in its epilogue (a3 / a4; round 6) --------------------------------------------
// One launch (gemm_p5_k<false, 3>) when the head dimension is 128 and the persistent frame takes the [M, hkv (qpk + 2) 128, K] product; otherwise the product
// into `qkv_scratch` followed by ie_qkv_rotary_fwd_scaled -- bit-identical results either way (same arithmetic on the same bf16-rounded products).
// ie_tune_qkv_rotary_fuse(0) forces the two launches (A/B, tests).
extern "C" int ie_gemm_qkv_rotary_dma_launch(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, IeRotaryEpi rot, void* stream);
extern "C" int ie_qkv_rotary_fwd_scaled(const void* qkv, const void* cos_, const void* sin_, const int64_t* pos, void* q_out, void* kv_out, int64_t T, int hkv,
                                        int q_per_kv, int d, int interleaved, float q_scale, void* stream);
static int g_qkv_rotary_fuse = 1;
extern "C" int ie_tune_qkv_rotary_fuse(int mode) {
    if (mode < 0 || mode > 1) return IE_ERR_INVALID;
    g_qkv_rotary_fuse = mode;
    return IE_OK;
}
static bool qkv_rotary_fusable(int64_t M, int64_t N, int64_t K, int d) {
    return g_qkv_rotary_fuse && d == 128 && K > 0 && K % 64 == 0 && M * K * 2 < (1ll << 32) && N * K * 2 < (1ll << 32) &&
           pick_variant(M, N, K, false, false) == 20 && !(g_tail_split && tail_split(M, N).on) && N != g_gemm_persistent_skip_n && ie_gemm_dma_persistent_takes(M, N, K);
}
extern "C" int ie_gemm_qkv_rotary_is_fused(int64_t M, int hkv, int q_per_kv, int d, int64_t K) {
    return qkv_rotary_fusable(M, (int64_t)hkv * (q_per_kv + 2) * d, K, d) ? 1 : 0;
}
extern "C" int ie_gemm_qkv_rotary_fwd(const void* x, int64_t ldx, const void* wqkv, int64_t ldw, const void* cos_, const void* sin_, const int64_t* pos, void* q_out,
                                      void* kv_out, void* qkv_scratch, int64_t ld_scratch, int64_t M, int hkv, int q_per_kv, int d, int64_t K, int interleaved,
                                      float q_scale, void* stream) {
    IE_CHECK_ARG(x && wqkv && cos_ && sin_ && pos && q_out && kv_out && qkv_scratch, "ie_gemm_qkv_rotary_fwd: null pointer");
    IE_CHECK_ARG(M >= 0 && hkv > 0 && q_per_kv > 0 && K >= 0 && q_scale > 0.f && M < (1ll << 30) && K < (1ll << 30), "ie_gemm_qkv_rotary_fwd: bad shape");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_gemm_qkv_rotary_fwd: head dim must be 64 or 128");
    const int64_t N = (int64_t)hkv * (q_per_kv + 2) * d;
    IE_CHECK_ARG(ldx >= K && ldw >= K && ld_scratch == N, "ie_gemm_qkv_rotary_fwd: leading dimensions (the scratch product must be contiguous)");
    if (M == 0) return IE_OK;
    const bool fuse = qkv_rotary_fusable(M, N, K, d) && ldx == K && ldw == K && aligned16(x) && aligned16(wqkv) && aligned16(cos_) && aligned16(sin_) &&
                      aligned16(q_out) && aligned16(kv_out);
    if (fuse) {
        IeRotaryEpi rot{cos_, sin_, pos, q_out, kv_out, hkv, q_per_kv, interleaved ? 1 : 0, q_scale};
        return ie_gemm_qkv_rotary_dma_launch(x, ldx, wqkv, ldw, M, N, K, rot, stream);
    }
    const int rc = gemm_dispatch(-1, x, ldx, 0, wqkv, ldw, 0, qkv_scratch, ld_scratch, M, N, K, 0, stream);
    if (rc != IE_OK) return rc;
    return ie_qkv_rotary_fwd_scaled(qkv_scratch, cos_, sin_, pos, q_out, kv_out, M, hkv, q_per_kv, d, interleaved, q_scale, stream);
}

extern "C" int ie_gemm_bf16_tile(int variant, const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C,
                                 int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate, void* stream) {
    return gemm_dispatch(variant, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, accumulate, stream);
}

extern "C" int ie_gemm_bf16_batched(const void* A, int64_t lda, int64_t stride_a, int a_kmajor, const void* B, int64_t ldb, int64_t stride_b,
                                    int b_kmajor, void* C, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K, int batch, int accumulate,
                                    void* stream) {
    IE_CHECK_ARG(batch >= 1, "ie_gemm_bf16_batched: batch must be >= 1");
    return gemm_dispatch(-1, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, accumulate, stream, IeGemmBatch{batch, stride_a, stride_b, stride_c});
}

extern "C" int ie_gemm_dma_set_persistent_grid(int blocks);
extern "C" int ie_tune_gemm_persistent(int mode) {
    if (mode < 0 || mode > 1024 || (mode > 1 && mode % 8)) return IE_ERR_INVALID;
    g_gemm_persistent = mode;
    return ie_gemm_dma_set_persistent_grid(mode == 0 ? 0 : mode > 1 ? mode : 256);
}

extern "C" int ie_gemm_set_wgrad_ksplit_workspace(void* ws, int64_t bytes) {
    if ((ws == nullptr) != (bytes == 0) || bytes < 0 || (((uintptr_t)ws) & 15u)) return IE_ERR_INVALID;
    g_ksplit_ws = ws;
    g_ksplit_ws_bytes = bytes;
    return IE_OK;
}

extern "C" int ie_tune_gemm_dgrad_refill_all(int on) {
    g_dgrad_refill_all = on ? 1 : 0;
    return IE_OK;
}

extern "C" int ie_tune_gemm_persistent_skip_n(int64_t n_cols) {
    if (n_cols < 0) return IE_ERR_INVALID;
    g_gemm_persistent_skip_n = n_cols;
    return IE_OK;
}

extern "C" int ie_tune_gemm_tail_split(int mode) {
    if (mode < 0 || mode > 3) return IE_ERR_INVALID;
    g_tail_split = mode;
    return IE_OK;
}

extern "C" int ie_colsum_bf16(const void* x, int64_t ld, void* out, int64_t rows, int64_t cols, void* stream) {
    IE_CHECK_ARG(x && out && rows >= 0 && cols >= 0 && ld >= cols, "ie_colsum_bf16: bad argument");
    if (cols == 0) return IE_OK;
    const bool vec = cols % 8 == 0 && ld % 8 == 0 && (((uintptr_t)x) & 15u) == 0;
    if (vec)
        hipLaunchKernelGGL(colsum_bf16_vec_k, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld,
                           (bf16_t*)out, rows, cols);
    else
        hipLaunchKernelGGL(colsum_bf16_k, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld,
                           (bf16_t*)out, rows, cols);
    return ie_launch_status("ie_colsum_bf16 launch");
}
