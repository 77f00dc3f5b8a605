// K5 RMSNorm forward / backward for gfx950.  HBM-bound: one wave64 owns one row, 16-byte loads,
// the whole row lives in registers between the statistics pass and the normalise pass, so every
// byte is read from HBM exactly once.  Algorithmic bytes/row: fwd 2*C in + 2*C out (+4 rstd);
// bwd 2*C (dy) + 2*C (x) [+2*C dres] in, 2*C out.
//
// Numerics follow the reference's own torch statement internlm/model/ops/norm.py:10-23
// (manual_rms_norm): variance in fp32, x*rsqrt(var+eps) in fp32, ROUNDED to the weight dtype when
// that is bf16, then multiplied by the weight and rounded again.
#include "ie_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

template <typename T> struct Ld;
template <> struct Ld<bf16_t> { static __device__ __forceinline__ float get(const void* p, int64_t i) { return bf2f(((const bf16_t*)p)[i]); } };
template <> struct Ld<float> { static __device__ __forceinline__ float get(const void* p, int64_t i) { return ((const float*)p)[i]; } };

__device__ __forceinline__ void store_out(void* p, int64_t i, float v, bool bf) {
    if (bf) ((bf16_t*)p)[i] = f2bf(v); else ((float*)p)[i] = v;
}

// ---------------------------------------------------------------- vectorised forward (bf16 x, bf16 w)
// cols == NCH * 512.  HAS_ADD: r = bf16(a + b) is the normalised tensor and is written to r_out.
template <int NCH, bool HAS_ADD>
__global__ __launch_bounds__(256) void rmsnorm_fwd_vec(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ r_out, const bf16_t* __restrict__ w,
                                                       bf16_t* __restrict__ y, float* __restrict__ rstd,
                                                       int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int C = NCH * 512;
    const bf16_t* ap = a + row * C;
    float xv[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 512 + lane * 8;
        uint4 va = ld16(ap + c);
        unpack8(va, xv[j]);
        if (HAS_ADD) {
            float bv[8];
            uint4 vb = ld16(b + row * C + c);
            unpack8(vb, bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[j][e] = rbf(xv[j][e] + bv[e]);
            st16(r_out + row * C + c, pack8(xv[j]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += xv[j][e] * xv[j][e];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)C + eps);
    if (lane == 0) rstd[row] = rs;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 512 + lane * 8;
        float wv[8], o[8];
        unpack8(ld16(w + c), wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = wv[e] * rbf(xv[j][e] * rs);
        st16(y + row * C + c, pack8(o));
    }
}

// ---------------------------------------------------------------- generic forward (any cols / dtypes)
template <typename XT, bool HAS_ADD>
__global__ __launch_bounds__(256) void rmsnorm_fwd_gen(const void* __restrict__ a, const void* __restrict__ b,
                                                       void* __restrict__ r_out, const void* __restrict__ w,
                                                       int w_bf16, void* __restrict__ y, float* __restrict__ rstd,
                                                       int64_t rows, int64_t cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * cols;
    float ss = 0.f;
    for (int64_t c = lane; c < cols; c += 64) {
        float v = Ld<XT>::get(a, base + c);
        if (HAS_ADD) {
            v = rbf(v + Ld<XT>::get(b, base + c));
            ((bf16_t*)r_out)[base + c] = f2bf(v);
        }
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)cols + eps);
    if (lane == 0) rstd[row] = rs;
    for (int64_t c = lane; c < cols; c += 64) {
        float v = Ld<XT>::get(a, base + c);
        if (HAS_ADD) v = rbf(v + Ld<XT>::get(b, base + c));
        float n = v * rs;
        float o;
        if (w_bf16) o = bf2f(((const bf16_t*)w)[c]) * rbf(n);
        else o = ((const float*)w)[c] * n;
        store_out(y, base + c, o, w_bf16 != 0);
    }
}

// ---------------------------------------------------------------- vectorised backward
template <int NCH, bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_bwd_vec(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                       const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                       const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                       float* __restrict__ dw_partial, int64_t rows) {
    constexpr int C = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    float wv[NCH][8];
    float dwacc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        unpack8(ld16(w + j * 512 + lane * 8), wv[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[j][e] = 0.f;
    }
    for (int64_t row = wave_global; row < rows; row += nwaves) {
        const float rs = rstd[row];
        float xh[NCH][8], g[NCH][8];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 512 + lane * 8;
            float dyv[8];
            unpack8(ld16(x + row * C + c), xh[j]);
            unpack8(ld16(dy + row * C + c), dyv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[j][e] *= rs;
                g[j][e] = dyv[e] * wv[j][e];
                dot += g[j][e] * xh[j][e];
                dwacc[j][e] += dyv[e] * rbf(xh[j][e]);
            }
        }
        dot = wave_sum(dot) / (float)C;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 512 + lane * 8;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (g[j][e] - xh[j][e] * dot);
            if (HAS_RES) {
                float rv[8];
                unpack8(ld16(dres + row * C + c), rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rbf(o[e]) + rv[e];
            }
            st16(dx + row * C + c, pack8(o));
        }
    }
    // block-level reduction of the 4 waves' dw accumulators through LDS, one partial row per block
    __shared__ float red[C];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int wv = 0; wv < kWavesPerBlock; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = j * 512 + lane * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (wv == 0) red[c + e] = dwacc[j][e];
                    else red[c + e] += dwacc[j][e];
                }
            }
        }
        __syncthreads();
    }
    float* prow = dw_partial + (int64_t)blockIdx.x * C;
    for (int c = threadIdx.x * 4; c < C; c += 256 * 4)
        *reinterpret_cast<float4*>(prow + c) = *reinterpret_cast<const float4*>(red + c);
}

// ---------------------------------------------------------------- generic backward
template <typename XT>
__global__ __launch_bounds__(64) void rmsnorm_bwd_gen(const bf16_t* __restrict__ dy_bf, const float* __restrict__ dy_f,
                                                       const void* __restrict__ x, const void* __restrict__ w, int w_bf16,
                                                       const float* __restrict__ rstd, const void* __restrict__ dres,
                                                       void* __restrict__ dx, float* __restrict__ dw_partial,
                                                       int64_t rows, int64_t cols) {
    // dy has w's dtype (it is the grad of y); dx and dres have x's dtype.  One wave per block.
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x;
    const int nwaves = gridDim.x;
    float* prow = dw_partial + (int64_t)wave_global * cols;
    for (int64_t c = lane; c < cols; c += 64) prow[c] = 0.f;
    for (int64_t row = wave_global; row < rows; row += nwaves) {
        const float rs = rstd[row];
        const int64_t base = row * cols;
        float dot = 0.f;
        for (int64_t c = lane; c < cols; c += 64) {
            const float xh = Ld<XT>::get(x, base + c) * rs;
            const float dyv = w_bf16 ? bf2f(dy_bf[base + c]) : dy_f[base + c];
            const float wv = w_bf16 ? bf2f(((const bf16_t*)w)[c]) : ((const float*)w)[c];
            dot += dyv * wv * xh;
        }
        dot = wave_sum(dot) / (float)cols;
        for (int64_t c = lane; c < cols; c += 64) {
            const float xh = Ld<XT>::get(x, base + c) * rs;
            const float dyv = w_bf16 ? bf2f(dy_bf[base + c]) : dy_f[base + c];
            const float wv = w_bf16 ? bf2f(((const bf16_t*)w)[c]) : ((const float*)w)[c];
            float o = rs * (dyv * wv - xh * dot);
            prow[c] += dyv * (w_bf16 ? rbf(xh) : xh);
            constexpr bool xbf = sizeof(XT) == 2;
            if (dres) {
                if (xbf) o = rbf(o) + bf2f(((const bf16_t*)dres)[base + c]);
                else o = o + ((const float*)dres)[base + c];
            }
            store_out(dx, base + c, o, xbf);
        }
    }
}

// 32 columns per block; 8 row-groups of partials are summed in parallel and combined through LDS.
__global__ __launch_bounds__(256) void rmsnorm_dw_reduce_k(const float* __restrict__ part, int64_t nparts, void* __restrict__ dw,
                                                            int w_bf16, int64_t cols, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c = (int64_t)blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < cols)
        for (int64_t p = ty; p < nparts; p += 8) s += part[p * cols + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][tx];
        if (w_bf16) {
            bf16_t* o = (bf16_t*)dw;
            float r = rbf(s);
            if (accumulate) r = bf2f(o[c]) + r;
            o[c] = f2bf(r);
        } else {
            float* o = (float*)dw;
            o[c] = accumulate ? o[c] + s : s;
        }
    }
}

template <bool HAS_ADD>
int launch_fwd_vec(int nch, dim3 grid, hipStream_t st, const bf16_t* a, const bf16_t* b, bf16_t* r, const bf16_t* w, bf16_t* y,
                   float* rstd, int64_t rows, float eps) {
#define IE_CASE(N)                                                                                        \
    case N:                                                                                               \
        hipLaunchKernelGGL((rmsnorm_fwd_vec<N, HAS_ADD>), grid, dim3(256), 0, st, a, b, r, w, y, rstd, rows, eps); \
        return 1;
    switch (nch) {
        IE_CASE(1) IE_CASE(2) IE_CASE(3) IE_CASE(4) IE_CASE(5) IE_CASE(6) IE_CASE(7) IE_CASE(8)
        IE_CASE(10) IE_CASE(12) IE_CASE(16)
        default: return 0;
    }
#undef IE_CASE
}

template <bool HAS_RES>
int launch_bwd_vec(int nch, dim3 grid, hipStream_t st, const bf16_t* dy, const bf16_t* x, const bf16_t* w, const float* rstd,
                   const bf16_t* dres, bf16_t* dx, float* part, int64_t rows) {
#define IE_CASE(N)                                                                                             \
    case N:                                                                                                    \
        hipLaunchKernelGGL((rmsnorm_bwd_vec<N, HAS_RES>), grid, dim3(256), 0, st, dy, x, w, rstd, dres, dx, part, rows); \
        return 1;
    switch (nch) {
        IE_CASE(1) IE_CASE(2) IE_CASE(3) IE_CASE(4) IE_CASE(5) IE_CASE(6) IE_CASE(7) IE_CASE(8)
        default: return 0;
    }
#undef IE_CASE
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" int ie_rmsnorm_fwd(const void* x, int x_dtype, const void* w, int w_dtype, void* y, float* rstd,
                              int64_t rows, int64_t cols, float eps, void* stream) {
    IE_CHECK_ARG(x && w && y && rstd, "ie_rmsnorm_fwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && cols > 0, "ie_rmsnorm_fwd: bad shape");
    IE_CHECK_ARG((x_dtype == IE_BF16 || x_dtype == IE_F32) && (w_dtype == IE_BF16 || w_dtype == IE_F32),
                 "ie_rmsnorm_fwd: bad dtype");
    if (rows == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock));
    if (x_dtype == IE_BF16 && w_dtype == IE_BF16 && cols % 512 == 0 && aligned16(x) && aligned16(w) && aligned16(y)) {
        if (launch_fwd_vec<false>((int)(cols / 512), grid, st, (const bf16_t*)x, nullptr, nullptr, (const bf16_t*)w,
                                  (bf16_t*)y, rstd, rows, eps))
            return ie_launch_status("ie_rmsnorm_fwd launch");
    }
    if (x_dtype == IE_BF16)
        hipLaunchKernelGGL((rmsnorm_fwd_gen<bf16_t, false>), grid, dim3(256), 0, st, x, nullptr, nullptr, w,
                           (int)(w_dtype == IE_BF16), y, rstd, rows, cols, eps);
    else
        hipLaunchKernelGGL((rmsnorm_fwd_gen<float, false>), grid, dim3(256), 0, st, x, nullptr, nullptr, w,
                           (int)(w_dtype == IE_BF16), y, rstd, rows, cols, eps);
    return ie_launch_status("ie_rmsnorm_fwd launch");
}

extern "C" int ie_add_rmsnorm_fwd(const void* a, const void* b, void* r_out, const void* w, void* y, float* rstd,
                                  int64_t rows, int64_t cols, float eps, void* stream) {
    IE_CHECK_ARG(a && b && r_out && w && y && rstd, "ie_add_rmsnorm_fwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && cols > 0, "ie_add_rmsnorm_fwd: bad shape");
    if (rows == 0) return IE_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock));
    if (cols % 512 == 0 && aligned16(a) && aligned16(b) && aligned16(r_out) && aligned16(w) && aligned16(y)) {
        if (launch_fwd_vec<true>((int)(cols / 512), grid, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)r_out,
                                 (const bf16_t*)w, (bf16_t*)y, rstd, rows, eps))
            return ie_launch_status("ie_add_rmsnorm_fwd launch");
    }
    hipLaunchKernelGGL((rmsnorm_fwd_gen<bf16_t, true>), grid, dim3(256), 0, st, a, b, r_out, w, 1, y, rstd, rows, cols, eps);
    return ie_launch_status("ie_add_rmsnorm_fwd launch");
}

static inline int64_t bwd_blocks(int64_t rows) {
    int64_t b = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return b;
}

// workspace rows: the generic path runs one wave per block with 4x as many blocks
extern "C" int64_t ie_rmsnorm_bwd_partials(int64_t rows) { return bwd_blocks(rows) * kWavesPerBlock; }

extern "C" int ie_rmsnorm_bwd(const void* dy, const void* x, int x_dtype, const void* w, int w_dtype, const float* rstd,
                              const void* dres, void* dx, float* dw_partial, void* dw, int accumulate, int64_t rows, int64_t cols,
                              void* stream) {
    IE_CHECK_ARG(dy && x && w && rstd && dx && dw_partial && dw, "ie_rmsnorm_bwd: null pointer");
    IE_CHECK_ARG(rows >= 0 && cols > 0, "ie_rmsnorm_bwd: bad shape");
    IE_CHECK_ARG((x_dtype == IE_BF16 || x_dtype == IE_F32) && (w_dtype == IE_BF16 || w_dtype == IE_F32),
                 "ie_rmsnorm_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    int64_t nparts = 0;
    bool done = false;
    if (x_dtype == IE_BF16 && w_dtype == IE_BF16 && cols % 512 == 0 && cols <= 4096 && aligned16(dy) && aligned16(x) &&
        aligned16(w) && aligned16(dx) && aligned16(dw_partial) && (!dres || aligned16(dres))) {
        dim3 grid((unsigned)bwd_blocks(rows));
        int ok = dres ? launch_bwd_vec<true>((int)(cols / 512), grid, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w,
                                             rstd, (const bf16_t*)dres, (bf16_t*)dx, dw_partial, rows)
                      : launch_bwd_vec<false>((int)(cols / 512), grid, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w,
                                              rstd, nullptr, (bf16_t*)dx, dw_partial, rows);
        if (ok) { done = true; nparts = grid.x; }
    }
    if (!done) {
        dim3 grid((unsigned)(bwd_blocks(rows) * kWavesPerBlock));
        nparts = grid.x;
        const bf16_t* dyb = w_dtype == IE_BF16 ? (const bf16_t*)dy : nullptr;
        const float* dyf = w_dtype == IE_F32 ? (const float*)dy : nullptr;
        if (x_dtype == IE_BF16)
            hipLaunchKernelGGL((rmsnorm_bwd_gen<bf16_t>), grid, dim3(64), 0, st, dyb, dyf, x, w, (int)(w_dtype == IE_BF16), rstd,
                               dres, dx, dw_partial, rows, cols);
        else
            hipLaunchKernelGGL((rmsnorm_bwd_gen<float>), grid, dim3(64), 0, st, dyb, dyf, x, w, (int)(w_dtype == IE_BF16), rstd,
                               dres, dx, dw_partial, rows, cols);
    }
    hipLaunchKernelGGL(rmsnorm_dw_reduce_k, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, st, dw_partial, nparts, dw,
                       (int)(w_dtype == IE_BF16), cols, accumulate);
    return ie_launch_status("ie_rmsnorm_bwd launch");
}
