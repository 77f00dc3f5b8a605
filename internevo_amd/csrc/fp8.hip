// fp8 (OCP e4m3, the format gfx950's conversion and MFMA instructions use) operands for the expert products ("fp8 linear has no
// reference => define own tolerance").  Per-tensor dynamic scaling, the scales device scalars from the quantiser to the product (no host read):
//   ie_fp8_amax       max |x| of a bf16 tensor into a device float (atomic max of the non-negative float bits; the caller zeroes it); `count` equal tensors
//                     laid out back to back (the experts' blocks) get one scale each
//   ie_fp8_quantize   q = e4m3(x * 448 / amax), round-to-nearest-even with saturation (v_cvt_pk_fp8_f32); dequant[0] = amax / 448 (1 for an all-zero tensor)
//   ie_gemm_fp8       C[M, N] bf16 = (A[M, K] e4m3)(B[N, K] e4m3)^T * dequant_a * dequant_b on v_mfma_f32_32x32x64_f8f6f4 (gemm_bf16_dma.hip, schedule -6:
//                     the bf16 forward kernel's LDS-DMA pipeline byte for byte, twice the k per tile)
#include "ie_common.h"

namespace {

constexpr float E4M3_MAX = 448.f;

__global__ __launch_bounds__(256) void fp8_amax_k(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ amax) {
    x += blockIdx.y * n;   // (tensor blockIdx.y of a batch of equal tensors, each with its own scale)
    amax += blockIdx.y;
    float m = 0.f;
    const int64_t n8 = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(ld16(x + 8 * i), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(f[e]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 8 * n8)) m = fmaxf(m, fabsf(bf2f(x[8 * n8 + threadIdx.x])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));   // (non-negative floats order like their bits)
    }
}

__device__ __forceinline__ unsigned cvt4(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}

__global__ __launch_bounds__(256) void fp8_quantize_k(const bf16_t* __restrict__ x, int64_t n, const float* __restrict__ amax, unsigned char* __restrict__ q,
                                                      float* __restrict__ dequant) {
    x += blockIdx.y * n;
    q += blockIdx.y * n;
    amax += blockIdx.y;
    dequant += blockIdx.y;
    const float a = *amax;
    const float scale = a > 0.f ? E4M3_MAX / a : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *dequant = a > 0.f ? a / E4M3_MAX : 1.f;
    const int64_t n8 = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float f[8];
        unpack8(ld16(x + 8 * i), f);
        // (|x * scale| <= 448 up to one rounding of the product; the conversion saturates)
        uint2 o;
        o.x = cvt4(fminf(fmaxf(f[0] * scale, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(f[1] * scale, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(f[2] * scale, -E4M3_MAX), E4M3_MAX),
                   fminf(fmaxf(f[3] * scale, -E4M3_MAX), E4M3_MAX));
        o.y = cvt4(fminf(fmaxf(f[4] * scale, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(f[5] * scale, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(f[6] * scale, -E4M3_MAX), E4M3_MAX),
                   fminf(fmaxf(f[7] * scale, -E4M3_MAX), E4M3_MAX));
        *reinterpret_cast<uint2*>(q + 8 * i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 8 * n8)) {
        const float v = fminf(fmaxf(bf2f(x[8 * n8 + threadIdx.x]) * scale, -E4M3_MAX), E4M3_MAX);
        q[8 * n8 + threadIdx.x] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
    }
}

int grid_for(int64_t n) {
    const int64_t blocks = (n / 8 + 255) / 256;
    return (int)(blocks < 1 ? 1 : blocks > 256 * 16 ? 256 * 16 : blocks);
}

}  // namespace

extern "C" int ie_gemm_fp8_dma_launch(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb, void* C, int64_t ldc, int64_t sc, int64_t count,
                                      int64_t M, int64_t N, int64_t K, const float* scale_a, const float* scale_b, int accumulate, void* stream);

extern "C" int ie_fp8_amax(const void* x, int64_t n, int64_t count, float* amax, void* stream) {
    IE_CHECK_ARG(n >= 0 && count >= 1 && count <= 65535 && (n == 0 || (x && amax)), "ie_fp8_amax: null pointer or bad count");
    IE_CHECK_ARG(((uintptr_t)x & 15) == 0 && (count == 1 || n % 8 == 0), "ie_fp8_amax: x must be 16-byte aligned (and n a multiple of 8 in a batch)");
    if (n == 0) return IE_OK;
    hipLaunchKernelGGL(fp8_amax_k, dim3(grid_for(n), (unsigned)count), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n, amax);
    return ie_launch_status("ie_fp8_amax launch");
}

extern "C" int ie_fp8_quantize(const void* x, int64_t n, int64_t count, const float* amax, void* q, float* dequant, void* stream) {
    IE_CHECK_ARG(n >= 0 && count >= 1 && count <= 65535 && amax && dequant && (n == 0 || (x && q)), "ie_fp8_quantize: null pointer or bad count");
    IE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0 && (count == 1 || n % 8 == 0),
                 "ie_fp8_quantize: x must be 16-byte, q 8-byte aligned (and n a multiple of 8 in a batch)");
    hipLaunchKernelGGL(fp8_quantize_k, dim3(grid_for(n), (unsigned)count), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n, amax, (unsigned char*)q, dequant);
    return ie_launch_status("ie_fp8_quantize launch");
}

static int fp8_gemm_checked(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb, void* C, int64_t ldc, int64_t sc, int64_t count, int64_t M,
                            int64_t N, int64_t K, const float* dequant_a, const float* dequant_b, int accumulate, void* stream) {
    IE_CHECK_ARG(A && B && C && dequant_a && dequant_b, "ie_gemm_fp8: null pointer");
    IE_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N && count >= 1, "ie_gemm_fp8: bad shape, leading dimension or count");
    IE_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0 && sa % 16 == 0 && sb % 16 == 0 && sc % 8 == 0,
                 "ie_gemm_fp8: pointers must be 16-byte aligned, lda / ldb / operand strides multiples of 16 and ldc / the output stride of 8 elements");
    IE_CHECK_SUPPORTED(K % 128 == 0 && N % 8 == 0 && M >= 8 && N >= 8, "ie_gemm_fp8: K must be a multiple of 128 (one LDS row of e4m3), N of 8");
    IE_CHECK_SUPPORTED(M * lda < (1ll << 32) && N * ldb < (1ll << 32), "ie_gemm_fp8: operands must be smaller than 4 GiB (32-bit buffer offsets)");
    return ie_gemm_fp8_dma_launch(A, lda, sa, B, ldb, sb, C, ldc, sc, count, M, N, K, dequant_a, dequant_b, accumulate, stream);
}

extern "C" int ie_gemm_fp8(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const float* dequant_a,
                           const float* dequant_b, int accumulate, void* stream) {
    return fp8_gemm_checked(A, lda, 0, B, ldb, 0, C, ldc, 0, 1, M, N, K, dequant_a, dequant_b, accumulate, stream);
}

extern "C" int ie_gemm_fp8_batched(const void* A, int64_t lda, int64_t stride_a, const void* B, int64_t ldb, int64_t stride_b, void* C, int64_t ldc, int64_t stride_c,
                                   int64_t count, int64_t M, int64_t N, int64_t K, const float* dequant_a, const float* dequant_b, int accumulate, void* stream) {
    return fp8_gemm_checked(A, lda, stride_a, B, ldb, stride_b, C, ldc, stride_c, count, M, N, K, dequant_a, dequant_b, accumulate, stream);
}
