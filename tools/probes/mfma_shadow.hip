// Probe: how many independent vector-ALU instructions does ONE wave hide in the shadow of a v_mfma_f32_32x32x16_bf16
// (32 cycles of matrix pipe per SIMD), depending on where the accumulator lives (arch VGPRs vs accumulation VGPRs) and on the
// kind of filler?  One wave per SIMD (256-thread blocks, 1 block per CU through a 128 KiB LDS allocation), 4 independent
// accumulators used round-robin, N fillers between two MFMAs.  Prints shader cycles per MFMA (s_memtime) for every combination.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shadow.hip -o tools/probes/mfma_shadow && tools/probes/mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int ACC_IN_AGPR> __device__ __forceinline__ void mfma(f32x16& d, const s16x8& a, const s16x8& b) {
    if constexpr (ACC_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// KIND 0: v_fma_f32 on private registers; 1: v_exp_f32; 2: v_cvt_pk_bf16_f32; 3: v_max3_f32; 4: the softmax mix (2 fma, 2 exp, 2 add, 1 cvt = 7)
template <int KIND, int N> __device__ __forceinline__ void fillers(float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float& v = x[i & 7];
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
        else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
        else if constexpr (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(x[(i + 1) & 7]));
        else if constexpr (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
    }
}
template <> __device__ __forceinline__ void fillers<4, 7>(float (&x)[8]) {
    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %2, %2, %1\n\t"
                 "v_cvt_pk_bf16_f32 %3, %0, %1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
                 : "v"(x[4]), "v"(x[5]));
}

template <int ACC_IN_AGPR, int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ unsigned char big[];
    f32x16 acc[4];
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            mfma<ACC_IN_AGPR>(acc[u & 3], a, b);
            fillers<KIND, N>(x);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + big[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int A, int K, int N> static void run(float* out, long long* cyc, const char* kind) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)probe<A, K, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL((probe<A, K, N>), dim3(blocks), dim3(256), 128 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    // s_memtime ticks at 100 MHz on this chip? report raw ticks per MFMA and let the reader compare rows (N = 0 row = 32 cycles)
    printf("{\"acc\": \"%s\", \"filler\": \"%s\", \"n\": %d, \"ticks_per_mfma\": %.3f}\n", A ? "agpr" : "vgpr", kind, N, m / (iters * 16.0));
    fflush(stdout);
}

template <int A> static void sweep(float* out, long long* cyc) {
    run<A, 0, 0>(out, cyc, "none");
    run<A, 0, 2>(out, cyc, "fma"); run<A, 0, 4>(out, cyc, "fma"); run<A, 0, 5>(out, cyc, "fma"); run<A, 0, 6>(out, cyc, "fma"); run<A, 0, 7>(out, cyc, "fma");
    run<A, 0, 8>(out, cyc, "fma"); run<A, 0, 10>(out, cyc, "fma"); run<A, 0, 12>(out, cyc, "fma");
    run<A, 1, 2>(out, cyc, "exp"); run<A, 1, 4>(out, cyc, "exp"); run<A, 1, 6>(out, cyc, "exp");
    run<A, 2, 4>(out, cyc, "cvt_pk"); run<A, 3, 4>(out, cyc, "max3");
    run<A, 4, 7>(out, cyc, "softmax_mix7");
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    sweep<0>(out, cyc);
    sweep<1>(out, cyc);
    return 0;
}
