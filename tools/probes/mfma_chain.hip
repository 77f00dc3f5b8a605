// Probe: cycles per v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into the SAME registers (a dependent chain), by the
// number of independent chains interleaved round-robin (1 = back-to-back dependent, 2 = the S / dP pair of the attention backward,
// 4, 8).  One wave per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o tools/probes/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ unsigned char big[];
    f32x16 acc[NACC];
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "v"(b));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s + big[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC> static void run(float* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)probe<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL((probe<NACC>), dim3(blocks), dim3(256), 128 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    printf("{\"probe\": \"mfma_chain\", \"independent_chains\": %d, \"ticks_per_mfma\": %.2f}\n", NACC, m / blocks / (iters * 16.0));
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    run<1>(out, cyc);
    run<2>(out, cyc);
    run<4>(out, cyc);
    run<8>(out, cyc);
    return 0;
}
