// Probe: what does a ds_read_b128 (1 KiB per wave) / a ds_read_b64_tr_b16 pair cost a wave that otherwise issues back-to-back
// v_mfma_f32_32x32x16_bf16 (one wave per SIMD, four waves per CU reading the same LDS)?  N reads behind every MFMA; cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_issue.hip -o tools/probes/lds_issue && tools/probes/lds_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

template <int KIND, int N>   // KIND 0: ds_read_b128, 1: ds_read_b64_tr_b16
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char big[];
    f32x16 acc[4];
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int i = threadIdx.x; i < 32768; i += 256) ((int*)big)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the GEMM's conflict-free pattern: row (lane & 31) of a 128-byte-row image, 16-byte chunk swizzled by the row
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)big + wave * 16384 + (lane & 31) * 128 +
                          ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    i32x4 sink = {0, 0, 0, 0};
    i32x2 sink2 = {0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int z = 0; z < N; ++z) {
                if (KIND == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink) : "v"(addr), "n"((z * 4096) & 0xffff));
                else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(sink2) : "v"(addr), "n"((z * 2048) & 0xffff));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)sink[0] + (float)sink2[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int N> static void run(float* out, long long* cyc, const char* what) {
    const int iters = 2000, blocks = 256;
    (void)hipFuncSetAttribute((const void*)probe<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<KIND, N>), dim3(blocks), dim3(256), 128 * 1024, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    printf("{\"probe\": \"lds_issue\", \"read\": \"%s\", \"reads_per_mfma\": %d, \"ticks_per_mfma\": %.2f}\n", what, N, m / blocks / (iters * 16.0));
    fflush(stdout);
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 256 * 8);
    run<0, 0>(out, cyc, "none");
    run<0, 1>(out, cyc, "ds_read_b128");
    run<0, 2>(out, cyc, "ds_read_b128");
    run<0, 3>(out, cyc, "ds_read_b128");
    run<1, 2>(out, cyc, "ds_read_b64_tr_b16");
    run<1, 4>(out, cyc, "ds_read_b64_tr_b16");
    return 0;
}
