// Probe: whole-chip dense bf16 MFMA throughput (v_mfma_f32_32x32x16_bf16, one wave per SIMD) when the matrix pipe is kept busy only a
// fraction of the time: IDLE wait states (s_nop) behind every MFMA.  The chip is power-limited under dense MFMA, so the clock rises as the
// duty cycle falls; this gives throughput(duty) -- the yardstick for "how busy is the matrix pipe in my kernel really".
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_duty.hip -o tools/probes/mfma_duty && tools/probes/mfma_duty
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int IDLE>   // idle cycles behind every MFMA (multiples of 16 through s_nop 15)
__global__ __launch_bounds__(256, 1) void burn(float* out, long long* cyc, int iters, int data) {
    s16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (short)(0x3c00 + ((threadIdx.x * 37 + i * 11 + j * 5 + data) & 0x3ff));
            b[j][i] = (short)(0xbc00 + ((threadIdx.x * 53 + i * 7 + j * 3 + data) & 0x3ff));
        }
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a[t & 3]), "v"(b[t >> 2]));
#pragma unroll
            for (int z = 0; z < IDLE / 16; ++z) asm volatile("s_nop 15");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 16; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int IDLE> static void run(float* out, long long* cyc) {
    const int blocks = 256, iters = 60000 * 32 / (32 + IDLE);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((burn<IDLE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    const double cyc_per = m / blocks / (iters * 16.0);
    const double tf = 16.0 * 2 * 32 * 32 * 16 * iters * blocks * 4 / (ms * 1e-3) * 1e-12;
    printf("{\"probe\": \"mfma_duty\", \"idle_cycles_per_mfma\": %d, \"cycles_per_mfma\": %.1f, \"duty\": %.2f, \"tflops\": %.1f, \"clock_ghz\": %.2f}\n", IDLE, cyc_per,
           32.0 / cyc_per, tf, cyc_per * iters * 16.0 / (ms * 1e-3) * 1e-9);
    fflush(stdout);
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 256 * 8);
    run<0>(out, cyc);
    run<16>(out, cyc);
    run<32>(out, cyc);
    run<48>(out, cyc);
    run<0>(out, cyc);
    return 0;
}
