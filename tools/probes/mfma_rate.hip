// Probe: sustained dense bf16 MFMA rate of the whole chip for the two instruction shapes, wall clock (hipEvents), long enough
// (~0.3 s per point) to sit at the power-limited clock.  Answers: is v_mfma_f32_16x16x32_bf16 (what hipBLASLt's kernels use)
// cheaper in power -- i.e. faster in practice -- than v_mfma_f32_32x32x16_bf16 (what this repo's GEMM uses)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate && tools/probes/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// SHAPE 0: 32x32x16 (16 accumulator tiles of 16 regs = one 128x128 wave tile); SHAPE 1: 16x16x32 (64 tiles of 4 regs)
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void burn(float* out, int iters, int data) {
    s16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            // bf16 bit patterns of small random-looking values (all-zero operands would understate the power)
            a[j][i] = (short)(0x3c00 + ((threadIdx.x * 37 + i * 11 + j * 5 + data) & 0x3ff));
            b[j][i] = (short)(0xbc00 + ((threadIdx.x * 53 + i * 7 + j * 3 + data) & 0x3ff));
        }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[16];
        for (int t = 0; t < 16; ++t)
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 3], b[t >> 2], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 16; ++t)
            for (int i = 0; i < 16; ++i) s += acc[t][i];
    } else {
        f32x4 acc[64];
        for (int t = 0; t < 64; ++t)
            for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 64; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 64; ++t)
            for (int i = 0; i < 4; ++i) s += acc[t][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE> static void run(float* out, const char* name) {
    const int blocks = 256, iters = 60000;
    const double flop_per_iter_wave = SHAPE == 0 ? 16.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32;   // both 524288
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((burn<SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_per_iter_wave * iters * blocks * 4 / (ms * 1e-3) * 1e-12;
        printf("{\"probe\": \"mfma_rate\", \"shape\": \"%s\", \"rep\": %d, \"ms\": %.1f, \"tflops\": %.1f}\n", name, rep, ms, tf);
        fflush(stdout);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    run<0>(out, "32x32x16");
    run<1>(out, "16x16x32");
    run<0>(out, "32x32x16");
    run<1>(out, "16x16x32");
    return 0;
}
