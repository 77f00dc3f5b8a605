// ABI bookkeeping (version, thread-local error string) and the MFMA fragment-layout probe.
#include "ie_common.h"

#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void ie_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* ie_last_error(void) { return g_err; }

extern "C" int ie_abi_version(void) { return IE_ABI_VERSION; }

// One wave, one v_mfma_f32_32x32x16_bf16.  Operand maps assumed by gemm_bf16.hip / flash_attn_*.hip:
//   A operand: lane l holds A[i = l & 31][k = 8*(l >> 5) + 0..7]
//   B operand: lane l holds B[k = 8*(l >> 5) + 0..7][j = l & 31]
//   C/D:       lane l, reg r holds D[i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)][j = l & 31]
__global__ void mfma_probe_k(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, float* __restrict__ c) {
    const int l = threadIdx.x;
    const int i = l & 31, kh = l >> 5;
    s16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fa[e] = (short)a[i * 16 + kh * 8 + e];
        fb[e] = (short)b[(kh * 8 + e) * 32 + i];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        c[row * 32 + i] = acc[r];
    }
}

extern "C" int ie_mfma_probe(const void* a, const void* b, float* c, void* stream) {
    IE_CHECK_ARG(a && b && c, "ie_mfma_probe: null pointer");
    hipLaunchKernelGGL(mfma_probe_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, c);
    return ie_launch_status("ie_mfma_probe launch");
}

// Diagnostic (bench.py --hold-cus): `blocks` workgroups that each take a whole CU out of the products' reach for `usec` microseconds and
// do nothing -- a stand-in, on a one-GPU box, for the CUs an RCCL collective holds while it runs beside the step (a GEMM workgroup of this library needs a CU
// to itself: one 512-register wave per SIMD and 144 KB of LDS, so any resident wave keeps it away; here 256 threads = one wave per SIMD and 96 KB of LDS = one
// such block per CU).  The blocks sleep between looks at the 100 MHz wall clock (s_sleep: no issue slots, no memory traffic, next to no power).
__global__ __launch_bounds__(256) void hold_cu_k(long long ticks) {
    __shared__ volatile unsigned char pad[96 * 1024];
    if (ticks < 0) pad[threadIdx.x] = 0;   // (keeps the allocation)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int ie_hold_cus(int blocks, int usec, void* stream) {
    IE_CHECK_ARG(blocks >= 1 && blocks <= 256 && usec >= 1 && usec <= 1000000, "ie_hold_cus: 1..256 blocks, 1..1e6 microseconds");
    hipLaunchKernelGGL(hold_cu_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)usec * 100ll);
    return ie_launch_status("ie_hold_cus launch");
}
