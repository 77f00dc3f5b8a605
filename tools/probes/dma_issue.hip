// Probe: what does ONE LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB) cost a wave that otherwise issues back-to-back
// v_mfma_f32_32x32x16_bf16 (one wave per SIMD, 4 waves per CU as in the 256x256 GEMM)?  The GEMM issues 16 pieces per wave per
// 64 MFMAs (one per 4); round 1 measured ~15 % lost matrix time for the pieces alone.  This probe times the loop
//     { MFMA x P ; one piece }  repeated, every block streaming its own rows of a large matrix (8 rows x 128 B per piece)
// for several ways of forming the instruction, and prints cycles per MFMA (32 = free).
//   MODE 0  no DMA
//   MODE 1  s_add m0 right before the load, soffset in an SGPR (what hipcc emits for the GEMM's BufSrc::issue_keep)
//   MODE 2  load first, m0 bumped after it (under the next MFMA), per-piece offset in the VGPR, soffset 0 (hipBLASLt's loop)
//   MODE 3  m0 never changes (same LDS destination), soffset 0
//   MODE 4  same as 2 but the data goes to VGPRs (no LDS-DMA), to separate "VMEM issue" from "LDS-DMA issue"
//   MODE 5  as 2, every load hits the same 1 KiB (cache-resident source)
//   MODE 6  as 1 with 16 rows x 64 B per piece (the k32 ring's k-contiguous pieces); MODE 7  32 rows x 32 B
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_issue.hip -o tools/probes/dma_issue && tools/probes/dma_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void mfma(f32x16& d, const s16x8& a, const s16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}

template <int MODE, int P>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* src, long long ld_bytes, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char big[];
    f32x16 acc[4];
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this block's rows: 256 rows of ld_bytes each; a piece = 8 rows x 128 B; k advances by 128 B per 16 pieces
    const unsigned char* base = src + (long long)blockIdx.x * 256 * ld_bytes;
    const int voff0 = MODE == 6 ? (lane >> 2) * (int)ld_bytes + (lane & 3) * 16 : MODE == 7 ? (lane >> 1) * (int)ld_bytes + (lane & 1) * 16
                                : (lane >> 3) * (int)ld_bytes + (lane & 7) * 16;
    uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)big + wave * 16384;
    i32x4 sink = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 64 MFMAs; a piece after every P-th
        const unsigned char* kb = MODE == 5 ? base : base + (long long)(it & 31) * 128;
        i32x4 rsrc;
        rsrc[0] = (int)(uint32_t)(uintptr_t)kb;
        rsrc[1] = (int)(((uintptr_t)kb >> 32) & 0xffff);
        rsrc[2] = 0x7fffffff;
        rsrc[3] = 0x00020000;
        rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]);
        rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
        if (MODE == 2 || MODE == 3 || MODE == 5) asm volatile("s_mov_b32 m0, %0" ::"s"(lds_base));
        int piece = 0;
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            mfma(acc[u & 3], a, b);
            // hipBLASLt's order: the m0 bump sits one MFMA AFTER the load that used the old value (never right behind it)
            if ((MODE == 2 || MODE == 5) && u >= 1 && ((u - 1) % P) == P - 1) asm volatile("s_add_u32 m0, m0, 0x400" ::: "memory", "scc");
            if (MODE != 0 && (u % P) == P - 1) {
                const int q = piece++;
                // rows of this piece: 8 per piece (16 / 32 in modes 6 / 7, which then take 2 / 4 pieces to cover the same 128-B columns)
                const int row_off = MODE == 5 ? 0 : MODE == 6 ? (16 * (wave + 4 * ((q >> 1) & 3))) * (int)ld_bytes + (q & 1) * 64
                                  : MODE == 7 ? (32 * (wave + 4 * ((q >> 2) & 1))) * (int)ld_bytes + (q & 3) * 32
                                              : (8 * (wave + 4 * (q & 7))) * (int)ld_bytes;
                if (MODE == 1 || MODE == 6 || MODE == 7) {
                    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base + (q & 15) * 1024), "v"(voff0), "s"(rsrc), "s"(row_off) : "memory");
                } else if (MODE == 2 || MODE == 5) {
                    asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff0 + row_off), "s"(rsrc) : "memory");
                } else if (MODE == 3) {
                    asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff0 + row_off), "s"(rsrc) : "memory");
                } else if (MODE == 4) {
                    i32x4 t;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(t) : "v"(voff0 + row_off), "s"(rsrc) : "memory");
                    sink = t;   // (the value is only kept alive; the wait below covers it)
                }
            }
        }
        // (no landing wait inside the loop: it would add one memory latency per 64 MFMAs and hide what is measured here;
        //  at most 16 pieces per ~2000 cycles are in flight, far below the 63 the counter can hold)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s + big[threadIdx.x] + (float)sink[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int P> static void run(const unsigned char* src, long long ld, float* out, long long* cyc, const char* what) {
    const int iters = 400, blocks = 256;
    hipFuncSetAttribute((const void*)probe<MODE, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, P>), dim3(blocks), dim3(256), 128 * 1024, 0, src, ld, out, cyc, iters);
        hipDeviceSynchronize();
    }
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    printf("{\"probe\": \"dma_issue\", \"mode\": %d, \"what\": \"%s\", \"mfma_per_piece\": %d, \"ticks_per_mfma\": %.2f}\n", MODE, what, P, m / (iters * 64.0));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long ld = 8192;                       // bytes per row (K = 4096 bf16)
    const size_t bytes = (size_t)256 * 256 * ld;     // 256 blocks x 256 rows
    unsigned char* src;
    float* out;
    long long* cyc;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int sel = argc > 1 ? atoi(argv[1]) : 0;
    if (sel == 0) run<0, 4>(src, ld, out, cyc, "no dma");
    if (sel == 1) { run<1, 4>(src, ld, out, cyc, "m0 before, soffset sgpr"); run<1, 2>(src, ld, out, cyc, "m0 before, soffset sgpr"); run<1, 8>(src, ld, out, cyc, "m0 before, soffset sgpr"); }
    if (sel == 2) { run<2, 4>(src, ld, out, cyc, "m0 after, voffset only"); run<2, 2>(src, ld, out, cyc, "m0 after, voffset only"); run<2, 8>(src, ld, out, cyc, "m0 after, voffset only"); }
    if (sel == 3) run<3, 4>(src, ld, out, cyc, "m0 fixed");
    if (sel == 4) { run<4, 4>(src, ld, out, cyc, "to vgpr"); run<4, 2>(src, ld, out, cyc, "to vgpr"); }
    if (sel == 6) { run<6, 4>(src, ld, out, cyc, "16 rows x 64 B"); run<6, 2>(src, ld, out, cyc, "16 rows x 64 B"); }
    if (sel == 7) { run<7, 4>(src, ld, out, cyc, "32 rows x 32 B"); run<7, 2>(src, ld, out, cyc, "32 rows x 32 B"); }
    if (sel == 5) { run<5, 4>(src, ld, out, cyc, "m0 after, cache-resident source"); run<5, 2>(src, ld, out, cyc, "m0 after, cache-resident source"); }
    return 0;
}
