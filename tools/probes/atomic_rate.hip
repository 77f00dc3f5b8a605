// Probe (round 5): what does it cost to push the dQ contributions of a FUSED attention backward through global memory?
// A fused kernel that owns a block of NK keys adds one [64 query rows][128] fp32 tile to dQ per (key block, query tile, q head):
// 4.3 GB of read-modify-write traffic per 4 x 4096-token call at NK = 128, 2.1 GB at NK = 256.  This probe replays exactly that
// address stream with nothing else in the kernel (InternLM2-7B attention: 4 sequences x 4096, 32 q heads / 8 kv heads, d = 128) as
//   mode 0: global_atomic_add_f32 (what hipcc emits for atomicAdd at agent AND workgroup scope on gfx950: no sc bits)
//   mode 1: plain load + add + store of the same addresses (not atomic: a bandwidth yardstick only)
//   mode 2: plain stores (the write side of a partial-buffer scheme)
// and with two block -> work maps: `xcd` = 1 puts all key blocks of one (sequence, kv head) on one XCD under round-robin dispatch
// (block b -> XCD b % 8), 0 spreads them over all XCDs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_rate.hip -o tools/probes/atomic_rate && tools/probes/atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int SEQS = 4, LEN = 4096, HQ = 32, HKV = 8, D = 128, GRP = HQ / HKV;

template <int MODE>
__global__ __launch_bounds__(256) void replay(float* dq, int nk, int xcd_local, int reverse) {
    const int nkb = LEN / nk;                      // key blocks per sequence
    const int b = blockIdx.x;
    int seq, hk, j;
    if (xcd_local) {                               // b % 8 = kv head (-> XCD); heaviest key blocks first
        hk = b % HKV;
        seq = (b / HKV) % SEQS;
        j = b / (HKV * SEQS);
    } else {                                       // consecutive blocks = consecutive key blocks of one (seq, kv head): spread over the XCDs
        j = b % nkb;
        hk = (b / nkb) % HKV;
        seq = b / (nkb * HKV);
    }
    const int qt0 = j * nk / 64, nqt = LEN / 64;
    const int col = threadIdx.x & 127, r0 = threadIdx.x >> 7;
    const float v = 1.0f;
    for (int g = 0; g < GRP; ++g) {
        const int h = hk * GRP + g;
        for (int s = 0; s < nqt - qt0; ++s) {
            const int qt = reverse ? nqt - 1 - s : qt0 + s;
            float* p = dq + ((size_t)(seq * LEN + qt * 64 + r0) * HQ + h) * D + col;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                float* a = p + (size_t)2 * k * HQ * D;
                if (MODE == 0) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 1) *a = *a + v;
                else *a = v;
            }
        }
    }
}

int main() {
    const size_t n = (size_t)SEQS * LEN * HQ * D;
    float* dq;
    OK(hipMalloc(&dq, n * 4));
    OK(hipMemset(dq, 0, n * 4));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    for (int nk : {128, 256})
        for (int mode = 0; mode < 3; ++mode)
            for (int xcd = 0; xcd < 2; ++xcd)
                for (int rev = 0; rev < 2; ++rev) {
                    const int blocks = SEQS * HKV * (LEN / nk);
                    const int nqt = LEN / 64;
                    double tiles = 0;
                    for (int j = 0; j < LEN / nk; ++j) tiles += nqt - j * nk / 64;
                    const double bytes = tiles * SEQS * HQ * 64.0 * D * 4;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; ++rep) {
                        OK(hipEventRecord(e0));
                        if (mode == 0) hipLaunchKernelGGL((replay<0>), dim3(blocks), dim3(256), 0, 0, dq, nk, xcd, rev);
                        else if (mode == 1) hipLaunchKernelGGL((replay<1>), dim3(blocks), dim3(256), 0, 0, dq, nk, xcd, rev);
                        else hipLaunchKernelGGL((replay<2>), dim3(blocks), dim3(256), 0, 0, dq, nk, xcd, rev);
                        OK(hipEventRecord(e1));
                        OK(hipEventSynchronize(e1));
                        float ms;
                        OK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    printf("{\"probe\": \"atomic_rate\", \"nk\": %d, \"mode\": \"%s\", \"xcd_local\": %d, \"reverse\": %d, \"blocks\": %d, \"GB\": %.2f, \"us\": %.0f, \"TBps\": %.2f}\n", nk,
                           mode == 0 ? "atomic_add_f32" : (mode == 1 ? "load_add_store" : "store"), xcd, rev, blocks, bytes * 1e-9, best * 1e3, bytes / (best * 1e-3) * 1e-12);
                    fflush(stdout);
                }
    // sanity of the atomic path: every element of every row got (number of key blocks that see it) x passes
    return 0;
}
