// Standalone hardware probes (gfx950): semantics of ds_read_b64_tr_b16 and of the LDS-DMA load
// (__builtin_amdgcn_global_load_lds).  Build: hipcc --offload-arch=gfx950 -O2 probe_lds.hip -o probe_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void tr_probe(const int* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)lds + (unsigned)addr[threadIdx.x];
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// each lane copies 16 bytes global -> LDS with the DMA path; src index per lane comes from `perm`
__global__ void glds_probe(const unsigned short* __restrict__ src, const int* __restrict__ perm, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xffff;
    __syncthreads();
    const unsigned short* g = src + perm[threadIdx.x] * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds + 512), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    int *d_addr; unsigned short *d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 4096 * 2);
    std::vector<int> addr(64); std::vector<unsigned short> out(4096);
    const char* names[3] = {"P1 addr=lane*8", "P2 addr=(lane&15)*32+(lane>>4)*8", "P3 rows: addr=((lane&15)>>2)*576+((lane&15)&3)*8+(lane>>4)*32"};
    for (int p = 0; p < 3; ++p) {
        for (int l = 0; l < 64; ++l) {
            if (p == 0) addr[l] = l * 8;
            else if (p == 1) addr[l] = (l & 15) * 32 + (l >> 4) * 8;
            else addr[l] = ((l & 15) >> 2) * 576 + ((l & 15) & 3) * 8 + (l >> 4) * 32;
        }
        hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(out.data(), d_out, 64 * 4 * 2, hipMemcpyDeviceToHost);
        printf("TR %s\n", names[p]);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %5d -> elems(idx) %5u %5u %5u %5u\n", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
        }
    }
    unsigned short* d_src; int* d_perm;
    hipMalloc(&d_src, 4096 * 2); hipMalloc(&d_perm, 256);
    std::vector<unsigned short> src(4096); for (int i = 0; i < 4096; ++i) src[i] = (unsigned short)i;
    std::vector<int> perm(64);
    for (int t = 0; t < 2; ++t) {
        for (int l = 0; l < 64; ++l) perm[l] = t == 0 ? l : (l ^ 5);
        hipMemcpy(d_src, src.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(d_perm, perm.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(glds_probe, dim3(1), dim3(64), 0, 0, d_src, d_perm, d_out);
        hipMemcpy(out.data(), d_out, 4096, hipMemcpyDeviceToHost);
        printf("GLDS perm %s: lds[512 + 8*l] for l=0..63 (expect src chunk perm[l]*8):\n ", t == 0 ? "identity" : "l^5");
        for (int l = 0; l < 64; ++l) printf(" %u", out[512 + 8 * l]);
        printf("\n  untouched before/after: %u %u\n", out[511], out[512 + 512]);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
