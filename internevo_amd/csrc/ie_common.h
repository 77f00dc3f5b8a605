// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libinternevo_hip.so.
// Not a portability layer: everything here assumes a 64-lane wavefront and gfx950 builtins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/internevo_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;  // 8 bf16 = one MFMA A/B fragment

extern "C" void ie_set_error(const char* msg);

#define IE_CHECK_ARG(cond, msg)  \
    do {                         \
        if (!(cond)) {           \
            ie_set_error(msg);   \
            return IE_ERR_INVALID; \
        }                        \
    } while (0)

#define IE_CHECK_SUPPORTED(cond, msg) \
    do {                              \
        if (!(cond)) {                \
            ie_set_error(msg);        \
            return IE_ERR_UNSUPPORTED; \
        }                             \
    } while (0)

static inline int ie_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[240];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        ie_set_error(buf);
        return IE_ERR_LAUNCH;
    }
    return IE_OK;
}

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved; identical to torch's cast) -------------
__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }

// gfx950 has a native RNE convert (v_cvt_pk_bf16_f32, two floats per instruction); hipcc selects it for the
// __bf16 fptrunc, so the conversions are written as plain casts rather than integer bit tricks.
typedef __attribute__((ext_vector_type(2))) float ie_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ie_bf16x2;

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const ie_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ie_bf16x2));
}

__device__ __forceinline__ float bflo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// round an f32 to the nearest bf16 and return it as f32 (mimics a bf16 intermediate in the reference)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x);
    f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z);
    f[6] = bflo(v.w); f[7] = bfhi(v.w);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]);
    v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]);
    v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- wave64 / block reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a block of NW waves; result valid in every thread. `scratch` holds >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += scratch[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float t = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, scratch[i]);
    return t;
}

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ uint2 ld8(const void* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ void st8(void* p, const uint2& v) { *reinterpret_cast<uint2*>(p) = v; }

// strided batch of equal GEMMs in one launch (element strides between consecutive products; count = 1: a single product)
struct IeGemmBatch {
    int count;
    int64_t sa, sb, sc;
    // fused SwiGLU epilogues of the FFN products (gemm_bf16_dma.hip, template parameter EPI; unused otherwise):
    //   EPI 1 (w1 | w3 forward): B = [2F, K] (w1 rows, then w3 rows); a 256-column tile = 128 gate columns + the SAME 128 up columns, the epilogue
    //     writes both to C = h13 [M, 2F] and silu(gate) * up to act [M, F];
    //   EPI 2 (w2 dgrad): the tile of d(act) never reaches memory: with gate | up read from h13 = aux the epilogue writes d(gate) | d(up) to C = dh13.
    int64_t f;            // F
    const void* aux;      // EPI 2: h13 [M, 2F]
    int64_t ld_aux;
    void* act;            // EPI 1: act [M, F]
    int64_t ld_act;
};

// EPI 3 of the persistent GEMM frame (gemm_p5_k<false, 3>, round 6): the wqkv product's epilogue does what qkv_rotary_fwd_k does -- GQA split, even / odd
// de-interleave, cos / sin gather by position, rotation -- and writes q / kv directly; the [T, hkv (qpk + 2) d] product never reaches memory.
struct IeRotaryEpi {
    const void* cos;      // bf16 [max position, d / 2]
    const void* sin;
    const int64_t* pos;   // [T] position of every token row
    void* q;              // bf16 [T, hkv * qpk, d]
    void* kv;             // bf16 [T, 2, hkv, d]
    int hkv, qpk, interleaved;
    float q_scale;
};

// SwiGLU element arithmetic shared by the elementwise kernels and the fused GEMM epilogues (bit-identical by construction).
// Reference: Silu(w1_o, w3_o) = F.silu(w1_o) * w3_o on bf16 tensors (model/utils.py:684-688): silu evaluated in fp32, rounded to bf16,
// product rounded to bf16; the backward sees the bf16 silu(a) autograd saved and the bf16 gradient of the product.
__device__ __forceinline__ float sigmoidf_(float a) { return 1.f / (1.f + __expf(-a)); }
// Backward of the rotary embedding on one (first half, second half) pair: the rotation by -theta (ApplyRotaryEmb.backward, modules/embedding.py:150-166:
// conjugate=True).  Every rounding written out: qkv_rotary_bwd_k and the attention backward's fused stores (flash_common.h) must give the same bits.
__device__ __forceinline__ void rot_conj1(float d1, float d2, float co, float si, float& x1, float& x2) {
#pragma clang fp contract(off)
    x1 = __builtin_fmaf(d1, co, d2 * si);
    x2 = __builtin_fmaf(d2, co, -(d1 * si));
}

__device__ __forceinline__ float swiglu_fwd1(float a, float b) { return rbf(a * sigmoidf_(a)) * b; }
__device__ __forceinline__ void swiglu_bwd1(float g, float a, float b, float& da, float& db, float& act) {
    const float sg = sigmoidf_(a);
    const float s = rbf(a * sg);                    // bf16 silu(a), as saved by autograd
    const float ds = rbf(g * b);                    // grad wrt silu output (bf16 mul backward)
    da = ds * (sg * (1.f + a * (1.f - sg)));        // silu'(a)
    db = g * s;
    act = s * b;
}
