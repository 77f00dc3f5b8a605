// Torch-free kernel micro-benchmark / A-B harness for libinternevo_hip.so (development tool, not on the product path).
//
// A gpurun call pays ~1-2 minutes for the first `import torch` on a fresh box; this binary starts in a second, so one call
// can compare many kernel variants.  It links nothing but the HIP runtime and dlopen()s the C ABI like any other host would.
//
//   kbench fwd  [--seqs 4 --len 4096 --hq 32 --hkv 8 --d 128 --iters 20 --variants 0,1,2]
//   kbench bwd  [same shape flags] [--variants ...]            (variant = value passed to the tune hooks, see below)
//   kbench gemm [--m 16384 --n 4096 --k 4096 --layout nt|nn|tn --variants -1,11,13]
//
// Every variant is checked against variant 0 (fwd / bwd: the round-1 kernels, which the test-suite pins on the oracle;
// gemm: an fp32-accumulating reference kernel compiled here) before it is timed, on uniform random data in [-1, 1).
// Output: one JSON line per variant (us per call, TFLOP/s of the algorithmic flops, max abs / rel difference).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#define HIP_OK(x)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

typedef unsigned short bf16_t;

static inline bf16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline float bf2f(bf16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    inline uint32_t next() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (uint32_t)(s >> 32);
    }
    inline float uni() { return (float)(next() >> 8) * (2.0f / 16777216.0f) - 1.0f; }
};

static bf16_t* dev_random_bf16(size_t n, uint64_t seed, float scale) {
    std::vector<bf16_t> h(n);
    Rng r(seed);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(r.uni() * scale);
    bf16_t* d;
    HIP_OK(hipMalloc(&d, n * 2));
    HIP_OK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}

template <class T> static T* dev_alloc(size_t n) {
    T* d;
    HIP_OK(hipMalloc(&d, n * sizeof(T)));
    HIP_OK(hipMemset(d, 0, n * sizeof(T)));
    return d;
}

template <class T> static std::vector<T> to_host(const T* d, size_t n) {
    std::vector<T> h(n);
    HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
}

struct Diff {
    double max_abs = 0, max_ref = 0, sum_sq = 0, ref_sq = 0;
    size_t bad = 0;
};
static Diff diff_bf16(const std::vector<bf16_t>& a, const std::vector<bf16_t>& ref) {
    Diff d;
    for (size_t i = 0; i < a.size(); ++i) {
        const double x = bf2f(a[i]), y = bf2f(ref[i]);
        if (!(std::isfinite(x))) d.bad++;
        const double e = fabs(x - y);
        if (e > d.max_abs) d.max_abs = e;
        if (fabs(y) > d.max_ref) d.max_ref = fabs(y);
        d.sum_sq += e * e;
        d.ref_sq += y * y;
    }
    return d;
}
static Diff diff_f32(const std::vector<float>& a, const std::vector<float>& ref) {
    Diff d;
    for (size_t i = 0; i < a.size(); ++i) {
        const double x = a[i], y = ref[i];
        if (std::isinf(y) && std::isinf(x)) continue;
        if (!(std::isfinite(x))) d.bad++;
        const double e = fabs(x - y);
        if (e > d.max_abs) d.max_abs = e;
        if (fabs(y) > d.max_ref) d.max_ref = fabs(y);
        d.sum_sq += e * e;
        d.ref_sq += y * y;
    }
    return d;
}

static void* g_lib = nullptr;
static void* g_ref_lib = nullptr;   // IE_REF_LIB=<path>: fwd / bwd reference outputs come from this library (default: the library under test)
template <class F> static F sym(const char* name) {
    void* p = dlsym(g_lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s\n", name);
        exit(2);
    }
    return (F)p;
}
template <class F> static F ref_sym(const char* name) {
    void* p = dlsym(g_ref_lib ? g_ref_lib : g_lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s in the reference library\n", name);
        exit(2);
    }
    return (F)p;
}
typedef const char* (*last_error_t)();
static last_error_t g_last_error;
#define IE_OKAY(x)                                                                       \
    do {                                                                                 \
        int rc_ = (x);                                                                   \
        if (rc_ != 0) {                                                                  \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, g_last_error()); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

static int g_flush = 0;
struct Args {
    int seqs = 4, len = 4096, hq = 32, hkv = 8, d = 128, iters = 20, causal = 1;
    int64_t m = 16384, n = 4096, k = 4096;
    std::string layout = "nt";
    std::vector<int> variants{0};
    int ragged = 0;
};

static std::vector<int> parse_list(const char* s) {
    std::vector<int> v;
    const char* p = s;
    while (*p) {
        v.push_back((int)strtol(p, (char**)&p, 10));
        if (*p == ',') ++p;
    }
    return v;
}

static Args parse(int argc, char** argv) {
    Args a;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string f = argv[i];
        const char* v = argv[i + 1];
        if (f == "--seqs") a.seqs = atoi(v);
        else if (f == "--len") a.len = atoi(v);
        else if (f == "--hq") a.hq = atoi(v);
        else if (f == "--hkv") a.hkv = atoi(v);
        else if (f == "--d") a.d = atoi(v);
        else if (f == "--iters") a.iters = atoi(v);
        else if (f == "--causal") a.causal = atoi(v);
        else if (f == "--ragged") a.ragged = atoi(v);
        else if (f == "--flush") g_flush = atoi(v);
        else if (f == "--m") a.m = atoll(v);
        else if (f == "--n") a.n = atoll(v);
        else if (f == "--k") a.k = atoll(v);
        else if (f == "--layout") a.layout = v;
        else if (f == "--variants") a.variants = parse_list(v);
        else {
            fprintf(stderr, "unknown flag %s\n", f.c_str());
            exit(2);
        }
    }
    return a;
}

// --flush 1: a 1-GiB streaming copy between the timed calls (every call then finds its inputs in HBM, not in the 256-MiB Infinity Cache a loop over one set of
// buffers keeps them in -- the training step's situation: 32 layers' worth of q / k / v); each call is timed by its own event pair
static char *g_flush_a = nullptr, *g_flush_b = nullptr;
template <class Fn> static double time_us_flushed(Fn&& fn, int iters) {
    const size_t n = (size_t)1 << 30;
    if (!g_flush_a) {
        HIP_OK(hipMalloc(&g_flush_a, n));
        HIP_OK(hipMalloc(&g_flush_b, n));
        HIP_OK(hipMemset(g_flush_a, 1, n));
    }
    hipEvent_t s, e;
    HIP_OK(hipEventCreate(&s));
    HIP_OK(hipEventCreate(&e));
    double total = 0;
    for (int i = 0; i < iters + 2; ++i) {
        HIP_OK(hipMemcpyAsync(g_flush_b, g_flush_a, n, hipMemcpyDeviceToDevice, 0));
        HIP_OK(hipEventRecord(s, 0));
        fn();
        HIP_OK(hipEventRecord(e, 0));
        HIP_OK(hipEventSynchronize(e));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, s, e));
        if (i >= 2) total += ms;
    }
    return total * 1e3 / iters;
}

template <class Fn> static double time_us(Fn&& fn, int iters) {
    if (g_flush) return time_us_flushed(fn, iters);
    for (int i = 0; i < 3; ++i) fn();
    HIP_OK(hipDeviceSynchronize());
    hipEvent_t s, e;
    HIP_OK(hipEventCreate(&s));
    HIP_OK(hipEventCreate(&e));
    HIP_OK(hipEventRecord(s, 0));
    for (int i = 0; i < iters; ++i) fn();
    HIP_OK(hipEventRecord(e, 0));
    HIP_OK(hipEventSynchronize(e));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, s, e));
    return ms * 1e3 / iters;
}

// ---------------------------------------------------------------------------------------------------------- attention
typedef int (*fwd_t)(const void*, int64_t, const void*, const void*, int64_t, void*, int64_t, float*, const int32_t*, int, int64_t, int, int,
                     int, int, float, int, void*);
typedef int (*bwd_t)(const void*, int64_t, const void*, int64_t, const void*, const void*, int64_t, const void*, int64_t, const float*, float*,
                     void*, int64_t, void*, void*, int64_t, const int32_t*, int, int64_t, int, int, int, int, float, int, void*);
typedef int64_t (*bwd_ws_t)(int64_t, int, int, int);
typedef int (*tune_t)(int);

struct AttnProblem {
    Args a;
    int64_t T;
    int max_len;
    std::vector<int32_t> cu;
    int32_t* cu_d;
    bf16_t *q, *kv, *dout;
    double flops_fwd;  // algorithmic: 4 * D * sum(len^2) / 2 per head (causal)
    explicit AttnProblem(const Args& a_) : a(a_) {
        cu.push_back(0);
        Rng r(7);
        max_len = 0;
        double pairs = 0;
        for (int s = 0; s < a.seqs; ++s) {
            int L = a.len;
            if (a.ragged) L = 1 + (int)(r.next() % (uint32_t)a.len);  // ragged packs: uneven lengths, not multiples of any tile
            cu.push_back(cu.back() + L);
            if (L > max_len) max_len = L;
            pairs += a.causal ? (double)L * (L + 1) / 2 : (double)L * L;
        }
        T = cu.back();
        flops_fwd = 4.0 * a.d * pairs * a.hq;
        HIP_OK(hipMalloc(&cu_d, cu.size() * 4));
        HIP_OK(hipMemcpy(cu_d, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
        q = dev_random_bf16((size_t)T * a.hq * a.d, 11, 1.0f);
        kv = dev_random_bf16((size_t)T * 2 * a.hkv * a.d, 12, 1.0f);
        dout = dev_random_bf16((size_t)T * a.hq * a.d, 13, 1.0f);
    }
};

static int run_fwd(const Args& a) {
    AttnProblem P(a);
    auto fwd = sym<fwd_t>("ie_flash_attn_fwd");
    auto tune = sym<tune_t>("ie_tune_flash_fwd_variant");
    const size_t no = (size_t)P.T * a.hq * a.d, nl = (size_t)a.hq * P.T;
    bf16_t* out = dev_alloc<bf16_t>(no);
    float* lse = dev_alloc<float>(nl);
    const float scale = 1.0f / sqrtf((float)a.d);
    auto call = [&]() {
        IE_OKAY(fwd(P.q, (int64_t)a.hq * a.d, P.kv, P.kv + (size_t)a.hkv * a.d, (int64_t)2 * a.hkv * a.d, out, (int64_t)a.hq * a.d, lse, P.cu_d,
                    a.seqs, P.T, P.max_len, a.hq, a.hkv, a.d, scale, a.causal, nullptr));
    };
    IE_OKAY(ref_sym<tune_t>("ie_tune_flash_fwd_variant")(0));
    IE_OKAY(ref_sym<fwd_t>("ie_flash_attn_fwd")(P.q, (int64_t)a.hq * a.d, P.kv, P.kv + (size_t)a.hkv * a.d, (int64_t)2 * a.hkv * a.d, out, (int64_t)a.hq * a.d, lse,
                                                P.cu_d, a.seqs, P.T, P.max_len, a.hq, a.hkv, a.d, scale, a.causal, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const auto ref_o = to_host(out, no);
    const auto ref_l = to_host(lse, nl);
    for (int v : a.variants) {
        IE_OKAY(tune(v));
        HIP_OK(hipMemset(out, 0xff, no * 2));
        HIP_OK(hipMemset(lse, 0xff, nl * 4));
        call();
        HIP_OK(hipDeviceSynchronize());
        const Diff dO = diff_bf16(to_host(out, no), ref_o), dL = diff_f32(to_host(lse, nl), ref_l);
        const double us = time_us(call, a.iters);
        printf("{\"bench\": \"flash_fwd\", \"variant\": %d, \"T\": %lld, \"seqs\": %d, \"ragged\": %d, \"hq\": %d, \"hkv\": %d, \"d\": %d, \"causal\": %d, "
               "\"us\": %.1f, \"tflops\": %.1f, \"o_max_abs_diff\": %.3g, \"o_rms_rel\": %.3g, \"o_nonfinite\": %zu, \"lse_max_abs_diff\": %.3g}\n",
               v, (long long)P.T, a.seqs, a.ragged, a.hq, a.hkv, a.d, a.causal, us, P.flops_fwd / us * 1e-6, dO.max_abs,
               sqrt(dO.sum_sq / (dO.ref_sq + 1e-30)), dO.bad, dL.max_abs);
        fflush(stdout);
    }
    IE_OKAY(tune(-1));
    return 0;
}

static int run_bwd(const Args& a) {
    AttnProblem P(a);
    auto fwd = sym<fwd_t>("ie_flash_attn_fwd");
    auto bwd = sym<bwd_t>("ie_flash_attn_bwd");
    auto ws = sym<bwd_ws_t>("ie_flash_attn_bwd_workspace");
    auto tune = sym<tune_t>("ie_tune_flash_bwd_variant");
    {   // variants with bit 1 (five-product backward) need a spill buffer
        typedef int64_t (*spill_bytes_t)(int, int, int, int);
        typedef int (*set_spill_t)(void*, int64_t);
        auto sb = (spill_bytes_t)dlsym(g_lib, "ie_flash_attn_bwd_spill_bytes");
        auto ss = (set_spill_t)dlsym(g_lib, "ie_flash_attn_bwd_set_spill");
        if (sb && ss) {
            const int64_t bytes = sb(a.seqs, P.max_len, a.hq, a.causal);
            void* buf;
            HIP_OK(hipMalloc(&buf, (size_t)bytes));
            IE_OKAY(ss(buf, bytes));
        }
    }
    const size_t no = (size_t)P.T * a.hq * a.d, nl = (size_t)a.hq * P.T, nkv = (size_t)P.T * a.hkv * a.d;
    bf16_t* out = dev_alloc<bf16_t>(no);
    float* lse = dev_alloc<float>(nl);
    bf16_t* dq = dev_alloc<bf16_t>(no);
    bf16_t* dkv = dev_alloc<bf16_t>(2 * nkv);
    float* delta = dev_alloc<float>((size_t)ws(P.T, a.hq, a.hkv, a.d));
    const float scale = 1.0f / sqrtf((float)a.d);
    const int64_t qts = (int64_t)a.hq * a.d, kvts = (int64_t)2 * a.hkv * a.d;
    IE_OKAY(fwd(P.q, qts, P.kv, P.kv + (size_t)a.hkv * a.d, kvts, out, qts, lse, P.cu_d, a.seqs, P.T, P.max_len, a.hq, a.hkv, a.d, scale, a.causal,
                nullptr));
    auto call = [&]() {
        IE_OKAY(bwd(P.dout, qts, P.q, qts, P.kv, P.kv + (size_t)a.hkv * a.d, kvts, out, qts, lse, delta, dq, qts, dkv, dkv + (size_t)a.hkv * a.d, kvts,
                    P.cu_d, a.seqs, P.T, P.max_len, a.hq, a.hkv, a.d, scale, a.causal, nullptr));
    };
    IE_OKAY(ref_sym<tune_t>("ie_tune_flash_bwd_variant")(0));
    IE_OKAY(ref_sym<bwd_t>("ie_flash_attn_bwd")(P.dout, qts, P.q, qts, P.kv, P.kv + (size_t)a.hkv * a.d, kvts, out, qts, lse, delta, dq, qts, dkv,
                                                dkv + (size_t)a.hkv * a.d, kvts, P.cu_d, a.seqs, P.T, P.max_len, a.hq, a.hkv, a.d, scale, a.causal, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const auto ref_dq = to_host(dq, no);
    const auto ref_dkv = to_host(dkv, 2 * nkv);
    for (int v : a.variants) {
        IE_OKAY(tune(v));
        HIP_OK(hipMemset(dq, 0xff, no * 2));
        HIP_OK(hipMemset(dkv, 0xff, 2 * nkv * 2));
        call();
        HIP_OK(hipDeviceSynchronize());
        const Diff d1 = diff_bf16(to_host(dq, no), ref_dq), d2 = diff_bf16(to_host(dkv, 2 * nkv), ref_dkv);
        const double us = time_us(call, a.iters);
        printf("{\"bench\": \"flash_bwd\", \"variant\": %d, \"T\": %lld, \"seqs\": %d, \"ragged\": %d, \"hq\": %d, \"hkv\": %d, \"d\": %d, \"causal\": %d, "
               "\"us\": %.1f, \"tflops_algorithmic\": %.1f, \"dq_max_abs_diff\": %.3g, \"dq_rms_rel\": %.3g, \"dkv_max_abs_diff\": %.3g, "
               "\"dkv_rms_rel\": %.3g, \"nonfinite\": %zu}\n",
               v, (long long)P.T, a.seqs, a.ragged, a.hq, a.hkv, a.d, a.causal, us, 2.5 * P.flops_fwd / us * 1e-6, d1.max_abs,
               sqrt(d1.sum_sq / (d1.ref_sq + 1e-30)), d2.max_abs, sqrt(d2.sum_sq / (d2.ref_sq + 1e-30)), d1.bad + d2.bad);
        fflush(stdout);
    }
    IE_OKAY(tune(0));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------- gemm
typedef int (*gemm_tile_t)(int, const void*, int64_t, int, const void*, int64_t, int, void*, int64_t, int64_t, int64_t, int64_t, int, void*);

// reference: one thread per output element, fp32 accumulate over a sampled set of rows (full reference would take seconds at 16k)
__global__ void ref_gemm_rows(const bf16_t* A, int64_t lda, int akm, const bf16_t* B, int64_t ldb, int bkm, float* C, int64_t N, int64_t K,
                              const int* rows, int nrows) {
    const int64_t n = blockIdx.x * 256ll + threadIdx.x;
    const int ri = blockIdx.y;
    if (n >= N || ri >= nrows) return;
    const int64_t m = rows[ri];
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) {
        const bf16_t a = akm ? A[k * lda + m] : A[m * lda + k];
        const bf16_t b = bkm ? B[k * ldb + n] : B[n * ldb + k];
        acc += __uint_as_float(((unsigned)a) << 16) * __uint_as_float(((unsigned)b) << 16);
    }
    C[(int64_t)ri * N + n] = acc;
}

static int run_gemm(const Args& a) {
    auto gemm = sym<gemm_tile_t>("ie_gemm_bf16_tile");
    const int akm = a.layout[0] == 't', bkm = a.layout[1] == 'n';  // "nt" fwd, "nn" dgrad, "tn" wgrad
    const int64_t M = a.m, N = a.n, K = a.k;
    const int64_t lda = akm ? M : K, ldb = bkm ? N : K;
    bf16_t* A = dev_random_bf16((size_t)M * K, 21, 1.0f);
    bf16_t* B = dev_random_bf16((size_t)N * K, 22, 1.0f);
    bf16_t* C = dev_alloc<bf16_t>((size_t)M * N);
    const int nrows = 64;
    std::vector<int> rows(nrows);
    Rng r(5);
    for (int i = 0; i < nrows; ++i) rows[i] = (int)(r.next() % (uint32_t)M);
    rows[0] = 0;
    rows[1] = (int)M - 1;
    int* rows_d;
    HIP_OK(hipMalloc(&rows_d, nrows * 4));
    HIP_OK(hipMemcpy(rows_d, rows.data(), nrows * 4, hipMemcpyHostToDevice));
    float* Cref = dev_alloc<float>((size_t)nrows * N);
    hipLaunchKernelGGL(ref_gemm_rows, dim3((unsigned)((N + 255) / 256), nrows), dim3(256), 0, 0, A, lda, akm, B, ldb, bkm, Cref, N, K, rows_d, nrows);
    HIP_OK(hipDeviceSynchronize());
    const auto ref = to_host(Cref, (size_t)nrows * N);
    for (int v : a.variants) {
        auto call = [&]() { IE_OKAY(gemm(v, A, lda, akm, B, ldb, bkm, C, N, M, N, K, 0, nullptr)); };
        HIP_OK(hipMemset(C, 0xff, (size_t)M * N * 2));
        call();
        HIP_OK(hipDeviceSynchronize());
        const auto got = to_host(C, (size_t)M * N);
        double max_rel = 0;
        size_t bad = 0;
        for (int i = 0; i < nrows; ++i)
            for (int64_t n = 0; n < N; ++n) {
                const double x = bf2f(got[(size_t)rows[i] * N + n]), y = ref[(size_t)i * N + n];
                if (!std::isfinite(x)) bad++;
                const double e = fabs(x - y) / (fabs(y) + sqrt((double)K) * 0.05);
                if (e > max_rel) max_rel = e;
            }
        const double us = time_us(call, a.iters);
        printf("{\"bench\": \"gemm\", \"variant\": %d, \"layout\": \"%s\", \"M\": %lld, \"N\": %lld, \"K\": %lld, \"us\": %.1f, \"tflops\": %.1f, "
               "\"max_scaled_err\": %.3g, \"nonfinite\": %zu}\n",
               v, a.layout.c_str(), (long long)M, (long long)N, (long long)K, us, 2.0 * M * N * K / us * 1e-6, max_rel, bad);
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: kbench fwd|bwd|gemm [flags]\n");
        return 2;
    }
    const char* path = getenv("IE_LIB");
    std::string lib = path ? path : "internevo_amd/csrc/libinternevo_hip.so";
    g_lib = dlopen(lib.c_str(), RTLD_NOW);
    if (!g_lib) {
        fprintf(stderr, "dlopen(%s): %s\n", lib.c_str(), dlerror());
        return 2;
    }
    if (const char* rp = getenv("IE_REF_LIB")) {
        g_ref_lib = dlopen(rp, RTLD_NOW | RTLD_LOCAL);
        if (!g_ref_lib) {
            fprintf(stderr, "dlopen(%s): %s\n", rp, dlerror());
            return 2;
        }
    }
    g_last_error = sym<last_error_t>("ie_last_error");
    HIP_OK(hipSetDevice(0));
    const Args a = parse(argc, argv);
    const std::string cmd = argv[1];
    if (cmd == "fwd") return run_fwd(a);
    if (cmd == "bwd") return run_bwd(a);
    if (cmd == "gemm") return run_gemm(a);
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
