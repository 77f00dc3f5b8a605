// K1 forward: varlen, causal / full, GQA flash attention for gfx950.
//
// Grid (q tiles of 128 rows [heaviest first], q heads, sequences); block = 4 wave64, each wave owns
// 32 query rows.  K/V tiles of 64 keys arrive by LDS-DMA into double-buffered natural-layout images
// (flash_common.h): the next tile is in flight under the current tile's MFMAs, one barrier per tile,
// no staging registers, no ds_write.  Scores are formed as S^T = K Q^T so every lane owns ONE query
// row: the online-softmax max/sum are in-lane reductions plus a single lane^32 exchange, and the
// rescale of O^T (also lane-per-row) is a plain multiply, skipped when no row maximum moved.
// P feeds the PV MFMA straight from the score accumulators; V^T fragments come from the natural V image
// through ds_read_b64_tr_b16.  The tile loop is unrolled by the two pipeline stages so every LDS address
// is a precomputed per-lane offset + an immediate (no address arithmetic in the loop).
//
// Roofline: MFMA-bound; causal flop = 4*D*sum_s(len_s^2)/2 per head; algorithmic bytes = q+k+v+o.
#include "flash_common.h"

#include <type_traits>

namespace {

using namespace fa;

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void flash_fwd_k(const bf16_t* __restrict__ q, int64_t q_ts, const bf16_t* __restrict__ k,
                                                      const bf16_t* __restrict__ v, int64_t kv_ts, bf16_t* __restrict__ out,
                                                      int64_t o_ts, float* __restrict__ lse, const int32_t* __restrict__ cu, int64_t T,
                                                      int hq, int hkv, float scale) {
    using G = Geo<D>;
    constexpr int STAGE = 2 * G::IMG_BYTES;                                       // K image, V image
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int seq = blockIdx.z;
    // dispatch order = x fastest: ALL heads of the heaviest (last) query tile first, then the next tile ... -- a global
    // longest-job-first order over the causal work (head-major order started the last heads' heavy tiles half-way through the
    // kernel and left most SIMD slots idle in the tail); neighbouring blocks are the q heads of one kv head -> shared K/V in L2
    // consecutive workgroups go to consecutive XCDs: x -> head so that the q heads of one kv head (x = hk, hk + hkv, ...) share an XCD's L2
    const int h = ((int)blockIdx.x % hkv) * (hq / hkv) + (int)blockIdx.x / hkv;
    const int qt = gridDim.y - 1 - blockIdx.y;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int q0 = qt * 128;
    if (q0 >= len) return;
    const int hk = h / (hq / hkv);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qw0 = q0 + wave * 32;            // first query row of this wave
    const int my_q = qw0 + (lane & 31);        // this lane's query row (within the sequence)
    const bool q_valid = my_q < len;

    const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
    const int ntiles = (kv_end + 63) / 64;
    const bf16_t* kbase = k + (int64_t)tok0 * kv_ts + (int64_t)hk * D;
    const bf16_t* vbase = v + (int64_t)tok0 * kv_ts + (int64_t)hk * D;

    TileSrc<D, 4> ksrc, vsrc;
    ksrc.init(kbase, kv_ts, T - tok0, D, wave, lane);
    vsrc.init(vbase, kv_ts, T - tok0, D, wave, lane);
    ksrc.issue(smem, 0, 0, wave);
    vsrc.issue(smem + G::IMG_BYTES, 0, 0, wave);

    FragOffs<D> fo;
    fo.init(lane);

    // Q fragments (B operand of S^T = K Q^T), loaded while the first tile is in flight
    s16x8 qf[G::KS];
    {
        const bf16_t* qp = q + (int64_t)(tok0 + my_q) * q_ts + (int64_t)h * D + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            union { uint4 u; s16x8 s; } cv;
            cv.u = q_valid ? ld16(qp + ks * 16) : z4();
            qf[ks] = cv.s;
        }
    }

    f32x16 oacc[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) oacc[db] = zero16();
    float m_run = -INFINITY, l_run = 0.f;     // running max of the RAW scores, running sum
    const float sc2 = scale * kLog2e;         // exp(scale * s) = exp2(sc2 * s)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](auto stage_c, int t) {
        constexpr int S = decltype(stage_c)::value;
        const int kv0 = t * 64;
        const unsigned char* Ks = smem + S * STAGE;
        const unsigned char* Vs = Ks + G::IMG_BYTES;
        if (t + 1 < ntiles) {
            unsigned char* nxt = smem + (1 - S) * STAGE;
            ksrc.issue(nxt, kv0 + 64, 0, wave);
            vsrc.issue(nxt + G::IMG_BYTES, kv0 + 64, 0, wave);
        }
        // wave-uniform: does this wave see any unmasked key in this tile?
        const bool active = !CAUSAL || kv0 <= qw0 + 31;
        if (active) {
            f32x16 sacc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sacc[c] = zero16();
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks)
                    sacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<D>(Ks, 32 * c, ks, fo), qf[ks], sacc[c], 0, 0, 0);
            }
            const bool need_mask = (CAUSAL && kv0 + 63 > qw0) || (kv0 + 64 > len);
            float mx = -INFINITY;
            if (need_mask) {  // wave-uniform: only the diagonal / last tiles pay for the masks (selects, no per-element branches)
                const int lim = CAUSAL ? min(len - 1, my_q) : len - 1;  // largest visible key of this lane's query
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 32 * c + creg_row(r, lane);
                        sacc[c][r] = key > lim ? -INFINITY : sacc[c][r];
                    }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[c][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * sc2);  // m_run = -inf -> 0
            const float moff = m_use * sc2;
            float rs = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[c][r], sc2, -moff));
                    sacc[c][r] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (!__all(alpha == 1.f)) {  // wave-uniform: some row's running max moved
#pragma unroll
                for (int db = 0; db < G::DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            }
            // O^T += V^T P^T
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const s16x8 pf = pack_frag(sacc[c], s2);
#pragma unroll
                    for (int db = 0; db < G::DB; ++db)
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trans_frag<D>(Vs, db, 2 * c + s2, fo), pf, oacc[db], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile landed (this wave's share)
        __syncthreads();
    };

    for (int t = 0; t < ntiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    }

    if (q_valid) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        bf16_t* op = out + (int64_t)(tok0 + my_q) * o_ts + (int64_t)h * D;
#pragma unroll
        for (int db = 0; db < G::DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(oacc[db][4 * g + 0] * inv, oacc[db][4 * g + 1] * inv);
                w.y = pack2bf(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
                st8(op + 32 * db + 8 * g + 4 * (lane >> 5), w);
            }
        // lse of the SCALED scores in natural log: scale * m + ln(l)
        if (lane < 32) lse[(int64_t)h * T + tok0 + my_q] = (l_run > 0.f) ? m_run * scale + logf(l_run) : -INFINITY;
    }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" int ie_flash_attn_fwd(const void* q, int64_t q_ts, const void* k, const void* v, int64_t kv_ts, void* out, int64_t o_ts,
                                 float* lse, const int32_t* cu_seqlens, int nseq, int64_t T, int max_seqlen, int hq, int hkv, int d,
                                 float softmax_scale, int causal, void* stream) {
    IE_CHECK_ARG(q && k && v && out && lse && cu_seqlens, "ie_flash_attn_fwd: null pointer");
    IE_CHECK_ARG(nseq >= 0 && T >= 0 && max_seqlen >= 0 && hq > 0 && hkv > 0 && hq % hkv == 0, "ie_flash_attn_fwd: bad shape");
    IE_CHECK_ARG(softmax_scale > 0.f, "ie_flash_attn_fwd: softmax_scale must be positive");
    IE_CHECK_SUPPORTED(d == 128 || d == 64, "ie_flash_attn_fwd: head dim must be 64 or 128");
    IE_CHECK_SUPPORTED(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && q_ts % 8 == 0 && kv_ts % 8 == 0 && o_ts % 4 == 0,
                       "ie_flash_attn_fwd: pointers must be 16-byte aligned and token strides multiples of 8");
    if (nseq == 0 || T == 0 || max_seqlen == 0) return IE_OK;
    dim3 grid((unsigned)hq, (unsigned)((max_seqlen + 127) / 128), (unsigned)nseq);
    hipStream_t st = (hipStream_t)stream;
#define IE_L(DD, CA)                                                                                                              \
    hipLaunchKernelGGL((flash_fwd_k<DD, CA>), grid, dim3(256), 0, st, (const bf16_t*)q, q_ts, (const bf16_t*)k, (const bf16_t*)v, kv_ts, \
                       (bf16_t*)out, o_ts, lse, cu_seqlens, T, hq, hkv, softmax_scale)
    if (d == 128) { if (causal) IE_L(128, true); else IE_L(128, false); }
    else          { if (causal) IE_L(64, true); else IE_L(64, false); }
#undef IE_L
    return ie_launch_status("ie_flash_attn_fwd launch");
}
