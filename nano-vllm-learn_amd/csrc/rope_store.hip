// Fused (optional per-head q/k RMSNorm ->) neox RoPE -> store_kvcache for gfx950: the step immediately before
// the attention call.  One launch replaces the reference's q_norm / k_norm
// (nanovllm/models/qwen3.py:108-114, layers/layernorm.py:17-27), rotary_emb (layers/rotary_embedding.py:6-55)
// and store_kvcache (layers/attention.py:19-55, :84-86) — about twenty elementwise launches per layer in eager
// PyTorch — and removes one K/V round trip through HBM.
//
// q and k are rotated IN PLACE inside the fused qkv projection output (row stride (H+2KVH)*D), so the
// attention call still sees q/k/v as views of that buffer; rotated k and untouched v rows are also written
// to the paged cache at slot_mapping[token] (slot < 0 or no cache: skipped).
//
// Mapping: a head row of D bf16 is handled by D/16 lanes; lane c owns elements [8c, 8c+8) of BOTH halves
// (x1 = head[0:D/2], x2 = head[D/2:D]), i.e. two 16-byte loads and two 16-byte stores (four with the cache
// copy), and the cos/sin it needs are 2 x 32 contiguous bytes of the fp32 table.  Memory bound and tiny at
// decode (B rows); at prefill it streams Σtokens*(H+2KVH)*D*2 bytes once.
//   y1 = x1*cos - x2*sin,  y2 = x2*cos + x1*sin   in fp32, each product and sum rounded separately
//   (rotary_embedding.py:12-15), then one rounding to bf16.
//   RMSNorm (Qwen3 only): x * rsqrt(mean(x^2) + eps) in fp32, rounded to bf16, THEN multiplied by the bf16
//   weight (layernorm.py:23-26) — the product is rounded to bf16 once more, as the reference's in-place mul does.
#include "common.h"
#include "kernels.h"

// The whole file: no fused multiply-add contraction.  RoPE products and sums must round separately to match the
// reference's elementwise fp32 ops bit for bit (an exact bf16 tie flipped under -ffp-contract=fast).
#pragma clang fp contract(off)

namespace nvh {

namespace {

template <int D, bool NORM>
__global__ __launch_bounds__(256) void rope_store_kernel(const RopeStoreArgs a) {
    constexpr int LPH = D / 16;                      // lanes per head row
    const int rows = a.h + 2 * a.kvh;                // q heads, k heads, v heads of one token
    const int64_t total = (int64_t)a.n_tokens * rows * LPH;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % LPH);
        const int64_t tr = idx / LPH;
        const int r = (int)(tr % rows);
        const int tok = (int)(tr / rows);
        uint16_t* head = a.qkv + tok * a.qkv_row_stride + (int64_t)r * D;
        u32x4 lo = *reinterpret_cast<const u32x4*>(head + 8 * c);                 // x1[8c .. 8c+8)
        u32x4 hi = *reinterpret_cast<const u32x4*>(head + D / 2 + 8 * c);         // x2[8c .. 8c+8)
        const bool is_v = r >= a.h + a.kvh;
        if (!is_v) {
            float x1[8], x2[8];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                x1[2 * w] = bf16_lo(lo[w]); x1[2 * w + 1] = bf16_hi(lo[w]);
                x2[2 * w] = bf16_lo(hi[w]); x2[2 * w + 1] = bf16_hi(hi[w]);
            }
            if constexpr (NORM) {
                const uint16_t* wgt = r < a.h ? a.q_norm_w : a.k_norm_w;
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x1[e] * x1[e] + x2[e] * x2[e];
                ss += pair_in_row<1>(ss);                      // D/16 = 4 or 8 lanes per head, aligned groups
                ss += pair_in_row<2>(ss);
                if constexpr (LPH == 8) ss += pair_in_row<4>(ss);
                const float inv = rsqrtf(ss * (1.f / D) + a.eps);
                const u32x4 wl = *reinterpret_cast<const u32x4*>(wgt + 8 * c);
                const u32x4 wh = *reinterpret_cast<const u32x4*>(wgt + D / 2 + 8 * c);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // round the normalised value to bf16, multiply by the bf16 weight, round again
                    x1[2 * w] = (float)(__bf16)((float)(__bf16)(x1[2 * w] * inv) * bf16_lo(wl[w]));
                    x1[2 * w + 1] = (float)(__bf16)((float)(__bf16)(x1[2 * w + 1] * inv) * bf16_hi(wl[w]));
                    x2[2 * w] = (float)(__bf16)((float)(__bf16)(x2[2 * w] * inv) * bf16_lo(wh[w]));
                    x2[2 * w + 1] = (float)(__bf16)((float)(__bf16)(x2[2 * w + 1] * inv) * bf16_hi(wh[w]));
                }
            }
            const int64_t pos = a.positions[tok];
            const float* cs = a.cos_sin + pos * D + 8 * c;                       // [cos(0..D/2) | sin(0..D/2)] per position
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + D / 2), s1 = *reinterpret_cast<const f32x4*>(cs + D / 2 + 4);
            float y1[8], y2[8];
            {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float co = e < 4 ? c0[e] : c1[e - 4], si = e < 4 ? s0[e] : s1[e - 4];
                    const float a1 = x1[e] * co, b1 = x2[e] * si, a2 = x2[e] * co, b2 = x1[e] * si;
                    y1[e] = a1 - b1;
                    y2[e] = a2 + b2;
                }
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                lo[w] = pack_bf16x2(y1[2 * w], y1[2 * w + 1]);
                hi[w] = pack_bf16x2(y2[2 * w], y2[2 * w + 1]);
            }
            *reinterpret_cast<u32x4*>(head + 8 * c) = lo;
            *reinterpret_cast<u32x4*>(head + D / 2 + 8 * c) = hi;
        }
        if (r >= a.h && a.k_cache) {                                              // k (rotated) or v row -> paged cache
            const int slot = a.slot_mapping ? a.slot_mapping[tok] : -1;
            if (slot >= 0) {
                const int kvr = is_v ? r - a.h - a.kvh : r - a.h;
                uint16_t* dst = (is_v ? a.v_cache : a.k_cache) + ((int64_t)slot * a.kvh + kvr) * D;
                *reinterpret_cast<u32x4*>(dst + 8 * c) = lo;
                *reinterpret_cast<u32x4*>(dst + D / 2 + 8 * c) = hi;
            }
        }
    }
}

template <int D>
int launch_d(const RopeStoreArgs& a, hipStream_t stream) {
    const int64_t total = (int64_t)a.n_tokens * (a.h + 2 * a.kvh) * (D / 16);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (a.q_norm_w) hipLaunchKernelGGL((rope_store_kernel<D, true>), dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((rope_store_kernel<D, false>), dim3(blocks), dim3(256), 0, stream, a);
    return check_launch("rope_store");
}

}  // namespace

int launch_rope_store(const RopeStoreArgs& a, hipStream_t stream) {
    if (a.n_tokens == 0) return 0;
    return a.hd == 64 ? launch_d<64>(a, stream) : launch_d<128>(a, stream);
}

}  // namespace nvh
