// Paged single-query (decode) attention for gfx950.
// Replaces flash_attn_with_kvcache at nanovllm/layers/attention.py:99-101
// (oracle: nanovllm/layers/attention_sdpa.py:122-182).
//
// HBM-bound: every K/V byte of the live context is read exactly once per kv head, shared by the
// G = H/KVH query heads of the group.  Algorithmic bytes per launch:
//     sum_b 2*ctx_b*KVH*D*2  +  2*B*H*D*2 (q in, o out)  +  4*(sum_b ceil(ctx_b/bs) + B)
// (SURVEY.md section 8d).  Split-KV partials and the combine pass are overhead, not algorithmic.
//
// Structure (flash-decoding, wave64):
//   split kernel   grid (num_splits, KVH, B), 256 threads = 4 waves.  A workgroup owns SPLIT
//                  consecutive tokens of one (sequence, kv head); each wave owns one tile of WT of
//                  them.  A token row (D bf16) is read by LPT = D/8 lanes x 16 B
//                  (global_load_dwordx4), so one wave instruction moves 1 KiB = 64/LPT whole rows;
//                  a wave issues all 8 K and 8 V loads of its tile (16 KiB in flight) before any use.
//                  QK^T: v_dot2c_f32_bf16 partial dots + DPP butterfly over the LPT lanes of a row.
//                  softmax: exp2 domain, tile max/sum by wavefront reductions (DPP + cross-row).
//                  PV: fp32 FMA into acc[G][8] per lane, slot lanes reduced once at the end.
//                  The 4 waves combine through LDS and write one (max, sum, acc[G][D]) partial.
//   combine kernel grid (B*H), D threads: merges the ceil(ctx/SPLIT) partials, normalises, rounds.
// Grids depend only on static shapes; splits past context_lens[b] exit at once (graph-safe),
// block-table entries past ceil(ctx/bs) are never read, cache offsets are 64-bit.
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

template <int D>
struct Geo {
    static constexpr int LPT = D / 8;        // lanes per token row (16 B per lane)
    static constexpr int TPI = 64 / LPT;     // token rows per wave load instruction
    static constexpr int NI = 8;             // K (and V) load instructions per wave tile
    static constexpr int WT = NI * TPI;      // tokens per wave tile: 64 (D=64), 32 (D=128)
    static constexpr int WAVES = 4;
    static constexpr int SPLIT = WAVES * WT; // tokens per workgroup: 256 / 128
};

template <int D, int G>
__global__ __launch_bounds__(256) void paged_decode_split_kernel(const DecodeArgs a) {
    using geo = Geo<D>;
    constexpr int LPT = geo::LPT, TPI = geo::TPI, NI = geo::NI, WT = geo::WT, SPLIT = geo::SPLIT;
    __shared__ __attribute__((aligned(16))) float lds_acc[geo::WAVES][G][D];
    __shared__ float lds_ml[geo::WAVES][G][2];

    const int split = blockIdx.x, kh = blockIdx.y, b = blockIdx.z;
    const int ctx = a.context_lens[b];
    const int wg_tok0 = split * SPLIT;
    if (wg_tok0 >= ctx) return;                                   // whole workgroup, before any barrier

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane % LPT;                                     // 16-byte chunk of the row
    const int s = lane / LPT;                                     // token slot inside one instruction
    const int tok0 = wg_tok0 + wave * WT;

    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    if (tok0 < ctx) {                                             // wave-uniform
        // one block per wave tile: WT divides block_size
        const int blk = tok0 / a.block_size;
        const int bid = a.block_tables[b * a.bt_row_stride + blk];
        const int off0 = tok0 - blk * a.block_size;
        const int64_t row = (int64_t)a.kvh * D;                   // elements per token row (all kv heads)
        const int64_t base = ((int64_t)bid * a.block_size + off0 + s) * row + (int64_t)kh * D + j * 8;
        const uint16_t* kp = a.k_cache + base;
        const uint16_t* vp = a.v_cache + base;

        u32x4 kreg[NI], vreg[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool ok = tok0 + i * TPI + s < ctx;
            kreg[i] = ok ? *reinterpret_cast<const u32x4*>(kp + (int64_t)i * TPI * row) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool ok = tok0 + i * TPI + s < ctx;
            vreg[i] = ok ? *reinterpret_cast<const u32x4*>(vp + (int64_t)i * TPI * row) : u32x4{0, 0, 0, 0};
        }
        // this lane's 16-byte chunk of each of the group's G query rows
        u32x4 qreg[G];
        const uint16_t* qp = a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D + j * 8;
#pragma unroll
        for (int g = 0; g < G; ++g) qreg[g] = *reinterpret_cast<const u32x4*>(qp + g * D);

        // ---- scores: s[i][g] = scale*log2e * <q_g, k_token(i,s)>, -inf for masked tokens
        float sc[NI][G];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool ok = tok0 + i * TPI + s < ctx;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) d = dot2_bf16(kreg[i][w], qreg[g][w], d);
                d = group_sum<LPT>(d);
                sc[i][g] = ok ? d * a.scale_log2 : -INFINITY;
            }
        }
        // ---- tile max per head (over NI instructions locally, then over the token-slot lanes)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = sc[0][g];
#pragma unroll
            for (int i = 1; i < NI; ++i) mx = fmaxf(mx, sc[i][g]);
            m[g] = slot_max<LPT>(mx);                             // finite: token tok0 is valid
        }
        // ---- p = 2^(s - m); PV
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float vf[8];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                vf[2 * w] = bf16_lo(vreg[i][w]);
                vf[2 * w + 1] = bf16_hi(vreg[i][w]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float p = fast_exp2(sc[i][g] - m[g]);
                l[g] += p;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(p, vf[e], acc[g][e]);
            }
        }
        // ---- fold the token-slot lanes (lanes sharing j)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            l[g] = slot_sum<LPT>(l[g]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = slot_sum<LPT>(acc[g][e]);
        }
    }

    // ---- combine the 4 waves through LDS
    if (s == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            *reinterpret_cast<f32x4*>(&lds_acc[wave][g][j * 8]) = f32x4{acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
            *reinterpret_cast<f32x4*>(&lds_acc[wave][g][j * 8 + 4]) = f32x4{acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
        }
        if (j == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                lds_ml[wave][g][0] = m[g];
                lds_ml[wave][g][1] = l[g];
            }
        }
    }
    __syncthreads();
    const int H = a.h;
    for (int idx = threadIdx.x; idx < G * D; idx += 256) {
        const int g = idx / D, d = idx - g * D;
        float mw[geo::WAVES], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < geo::WAVES; ++w) {
            mw[w] = lds_ml[w][g][0];
            M = fmaxf(M, mw[w]);
        }
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int w = 0; w < geo::WAVES; ++w) {
            const float f = fast_exp2(mw[w] - M);                 // 0 for waves past the context (m = -inf)
            o = fmaf(lds_acc[w][g][d], f, o);
            L = fmaf(lds_ml[w][g][1], f, L);
        }
        const int64_t part = ((int64_t)b * H + kh * G + g) * a.num_splits + split;
        a.ws_acc[part * D + d] = o;
        if (d == 0) {
            a.ws_ml[part * 2] = M;
            a.ws_ml[part * 2 + 1] = L;
        }
    }
}

// grid (B*H), D threads: merge the live splits of one (sequence, head) and normalise.
template <int D>
__global__ __launch_bounds__(D) void paged_decode_combine_kernel(const DecodeArgs a) {
    constexpr int SPLIT = Geo<D>::SPLIT;
    const int bh = blockIdx.x;
    const int b = bh / a.h;
    const int d = threadIdx.x;
    const int ctx = a.context_lens[b];
    const int n = (ctx + SPLIT - 1) / SPLIT;                       // live splits; 0 for padding rows
    const float* ml = a.ws_ml + (int64_t)bh * a.num_splits * 2;
    const float* pa = a.ws_acc + (int64_t)bh * a.num_splits * D + d;
    float M = -INFINITY;
    for (int i = 0; i < n; ++i) M = fmaxf(M, ml[2 * i]);
    float o = 0.f, L = 0.f;
    for (int i = 0; i < n; ++i) {
        const float f = fast_exp2(ml[2 * i] - M);
        o = fmaf(pa[(int64_t)i * D], f, o);
        L = fmaf(ml[2 * i + 1], f, L);
    }
    const float r = n > 0 ? o / L : 0.f;                           // ctx == 0 -> zeros (oracle behaviour)
    if (a.out_f32) {
        reinterpret_cast<float*>(a.out)[(int64_t)bh * D + d] = r;
    } else {
        reinterpret_cast<__bf16*>(a.out)[(int64_t)bh * D + d] = (__bf16)r;
    }
}

template <int D, int G>
int launch_dg(const DecodeArgs& a, hipStream_t stream) {
    dim3 grid(a.num_splits, a.kvh, a.batch);
    hipLaunchKernelGGL((paged_decode_split_kernel<D, G>), grid, dim3(256), 0, stream, a);
    int rc = check_launch("paged_decode_split");
    if (rc) return rc;
    hipLaunchKernelGGL((paged_decode_combine_kernel<D>), dim3(a.batch * a.h), dim3(D), 0, stream, a);
    return check_launch("paged_decode_combine");
}

template <int D>
int launch_d(const DecodeArgs& a, int g, hipStream_t stream) {
    switch (g) {
        case 1: return launch_dg<D, 1>(a, stream);
        case 2: return launch_dg<D, 2>(a, stream);
        case 3: return launch_dg<D, 3>(a, stream);
        case 4: return launch_dg<D, 4>(a, stream);
        case 5: return launch_dg<D, 5>(a, stream);
        case 6: return launch_dg<D, 6>(a, stream);
        case 7: return launch_dg<D, 7>(a, stream);
        case 8: return launch_dg<D, 8>(a, stream);
    }
    set_error("paged_decode: group size %d not in 1..8", g);
    return -2;
}

}  // namespace

int decode_split_tokens(int hd) { return hd == 64 ? Geo<64>::SPLIT : Geo<128>::SPLIT; }

int launch_paged_decode(const DecodeArgs& a, hipStream_t stream) {
    if (a.batch == 0) return 0;
    const int g = a.h / a.kvh;
    return a.hd == 64 ? launch_d<64>(a, g, stream) : launch_d<128>(a, g, stream);
}

}  // namespace nvh
