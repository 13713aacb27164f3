// Paged single-query (decode) attention for gfx950.
// Replaces flash_attn_with_kvcache at nanovllm/layers/attention.py:99-101
// (oracle: nanovllm/layers/attention_sdpa.py:122-182).
//
// HBM-bound: every K/V byte of the live context is read exactly once per kv head and shared by the
// G = H/KVH query heads of the group.  Algorithmic bytes per launch:
//     sum_b 2*ctx_b*KVH*D*2  +  2*B*H*D*2 (q in, o out)  +  4*(sum_b ceil(ctx_b/bs) + B)
// Split-KV partials and the combine pass are overhead, not algorithmic.
//
// Structure (flash-decoding, wave64, no MFMA: M = G <= 8 rows is not a dense contraction):
//   split kernel   grid (num_splits, KVH, B), 256 threads = 4 waves.  A workgroup owns 256 consecutive
//                  tokens of one (sequence, kv head); each wave owns a tile of 64 of them and issues ALL
//                  its loads up front (K: 8 or 16 KiB by LDS-DMA, V: the same again into VGPRs).
//     K path       global_load_lds_dwordx4 (HBM -> LDS, no VGPRs): one instruction = 1 KiB = 64/LPT whole
//                  token rows, fully coalesced.  The LDS image is row-major [64 tokens][D bf16]; the
//                  16-byte chunk order inside a row is XOR-swizzled on the SOURCE address so that the
//                  per-token row reads below are bank-conflict free.
//     QK^T         lane = token.  Each lane reads its own K row from LDS chunk by chunk (ds_read_b128) and
//                  the group's q chunks as LDS broadcasts; v_dot2c_f32_bf16 accumulates in fp32.  No
//                  cross-lane reduction, one exp2 per (token, head).
//     softmax      tile max per head by a wavefront reduction (DPP inside 16-lane rows, permlane swaps
//                  across rows); p = 2^(s - m) goes to an LDS tile [token][head].
//     PV           lane = (token slot, 16-byte dim chunk): V rows are still in the registers the coalesced
//                  loads filled; p is an LDS broadcast read; acc[G][8] fp32 FMAs per lane.
//     epilogue     token slots folded by DPP + LDS, the 4 waves merged with their own maxima, one
//                  (max, sum, acc[G][D]) partial per workgroup written to the workspace.
//   combine kernel one thread per output element: loads every live partial in one round trip, merges,
//                  normalises, rounds to bf16 (or fp32 for parity checks).
// Grids depend only on static shapes; splits past context_lens[b] exit at once (graph-safe), block-table
// entries past ceil(ctx/bs) are never read, cache offsets are 64-bit.
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

constexpr int WAVES = 4;
constexpr int WT = 64;                  // tokens per wave tile (one per lane in the QK^T phase)
constexpr int SPLIT = WAVES * WT;       // tokens per workgroup

template <int D>
struct Geo {
    static constexpr int LPT = D / 8;           // lanes per token row in a coalesced load (16 B per lane)
    static constexpr int TPI = 64 / LPT;        // token rows per wave load instruction
    static constexpr int NI = WT / TPI;         // load instructions per tile (K and V each): 8 / 16
    static constexpr int ROWB = D * 2;          // bytes per token row in the LDS K image
    static constexpr int ROWS = 4;              // partial sets a wave leaves in LDS (one per 16-lane row)
};

// swizzle of the 16-byte chunk index inside token row T (conflict-free per-token ds_read_b128)
template <int LPT>
__device__ __forceinline__ int chunk_swizzle(int T) {
    return LPT == 8 ? ((T >> 1) & 7) : (T & 15);
}

__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, pair_in_row<1>(x));
    x = fmaxf(x, pair_in_row<2>(x));
    x = fmaxf(x, pair_in_row<4>(x));
    x = fmaxf(x, pair_in_row<8>(x));
    {   // rows 1,3 <-> rows 0,2
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {   // lanes 32..63 <-> lanes 0..31
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return x;
}

template <int D, int G>
__global__ __launch_bounds__(256) void paged_decode_split_kernel(const DecodeArgs a) {
    using geo = Geo<D>;
    constexpr int LPT = geo::LPT, TPI = geo::TPI, NI = geo::NI, ROWB = geo::ROWB, ROWS = geo::ROWS;
    constexpr int K_BYTES = WT * ROWB;                       // per wave: 8 KiB (D=64) / 16 KiB (D=128)
    constexpr int FIN_BYTES = ROWS * G * D * 4;              // per wave, aliases its K image
    static_assert(FIN_BYTES <= K_BYTES, "epilogue partials must fit in the wave's K image");
    constexpr int P_BYTES = WT * 8 * 4;                      // per wave: p[token][8 heads] fp32
    constexpr int QN = LPT * G;                              // 16-byte q chunks of the group: [chunk][head]
    constexpr int QI = (QN + 63) / 64;                       // LDS-DMA instructions to fetch them
    constexpr int Q_BYTES = QI * 1024;                       // per wave (each wave keeps its own copy: no barrier)
    constexpr int WAVE_BYTES = K_BYTES + P_BYTES + Q_BYTES;
    // one LDS array (a second __shared__ object next to LDS-DMA staging can force vmcnt(0) waits)
    __shared__ __attribute__((aligned(16))) unsigned char lds[WAVES * WAVE_BYTES + WAVES * (8 * 4 + ROWS * 8 * 4)];
    float* const lds_m = reinterpret_cast<float*>(lds + WAVES * WAVE_BYTES);     // [WAVES][8]
    float* const lds_l = lds_m + WAVES * 8;                                       // [WAVES][ROWS][8]

    const int split = blockIdx.x, kh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg_tok0 = split * SPLIT;
    const int tok0 = wg_tok0 + wave * WT;
    // context length and this tile's block id are fetched together (the id is only USED if the tile is live)
    int blk = tok0 / a.block_size;                                // WT divides block_size: one block per tile
    blk = blk < a.max_blocks ? blk : a.max_blocks - 1;
    const int ctx = a.context_lens[b];
    const int bid = a.block_tables[(int64_t)b * a.bt_row_stride + blk];
    asm volatile("" ::"s"(bid), "s"(ctx));                        // both scalar loads in flight before the branch
    if (wg_tok0 >= ctx) return;                                   // whole workgroup, before any barrier

    const int j = lane % LPT;                                     // 16-byte chunk of a row (load / PV mapping)
    const int sl = lane / LPT;                                    // token slot inside one load instruction
    const int n_live = ctx - tok0;                                // live tokens of this wave's tile (may be <= 0)
    const bool active = n_live > 0;                               // wave-uniform
    unsigned char* const lds_k = lds + wave * WAVE_BYTES;
    float* const lds_p = reinterpret_cast<float*>(lds_k + K_BYTES);
    unsigned char* const lds_q = lds_k + K_BYTES + P_BYTES;

    // ---- issue every load of the tile, branch-free: q and K by LDS-DMA (in that order), then V into VGPRs.
    // Rows past the live range are clamped to the tile's last live row (finite data); their scores are
    // masked and their probabilities are exactly 0, so what they load never reaches the output.
    u32x4 vreg[NI];
    if (active) {
        const int off0 = tok0 - (tok0 / a.block_size) * a.block_size;
        const int64_t row = (int64_t)a.kvh * D;                    // elements per token (all kv heads)
        const int64_t base = ((int64_t)bid * a.block_size + off0) * row + (int64_t)kh * D;
        const uint16_t* kp = a.k_cache + base;
        const uint16_t* vp = a.v_cache + base;
        const uint16_t* qp = a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D;
        const int last = n_live - 1;
#pragma unroll
        for (int i = 0; i < QI; ++i) {                             // LDS entry e = c*G + g  <-  q[head g][chunk c]
            int e = i * 64 + lane;
            e = e < QN ? e : QN - 1;
            const int c = e / G, g = e - c * G;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp + g * D + c * 8),
                                             (__attribute__((address_space(3))) void*)(lds_q + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int T = i * TPI + sl;                            // token of this lane in instruction i
            const int Tc = T < last ? T : last;
            const int src_chunk = j ^ chunk_swizzle<LPT>(T);       // LDS position j of row T holds this chunk
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp + Tc * row + src_chunk * 8),
                                             (__attribute__((address_space(3))) void*)(lds_k + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int T = i * TPI + sl;
            const int Tc = T < last ? T : last;
            vreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vp + Tc * row + j * 8));
        }
        // q and K have landed once at most the NI V loads are still outstanding (vmcnt retires in order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    }

    float acc[G][8], lsum[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        lsum[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    if (active) {
        // ---- QK^T: lane = token `lane` of the tile
        float sc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) sc[g] = 0.f;
        const int swz = chunk_swizzle<LPT>(lane);
        // two chunks per iteration: enough LDS reads in flight to cover their latency, few enough live
        // q registers (2 x G x 4) that the tile keeps >= 2 waves per SIMD
#pragma unroll 2
        for (int c = 0; c < LPT; ++c) {
            const u32x4 kc = *reinterpret_cast<const u32x4*>(lds_k + lane * ROWB + ((c ^ swz) * 16));
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 qc = *reinterpret_cast<const u32x4*>(lds_q + (c * G + g) * 16);   // broadcast
#pragma unroll
                for (int w = 0; w < 4; ++w) sc[g] = dot2_bf16(kc[w], qc[w], sc[g]);
            }
        }
        // ---- softmax numerators against the tile max (log2 domain)
        const bool live = lane < n_live;
        float pr[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) pr[g] = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float s = live ? sc[g] * a.scale_log2 : -INFINITY;
            const float m = wave_max(s);                           // finite: token 0 of the tile is live
            pr[g] = fast_exp2(s - m);
            if (lane == 0) lds_m[wave * 8 + g] = m;
        }
        *reinterpret_cast<f32x4*>(lds_p + lane * 8) = f32x4{pr[0], pr[1], pr[2], pr[3]};
        if constexpr (G > 4) *reinterpret_cast<f32x4*>(lds_p + lane * 8 + 4) = f32x4{pr[4], pr[5], pr[6], pr[7]};
        // (same wave writes and reads lds_p; the compiler orders the LDS accesses, no barrier needed)

        // ---- PV: lane = (token slot sl, dim chunk j).  p of the next instruction's token is fetched one
        // iteration ahead; the scheduling fences keep the compiler from hoisting all NI fetches at once.
        auto load_p = [&](int i, float (&p)[8]) {
            const int T = i * TPI + sl;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(lds_p + T * 8);
            p[0] = p0[0]; p[1] = p0[1]; p[2] = p0[2]; p[3] = p0[3];
            if constexpr (G > 4) {
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(lds_p + T * 8 + 4);
                p[4] = p1[0]; p[5] = p1[1]; p[6] = p1[2]; p[7] = p1[3];
            }
        };
        float pcur[8], pnxt[8];
        load_p(0, pcur);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) load_p(i + 1, pnxt);
            float vf[8];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                vf[2 * w] = bf16_lo(vreg[i][w]);
                vf[2 * w + 1] = bf16_hi(vreg[i][w]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                lsum[g] += pcur[g];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(pcur[g], vf[e], acc[g][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) pcur[g] = pnxt[g];
        }
        // ---- fold token slots: inside a 16-lane row by DPP (D=64 only: two slots per row), rows via LDS
        if constexpr (LPT == 8) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                lsum[g] += pair_in_row<8>(lsum[g]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] += pair_in_row<8>(acc[g][e]);
            }
        }
        const int rowi = lane >> 4;
        float* const fin = reinterpret_cast<float*>(lds_k) + rowi * (G * D);          // aliases this wave's K image
        if ((lane & 15) < LPT) {                                   // one representative lane per (row, chunk)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                *reinterpret_cast<f32x4*>(fin + g * D + j * 8) = f32x4{acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
                *reinterpret_cast<f32x4*>(fin + g * D + j * 8 + 4) = f32x4{acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
            }
            if (j == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) lds_l[(wave * ROWS + rowi) * 8 + g] = lsum[g];
            }
        }
    }
    __syncthreads();

    // ---- merge rows and waves, write the workgroup's partial
    const int n_waves = min(WAVES, (ctx - wg_tok0 + WT - 1) / WT); // live waves of this workgroup
    for (int idx = tid; idx < G * D; idx += 256) {
        const int g = idx / D, d = idx - g * D;
        float M = -INFINITY;
        for (int w = 0; w < n_waves; ++w) M = fmaxf(M, lds_m[w * 8 + g]);
        float o = 0.f, L = 0.f;
        for (int w = 0; w < n_waves; ++w) {
            const float f = fast_exp2(lds_m[w * 8 + g] - M);
            const float* fw = reinterpret_cast<const float*>(lds + w * WAVE_BYTES) + g * D + d;
            float so = 0.f, sl_ = 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                so += fw[r * (G * D)];
                sl_ += lds_l[(w * ROWS + r) * 8 + g];
            }
            o = fmaf(so, f, o);
            L = fmaf(sl_, f, L);
        }
        const int64_t part = ((int64_t)b * a.h + kh * G + g) * a.num_splits + split;
        a.ws_acc[part * D + d] = o;
        if (d == 0) {
            a.ws_ml[part * 2] = M;
            a.ws_ml[part * 2 + 1] = L;
        }
    }
}

// One thread per output element (b, h, d): all live partials are requested before any is used.
template <int D>
__global__ __launch_bounds__(256) void paged_decode_combine_kernel(const DecodeArgs a) {
    constexpr int CH = 16;                                         // partials per unrolled round trip
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.batch * a.h * D) return;
    const int64_t bh = idx / D;
    const int d = (int)(idx - bh * D);
    const int b = (int)(bh / a.h);
    const int ctx = a.context_lens[b];
    const int n = (ctx + SPLIT - 1) / SPLIT;                       // live splits; 0 for padding rows
    const float* ml = a.ws_ml + bh * a.num_splits * 2;
    const float* pa = a.ws_acc + bh * a.num_splits * D + d;
    float M = -INFINITY, o = 0.f, L = 0.f;
    for (int i0 = 0; i0 < n; i0 += CH) {
        float mv[CH], lv[CH], av[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool ok = i0 + i < n;
            mv[i] = ok ? ml[2 * (i0 + i)] : -INFINITY;
            lv[i] = ok ? ml[2 * (i0 + i) + 1] : 0.f;
            av[i] = ok ? pa[(int64_t)(i0 + i) * D] : 0.f;
        }
        float Mc = M;
#pragma unroll
        for (int i = 0; i < CH; ++i) Mc = fmaxf(Mc, mv[i]);
        const float fo = fast_exp2(M - Mc);                        // M = -inf on the first chunk -> 0
        o *= fo;
        L *= fo;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float f = fast_exp2(mv[i] - Mc);
            o = fmaf(av[i], f, o);
            L = fmaf(lv[i], f, L);
        }
        M = Mc;
    }
    const float r = n > 0 ? o / L : 0.f;                           // ctx == 0 -> zeros (oracle behaviour)
    if (a.out_f32) reinterpret_cast<float*>(a.out)[idx] = r;
    else reinterpret_cast<__bf16*>(a.out)[idx] = (__bf16)r;
}

template <int D, int G>
int launch_dg(const DecodeArgs& a, hipStream_t stream) {
    dim3 grid(a.num_splits, a.kvh, a.batch);
    hipLaunchKernelGGL((paged_decode_split_kernel<D, G>), grid, dim3(256), 0, stream, a);
    int rc = check_launch("paged_decode_split");
    if (rc) return rc;
    const int64_t total = (int64_t)a.batch * a.h * D;
    hipLaunchKernelGGL((paged_decode_combine_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
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

int decode_split_tokens(int hd) { (void)hd; return SPLIT; }

int launch_paged_decode(const DecodeArgs& a, hipStream_t stream) {
    if (a.batch == 0) return 0;
    const int g = a.h / a.kvh;
    return a.hd == 64 ? launch_d<64>(a, g, stream) : launch_d<128>(a, g, stream);
}

}  // namespace nvh
