// Paged single-query (decode) attention for gfx950.
// Replaces flash_attn_with_kvcache at nanovllm/layers/attention.py:99-101
// (oracle: nanovllm/layers/attention_sdpa.py:122-182).
//
// HBM-bound: every K/V byte of the live context is read exactly once per kv head and shared by the
// G = H/KVH query heads of the group.  Algorithmic bytes per launch:
//     sum_b 2*ctx_b*KVH*D*2  +  2*B*H*D*2 (q in, o out)  +  4*(sum_b ceil(ctx_b/bs) + B)
// Split-KV partials and the combine pass are overhead, not algorithmic.
//
// Structure (flash-decoding, wave64, VALU only: M = G <= 8 query rows is not a dense contraction):
//   split kernel   grid (num_splits, KVH, B), 512 threads = 8 waves; a workgroup owns SPLIT consecutive
//                  tokens of one (sequence, kv head), each wave a tile of WT = SPLIT/8 of them.
//     loads        a token row (D bf16) is read by LPT = D/8 lanes x 16 B, so one global_load_dwordx4 wave
//                  instruction moves 1 KiB = TPI = 64/LPT whole rows, fully coalesced.  A wave issues its
//                  NI K loads and NI V loads back to back (8 KiB in flight per wave, ~3 waves per SIMD) and
//                  the compiler's counted vmcnt waits let it start on token slot 0 as soon as that load
//                  lands.  K/V go straight to VGPRs: read once, no reuse, so no LDS staging (measured: the
//                  LDS-DMA + lane-per-token variant lost to exposed LDS latency).
//     QK^T         the lane's 16-byte chunk of each of the G query rows stays in registers (q is stationary);
//                  v_dot2c_f32_bf16 partial dots, then a DPP butterfly over the LPT lanes of the row.
//     softmax      exp2 domain; tile max per head = local max over the NI slots + DPP/permlane all-reduce
//                  over the token-slot lanes (wavefront-level reductions, no LDS).
//     PV           packed fp32 FMAs into acc[G][8] per lane.
//     fold         token-slot lanes folded by one DPP step and a permlane reduce-scatter (one swap + one
//                  add per two values); the 8 waves merge through a small LDS tile and the workgroup
//                  writes one (max, sum, acc[G][D]) partial.
//   combine kernel one thread per output element: every live partial requested in one round trip, merged,
//                  normalised, rounded to bf16 (or kept fp32 for parity checks).
// Grids depend only on static shapes; splits past context_lens[b] exit at once (graph-safe), block-table
// entries past ceil(ctx/bs) are never used, cache offsets are 64-bit.
#include <stdlib.h>

#ifndef NVH_DMA_AUX
#define NVH_DMA_AUX 2        // cache policy of the once-read LDS-DMA streams (weights, K/V): 2 = nt, 0 = default.
                             // nt measured -4.7 % on the decode step, -0.6 us per attention call (same box A/B, round 1)
#endif
#include "common.h"
#include "kernels.h"
#include "decode_chunked.h"

#ifndef NVH_D128_HALF_RULE
#define NVH_D128_HALF_RULE(chunks) ((chunks) <= 8)   // the chunk counts at which head_dim 128 takes 64-token passes by default.  Measured (same box,
                             // B = 32, 64- against 128-token passes, ctx 1034 / 1536 / 2048 / 3072): 7/1/128 (8 chunks) 8.02 / 8.81 / 9.95 / 12.76 against
                             // 8.25 / 9.27 / 10.23 / 12.74 us; 28/4/128 (2 chunks) -3 % everywhere; 16/8/128 (1 chunk) -0.2 .. -1 %; 7/1/128 at B = 64 (4 chunks)
                             // -1.3 %, B = 16 (16 chunks) +1.3 %, B = 8 (32 chunks) +9 %: with 128-token passes twelve passes over eight chunks leave half the
                             // workgroups with twice the tokens of the others; with many chunks the finer passes only add per-pass work
#endif
#ifndef NVH_D128_WAVES
#define NVH_D128_WAVES 4     // waves per workgroup of the chunked kernel at head_dim 128 when the caller does not choose (4 or 8).
                             // With 4-byte records 8 waves won (7/1/128 ctx 1536: 11.1 vs 12.1 us); since the records move as 16-byte items
                             // 4 waves (32-token tiles, the k = 32 MFMA) win everywhere: 7/1/128 9.2 vs 9.6, 28/4/128 19.8 vs 20.6, 16/8/128 33.3 vs 34.0
#endif

namespace nvh {

namespace {

constexpr int WAVES = 8;                        // VALU kernel: waves per workgroup
constexpr int NI = 4;                           // VALU kernel: K (and V) load instructions per wave tile

template <int D>
struct Geo {
    static constexpr int LPT = D / 8;           // lanes per token row (16 B per lane)
    static constexpr int TPI = 64 / LPT;        // token rows per wave load instruction
    static constexpr int WT = NI * TPI;         // tokens per wave tile: 32 (D=64) / 16 (D=128)
    static constexpr int SPLIT = WAVES * WT;    // tokens per workgroup: 256 / 128
};

template <int D, int G>
__global__ __launch_bounds__(WAVES * 64) void paged_decode_split_valu_kernel(const DecodeArgs a) {
    using geo = Geo<D>;
    constexpr int LPT = geo::LPT, TPI = geo::TPI, WT = geo::WT, SPLIT = geo::SPLIT;
    constexpr int NV = G * 8;                                 // accumulator values per lane (multiple of 4)
    constexpr int NQ = NV / 4;                                // values per lane after the reduce-scatter
    constexpr int NH = 2 * NQ;
    __shared__ __attribute__((aligned(16))) float lds_fin[WAVES][G * D];
    __shared__ float lds_ml[WAVES][2][8];                     // [wave][max | sum][head]

    const int split = blockIdx.x, kh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg_tok0 = split * SPLIT;
    const int tok0 = wg_tok0 + wave * WT;
    NVH_STAMP(0);
    // context length and this tile's block id are fetched together (the id is only USED if the tile is live)
    int blk = tok0 / a.block_size;                            // WT divides block_size: one block per tile
    blk = blk < a.max_blocks ? blk : a.max_blocks - 1;
    const int ctx = a.context_lens[b];
    const int bid = a.block_tables[(int64_t)b * a.bt_row_stride + blk];
    asm volatile("" ::"s"(bid), "s"(ctx));                    // both scalar loads in flight before the branch
    if (wg_tok0 >= ctx) return;                               // whole workgroup, before any barrier
    NVH_STAMP(1);

    const int j = lane % LPT;                                 // 16-byte chunk of the row owned by this lane
    const int sl = lane / LPT;                                // token slot inside one load instruction
    const int n_live = ctx - tok0;                            // live tokens of this wave's tile (may be <= 0)

    if (n_live > 0) {                                         // wave-uniform
        // ---- loads, branch-free: rows past the live range are clamped to the tile's last live row (finite
        // data); their scores are masked to -inf so their probabilities are exactly 0.
        const int off0 = tok0 - (tok0 / a.block_size) * a.block_size;
        const int64_t row = (int64_t)a.kvh * D;               // elements per token (all kv heads)
        const int64_t base = ((int64_t)bid * a.block_size + off0) * row + (int64_t)kh * D + j * 8;
        const uint16_t* kp = a.k_cache + base;
        const uint16_t* vp = a.v_cache + base;
        const int last = n_live - 1;
        u32x4 kreg[NI], vreg[NI], qreg[G];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int T = i * TPI + sl;
            kreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kp + (T < last ? T : last) * row));
        }
        {
            const uint16_t* qp = a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D + j * 8;
#pragma unroll
            for (int g = 0; g < G; ++g) qreg[g] = *reinterpret_cast<const u32x4*>(qp + g * D);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int T = i * TPI + sl;
            vreg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vp + (T < last ? T : last) * row));
        }
        __builtin_amdgcn_sched_barrier(0);                    // every load issued before any of the math below
        NVH_STAMP(2);

        // ---- scores: sc[i][g] = scale*log2e * <q_g, k_token(i,sl)>, -inf for masked tokens
        float sc[NI][G];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool live = i * TPI + sl < n_live;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) d = dot2_bf16(kreg[i][w], qreg[g][w], d);
                d = group_sum<LPT>(d);
                sc[i][g] = live ? d * a.scale_log2 : -INFINITY;
            }
        }
        NVH_STAMP(3);
        // ---- tile max per head: local over the NI slots, then over the token-slot lanes
        float m[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = sc[0][g];
#pragma unroll
            for (int i = 1; i < NI; ++i) mx = fmaxf(mx, sc[i][g]);
            m[g] = slot_max<LPT>(mx);                         // finite: token 0 of the tile is live
        }
        NVH_STAMP(4);
        // ---- p = 2^(s - m); PV
        float acc[NV], lsum[8];
#pragma unroll
        for (int x = 0; x < NV; ++x) acc[x] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) lsum[g] = 0.f;
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
                lsum[g] += p;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g * 8 + e] = fmaf(p, vf[e], acc[g * 8 + e]);
            }
        }
        NVH_STAMP(5);
        // ---- fold the token-slot lanes.  D=64: two slots share a 16-lane row -> one DPP step first.
        if constexpr (LPT == 8) {
#pragma unroll
            for (int x = 0; x < NV; ++x) acc[x] += pair_in_row<8>(acc[x]);
#pragma unroll
            for (int g = 0; g < G; ++g) lsum[g] += pair_in_row<8>(lsum[g]);
        }
        // reduce-scatter over the 4 rows: afterwards row r of the wave holds accumulator values
        // r*NQ .. r*NQ+NQ-1 and heads 2r, 2r+1 of lsum, each summed over all rows.
        float h32[NH], fin[NQ];
#pragma unroll
        for (int x = 0; x < NH; ++x) h32[x] = fold32(acc[x], acc[x + NH]);
#pragma unroll
        for (int x = 0; x < NQ; ++x) fin[x] = fold16(h32[x], h32[x + NQ]);
        float l32[4], lfin[2];
#pragma unroll
        for (int x = 0; x < 4; ++x) l32[x] = fold32(lsum[x], lsum[x + 4]);
#pragma unroll
        for (int x = 0; x < 2; ++x) lfin[x] = fold16(l32[x], l32[x + 2]);

        const int rowi = lane >> 4;
        if ((lane & 15) < LPT) {                              // one representative lane per (row, chunk)
#pragma unroll
            for (int x = 0; x < NQ; ++x) {
                const int v = rowi * NQ + x;                  // flat accumulator index g*8 + e
                lds_fin[wave][(v >> 3) * D + j * 8 + (v & 7)] = fin[x];
            }
            if (j == 0) {
                lds_ml[wave][1][2 * rowi] = lfin[0];
                lds_ml[wave][1][2 * rowi + 1] = lfin[1];
                if (rowi == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) lds_ml[wave][0][g] = m[g];
                }
            }
        }
        NVH_STAMP(6);
    }
    __syncthreads();

    // ---- merge the live waves with their own maxima, write the workgroup's partial
    const int n_waves = min(WAVES, (ctx - wg_tok0 + WT - 1) / WT);
    for (int idx = tid; idx < G * D; idx += WAVES * 64) {
        const int g = idx / D, d = idx - g * D;
        float M = -INFINITY;
        for (int w = 0; w < n_waves; ++w) M = fmaxf(M, lds_ml[w][0][g]);
        float o = 0.f, L = 0.f;
        for (int w = 0; w < n_waves; ++w) {
            const float f = fast_exp2(lds_ml[w][0][g] - M);
            o = fmaf(lds_fin[w][idx], f, o);
            L = fmaf(lds_ml[w][1][g], f, L);
        }
        const int64_t part = ((int64_t)b * a.h + kh * G + g) * a.num_splits + split;
        a.ws_acc[part * D + d] = o;
        if (d == 0) {
            a.ws_ml[part * 2] = M;
            a.ws_ml[part * 2 + 1] = L;
        }
    }
    NVH_STAMP(7);
}

// =====================================================================================================
// MFMA split kernel (default).  Under GQA the G query rows that share a kv head form a dense
// [G x D].[D x T] contraction per (sequence, kv head): G is padded to the 16 columns of
// v_mfma_f32_16x16x32_bf16 and the tile costs 8 + 16 MFMAs per 64 tokens instead of ~2000 VALU issues
// (measured: the VALU kernel below is issue-bound at 11.6-13.9 us per layer).
//   grid (num_splits, KVH, B), 256 threads = 4 waves; a workgroup owns SPLIT tokens, a wave a tile of WT.
//   K, V and q are staged through LDS by LDS-DMA (global_load_lds_dwordx4): every wave instruction moves
//   1 KiB = whole token rows, fully coalesced, no VGPR round trip.  Images are row-major; the 16-byte chunk
//   order inside a row is XOR-swizzled on the SOURCE address so the MFMA operand reads are conflict free.
//   S^T = K Q^T   A = K rows (ds_read_b128), B = the group's q rows.  C: lane l, reg r holds
//                 S^T[token 4(l>>4)+r][head l&15]: per-head max/sum = local ops + 2 permlane steps.
//   O^T = V^T P^T B = P^T straight from the S^T accumulators (split hi + lo bf16, so P keeps ~16 mantissa
//                 bits and the 1e-3 parity bar holds at |o| ~ 3); A = V^T by ds_read_b64_tr_b16.
//   The 4 waves merge through LDS (aliasing the wave's K image) into one partial per workgroup.

template <int D>
__global__ __launch_bounds__(MW * 64) void paged_decode_split_mfma_kernel(const DecodeArgs a, const int G) {
    using geo = MGeo<D>;
    constexpr int LPT = geo::LPT, TPI = geo::TPI, WT = geo::WT, NI = geo::NI, ROWB = geo::ROWB, NT = geo::NT;
    constexpr int NHALF = geo::NHALF, STEPS = geo::STEPS, DT = geo::DT, QI = geo::QI, IMG = geo::IMG;
    constexpr int WAVES = MW, SPLIT = geo::SPLIT;
    static_assert(16 * D * 4 <= IMG, "merge tile must fit in the wave's K image");
    // one LDS array (a second __shared__ object beside LDS-DMA staging can force vmcnt(0) waits)
    __shared__ __attribute__((aligned(16))) unsigned char lds[MW * geo::WAVE_BYTES + QI * 1024 + MW * 2 * 16 * 4];
    // every wave DMAs the same q bytes into this one image (identical writes; each wave waits on its own DMA)
    unsigned char* const lds_q = lds + MW * geo::WAVE_BYTES;
    float* const lds_ml = reinterpret_cast<float*>(lds_q + QI * 1024);              // [wave][max | sum][16 heads]

    // grid (KVH*B, num_splits): workgroups are dispatched x-fastest, so every sequence's split 0 goes first and the
    // splits past the live contexts (graph replay uses a fixed-width block table) are dispatched last and retire at once
    const int split = blockIdx.y, kh = blockIdx.x % a.kvh, b = blockIdx.x / a.kvh;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg_tok0 = split * SPLIT;
    const int tok0 = wg_tok0 + wave * WT;
    NVH_STAMP(0);
    int blk = tok0 / a.block_size;                            // WT divides block_size: one block per tile
    blk = blk < a.max_blocks ? blk : a.max_blocks - 1;
    const int ctx = a.context_lens[b];
    const int bid = a.block_tables[(int64_t)b * a.bt_row_stride + blk];
    asm volatile("" ::"s"(bid), "s"(ctx));                    // both scalar loads in flight before the branch
    if (wg_tok0 >= ctx) return;                               // whole workgroup, before any barrier
    NVH_STAMP(1);

    const int n_live = ctx - tok0;                            // live tokens of this wave's tile (may be <= 0)
    unsigned char* const lds_k = lds + wave * geo::WAVE_BYTES;
    unsigned char* const lds_v = lds_k + IMG;
    const int lq = lane & 15;                                 // head column of the MFMA tiles
    const int lg = lane >> 4;                                 // lane group: k-block of operands / row block of C

    if (n_live > 0) {                                         // wave-uniform; EXEC stays all ones inside
        // ---- LDS-DMA, branch-free: q (16 rows, rows >= G repeat the last head), then K, then V.  Rows past the
        // live range repeat the tile's last live row (finite data): their scores are masked, p is exactly 0.
        {
            const int p = lane % LPT, r = lane / LPT;         // chunk position / row inside one DMA instruction
            const int off0 = tok0 - (tok0 / a.block_size) * a.block_size;
            const int64_t row = (int64_t)a.kvh * D;           // elements per token (all kv heads)
            const int64_t base = ((int64_t)bid * a.block_size + off0) * row + (int64_t)kh * D;
            const uint16_t* kp = a.k_cache + base;
            const uint16_t* vp = a.v_cache + base;
            const uint16_t* qp = a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D;
            const int last = n_live - 1;
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int R = i * TPI + r;
                const int g = R < G ? R : G - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp + g * D + (p ^ chunk_swizzle<LPT>(R)) * 8),
                                                 (__attribute__((address_space(3))) void*)(lds_q + i * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int T = i * TPI + r;
                const int Tc = T < last ? T : last;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp + Tc * row + (p ^ chunk_swizzle<LPT>(T)) * 8),
                                                 (__attribute__((address_space(3))) void*)(lds_k + i * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int T = i * TPI + r;
                const int Tc = T < last ? T : last;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp + Tc * row + (p ^ chunk_swizzle<LPT>(T)) * 8),
                                                 (__attribute__((address_space(3))) void*)(lds_v + i * 1024), 16, 0, 0);
            }
        }
        NVH_STAMP(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");          // q and K landed; V may still be in flight
        NVH_STAMP(3);

        // ---- S^T = K Q^T
        bf16x8 qf[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
            qf[st] = *reinterpret_cast<const bf16x8*>(lds_q + lq * ROWB + (((4 * st + lg) ^ chunk_swizzle<LPT>(lq)) * 16));
        f32x4 sT[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            sT[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int R = 16 * tt + lq;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(lds_k + R * ROWB + (((4 * st + lg) ^ chunk_swizzle<LPT>(R)) * 16));
                sT[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[st], sT[tt], 0, 0, 0);
            }
        }
        // ---- softmax numerators against the tile max of this lane's head (log2 domain)
        float mx = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = 16 * tt + 4 * lg + r < n_live;
                sT[tt][r] = live ? sT[tt][r] * a.scale_log2 : -INFINITY;
                mx = fmaxf(mx, sT[tt][r]);
            }
        mx = max_xor16(mx);
        mx = max_xor32(mx);                                    // finite: token 0 of the tile is live
        float lsum = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sT[tt][r] = fast_exp2(sT[tt][r] - mx);
                lsum += sT[tt][r];
            }
        NVH_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // V landed
        NVH_STAMP(5);

        // ---- O^T += V^T P^T, P as hi + lo bf16
        f32x4 o[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int vq = lq >> 2, vp = lq & 3;                  // lane 4q+p of its group addresses key row q, dims 4p..4p+3
#pragma unroll
        for (int hh = 0; hh < NHALF; ++hh) {
            bf16x8 p_hi, p_lo;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pv = sT[2 * hh + (i >> 2)][i & 3];
                p_hi[i] = (__bf16)pv;
                p_lo[i] = (__bf16)(pv - (float)p_hi[i]);
            }
            const int R = 32 * hh + 4 * lg + vq;              // chunk_swizzle(R) == chunk_swizzle(R + 16)
            const unsigned char* vrow = lds_v + R * ROWB + (vp & 1) * 8;
            const int swz = chunk_swizzle<LPT>(R);
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int off = ((2 * t + (vp >> 1)) ^ swz) * 16;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vrow + off));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vrow + 16 * ROWB + off));
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p_hi, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, p_lo, o[t], 0, 0, 0);
            }
        }
        lsum = sum_xor16(lsum);
        lsum = sum_xor32(lsum);
        NVH_STAMP(6);
        // ---- this wave's (max, sum, O) into LDS: O^T[dim 16t+4lg+r][head lq] -> fin[head][dim], aliasing the K image
        if (lq < G) {
            float* const fin = reinterpret_cast<float*>(lds_k) + lq * D + 4 * lg;
#pragma unroll
            for (int t = 0; t < DT; ++t) *reinterpret_cast<f32x4*>(fin + 16 * t) = o[t];
            if (lg == 0) {
                lds_ml[(wave * 2 + 0) * 16 + lq] = mx;
                lds_ml[(wave * 2 + 1) * 16 + lq] = lsum;
            }
        }
    }
    __syncthreads();

    // ---- merge the live waves with their own maxima, write the workgroup's partial
    const int n_waves = min(WAVES, (ctx - wg_tok0 + WT - 1) / WT);
    for (int idx = tid; idx < G * D; idx += WAVES * 64) {
        const int g = idx / D, d = idx - g * D;
        float M = -INFINITY;
        for (int w = 0; w < n_waves; ++w) M = fmaxf(M, lds_ml[(w * 2 + 0) * 16 + g]);
        float o = 0.f, L = 0.f;
        for (int w = 0; w < n_waves; ++w) {
            const float f = fast_exp2(lds_ml[(w * 2 + 0) * 16 + g] - M);
            o = fmaf(reinterpret_cast<const float*>(lds + w * geo::WAVE_BYTES)[idx], f, o);
            L = fmaf(lds_ml[(w * 2 + 1) * 16 + g], f, L);
        }
        const int64_t part = ((int64_t)b * a.h + kh * G + g) * a.num_splits + split;
        a.ws_acc[part * D + d] = o;
        if (d == 0) {
            a.ws_ml[part * 2] = M;
            a.ws_ml[part * 2 + 1] = L;
        }
    }
    NVH_STAMP(7);
}

// Chunked MFMA kernel (default; what nvh_paged_decode launches): the body lives in decode_chunked.h, shared with the fused
// qkv + attention launch of qkv_attend.hip.
template <int D, int NW, int PASS = MGeo<D>::SPLIT>
__global__ __launch_bounds__(NW * 64) void paged_decode_chunked_kernel(
    // the operands on the way to the first DMA come first and flat: with -amdgpu-kernarg-preload-count they are in SGPRs when the
    // wave starts (14 user SGPRs), instead of behind a kernarg s_load; the rest of the descriptor follows by reference
    const int32_t* __restrict__ p_context_lens, const int32_t* __restrict__ p_block_tables, const uint16_t* __restrict__ p_k_cache,
    const uint16_t* __restrict__ p_v_cache, const int p_kvh, const int p_block_size, const int p_max_blocks, const int p_chunks,
    const int p_bt_stride, const int p_bs_shift, const DecodeArgs a, const int G) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[chunked_lds_bytes<D, NW, PASS>()];
    // grid (kv head, sequence, chunk): the same linear workgroup order as (kv head + kvh * sequence, chunk) without the division
    decode_chunked_body<D, NW, PASS, false>(lds, (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y, p_context_lens, p_block_tables, p_k_cache, p_v_cache,
                                            p_kvh, p_block_size, p_max_blocks, p_chunks, p_bt_stride, p_bs_shift, a, G, FusedSync{});
}

// One thread per output element (b, h, d): all live partials are requested before any is used.
template <int D>
__global__ __launch_bounds__(256) void paged_decode_combine_kernel(const DecodeArgs a) {
    constexpr int SPLIT = Geo<D>::SPLIT;
    constexpr int CH = 16;                                     // partials per unrolled round trip
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.batch * a.h * D) return;
    const int64_t bh = idx / D;
    const int d = (int)(idx - bh * D);
    const int b = (int)(bh / a.h);
    const int ctx = a.context_lens[b];
    const int n = (ctx + SPLIT - 1) / SPLIT;                   // live splits; 0 for padding rows
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
        const float fo = fast_exp2(M - Mc);                    // M = -inf on the first chunk -> 0
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
    const float r = n > 0 ? o / L : 0.f;                       // ctx == 0 -> zeros (oracle behaviour)
    if (a.out_f32) reinterpret_cast<float*>(a.out)[idx] = r;
    else reinterpret_cast<__bf16*>(a.out)[idx] = (__bf16)r;
}

template <int D>
int launch_combine(const DecodeArgs& a, hipStream_t stream) {
    const int64_t total = (int64_t)a.batch * a.h * D;
    hipLaunchKernelGGL((paged_decode_combine_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    return check_launch("paged_decode_combine");
}

template <int D, int G>
int launch_valu(const DecodeArgs& a, hipStream_t stream) {
    dim3 grid(a.num_splits, a.kvh, a.batch);
    hipLaunchKernelGGL((paged_decode_split_valu_kernel<D, G>), grid, dim3(WAVES * 64), 0, stream, a);
    int rc = check_launch("paged_decode_split_valu");
    return rc ? rc : launch_combine<D>(a, stream);
}

template <int D>
int launch_valu_d(const DecodeArgs& a, int g, hipStream_t stream) {
    switch (g) {
        case 1: return launch_valu<D, 1>(a, stream);
        case 2: return launch_valu<D, 2>(a, stream);
        case 3: return launch_valu<D, 3>(a, stream);
        case 4: return launch_valu<D, 4>(a, stream);
        case 5: return launch_valu<D, 5>(a, stream);
        case 6: return launch_valu<D, 6>(a, stream);
        case 7: return launch_valu<D, 7>(a, stream);
        case 8: return launch_valu<D, 8>(a, stream);
    }
    set_error("paged_decode (valu): group size %d not in 1..8", g);
    return -2;
}

template <int D>
int launch_chunked(const DecodeArgs& a, int g, hipStream_t stream) {
    // D = 64: 8 waves (two per SIMD; 32-token tiles) by default: 6.97 vs 7.44 us at ctx 1034, 8.36 vs 8.47 at 1536; D = 128: NVH_D128_WAVES;
    // a.waves = 4 / 8 (nvh_paged_decode_variant) selects either shape.
    const int waves = a.waves == 4 ? 4 : (a.waves == 8 ? 8 : (D == 64 ? 8 : NVH_D128_WAVES));
    dim3 grid(a.kvh, a.batch, a.chunks);
    const int bs_shift = (a.block_size & (a.block_size - 1)) == 0 ? __builtin_ctz(a.block_size) : -1;
    if constexpr (D == 64) {
        // Pass size at D = 64.  A pair split over 3-5 workgroups leaves each with one or two 256-token passes: whenever the pass count is not a
        // multiple of the chunk count some workgroups carry twice the tokens of the others and the call takes as long as they do.  128-token
        // passes (16-token wave tiles, the k = 16 MFMA) balance them: B = 32 (4 chunks) -0.5 / -1.9 / -6.6 / -4.0 / +1.9 / -3.7 / -1.9 % at ctx
        // 600 / 1034 / 1300 / 1536 / 1800 / 2500 / 3300, B = 40 (3 chunks) -2.9 % on average, B = 24 (5) -1.2 %.  With 8 and more chunks (small
        // batches) the passes outnumber nothing and the doubled per-pass work costs +1..12 %; with 2 chunks (B = 64) it is a wash.
        const bool half_passes = a.pass_tokens == 128 || (a.pass_tokens == 0 && waves == 8 && a.chunks >= 3 && a.chunks <= 5);
        if (half_passes) {
            hipLaunchKernelGGL((paged_decode_chunked_kernel<D, 8, 128>), grid, dim3(8 * 64), 0, stream, a.context_lens, a.block_tables, a.k_cache, a.v_cache,
                               a.kvh, a.block_size, a.max_blocks, a.chunks, (int)a.bt_row_stride, bs_shift, a, g);
            return check_launch("paged_decode_chunked");
        }
    }
    if constexpr (D == 128) {
        // Pass size at D = 128: 64-token passes (four waves x 16-token tiles, the k = 16 MFMA) where 128-token passes leave some workgroups of a pair
        // with twice the tokens of the others (NVH_D128_HALF_RULE below; a.pass_tokens = 64 / 128 forces either)
        const bool half_passes = a.pass_tokens == 64 || (a.pass_tokens == 0 && a.waves == 0 && NVH_D128_HALF_RULE(a.chunks));   // (pass_tokens 128 forces the full pass)
        if (half_passes) {
            hipLaunchKernelGGL((paged_decode_chunked_kernel<D, 4, 64>), grid, dim3(4 * 64), 0, stream, a.context_lens, a.block_tables, a.k_cache, a.v_cache,
                               a.kvh, a.block_size, a.max_blocks, a.chunks, (int)a.bt_row_stride, bs_shift, a, g);
            return check_launch("paged_decode_chunked");
        }
    }
    if (waves == 8) {
        hipLaunchKernelGGL((paged_decode_chunked_kernel<D, 8>), grid, dim3(8 * 64), 0, stream, a.context_lens, a.block_tables, a.k_cache, a.v_cache,
                           a.kvh, a.block_size, a.max_blocks, a.chunks, (int)a.bt_row_stride, bs_shift, a, g);
        return check_launch("paged_decode_chunked");
    }
    hipLaunchKernelGGL((paged_decode_chunked_kernel<D, 4>), grid, dim3(4 * 64), 0, stream, a.context_lens, a.block_tables, a.k_cache, a.v_cache, a.kvh,
                       a.block_size, a.max_blocks, a.chunks, (int)a.bt_row_stride, bs_shift, a, g);
    return check_launch("paged_decode_chunked");
}

template <int D>
int launch_mfma(const DecodeArgs& a, int g, hipStream_t stream) {
    dim3 grid(a.kvh * a.batch, a.num_splits);
    hipLaunchKernelGGL((paged_decode_split_mfma_kernel<D>), grid, dim3(MW * 64), 0, stream, a, g);
    int rc = check_launch("paged_decode_split_mfma");
    return rc ? rc : launch_combine<D>(a, stream);
}

}  // namespace

int decode_split_tokens(int hd) { return hd == 64 ? MGeo<64>::SPLIT : MGeo<128>::SPLIT; }
static_assert(MGeo<64>::SPLIT == Geo<64>::SPLIT && MGeo<128>::SPLIT == Geo<128>::SPLIT, "both split kernels share one workspace layout");

int decode_max_group(void) { return 16; }

// compute units of the current device (256 on MI355X), read once: a device property, not a tuning knob
static int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// workgroups per (sequence, kv head): one wave of workgroups over the chip, at most one per pass; `forced` > 0 (the
// variant entry point, tests and A/B runs) overrides the choice
int decode_chunks(int batch, int kvh, int num_splits, int forced) {
    if ((int64_t)batch * kvh > kMaxSplitPairs) return 1;       // (more pairs than tickets: far more than CUs anyway)
    const int cus = device_cus();
    // floor, never nearest: every workgroup needs a CU to itself (134 KB of LDS), so a launch of more workgroups than CUs runs its
    // surplus as a second round (measured, B = 48 / 28 / 36 at ctx 1536: 3 / 5 / 4 chunks 13.8 / 11.0 / 10.8 us against
    // 10.1 / 8.0 / 8.5 with 2 / 4 / 3)
    int c = forced > 0 ? forced : cus / (batch * kvh);
    if (c < 1) c = 1;
    return c > num_splits ? num_splits : c;
}

int launch_paged_decode(const DecodeArgs& a, hipStream_t stream) {
    if (a.batch == 0) return 0;
    const int g = a.h / a.kvh;
    // a.impl (nvh_paged_decode_variant): 0 = the chunked MFMA kernel (one launch; what nvh_paged_decode runs), 1 = single-pass MFMA
    // split kernel + combine, 2 = VALU split kernel (groups <= 8) + combine — the older formulations, kept for A/B and parity-tested
    const int impl = a.impl;
    if (impl != 0 && a.out_packed) { set_error("paged_decode: out_packed needs the chunked kernel"); return -2; }
    if (impl == 2) return a.hd == 64 ? launch_valu_d<64>(a, g, stream) : launch_valu_d<128>(a, g, stream);
    if (impl == 1) return a.hd == 64 ? launch_mfma<64>(a, g, stream) : launch_mfma<128>(a, g, stream);
    return a.hd == 64 ? launch_chunked<64>(a, g, stream) : launch_chunked<128>(a, g, stream);
}

}  // namespace nvh
