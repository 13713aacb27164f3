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

#ifndef NVH_D128_WAVES
#define NVH_D128_WAVES 4     // waves per workgroup of the chunked kernel at head_dim 128 when the caller does not choose (4 or 8).
                             // With 4-byte records 8 waves won (7/1/128 ctx 1536: 11.1 vs 12.1 us); since the records move as 16-byte items
                             // 4 waves (32-token tiles, the k = 32 MFMA) win everywhere: 7/1/128 9.2 vs 9.6, 28/4/128 19.8 vs 20.6, 16/8/128 33.3 vs 34.0
#endif

namespace nvh {

namespace {

// Diagnostic build only (-DNVH_STAMPS, tools/probes/stamp_decode.py): clock stamps per wave into a debug
// buffer that nothing else reads.  Never compiled into the shipped library.
#ifdef NVH_STAMPS
#define NVH_STAMP(k) NVH_STAMP_IF(NVH_STAMP_HEAD(k), k)
#ifdef NVH_STAMPS_TAIL                          // slots 1..5 follow the hand-off tail of the chunked kernel instead of its first pass
#define NVH_STAMP_HEAD(k) ((k) == 0 || (k) >= 6)
#define NVH_TSTAMP(k) NVH_STAMP_IF(true, k)
#else
#define NVH_STAMP_HEAD(k) true
#define NVH_TSTAMP(k) do {} while (0)
#endif
#define NVH_STAMP_IF(on, k)                                                                              \
    do {                                                                                                 \
        if (!(on)) break;                                                                                \
        unsigned long long t_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (a.stamps && lane == 0)                                                                       \
            a.stamps[(((int64_t)b * a.kvh + kh) * a.num_splits + split) * (WAVES * 8) + wave * 8 + (k)] = t_; \
    } while (0)
#else
#define NVH_STAMP(k) do {} while (0)
#define NVH_TSTAMP(k) do {} while (0)
#endif

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
constexpr int MW = 4;
#ifndef NVH_TICKET_WORDS
#define NVH_TICKET_WORDS 32                      // A/B builds: 1 = dense tickets (the round-1 layout)
#endif
#ifndef NVH_REC_ALIGN
#define NVH_REC_ALIGN 64                         // floats; A/B builds: 1 = records packed back to back (the round-1 layout)
#endif
constexpr int kTicketStride = NVH_TICKET_WORDS; // uint32 words between the tickets of two (sequence, kv head) pairs (= 128 bytes)
constexpr int kMaxSplitPairs = 65536 / (4 * kTicketStride);   // tickets in the workspace's 64 KiB header: 512

template <int D>
struct MGeo {
    static constexpr int LPT = D / 8;            // 16-byte chunks per token row
    static constexpr int TPI = 64 / LPT;         // rows per LDS-DMA instruction (1 KiB)
    static constexpr int WT = 4096 / D;          // tokens per wave tile: 64 (D=64) / 32 (D=128) -> 8 KiB images
    static constexpr int NI = WT / TPI;          // DMA instructions per image (8)
    static constexpr int ROWB = D * 2;           // bytes per row
    static constexpr int NT = WT / 16;           // 16-token MFMA tiles of S^T
    static constexpr int NHALF = WT / 32;        // 32-token groups of the PV contraction
    static constexpr int STEPS = D / 32;         // k-steps of QK^T
    static constexpr int DT = D / 16;            // 16-dim tiles of O^T
    static constexpr int QI = 16 / TPI;          // DMA instructions for the 16-row q image
    static constexpr int IMG = WT * ROWB;        // 8192
    static constexpr int WAVE_BYTES = 2 * IMG;   // K + V image per wave; the q image is shared by the workgroup
    static constexpr int SPLIT = MW * WT;
};

// swizzle of the 16-byte chunk position inside row T of an LDS image
template <int LPT>
__device__ __forceinline__ int chunk_swizzle(int T) {
    return LPT == 8 ? ((T >> 1) & 7) : (T & 15);
}

// V image swizzle of the chunked kernel (see prefill_mfma.hip chunk_swz_v): makes the transposed V reads conflict free
template <int LPT>
__device__ __forceinline__ int chunk_swizzle_v(int T) {
    return LPT == 8 ? (T & 6) : ((2 * T) & 14);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16 through inline asm: with the builtin hipcc waits vmcnt(0) before the read (it cannot prove the read does
// not alias the LDS-DMA of the NEXT pass still in flight).  The caller batches these, then `s_waitcnt lgkmcnt(0)` + a
// scheduling fence before the first use.
__device__ __forceinline__ u32x2 ds_read_tr16_b64_asm(uint32_t lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// one v_max3_f32: plain fmaxf on MFMA outputs makes hipcc emit a canonicalising v_max x,x before every use
__device__ __forceinline__ float max3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

// wave-uniform int32 load through the constant address space: stays a scalar load (lgkmcnt) inside loops that also hold
// stores and LDS-DMA, where a plain load would be a vector load whose vmcnt wait drains the DMA queue
__device__ __forceinline__ int32_t load_uniform_i32(const int32_t* p) {
    return *(const __attribute__((address_space(4))) int32_t*)(uintptr_t)p;
}

// The chunk hand-off's two flavours.  Default: the records move as write-through stores / L1-bypassing loads (sc1) and no fence is
// needed.  -DNVH_HANDOFF_FENCES (cross-check build, never shipped; tools/probes/run_fence_crosscheck.sh): the SAME
// hand-off in the HIP memory model's textbook form — plain stores, an agent-scope release fence before the ticket, an agent-scope
// acquire fence behind it, plain loads.  +3-4 us per launch; the parity suite is run against it and its results are compared bit for
// bit with the shipped form's.
#ifdef NVH_HANDOFF_FENCES
#define NVH_HANDOFF_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define NVH_HANDOFF_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define NVH_HANDOFF_RELEASE() do {} while (0)
#define NVH_HANDOFF_ACQUIRE() do {} while (0)
#endif

// The record accesses: 16-byte items (buffer_store_dwordx4 / buffer_load_dwordx4 with sc1 through a raw
// buffer descriptor of the (sequence, kv head)'s record group: the compiler sees them as memory operations and counts their vmcnt,
// which inline-asm global_* accesses would leave to hand-placed waits).  Byte offsets; out-of-range accesses cannot happen (the
// descriptor spans exactly the group).
#ifdef NVH_HANDOFF_FENCES
constexpr int kRecAux = 0;
#else
constexpr int kRecAux = 16;                               // gfx940+ cache-policy bits of the buffer intrinsics: 1 = sc0, 2 = nt, 16 = sc1
#endif
typedef __amdgpu_buffer_rsrc_t RecBuf;
__device__ __forceinline__ RecBuf rec_buffer(float* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);     // raw buffer, 32-bit data format word of gfx9
}
__device__ __forceinline__ void st16_sc1(RecBuf rb, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rb, (int)byte_off, 0, kRecAux);
}
__device__ __forceinline__ f32x4 ld16_sc1(RecBuf rb, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)byte_off, 0, kRecAux));
}

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

// =====================================================================================================
// Chunked MFMA kernel (default): the split kernel above restructured so that one launch does the whole call.
//   * grid (KVH*B, chunks): a workgroup walks PASSES of SPLIT tokens (pass p belongs to chunk p % chunks, so the live
//     passes of a sequence are dealt evenly whatever its length); the host picks chunks ~ 256 / (B*KVH): the launch is
//     one wave of workgroups over the 256 CUs and the per-launch fixed costs (arguments, dispatch ramp, first-byte
//     latency, merge, epilogue) are paid once per 2-8 passes instead of once per pass.
//   * each wave owns two K + V image pairs (double buffer) and runs the online softmax over its passes without any
//     workgroup barrier: pass p+1's LDS-DMA is issued before pass p is consumed, behind counted vmcnt waits.
//   * the waves merge through LDS once; with more than one live chunk the workgroup publishes its (max, sum, O) record
//     write-through (16-byte sc1 stores, vmcnt(0), barrier, ticket by a relaxed agent atomic) and the LAST ARRIVER of the
//     (sequence, kv head) merges all records (16-byte sc1 loads) and writes the output: no combine launch (-4.7 us per layer),
//     deterministic merge order.  The release/acquire-fence form of this hand-off cost more than the launch it saved
//     (profiles/r01_gemm_phase_stamps.txt has the same measurement for the split-K GEMM).
template <int D, int NW, int PASS = MGeo<D>::SPLIT>
__global__ __launch_bounds__(NW * 64) void paged_decode_chunked_kernel(
    // the operands on the way to the first DMA come first and flat: with -amdgpu-kernarg-preload-count they are in SGPRs when the
    // wave starts (14 user SGPRs), instead of behind a kernarg s_load; the rest of the descriptor follows by reference
    const int32_t* __restrict__ p_context_lens, const int32_t* __restrict__ p_block_tables, const uint16_t* __restrict__ p_k_cache,
    const uint16_t* __restrict__ p_v_cache, const int p_kvh, const int p_block_size, const int p_max_blocks, const int p_chunks,
    const int p_bt_stride, const int p_bs_shift, const DecodeArgs a, const int G) {
    // NW waves share a pass of SPLIT tokens: NW = 4 -> 64-token (D=64) / 32-token (D=128) tiles; NW = 8 -> 32-token (D=64) /
    // 16-token (D=128) tiles, two waves per SIMD covering each other's LDS / MFMA latencies, same LDS footprint.  A 16-token tile
    // contracts P V over 16 keys with v_mfma_f32_16x16x16_bf16 (one transposed V read per dim tile) instead of 32 with 16x16x32.
    constexpr int MW = NW;
    // PASS = tokens of one pass of the workgroup (default 256 at D = 64, 128 at D = 128).  D = 64 with PASS = 128 (16-token wave tiles,
    // the k = 16 MFMA) deals a context in half-size passes: chunks whose pass counts differ by one then differ by 128 tokens, not 256
    constexpr int SPLIT = PASS, WT = SPLIT / NW;
    constexpr int LPT = D / 8, TPI = 64 / LPT, NI = WT / TPI, ROWB = D * 2, NT = WT / 16, NHALF = WT / 32;
    constexpr int STEPS = D / 32, DT = D / 16, QI = 16 / TPI, IMG = WT * ROWB;
    constexpr int WAVES = NW;
    constexpr int WAVE_BYTES = 2 * IMG;                       // K + V image of one tile
    constexpr int WAVE_LDS = 2 * WAVE_BYTES;                  // two (K, V) image pairs per wave
    constexpr bool HALF = WT == 16;                           // 16-token tiles: one 16-key PV step on the k = 16 MFMA
    static_assert((NHALF >= 1 || HALF) && NI >= 1 && NT >= 1, "a wave tile is 16 tokens or a multiple of 32");
    static_assert(16 * D * 4 <= WAVE_LDS, "merge tile must fit in the wave's images (all landed and read by then)");
    __shared__ __attribute__((aligned(16))) unsigned char lds[MW * WAVE_LDS + QI * 1024 + MW * 2 * 16 * 4 + 16];
    unsigned char* const lds_q = lds + MW * WAVE_LDS;
    float* const lds_ml = reinterpret_cast<float*>(lds_q + QI * 1024);              // [wave][max | sum][16 heads]
    unsigned* const lds_ticket = reinterpret_cast<unsigned*>(lds_q + QI * 1024 + MW * 2 * 16 * 4);

    // grid (kv head, sequence, chunk): the same linear workgroup order as (kv head + kvh * sequence, chunk) without the division;
    // block_size is a power of two in every engine configuration: p_bs_shift >= 0 then replaces the divisions by it (each a
    // ~30-instruction sequence on the way to the first DMA)
    const int split = blockIdx.z, kh = blockIdx.x, b = blockIdx.y;   // `split` = chunk index
    auto div_bs = [&](int x) { return p_bs_shift >= 0 ? x >> p_bs_shift : x / p_block_size; };
    const int NC = p_chunks;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    NVH_STAMP(0);
    // the block ids of the first two passes do not depend on the context length (the index is clamped to the table row, whose
    // entries past the live range are never used): fetch them together with it, one scalar round trip instead of two in a row
    const int64_t bt_row = (int64_t)b * p_bt_stride;
    const int wtok = wave * WT;
    int pass = split;
    int tok0 = pass * SPLIT + wtok;
    int bid = load_uniform_i32(p_block_tables + bt_row + min(div_bs(tok0), p_max_blocks - 1));
    int bid_next = load_uniform_i32(p_block_tables + bt_row + min(div_bs(tok0 + NC * SPLIT), p_max_blocks - 1));
    const int ctx = load_uniform_i32(p_context_lens + b);
    const int live_passes = (ctx + SPLIT - 1) / SPLIT;
    if (split >= live_passes) {
        if (split == 0) {                                     // ctx == 0 (padding row): zeros, as the oracle
            for (int idx = tid; idx < G * D; idx += WAVES * 64) {
                const int64_t o = ((int64_t)b * a.h + kh * G) * D + idx;
                if (a.out_f32) reinterpret_cast<float*>(a.out)[o] = 0.f;
                else reinterpret_cast<uint16_t*>(a.out)[o] = 0;
                if (a.out_packed) a.out_packed[pack_index(b, kh * G * D + idx, a.h * D)] = 0;
            }
        }
        return;                                               // whole workgroup, before any barrier
    }
    // this wave's tile in pass p starts at token p*SPLIT + wave*WT; block ids are fetched one pass ahead
    NVH_STAMP(1);

    unsigned char* const lds_w = lds + wave * WAVE_LDS;
    const int lq = lane & 15;                                 // head column of the MFMA tiles
    const int lg = lane >> 4;                                 // lane group: k-block of operands / row block of C
    const int dp = lane % LPT, dr = lane / LPT;               // DMA: chunk position / row inside one instruction
    const int64_t row = (int64_t)p_kvh * D;                   // elements per token (all kv heads)

    // K (which = 1), V (2) or both (3, K first) images of the tile starting at token t0
    auto issue_kv = [&](int t0, int block_id, int buf, int which = 3) {
        const int off0 = t0 - div_bs(t0) * p_block_size;
        const int64_t base = ((int64_t)block_id * p_block_size + off0) * row + (int64_t)kh * D;
        const int last = ctx - t0 - 1;                        // rows past the live range repeat the last live row
        unsigned char* const kimg = lds_w + buf * WAVE_BYTES;
        if (which & 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int T = i * TPI + dr;
                const int Tc = T < last ? T : last;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_k_cache + base + Tc * row + (dp ^ chunk_swizzle<LPT>(T)) * 8),
                                                 (__attribute__((address_space(3))) void*)(kimg + i * 1024), 16, 0, NVH_DMA_AUX);
            }
        }
        if (which & 2) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int T = i * TPI + dr;
                const int Tc = T < last ? T : last;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_v_cache + base + Tc * row + (dp ^ chunk_swizzle_v<LPT>(T)) * 8),
                                                 (__attribute__((address_space(3))) void*)(kimg + IMG + i * 1024), 16, 0, NVH_DMA_AUX);
            }
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // descriptor fields of the hand-off and the output, fetched under the first K/V images (pinned below): left to the compiler
    // their kernarg loads sit in front of the first use, a scalar round trip each in the tail of the launch
    float* const e_ws_acc = a.ws_acc;
    unsigned* const e_counters = a.counters;
    void* const e_out = a.out;
    uint16_t* const e_out_packed = a.out_packed;
    const int e_out_f32 = a.out_f32, e_h = a.h;
    // scalars of the tail (record group, ticket, output row), computed under the first K/V images as well: ~40 scalar instructions
    // (64-bit multiplies) that otherwise sit between the LDS merge and the record stores, on every workgroup's way to its ticket
    // (offsets, not pointers, go through the pin: a pointer that has passed an asm statement has lost its address space and its
    // accesses become flat_ instructions)
    int t_rec;
    int64_t t_recs_off, t_ctr_off, t_orow;
    auto tail_scalars = [&]() {
        // floats per record: G rows of [D floats of O | max | sum | 0 | 0], padded so that every record starts on a 256-byte boundary
        t_rec = (G * (D + 4) + NVH_REC_ALIGN - 1) / NVH_REC_ALIGN * NVH_REC_ALIGN;
        const int64_t pair = (int64_t)b * p_kvh + kh;
        t_recs_off = pair * NC * t_rec;
        t_ctr_off = pair * kTicketStride;
        t_orow = ((int64_t)b * e_h + kh * G) * D;
        asm volatile("" : "+s"(t_recs_off), "+s"(t_ctr_off), "+s"(t_rec), "+s"(t_orow));
    };
    if (tok0 < ctx) {                                         // wave-uniform; EXEC stays all ones inside
        // first K image, then q, then the first V image: everything the K DMA needs arrived with the wave (preloaded arguments,
        // the block id), while q's pointer is still behind a kernarg load; the first wait below (q and K landed) counts on q
        // being older than V
        issue_kv(tok0, bid, 0, 1);
        {
            const uint16_t* qp = a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D;
#pragma unroll
            for (int i = 0; i < QI; ++i) {                    // q: 16 rows, rows >= G repeat the last head
                const int R = i * TPI + dr;
                const int g = R < G ? R : G - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qp + g * D + (dp ^ chunk_swizzle<LPT>(R)) * 8),
                                                 (__attribute__((address_space(3))) void*)(lds_q + i * 1024), 16, 0, 0);
            }
        }
        issue_kv(tok0, bid, 0, 2);
        NVH_STAMP(2);
        asm volatile("" ::"s"(e_ws_acc), "s"(e_counters), "s"(e_out), "s"(e_out_packed), "s"(e_out_f32), "s"(e_h));
        tail_scalars();
        bf16x8 qf[STEPS];
        for (int buf = 0;; buf ^= 1) {
            const int tok_next = tok0 + NC * SPLIT;
            const bool has_next = tok_next < ctx;             // wave-uniform
            int bid_nn = 0;
            if (has_next) {
                issue_kv(tok_next, bid_next, buf ^ 1);
                bid_nn = load_uniform_i32(p_block_tables + bt_row + min(div_bs(tok_next + NC * SPLIT), p_max_blocks - 1));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");      // q and this pass's K landed
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            }
            const unsigned char* const lds_k = lds_w + buf * WAVE_BYTES;
            const unsigned char* const lds_v = lds_k + IMG;
            const int n_live = ctx - tok0;
            if (pass == split) {
                NVH_STAMP(3);
#pragma unroll
                for (int st = 0; st < STEPS; ++st)
                    qf[st] = *reinterpret_cast<const bf16x8*>(lds_q + lq * ROWB + (((4 * st + lg) ^ chunk_swizzle<LPT>(lq)) * 16));
            }
            // ---- S^T = K Q^T
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
            // ---- online softmax (log2 domain): tile max per head on the RAW scores (scale > 0 commutes with max), scale and
            // max subtraction in one FMA inside the exp2; masking only on a ragged tile; rescale only when a max moved
            if (n_live < WT) {                                 // wave-uniform: the context ends inside this tile
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * tt + 4 * lg + r >= n_live) sT[tt][r] = -INFINITY;
            }
            // The max chain below reads the S^T accumulators through inline asm (max3), which hipcc does not pad: a VALU read of an
            // MFMA result needs its wait states (8-pass XDL: 11) and the compiler only inserts them for instructions it can see.
            // One statement that takes EVERY accumulator as an operand (so it follows every MFMA of the tile) carries the pad;
            // without it a 16-token tile (one S^T accumulator, the max3 right behind its last MFMA) read a half-written
            // accumulator now and then: a wrong running max, i.e. a correct softmax in another rounding (found as run-to-run
            // differences of 1e-7 at D = 128 with 8 waves).
            if constexpr (NT == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sT[0]));
            else if constexpr (NT == 2) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sT[0]), "+v"(sT[1]));
            else asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sT[0]), "+v"(sT[1]), "+v"(sT[2]), "+v"(sT[3]));
            static_assert(NT == 1 || NT == 2 || NT == 4, "tile shapes of the chunked kernel");
            float mx = sT[0][0];
#pragma unroll
            for (int i = 1; i + 1 < 4 * NT; i += 2) mx = max3(mx, sT[i >> 2][i & 3], sT[(i + 1) >> 2][(i + 1) & 3]);
            mx = max2(mx, sT[NT - 1][3]);
            mx = max_xor16(mx);
            mx = max_xor32(mx);                                // finite: token 0 of the tile is live
            const float m_new = max2(m_run, mx * a.scale_log2);
            float lsum = 0.f;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sT[tt][r] = fast_exp2(fmaf(sT[tt][r], a.scale_log2, -m_new));
                    lsum += sT[tt][r];
                }
            if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {             // wave-uniform branch; always taken on the first pass
                const float alpha = fast_exp2(m_run - m_new);                   // first pass: exp2(-inf) = 0
                l_run *= alpha;                                // per-lane partial sums; alpha is uniform over a head's lanes
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = o[t] * alpha;
            }
            l_run += lsum;
            m_run = m_new;
            if (pass == split) NVH_STAMP(4);
            if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");   // this pass's V landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (pass == split) NVH_STAMP(5);
            // ---- O^T += V^T P^T, P as hi + lo bf16
            const int vq = lq >> 2, vp = lq & 3;              // lane 4q+p of its group addresses key row q, dims 4p..4p+3
            if constexpr (HALF) {
                // 16 keys: B = P^T straight from the one S^T accumulator (k-slot j -> key 4 lg + j), A = V^T by ONE transposed read
                // per dim tile, both in the k order of v_mfma_f32_16x16x16_bf16
                bf16x4 p_hi, p_lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = sT[0][i];
                    p_hi[i] = (__bf16)pv;
                    p_lo[i] = (__bf16)(pv - (float)p_hi[i]);
                }
                const int R = 4 * lg + vq;
                const uint32_t vrow = lds_offset(lds_v + R * ROWB + (vp & 1) * 8);
                const int swz = chunk_swizzle_v<LPT>(R);
                u32x2 vt[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) vt[t] = ds_read_tr16_b64_asm(vrow + ((2 * t + (vp >> 1)) ^ swz) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                const s16x4 ph = __builtin_bit_cast(s16x4, p_hi), pl = __builtin_bit_cast(s16x4, p_lo);
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vt[t]), ph, o[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vt[t]), pl, o[t], 0, 0, 0);
            }
#pragma unroll
            for (int hh = 0; hh < NHALF; ++hh) {
                bf16x8 p_hi, p_lo;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float pv = sT[2 * hh + (i >> 2)][i & 3];
                    p_hi[i] = (__bf16)pv;
                    p_lo[i] = (__bf16)(pv - (float)p_hi[i]);
                }
                const int R = 32 * hh + 4 * lg + vq;          // chunk_swizzle_v(R) == chunk_swizzle_v(R + 16)
                const uint32_t vrow = lds_offset(lds_v + R * ROWB + (vp & 1) * 8);
                const int swz = chunk_swizzle_v<LPT>(R);
                u32x2 vlo[DT], vhi[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const uint32_t off = ((2 * t + (vp >> 1)) ^ swz) * 16;
                    vlo[t] = ds_read_tr16_b64_asm(vrow + off);
                    vhi[t] = ds_read_tr16_b64_asm(vrow + 16 * ROWB + off);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 vf[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const u32x4 raw = {vlo[t][0], vlo[t][1], vhi[t][0], vhi[t][1]};
                    vf[t] = *reinterpret_cast<const bf16x8*>(&raw);
                }
                // all hi products, then all lo: the two MFMAs on one accumulator are DT-1 independent MFMAs apart
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t], p_hi, o[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t], p_lo, o[t], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // image reads done: the buffer may be refilled
            if (!has_next) break;
            pass += NC;
            tok0 = tok_next;
            bid_next = bid_nn;
        }
        l_run = sum_xor16(l_run);
        l_run = sum_xor32(l_run);
        NVH_STAMP(6);
        // ---- this wave's (max, sum, O) into LDS: O^T[dim 16t+4lg+r][head lq] -> fin[head][dim], aliasing its first K image
        if (lq < G) {
            float* const fin = reinterpret_cast<float*>(lds_w) + lq * D + 4 * lg;
#pragma unroll
            for (int t = 0; t < DT; ++t) *reinterpret_cast<f32x4*>(fin + 16 * t) = o[t];
            if (lg == 0) {
                lds_ml[(wave * 2 + 0) * 16 + lq] = m_run;
                lds_ml[(wave * 2 + 1) * 16 + lq] = l_run;
            }
        }
    } else {
        tail_scalars();
    }
    __syncthreads();

    // ---- merge the live waves (those with a live tile in the workgroup's first pass), then the live chunks
    // A thread owns ITEMS of four consecutive dims of one head (16 bytes): the LDS reads, the record stores and loads of the
    // hand-off and the output stores are all 16-byte accesses (a write-through store is one fabric write per LANE whatever its
    // width: a record of G*D floats is G*D/4 writes instead of G*D)
    const int n_waves = min(WAVES, (ctx - split * SPLIT + WT - 1) / WT);
    const int live_chunks = min(NC, live_passes);
    constexpr int IPT = (16 * D / 4 + MW * 64 - 1) / (MW * 64);   // items per thread when G == 16
    constexpr int DSH = D == 64 ? 6 : 7;
    const int n_items = G * (D / 4);
    float Mv[IPT], Lv[IPT];
    f32x4 Ov[IPT];
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int it = tid + e * WAVES * 64;
        Mv[e] = -INFINITY; Lv[e] = 0.f; Ov[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (it < n_items) {
            const int g = (4 * it) >> DSH;
            // every wave's (max, sum, O) requested at once (dead waves re-read the last live one and are masked): one LDS
            // latency instead of one per wave
            float mw[WAVES], lw[WAVES];
            f32x4 ow[WAVES];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const int wc = w < n_waves ? w : n_waves - 1;
                mw[w] = lds_ml[(wc * 2 + 0) * 16 + g];
                lw[w] = lds_ml[(wc * 2 + 1) * 16 + g];
                ow[w] = reinterpret_cast<const f32x4*>(lds + wc * WAVE_LDS)[it];
            }
            float M = mw[0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) M = fmaxf(M, mw[w]);      // (a repeated wave does not change the max)
            f32x4 ov = f32x4{0.f, 0.f, 0.f, 0.f};
            float L = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const float f = w < n_waves ? fast_exp2(mw[w] - M) : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = fmaf(ow[w][j], f, ov[j]);
                L = fmaf(lw[w], f, L);
            }
            Mv[e] = M; Lv[e] = L; Ov[e] = ov;
        }
    }
    NVH_TSTAMP(1);
    if (live_chunks > 1) {
        // record: G rows of [D floats of O | max | sum | 0 | 0], padded so that every record starts on a 256-byte boundary (no
        // line shared with another workgroup's record); byte offsets inside the (sequence, kv head)'s record group
        const int rec = t_rec;
        float* const recs = e_ws_acc + t_recs_off;
        const RecBuf rb = rec_buffer(recs, (uint32_t)(NC * rec * 4));
        const uint32_t mine = (uint32_t)(split * rec * 4);
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int it = tid + e * WAVES * 64;
            if (it < n_items) {
                const int g = (4 * it) >> DSH;
                st16_sc1(rb, mine + 16 * (it + g), Ov[e]);                        // row g starts at float g * (D + 4)
                if (((4 * it) & (D - 1)) == 0) st16_sc1(rb, mine + 4 * (g * (D + 4) + D), f32x4{Mv[e], Lv[e], 0.f, 0.f});
            }
        }
        NVH_TSTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NVH_TSTAMP(3);
        if (tid == 0) {
            NVH_HANDOFF_RELEASE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // one ticket per 128-byte line: the memory side executes the adds on one line one after the other (~12 ns each); with
            // dense tickets the 16-32 pairs that share a line made every pair's last arriver queue behind all their adds
            unsigned* const ctr = e_counters + t_ctr_off;
            const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *lds_ticket = old;
            NVH_HANDOFF_ACQUIRE();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        NVH_TSTAMP(4);
        if (*lds_ticket != (unsigned)live_chunks - 1) return;  // workgroup-uniform
        // the last arriver zeroes the ticket for the next launch.  Issued here, behind the barrier, and not next to the add: the
        // barrier's wait would hold the whole workgroup until this store is acknowledged (~0.2 us on the launch's critical path);
        // now it completes under the record loads (the kernel's end waits for it like for the output stores)
        if (tid == 0)
            __hip_atomic_store(e_counters + t_ctr_off, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // records are requested CB at a time (every load of a batch in flight together); four chunks, the common shape of a
        // full launch, are one batch of 8 loads per thread rather than an 8-wide batch with half of it repeated
        auto merge_chunks = [&](auto cb_tag) {
            constexpr int CB = decltype(cb_tag)::value;
#pragma unroll
            for (int e = 0; e < IPT; ++e) {
                const int it = tid + e * WAVES * 64;
                if (it < n_items) {
                    const int g = (4 * it) >> DSH;
                    float M = -INFINITY, L = 0.f;
                    f32x4 ov = f32x4{0.f, 0.f, 0.f, 0.f};
                    for (int c0 = 0; c0 < live_chunks; c0 += CB) {
                        f32x4 ml[CB], av[CB];
#pragma unroll
                        for (int i = 0; i < CB; ++i) {
                            const int c = c0 + i < live_chunks ? c0 + i : live_chunks - 1;
                            const uint32_t r = (uint32_t)(c * rec * 4);
                            ml[i] = ld16_sc1(rb, r + 4 * (g * (D + 4) + D));
                            av[i] = ld16_sc1(rb, r + 16 * (it + g));
                        }
                        float Mc = M;
#pragma unroll
                        for (int i = 0; i < CB; ++i)
                            if (c0 + i < live_chunks) Mc = fmaxf(Mc, ml[i][0]);
                        const float fo = fast_exp2(M - Mc);        // M = -inf on the first group -> 0
#pragma unroll
                        for (int j = 0; j < 4; ++j) ov[j] *= fo;
                        L *= fo;
#pragma unroll
                        for (int i = 0; i < CB; ++i)
                            if (c0 + i < live_chunks) {
                                const float f = fast_exp2(ml[i][0] - Mc);
#pragma unroll
                                for (int j = 0; j < 4; ++j) ov[j] = fmaf(av[i][j], f, ov[j]);
                                L = fmaf(ml[i][1], f, L);
                            }
                        M = Mc;
                    }
                    Lv[e] = L; Ov[e] = ov;
                }
            }
        };
        if (live_chunks <= 4) merge_chunks(std::integral_constant<int, 4>{});
        else merge_chunks(std::integral_constant<int, 8>{});
        NVH_TSTAMP(5);
    }
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int it = tid + e * WAVES * 64;
        if (it < n_items) {
            // one reciprocal per item (v_rcp_f32: 1 ulp; then a product per dim: <= 2 ulp of fp32, far inside the bf16 output's rounding
            // and the 1e-3 bar of the fp32 output) instead of four IEEE divisions in a row at the very end of the launch's critical path
            const float inv_l = __builtin_amdgcn_rcpf(Lv[e]);
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = Ov[e][j] * inv_l;
            const int64_t oidx = t_orow + 4 * it;
            bf16x4 rb16;
#pragma unroll
            for (int j = 0; j < 4; ++j) rb16[j] = (__bf16)r[j];
            if (e_out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e_out) + oidx) = r;
            else *reinterpret_cast<bf16x4*>(reinterpret_cast<uint16_t*>(e_out) + oidx) = rb16;
            if (e_out_packed) *reinterpret_cast<bf16x4*>(e_out_packed + pack_index(b, kh * G * D + 4 * it, e_h * D)) = rb16;
        }
    }
    NVH_STAMP(7);
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
