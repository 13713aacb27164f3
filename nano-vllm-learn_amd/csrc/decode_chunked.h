// The chunked MFMA decode-attention body (one workgroup = one chunk of one (sequence, kv head) pair), shared by
//   * paged_decode.hip   paged_decode_chunked_kernel: the stand-alone launch behind nvh_paged_decode (FUSED = false), and
//   * qkv_attend.hip     qkv_attend_kernel: the same body as the CONSUMER role of the launch that also computes q and this step's
//                        K / V rows (FUSED = true): the K/V stream starts with the launch, q and the newest row arrive by a hand-off.
// Replaces flash_attn_with_kvcache at nanovllm/layers/attention.py:99-101 (oracle: nanovllm/layers/attention_sdpa.py:122-182).
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

#ifndef NVH_DMA_AUX
#define NVH_DMA_AUX 2        // cache policy of the once-read LDS-DMA streams (weights, K/V): 2 = nt, 0 = default.
                             // nt measured -4.7 % on the decode step, -0.6 us per attention call (same box A/B, round 1)
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
// the same for any wave-uniform range, always L1-bypassing (the fused launch's hand-off reads q and the newest K / V row this way)
typedef __amdgpu_buffer_rsrc_t RawBuf;
__device__ __forceinline__ RawBuf raw_buffer(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16_raw_sc1(RawBuf rb, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rb, (int)byte_off, 0, 16);
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
// What a FUSED consumer waits on (qkv_attend.hip).  The producers of kv head kh's group (the workgroups that compute the q heads of the
// group and this step's K and V row of the head) each add one to ready[kh] once their rows are globally visible (write-through
// stores, drained, workgroup barrier: the same producer side as the chunk hand-off below); a consumer polls until `need` of them have.
// Every consumer adds to done[kh] behind its wait, and the last of the `consumers` returns both counters to zero for the next launch.
struct FusedSync {
    unsigned* ready;             // [kvh], one counter per 128-byte line (kTicketStride words apart)
    unsigned* done;              // [kvh], likewise
    unsigned* status;            // [0] != 0 after a wait that ran out (a producer that never arrived): outputs of that launch are NaN
    unsigned need;               // producer workgroups per kv head
    unsigned consumers;          // consumer workgroups per kv head
    unsigned spin_limit;         // polls before giving up (each ~ one memory-side round trip)
    const void* pf_ptr;          // nullable: bytes a LATER launch streams once (the output projection's weights), touched while waiting
    int64_t pf_bytes;
};

template <int D, int NW, int PASS>
constexpr int chunked_lds_bytes() {
    constexpr int WT = PASS / NW, IMG = WT * D * 2, QI = 16 / (64 / (D / 8));
    return NW * 4 * IMG + QI * 1024 + NW * 2 * 16 * 4 + 16;
}

// The poll of one wave (all lanes load the same dword: one request): FOUR polls in flight, a new one issued as the oldest returns, so
// that a counter that completes is seen one round trip later at most (a single poll in flight sees it up to two round trips later: the
// poll that just missed has to come back first).  One asm statement: the in-flight destination registers are nobody else's.
// Returns 1 when the counter reached `need`, 0 after `limit` rounds of four polls.  Older vector-memory operations of the wave complete
// first (vmcnt counts in order).
__device__ __forceinline__ unsigned poll_counter(const unsigned* ctr, unsigned need, unsigned limit) {
    unsigned ok, cnt, r0, r1, r2, r3;
    asm volatile(
        "s_mov_b32 %[cnt], %[limit]\n\t"
        "global_load_dword %[r0], %[p], off sc1\n\t"
        "s_sleep 3\n\t"
        "global_load_dword %[r1], %[p], off sc1\n\t"
        "s_sleep 3\n\t"
        "global_load_dword %[r2], %[p], off sc1\n\t"
        "s_sleep 3\n\t"
        "global_load_dword %[r3], %[p], off sc1\n\t"
        "1:\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_readfirstlane_b32 %[ok], %[r0]\n\t"
        "s_cmp_ge_u32 %[ok], %[need]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dword %[r0], %[p], off sc1\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_readfirstlane_b32 %[ok], %[r1]\n\t"
        "s_cmp_ge_u32 %[ok], %[need]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dword %[r1], %[p], off sc1\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_readfirstlane_b32 %[ok], %[r2]\n\t"
        "s_cmp_ge_u32 %[ok], %[need]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dword %[r2], %[p], off sc1\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_readfirstlane_b32 %[ok], %[r3]\n\t"
        "s_cmp_ge_u32 %[ok], %[need]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dword %[r3], %[p], off sc1\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[ok], 0\n\t"
        "s_branch 3f\n\t"
        "2:\n\t"
        "s_mov_b32 %[ok], 1\n\t"
        "3:\n\t"
        "s_waitcnt vmcnt(0)"
        : [ok] "=&s"(ok), [cnt] "=&s"(cnt), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
        : [p] "v"(ctr), [need] "s"(need), [limit] "s"(limit)
        : "memory", "scc");
    return ok;
}

// Wave 0 polls (bounded) and, once the producers are in, fetches the group's q rows into the workgroup's q image (L1-bypassing loads of
// ONE wave: the eight waves of 256 workgroups asking the memory side for the same 57 KB at once queued for ~1.5 us); every wave of the
// workgroup leaves through the barrier (an LDS-only barrier: the waves' LDS-DMA stays in flight across it).  Returns false when the wait
// ran out.  The poller then draws the head's consumer count; fused_release() — at the END of the workgroup's work, so that the returning
// atomic's round trip is nobody's wait — returns the counters to zero in the workgroup that drew the last one.
template <typename StageQ>
__device__ __forceinline__ bool fused_wait(const FusedSync& fs, int kh, int tid, unsigned* lds_word, unsigned& drawn, StageQ stage_q) {
    drawn = 0;
    if (tid < 64) {                                              // wave 0, all lanes
        const unsigned ok = poll_counter(fs.ready + (int64_t)kh * kTicketStride, fs.need, (fs.spin_limit + 3) / 4);
        stage_q();
        if (tid == 0) *lds_word = ok;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool ok = *lds_word != 0;
    if (tid == 0) {
        if (!ok) __hip_atomic_store(fs.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drawn = __hip_atomic_fetch_add(fs.done + (int64_t)kh * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return ok;
}
__device__ __forceinline__ void fused_release(const FusedSync& fs, int kh, int tid, unsigned drawn) {
    if (tid == 0 && drawn == fs.consumers - 1) {                 // every consumer of the head is past its wait: reset for the next launch
        __hip_atomic_store(fs.ready + (int64_t)kh * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fs.done + (int64_t)kh * kTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifndef NVH_FUSED_DELAY
#define NVH_FUSED_DELAY 0    // consumers of the fused launch hold their K/V stream back by this many s_sleep(8) (512 clocks each)
#endif

// lds: chunked_lds_bytes<D, NW, PASS>() bytes, 16-byte aligned.  (split, kh, b) = (chunk, kv head, sequence) of this workgroup.
template <int D, int NW, int PASS, bool FUSED>
__device__ __forceinline__ void decode_chunked_body(
    unsigned char* const lds, const int split, const int kh, const int b,
    const int32_t* __restrict__ p_context_lens, const int32_t* __restrict__ p_block_tables, const uint16_t* __restrict__ p_k_cache,
    const uint16_t* __restrict__ p_v_cache, const int p_kvh, const int p_block_size, const int p_max_blocks, const int p_chunks,
    const int p_bt_stride, const int p_bs_shift, const DecodeArgs& a, const int G, const FusedSync& fs) {
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
    static_assert(MW * WAVE_LDS + QI * 1024 + MW * 2 * 16 * 4 + 16 == chunked_lds_bytes<D, NW, PASS>(), "LDS layout");
    unsigned char* const lds_q = lds + MW * WAVE_LDS;
    float* const lds_ml = reinterpret_cast<float*>(lds_q + QI * 1024);              // [wave][max | sum][16 heads]
    unsigned* const lds_ticket = reinterpret_cast<unsigned*>(lds_q + QI * 1024 + MW * 2 * 16 * 4);

    // grid (kv head, sequence, chunk): the same linear workgroup order as (kv head + kvh * sequence, chunk) without the division;
    // block_size is a power of two in every engine configuration: p_bs_shift >= 0 then replaces the divisions by it (each a
    // ~30-instruction sequence on the way to the first DMA)
    // (`split` = chunk index)
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
        if constexpr (FUSED) {                                // (it still counts among the head's consumers)
            unsigned drawn;
            fused_wait(fs, kh, tid, lds_ticket, drawn, [] {});
            fused_release(fs, kh, tid, drawn);
        }
        return;                                               // whole workgroup, uniformly
    }
    // this wave's tile in pass p starts at token p*SPLIT + wave*WT; block ids are fetched one pass ahead
    NVH_STAMP(1);

    unsigned char* const lds_w = lds + wave * WAVE_LDS;
    const int lq = lane & 15;                                 // head column of the MFMA tiles
    const int lg = lane >> 4;                                 // lane group: k-block of operands / row block of C
    const int dp = lane % LPT, dr = lane / LPT;               // DMA: chunk position / row inside one instruction
    const int64_t row = (int64_t)p_kvh * D;                   // elements per token (all kv heads)

    // K (which = 1), V (2) or both (3, K first) images of the tile starting at token t0
    // FUSED: row ctx - 1 (the token this very launch produces) is never read here: it and the rows past it repeat the row BEFORE
    // it, and patch_new_row() below puts the hand-off's bytes in their place; a tile that holds nothing older is not fetched at all
    auto issue_kv = [&](int t0, int block_id, int buf, int which = 3) {
        const int off0 = t0 - div_bs(t0) * p_block_size;
        const int64_t base = ((int64_t)block_id * p_block_size + off0) * row + (int64_t)kh * D;
        int last = ctx - t0 - 1;                              // rows past the live range repeat the last live row
        if constexpr (FUSED) {
            if (last < WT) {                                  // wave-uniform
                if (last <= 0) return;
                --last;
            }
        }
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
    bf16x8 qf[STEPS];
    // ---- FUSED: everything that does not depend on this launch's own q / K / V rows starts NOW; then the hand-off wait
    constexpr int PN = (WT * LPT + 63) / 64;                  // (row, chunk) slots of one image per lane
    bool fused_ok = true, holds_new = false;                  // holds_new: one of this wave's tiles contains token ctx - 1
    unsigned fused_drawn = 0;
    int new_tok0 = 0, new_last = 0;                           // that tile's first token; the new row's index inside it
    u32x4 nk[PN], nv[PN];
    if constexpr (FUSED) {
        const int stride = NC * SPLIT;
        uint32_t sink = 0;
        int64_t new_row_off = 0;
        for (int i = 0; i < NVH_FUSED_DELAY; ++i) __builtin_amdgcn_s_sleep(8);   // (the producers' loads go first)
        if (tok0 < ctx) {                                     // wave-uniform
            issue_kv(tok0, bid, 0);
            if (tok0 + stride < ctx) issue_kv(tok0 + stride, bid_next, 1);
            // where the newest token sits (scalar work, under the DMA)
            const int d = ctx - 1 - tok0, kq = d / stride, r = d - kq * stride;
            holds_new = r < WT;
            new_tok0 = tok0 + kq * stride;
            new_last = r;
            if (holds_new) {
                const int nb = load_uniform_i32(p_block_tables + bt_row + div_bs(ctx - 1));
                new_row_off = ((int64_t)nb * p_block_size + (ctx - 1 - div_bs(ctx - 1) * p_block_size)) * row + (int64_t)kh * D;
            }
            // the tiles of this wave's passes 2.. : one dword per 32 bytes, default cache policy — the lines wait in the XCD's L2 for the
            // LDS-DMA that can only be issued once an image buffer is free, i.e. behind the hand-off (never the newest row: see issue_kv)
            for (int t = tok0 + 2 * stride; t < ctx; t += stride) {
                const int bidt = load_uniform_i32(p_block_tables + bt_row + div_bs(t));
                int last = ctx - t - 1;
                if (last < WT) {
                    if (last <= 0) break;
                    --last;
                }
                const int64_t base = ((int64_t)bidt * p_block_size + (t - div_bs(t) * p_block_size)) * row + (int64_t)kh * D;
                constexpr int PER_ROW = ROWB / 32;
#pragma unroll
                for (int i = 0; i < (WT * PER_ROW + 63) / 64; ++i) {
                    const int idx = i * 64 + lane, T = idx / PER_ROW, Tc = T < last ? T : last;
                    if (T < WT) {
                        sink |= *reinterpret_cast<const uint32_t*>(p_k_cache + base + Tc * row + (idx % PER_ROW) * 16);
                        sink |= *reinterpret_cast<const uint32_t*>(p_v_cache + base + Tc * row + (idx % PER_ROW) * 16);
                    }
                }
            }
        }
        asm volatile("" ::"s"(e_ws_acc), "s"(e_counters), "s"(e_out), "s"(e_out_packed), "s"(e_out_f32), "s"(e_h));
        tail_scalars();
        if (fs.pf_ptr) {                                      // this workgroup's share of the next launch's once-read operand
            const int64_t lines = fs.pf_bytes >> 5;
            const int nwg = (int)fs.consumers * p_kvh, iwg = (split * (int)gridDim.y + b) * p_kvh + kh;
            const int64_t per = (lines + nwg - 1) / nwg, l0 = iwg * per, l1 = l0 + per < lines ? l0 + per : lines;
            for (int64_t l = l0 + tid; l < l1; l += WAVES * 64) sink |= *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(fs.pf_ptr) + (l << 5));
        }
        NVH_STAMP(2);
        // the q image of the workgroup, filled by wave 0 behind its poll: row R = head min(R, G - 1), chunk order swizzled as the LDS-DMA of
        // the stand-alone kernel leaves it (the operand reads below are the same)
        fused_ok = fused_wait(fs, kh, tid, lds_ticket, fused_drawn, [&] {
            const RawBuf qb = raw_buffer(a.q + (int64_t)b * a.q_row_stride + (int64_t)(kh * G) * D, (uint32_t)(G * D * 2));
            u32x4 qv[QI];
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int R = i * TPI + dr, g = R < G ? R : G - 1;
                qv[i] = ld16_raw_sc1(qb, (uint32_t)((g * D + (dp ^ chunk_swizzle<LPT>(R)) * 8) * 2));
            }
#pragma unroll
            for (int i = 0; i < QI; ++i) *reinterpret_cast<u32x4*>(lds_q + i * 1024 + lane * 16) = qv[i];
        });
        NVH_STAMP(3);
        if (tok0 < ctx) {
            // the newest K / V row for the wave that holds its tile: the handed-off bytes are read by L1-bypassing (sc1) loads only
#pragma unroll
            for (int st = 0; st < STEPS; ++st)
                qf[st] = *reinterpret_cast<const bf16x8*>(lds_q + lq * ROWB + (((4 * st + lg) ^ chunk_swizzle<LPT>(lq)) * 16));
            if (holds_new) {
                const RawBuf kb = raw_buffer(p_k_cache + new_row_off, (uint32_t)ROWB), vb = raw_buffer(p_v_cache + new_row_off, (uint32_t)ROWB);
#pragma unroll
                for (int i = 0; i < PN; ++i) {
                    const int idx = i * 64 + lane, T = (idx / LPT) & (WT - 1), pch = idx % LPT;
                    nk[i] = ld16_raw_sc1(kb, (uint32_t)((pch ^ chunk_swizzle<LPT>(T)) * 16));
                    nv[i] = ld16_raw_sc1(vb, (uint32_t)((pch ^ chunk_swizzle_v<LPT>(T)) * 16));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink) : "memory");   // this wave's two passes, its touches, q (and the newest row) have landed
    }
    if (tok0 < ctx) {                                         // wave-uniform; EXEC stays all ones inside
        if constexpr (!FUSED) {
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
        }
        for (int buf = 0;; buf ^= 1) {
            const int tok_next = tok0 + NC * SPLIT;
            const bool has_next = tok_next < ctx;             // wave-uniform
            // FUSED: a tile that holds nothing but the newest row is not fetched (issue_kv), and pass 1 was issued in front of the hand-off
            const bool next_fetched = has_next && !(FUSED && ctx - tok_next - 1 == 0);
            int bid_nn = 0;
            if (has_next) {
                if (!(FUSED && pass == split)) issue_kv(tok_next, bid_next, buf ^ 1);
                bid_nn = load_uniform_i32(p_block_tables + bt_row + min(div_bs(tok_next + NC * SPLIT), p_max_blocks - 1));
            }
            if (next_fetched) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");      // q and this pass's K landed
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            unsigned char* const lds_k = lds_w + buf * WAVE_BYTES;
            unsigned char* const lds_v = lds_k + IMG;
            const int n_live = ctx - tok0;
            if constexpr (FUSED) {
                if (holds_new && tok0 == new_tok0) {           // wave-uniform: the wave's last pass
                    // the newest row and the dead rows behind it (so that P = 0 never meets a non-finite V): from the hand-off's bytes
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this tile's K and V DMA have landed
#pragma unroll
                    for (int i = 0; i < PN; ++i) {
                        const int idx = i * 64 + lane, T = idx / LPT, pch = idx % LPT;
                        if (T >= new_last && T < WT) {
                            *reinterpret_cast<u32x4*>(lds_k + T * ROWB + pch * 16) = nk[i];
                            *reinterpret_cast<u32x4*>(lds_v + T * ROWB + pch * 16) = nv[i];
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else {
            if (pass == split) {
                NVH_STAMP(3);
#pragma unroll
                for (int st = 0; st < STEPS; ++st)
                    qf[st] = *reinterpret_cast<const bf16x8*>(lds_q + lq * ROWB + (((4 * st + lg) ^ chunk_swizzle<LPT>(lq)) * 16));
            }
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
            if (next_fetched) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");   // this pass's V landed
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
        if constexpr (FUSED) {
            if (!fused_ok) l_run = __builtin_nanf("");        // a producer never arrived: the row is NaN, never a plausible number
        }
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
        if constexpr (!FUSED) tail_scalars();
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
        if (*lds_ticket != (unsigned)live_chunks - 1) {        // workgroup-uniform
            if constexpr (FUSED) fused_release(fs, kh, tid, fused_drawn);
            return;
        }
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
                        // every load of the batch is ISSUED here, before any is consumed: left alone, hipcc sank the last load of an 8-wide batch
                        // into the `c0 + i < live_chunks` block that uses it — behind the wait for the other fifteen, a second memory-side round
                        // trip (0.6 us) in the last arriver of every pair split over 5-8 workgroups (stamped: 1.24 us for this step at 7/1/128)
#pragma unroll
                        for (int i = 0; i < CB; ++i) asm volatile("" : "+v"(ml[i]), "+v"(av[i]));
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
    if constexpr (FUSED) fused_release(fs, kh, tid, fused_drawn);
}

}  // namespace

}  // namespace nvh
