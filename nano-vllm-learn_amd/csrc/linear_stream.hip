// Weight-streaming linear layer for decode-sized batches (M <= 64 rows), single-pass form — engine widening around the
// attention call, not part of the attention parity bar.  Same contract as skinny_gemm.hip (prologue NORM 0 / 2, the four
// epilogues); that file stays the general fallback (exact-norm prologue, K loops, no workspace).
//
// Why a second kernel: stamps of the loop kernel (tools/probes/stamp_gemm.py, profiles/r01_gemm_phase_stamps.txt) show a
// decode GEMM is a chain of latencies, not a bandwidth problem: 0.5 us kernel arguments, 1.2 us (p90 4.5) ISSUING the x
// operand as fragment-shaped loads (16 rows x 64 B per instruction: four rows per lane quad, a quarter of the address rate),
// 1.2 us issuing the W DMA queued behind them, 1.3 us landing, then one such round per K group (down_proj: five rounds on 56
// workgroups).  This kernel makes the chain as short as the hardware allows:
//   * every wave owns at most PMAX = 4 pieces (64 K-elements each) of ONE weight tile, so the whole K range is in flight at
//     once; long K is split over workgroups (grid.y = ksplit) and the tile's LAST-ARRIVING workgroup sums the fp32 partials
//     in split order (deterministic) and runs the epilogue: no second launch, no float atomics.
//   * W DMA is issued first (addresses need only the kernel arguments), x second.
//   * x can arrive PACKED in MFMA-fragment order (kernels.h pack_index): one coalesced 1 KiB load per fragment; the
//     epilogues can write that layout for the next GEMM (out_packed), so inside the fused decoder layer activations never
//     take the quarter-rate path.
//   * the folded-norm row sums use v_dot2_f32_bf16 on the packed pairs; the residual operand of RESADD is fetched while W
//     is in flight.
//   * wide N (LM head): a workgroup walks several tiles with x held in registers and the next tile's W DMA in flight
//     (MULTI), so x is fetched once per workgroup instead of once per tile; with candidate buffers it also keeps the
//     running arg-max of its columns, so a greedy decode step never writes the logits.
//   * WAVES: 8 per workgroup by default (two per SIMD, half the pieces each): a wave spends most of its life ISSUING (24 DMA +
//     16 x loads on gate_up ~ 2.8 us at one wave per SIMD) and two waves per SIMD overlap those phases: decode step -4.7 %
//     (same-box A/B, NVH_GEMM_WAVES=4 restores the old shape).  The multi-tile LM head (two workgroups per CU) stays at 4.
//   * LAUNCH BALANCE: every workgroup streams at about the same ~45 GB/s whatever it does, so the most loaded CU sets a
//     launch's load phase.  Launches are therefore shaped to at most 256 EQUAL workgroups where the shape allows: down_proj
//     takes 5 pieces per wave (PM) so that 56 tiles x 4 splits = 224 (not 280), gate_up uses WIDE tiles of 24 + 24
//     columns -> 203 workgroups (not 304).  (Fewer, fatter workgroups than that lose: see DESIGN.md section 8.)
// Hand-off (split-K), fence-free form: the partials are stored write-through (relaxed agent-scope atomic stores = `sc1`
// stores), every storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 takes the ticket (relaxed agent
// atomic add); the workgroup whose add came last reads all partials with `sc1` loads (relaxed agent-scope atomic loads) after
// a workgroup barrier the ticket holder joins.  A release/acquire fence pair here cost 4-5 us per launch (whole-L2
// write-back + invalidate under 280 workgroups; profiles/r01_gemm_phase_stamps.txt).
#include <cstdlib>
#ifndef NVH_DMA_AUX
#define NVH_DMA_AUX 2        // cache policy of the once-read LDS-DMA streams (weights, K/V): 2 = nt, 0 = default.
                             // nt measured -4.7 % on the decode step, -0.6 us per attention call (same box A/B, round 1)
#endif
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef NVH_TICKET_WORDS
#define NVH_TICKET_WORDS 32                        // A/B builds: 1 = dense tickets (the round-1 layout)
#endif
constexpr int kTicketWords = NVH_TICKET_WORDS;    // uint32 words between two tiles' arrival tickets: one ticket per 128-byte line
constexpr int SW = 4;                          // waves per workgroup
constexpr unsigned kMultiWgs = 512;            // workgroups of a MULTI launch (the LM head): two per CU, a power of two
constexpr int PMAX = 4;                        // pieces (2 k-steps = 64 K-elements = one 128-byte line per weight row) per wave

#ifdef NVH_STAMPS
#define LS_STAMP(k)                                                                                                        \
    do {                                                                                                                   \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                    \
        if (a.stamps && lane == 0) a.stamps[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + (k)] = t_;   \
    } while (0)
#else
#define LS_STAMP(k) do {} while (0)
#endif

// write-through store / cache-bypassing load of one float (compile to global_store_dword / global_load_dword with sc1)
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// compute units of the current device (256 on MI355X), read once
static int device_cus_ls() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// workgroup barrier behind this wave's LDS traffic only.  __syncthreads() also waits vmcnt(0): inside the multi-tile loop that is the NEXT
// tile's weight DMA, issued at the top of the iteration precisely so that it travels under this tile's reduction and epilogue
// (measured neutral on the LM head, 0.729 vs 0.731 ms per decode step: two workgroups per CU already cover each other's bubbles)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// leave the newest `np` pieces' DMA (np * PER instructions) in flight
template <int PER>
__device__ __forceinline__ void wait_all_but_pieces(int np) {
    switch (np) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<PER>(); break;
        case 2: wait_vm<2 * PER>(); break;
        case 3: wait_vm<3 * PER>(); break;
        case 4: wait_vm<4 * PER>(); break;
        default: wait_vm<5 * PER>(); break;
    }
}

// Prefetch role (workgroups with blockIdx.x >= tiles, on CUs the launch would leave idle): touch one dword of every NVH_PF_STRIDE bytes
// of this workgroup's share of [base, base + bytes) with default-policy loads and drop the data.
#ifndef NVH_PF_STRIDE
#define NVH_PF_STRIDE 32
#endif
__device__ __forceinline__ void prefetch_lines(const void* base, int64_t bytes, int idx, int n, int tid, int nthreads) {
    constexpr int SH = NVH_PF_STRIDE == 32 ? 5 : (NVH_PF_STRIDE == 64 ? 6 : 7);
    const int64_t lines = bytes >> SH;
    const int64_t per = (lines + n - 1) / n;
    const int64_t l0 = idx * per, l1 = l0 + per < lines ? l0 + per : lines;
    const unsigned char* const b = reinterpret_cast<const unsigned char*>(base);
    // compiler-visible loads OR-ed into one value that an empty asm consumes behind the loop: the destination registers stay live until
    // their data has landed (an inline-asm load into a dead register could be handed to another value while still in flight)
    uint32_t sink = 0;
    for (int64_t l = l0 + tid; l < l1; l += 8 * nthreads) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t ll = l + (int64_t)u * nthreads;
            if (ll < l1) sink |= *reinterpret_cast<const uint32_t*>(b + (ll << SH));
        }
    }
    asm volatile("" ::"v"(sink));
}

// The same for the first K/V images of the decode attention launch that follows (LinearArgs::KvPrefetch): attention workgroup w = (kv head
// w % kvh, sequence (w / kvh) % batch, chunk w / (kvh * batch)) starts with pass `chunk` of its sequence (and goes on with chunk + chunks);
// a pass lies inside one cache block (pass_tokens divides block_size).  One dword per 32 bytes of every live row before token ctx - 1.
// Two phases, so that a workgroup pays two memory round trips in all: (1) one thread per (attention workgroup, pass) of this workgroup's
// share loads the context length and the block id TOGETHER (the table index does not depend on the length) and leaves (live rows, row
// offset) in LDS; (2) every thread touches its share of those rows.
__device__ __forceinline__ void prefetch_kv_images(const LinearArgs::KvPrefetch& p, int idx, int n, int tid, int nthreads, unsigned char* lds) {
    constexpr int MAXC = 64;                                      // (attention workgroup, pass) combinations per prefetching workgroup
    int* const c_live = reinterpret_cast<int*>(lds);
    int64_t* const c_base = reinterpret_cast<int64_t*>(lds + MAXC * 4);
    const int total = p.kvh * p.batch * p.chunks;
    const int mine = idx < total ? (total - idx + n - 1) / n : 0;  // attention workgroups idx, idx + n, ...
    const int combos = min(mine * p.passes, MAXC);
    const int64_t row = (int64_t)p.kvh * p.hd;
    if (tid < combos) {
        const int w = idx + (tid / p.passes) * n, ps = tid % p.passes;
        const int kh = w % p.kvh, b = (w / p.kvh) % p.batch, chunk = w / (p.kvh * p.batch);
        const int t0 = (chunk + ps * p.chunks) * p.pass_tokens;
        const int slot = min(t0 / p.block_size, (int)p.bt_stride - 1);
        const int ctx = p.context_lens[b];
        const int blk = p.block_tables[(int64_t)b * p.bt_stride + slot];
        const int live = min(p.pass_tokens, ctx - 1 - t0);         // rows before the newest token
        c_live[tid] = live > 0 ? live : 0;
        c_base[tid] = ((int64_t)blk * p.block_size + t0 % p.block_size) * row + (int64_t)kh * p.hd;
    }
    __syncthreads();
    const int per_row = p.hd * 2 / 32;                            // touches per row of one head
    uint32_t sink = 0;
    for (int c = 0; c < combos; ++c) {
        const int live = c_live[c];
        const int64_t base = c_base[c];
        for (int i = tid; i < live * per_row; i += nthreads) {
            const int64_t off = base + (i / per_row) * row + (i % per_row) * 16;
            sink |= *reinterpret_cast<const uint32_t*>(p.k_cache + off);
            sink |= *reinterpret_cast<const uint32_t*>(p.v_cache + off);
        }
    }
    asm volatile("" ::"v"(sink));
}

// PM = most pieces one wave may own (4; 5 where it lets a split-K launch fit the 256 CUs with equal workgroups)
// WIDE (SILU only): a tile is 24 gate + 24 up columns held as four 16-row blocks {gate 0-15, gate 16-23, up 0-15, up 16-23}
// (the half blocks fetch 8 rows; their other 8 MFMA columns are never stored).  4864 / 24 -> 203 equal workgroups on the 256
// CUs instead of 304 workgroups of 16 + 16 columns, of which 48 CUs carried two.
// NWV = waves per workgroup (4, or 8 with half the pieces per wave: two waves per SIMD overlap each other's issue phases)
// (p_pq, p_pr) = K pieces per split: quotient and remainder, so that no workgroup divides on the way to its first DMA
// RESADD launches use neither `inter` nor `hd`: their two preloaded slots carry M and the row stride of the residual stream instead, and
// the 13th / 14th preloaded dwords its address, so that the residual operand can be requested right behind the weight DMA without a
// kernarg round trip (14 dwords = the preload count of build.py)
#define LS_FLAT(a) (a).w, (a).x, (a).K, (a).N, (a).ksplit, (a).tiles, ((a).epi == EPI_RESADD ? (a).M : (a).inter),       \
                   ((a).epi == EPI_RESADD ? (int)(a).out_stride : (a).hd), ((a).K / 64) / (a).ksplit, ((a).K / 64) % (a).ksplit, (a).out, (a)
template <int MT, int EPI, int NORM, bool XPACK, bool MULTI, int PM = 4, bool WIDE = false, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void linear_stream_kernel(
    // what the first W DMA and the x loads need comes first and flat: with -amdgpu-kernarg-preload-count these are in SGPRs when the
    // wave starts instead of behind a kernarg s_load (build.py); the rest of the descriptor follows by reference
    const uint16_t* __restrict__ p_w, const uint16_t* __restrict__ p_x, const int p_K, const int p_N, const int p_ksplit, const int p_tiles,
    const int p_inter, const int p_hd, const int p_pq, const int p_pr, void* const p_out, const LinearArgs a) {
    constexpr int SW = NWV;                                                    // (shadows the file-scope default of 4)
    constexpr int TPB = NWV * 64;                                              // threads per workgroup
    constexpr int EPT = (MT * 256 + TPB - 1) / TPB;                            // output element slots per thread
    static_assert(!WIDE || (EPI == EPI_SILU && !MULTI), "wide tiles: SiLU gate_up, one tile per workgroup");
    constexpr int PMAX = PM;
    constexpr int NB = WIDE ? 4 : (EPI == EPI_SILU || EPI == EPI_ROPE) ? 2 : 1;   // weight row blocks per tile
    constexpr int NBUF = MULTI ? 2 : 1;
    constexpr int STAGE = NB * PMAX * 2048;                                    // W staging bytes per wave per buffer
    constexpr int PSTRIDE = NB * MT * 256 + MT * 16;                           // floats per (tile, split) partial record
    static_assert(NB * MT * 1024 <= STAGE, "the wave's reduction tile aliases the staging buffer it has just consumed");
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SW * NBUF * STAGE + SW * MT * 16 * 4 + 16];
    typedef float red_t[NB][MT][64][4];
    typedef float ss_t[MT][16];
    ss_t* const lds_ss = reinterpret_cast<ss_t*>(lds_raw + SW * NBUF * STAGE);
    unsigned* const lds_ticket = reinterpret_cast<unsigned*>(lds_raw + SW * NBUF * STAGE + SW * MT * 16 * 4);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    if constexpr (!MULTI) {
        if ((int)blockIdx.x >= p_tiles) {                        // workgroup-uniform: this workgroup only prefetches (launcher: pf_ptr set)
            const int pf_idx = (int)(blockIdx.y * (gridDim.x - p_tiles) + (blockIdx.x - p_tiles)), pf_n = (int)((gridDim.x - p_tiles) * gridDim.y);
            if (a.pf_kv.k_cache) prefetch_kv_images(a.pf_kv, pf_idx, pf_n, tid, TPB, lds_raw);     // (first: the attention launch is the next one)
            if (a.pf_ptr) prefetch_lines(a.pf_ptr, a.pf_bytes, pf_idx, pf_n, tid, TPB);
            return;
        }
    }
    LS_STAMP(0);

    // ---- this wave's K range: the tile's pieces are split over the ksplit workgroups, then over the 4 waves
    // (the first p_pr splits carry one piece more; integer division costs hundreds of cycles here, in front of the first DMA)
    const int split = blockIdx.y;
    const int wp0 = p_pq * split + min(split, p_pr), wp1 = wp0 + p_pq + (split < p_pr ? 1 : 0);
    const int p0 = __builtin_amdgcn_readfirstlane(wp0 + (wp1 - wp0) * wave / SW);
    const int np = __builtin_amdgcn_readfirstlane(wp0 + (wp1 - wp0) * (wave + 1) / SW - p0);      // 0..PMAX (host guarantees)
    const int KS = p_K / 32;
    const int tile_first = MULTI ? (int)((unsigned)(p_tiles * blockIdx.x) / kMultiWgs) : (int)blockIdx.x;   // (tiles * 512 < 2^31: host)
    const int tile_end = MULTI ? (int)((unsigned)(p_tiles * (blockIdx.x + 1)) / kMultiWgs) : tile_first + 1;

    // ---- W DMA: one instruction = 8 rows x 128 B (one whole line per row); LDS image of a piece = [16 rows][128 B] with the
    // 16-byte chunk order XOR-swizzled on the SOURCE so the operand reads are conflict free (as skinny_gemm.hip)
    const int dr = lane >> 3, dp = lane & 7;
    const int rswz = (lq >> 1) & 7;
    auto tile_rows = [&](int tile, int& n0, int& n1, int& head, int& hi0) {
        n1 = 0; head = 0; hi0 = 0;
        if constexpr (EPI == EPI_ROPE) {
            const int ph_shift = p_hd == 128 ? 2 : 1;                // tiles per head (hd / 32 = 2 or 4): columns i and i + D/2 together
            head = tile >> ph_shift;
            hi0 = 16 * (tile & ((1 << ph_shift) - 1));
            n0 = head * p_hd + hi0;
            n1 = n0 + p_hd / 2;
        } else {
            n0 = tile * 16;
            if constexpr (EPI == EPI_SILU) n1 = p_inter + n0;
        }
    };
    auto issue_w = [&](int tile, int buf) {
        int n0, n1, head, hi0;
        tile_rows(tile, n0, n1, head, hi0);
        unsigned char* const stage = lds_raw + (wave * NBUF + buf) * STAGE;
#pragma unroll
        for (int pi = 0; pi < PMAX; ++pi) {
            if (pi < np) {                                           // wave-uniform
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int hh = 0; hh < (WIDE && (nb & 1) ? 1 : 2); ++hh) {      // WIDE: blocks 1 and 3 are the 8-row halves
                        const int row = 8 * hh + dr;
                        int wrow;                                    // weight row (= output column) this lane fetches
                        if constexpr (WIDE) {
                            wrow = (nb >> 1) * p_inter + 24 * tile + 16 * (nb & 1) + row;
                            wrow = wrow < p_N ? wrow : p_N - 1;      // the last tile's half blocks run past the matrix: clamped, unused
                        } else {
                            wrow = (nb == 0 ? n0 : n1) + row;
                        }
                        const uint16_t* src = p_w + (int64_t)wrow * p_K + (p0 + pi) * 64 + (dp ^ ((row >> 1) & 7)) * 8;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(stage + ((pi * NB + nb) * 2 + hh) * 1024), 16, 0, NVH_DMA_AUX);
                    }
            }
        }
    };

    issue_w(tile_first, 0);
    LS_STAMP(1);

    // residual operand of RESADD for the elements this thread finishes (value v = tid + TPB j, see below), requested between the weight DMA
    // and the x loads from preloaded arguments only (address, M, row stride: LS_FLAT).  It used to follow the x loads behind a kernarg
    // round trip and was waited for on the spot: the last load of the launch was issued ~0.5 us late, and the MFMAs with it.  Rows past M
    // read row M - 1 (in bounds, never used): no lane-conditional load, hence no conversion-and-wait in a block of its own
    uint16_t resid[EPT];
    if constexpr (EPI == EPI_RESADD && !MULTI) {
        const int r_m = p_inter, r_stride = p_hd;                             // (see LS_FLAT)
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int v = tid + TPB * j, l = (v >> 2) & 63;
            const int row = min(16 * (v >> 8) + 4 * (l >> 4) + (v & 3), r_m - 1);
            resid[j] = reinterpret_cast<const uint16_t*>(p_out)[(int64_t)row * r_stride + tile_first * 16 + (l & 15)];
        }
    }
    // RoPE epilogue operands of the elements this thread finishes, first half: the loads that depend on nothing but the kernel
    // arguments (bias pair as raw bf16, cache slot, POSITION).  They are issued here, between the weight DMA and the x loads: the
    // cos / sin loads further down hang on the position, and waiting for a position that was requested after the x loads meant
    // draining every x load and then paying a whole round trip with nothing in flight, right in front of the MFMAs
    uint16_t rp_b1[EPT], rp_b2[EPT];
    int64_t rp_pos[EPT];
    int rp_slot[EPT];
    if constexpr (EPI == EPI_ROPE && !MULTI) {
        int n0, n1, head, hi0;
        tile_rows(tile_first, n0, n1, head, hi0);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int v = tid + TPB * j, l = (v >> 2) & 63, c = l & 15;
            const int row = min(16 * (v >> 8) + 4 * (l >> 4) + (v & 3), a.M - 1);
            rp_pos[j] = a.positions[row];
            rp_slot[j] = head >= a.h ? a.slots[row] : 0;
            // unconditional (a launch without bias reads two weights instead and never uses them): a load inside `if (a.bias)` made the
            // compiler convert the value in that very block, i.e. wait for it there with vmcnt(0) — draining the weight DMA in front of the x loads
            const uint16_t* const bsrc = a.bias ? a.bias : p_w;
            rp_b1[j] = bsrc[n0 + c];
            rp_b2[j] = bsrc[n1 + c];
        }
    }

    // ---- x fragments of this wave's K range, kept in registers for every tile of the workgroup
    u32x4 araw[2 * PMAX][MT];
#pragma unroll
    for (int pi = 0; pi < PMAX; ++pi) {
        if (pi < np) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ks = 2 * (p0 + pi) + j;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if constexpr (XPACK) {
                        araw[2 * pi + j][m] = *reinterpret_cast<const u32x4*>(p_x + (((int64_t)m * KS + ks) * 64 + lane) * 8);
                    } else {
                        const int r = 16 * m + lq;                   // rows past M repeat the last row; discarded
                        araw[2 * pi + j][m] = *reinterpret_cast<const u32x4*>(p_x + (int64_t)(r < a.M ? r : a.M - 1) * a.x_stride + ks * 32 + lg * 8);
                    }
                }
            }
        }
    }
    // RoPE operands, second half: cos and sin of the row's position (the position was requested before the x loads: waiting for it here
    // leaves those in flight); not needed before the epilogue, see the counted wait in front of the MFMAs
    float rp_co[EPT], rp_si[EPT];
    if constexpr (EPI == EPI_ROPE && !MULTI) {
        int n0, n1, head, hi0;
        tile_rows(tile_first, n0, n1, head, hi0);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int c = ((tid + TPB * j) >> 2) & 15;
            const float* cs = a.cos_sin + rp_pos[j] * p_hd;
            rp_co[j] = cs[hi0 + c];
            rp_si[j] = cs[p_hd / 2 + hi0 + c];
        }
    }
    LS_STAMP(2);
    // the epilogue's descriptor fields are fetched HERE, under the weight stream: left to the compiler their kernarg loads sit
    // right in front of the first use (a scalar round trip in the tail of every launch)
    void* const e_out = a.out;
    uint16_t* const e_out_packed = a.out_packed;
    const int64_t e_out_stride = a.out_stride;
    float* const e_ws = a.ws;
    unsigned* const e_counters = a.counters;
    const float e_norm_eps = a.norm_eps;
    uint16_t* const e_k_cache = a.k_cache;
    uint16_t* const e_v_cache = a.v_cache;
    asm volatile("" ::"s"(e_out), "s"(e_out_packed), "s"(e_out_stride), "s"(e_ws), "s"(e_counters), "s"(e_norm_eps), "s"(e_k_cache), "s"(e_v_cache));

    float ss2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ss2[m] = 0.f;
    bool ss_done = false;
    float best_v[EPT];                                          // greedy candidates (NONE + cand_val): running max per owned element slot
    int best_i[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { best_v[j] = -INFINITY; best_i[j] = 0x7fffffff; }

    if constexpr (MULTI) {
        // The x fragments are loaded once and used by every tile.  Left alone, the compiler waits for them inside the loop — with vmcnt(0),
        // because the DMA counts behind them are run-time values — in EVERY iteration, i.e. it drains the next tile's weight DMA that was
        // issued at the top of the iteration to travel under this tile's work (the double buffer never overlapped: found in the ISA of the
        // LM head).  One full wait here (the first tile needs x and its weights anyway), then the fragments pass an asm pin and are plain
        // register values for the loop, whose only vector-memory waits are then the counted ones.
        wait_vm<0>();
#pragma unroll
        for (int i = 0; i < 2 * PMAX; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(araw[i][m]));
    }
    for (int tile = tile_first, buf = 0; tile < tile_end; ++tile, buf ^= (NBUF - 1)) {
        if (MULTI && tile + 1 < tile_end) {
            issue_w(tile + 1, buf ^ 1);
            wait_all_but_pieces<NB * 2>(np);
        } else {
            // RoPE epilogue: the cos / sin loads hang on positions[row] and are therefore issued last, a round trip behind everything
            // else; they are not needed before the epilogue, so the wait in front of the MFMAs leaves the 2 * EPT youngest vector-memory
            // operations in flight (every weight DMA and x load is older than they are: the DMA is issued first of all)
            wait_vm<(EPI == EPI_ROPE && !MULTI) ? 2 * EPT : 0>();
        }
        if (tile == tile_first) LS_STAMP(3);
        unsigned char* const stage = lds_raw + (wave * NBUF + buf) * STAGE;
        f32x4 acc[NB][MT];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[nb][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pi = 0; pi < PMAX; ++pi) {
            if (pi < np) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 braw[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        braw[nb] = *reinterpret_cast<const u32x4*>(stage + (pi * NB + nb) * 2048 + lq * 128 + (((4 * j + lg) ^ rswz) * 16));
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const u32x4 av = araw[2 * pi + j][m];
                        if constexpr (NORM == 2) {
                            if (!ss_done) {
#pragma unroll
                                for (int w = 0; w < 4; ++w) ss2[m] = dot2_bf16(av[w], av[w], ss2[m]);
                            }
                        }
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&av);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *reinterpret_cast<const bf16x8*>(&braw[nb]), acc[nb][m], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // this wave's staging reads are done: reuse the buffer
        if (tile == tile_first) LS_STAMP(4);
        // ---- reduce the 4 waves through LDS (the wave's reduction tile aliases the staging buffer it has just consumed)
        if constexpr (NORM == 2) {
            if (!ss_done) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float t = sum_xor16(ss2[m]);                              // fold the 4 lane groups (k sub-blocks) of the row
                    t = sum_xor32(t);
                    if (lg == 0) lds_ss[wave][m][lq] = t;
                }
                ss_done = true;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>((*reinterpret_cast<red_t*>(stage))[nb][m][lane]) = acc[nb][m];
        if constexpr (MULTI) lds_barrier(); else __syncthreads();
        int n0, n1, head, hi0;
        tile_rows(tile, n0, n1, head, hi0);
        // value v = (m tile, lane, r): product[16 * mt + 4 * (lane >> 4) + r][column lane & 15 of each weight row block]
        float s[EPT][NB], rowss[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int v = tid + TPB * j, mt = min(v >> 8, MT - 1), l = (v >> 2) & 63, r = v & 3;   // (slots past MT*256 are never stored)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                s[j][nb] = 0.f;
#pragma unroll
                for (int w = 0; w < SW; ++w) s[j][nb] += (*reinterpret_cast<const red_t*>(lds_raw + (w * NBUF + buf) * STAGE))[nb][mt][l][r];
            }
            rowss[j] = 0.f;
            if constexpr (NORM == 2) {
#pragma unroll
                for (int w = 0; w < SW; ++w) rowss[j] += lds_ss[w][mt][4 * (l >> 4) + r];
            }
        }
        if (p_ksplit > 1) {
            // ---- split-K: publish the partial, take a ticket; the last arriver sums all partials in split order
            float* const part = e_ws + ((int64_t)tile * p_ksplit + split) * PSTRIDE;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                if (tid + TPB * j < MT * 256) {                  // (8 waves, one row tile: the upper half of the threads owns no element)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) st_sc1(part + nb * MT * 256 + tid + TPB * j, s[j][nb]);
                }
            }
            if constexpr (NORM == 2) {
                if (tid < MT * 16) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < SW; ++w) t += lds_ss[w][tid >> 4][tid & 15];
                    st_sc1(part + NB * MT * 256 + tid, t);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(&e_counters[tile * kTicketWords], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *lds_ticket = old;
            }
            __syncthreads();
            if (*lds_ticket != (unsigned)p_ksplit - 1) return;       // workgroup-uniform; MULTI never splits K
            // zero the ticket for the next launch: behind the barrier (whose wait would otherwise hold the workgroup until the store
            // is acknowledged), so that it completes under the partial loads
            if (tid == 0) __hip_atomic_store(&e_counters[tile * kTicketWords], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float* const base = e_ws + (int64_t)tile * p_ksplit * PSTRIDE;
            // partials are requested SB splits at a time, all loads of a batch in flight together; up to four splits (the
            // balanced down_proj shape) are one batch instead of an 8-wide one with half of it repeated
            auto sum_splits = [&](auto sb_tag) {
                constexpr int SB = decltype(sb_tag)::value;
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    const int v = tid + TPB * j, mt = v >> 8, l = (v >> 2) & 63, r = v & 3;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) s[j][nb] = 0.f;
                    rowss[j] = 0.f;
                    if (v >= MT * 256) continue;
                    for (int sp0 = 0; sp0 < p_ksplit; sp0 += SB) {
                        float tmp[SB][NB + 1];
#pragma unroll
                        for (int i = 0; i < SB; ++i) {
                            const int sp = sp0 + i < p_ksplit ? sp0 + i : p_ksplit - 1;
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) tmp[i][nb] = ld_sc1(base + (int64_t)sp * PSTRIDE + nb * MT * 256 + v);
                            tmp[i][NB] = NORM == 2 ? ld_sc1(base + (int64_t)sp * PSTRIDE + NB * MT * 256 + mt * 16 + 4 * (l >> 4) + r) : 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < SB; ++i) {
                            if (sp0 + i < p_ksplit) {
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb) s[j][nb] += tmp[i][nb];
                                rowss[j] += tmp[i][NB];
                            }
                        }
                    }
                }
            };
            if (p_ksplit <= 4) sum_splits(std::integral_constant<int, 4>{});
            else sum_splits(std::integral_constant<int, 8>{});
        }
        if (tile == tile_first) LS_STAMP(5);
        // ---- epilogue
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int v = tid + TPB * j, mt = v >> 8, l = (v >> 2) & 63, r = v & 3;
            const int row = 16 * mt + 4 * (l >> 4) + r;
            if (row >= a.M) continue;
            const int c = l & 15;
            float y[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) y[nb] = s[j][nb];
            if constexpr (NORM == 2) {                                 // x.(g*W)^T * rsqrt(mean(x^2)+eps) == RMSNorm(x).W^T without the two bf16 roundings
                const float inv_row = rsqrtf(rowss[j] / p_K + e_norm_eps);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) y[nb] *= inv_row;
            }
            __bf16* const out = reinterpret_cast<__bf16*>(e_out);
            if constexpr (EPI == EPI_SILU && WIDE) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {                           // full block (16 columns), half block (8 columns)
                    const int col = 24 * tile + 16 * hb + c;
                    if ((hb == 1 && c >= 8) || col >= p_inter) continue;
                    const float g = (float)(__bf16)y[hb], u = (float)(__bf16)y[2 + hb];
                    const __bf16 o = (__bf16)((float)(__bf16)(g / (1.f + __expf(-g))) * u);
                    if (out) out[(int64_t)row * e_out_stride + col] = o;
                    if (e_out_packed) e_out_packed[pack_index(row, col, p_inter)] = __builtin_bit_cast(uint16_t, o);
                }
            } else if constexpr (EPI == EPI_SILU) {
                const float g = (float)(__bf16)y[0], u = (float)(__bf16)y[1];          // the projection output is bf16 in the reference
                const __bf16 o = (__bf16)((float)(__bf16)(g / (1.f + __expf(-g))) * u);
                if (out) out[(int64_t)row * e_out_stride + n0 + c] = o;
                if (e_out_packed) e_out_packed[pack_index(row, n0 + c, p_inter)] = __builtin_bit_cast(uint16_t, o);
            } else if constexpr (EPI == EPI_RESADD) {
                __bf16* p = out + (int64_t)row * e_out_stride + n0 + c;
                uint32_t rraw = resid[j];
                asm volatile("" : "+v"(rraw));                                 // the bf16 -> f32 shift stays here, behind the MFMAs
                const float old = MULTI ? (float)*p : __builtin_bit_cast(float, rraw << 16);
                const __bf16 o = (__bf16)(y[0] + old);
                *p = o;
                if (e_out_packed) e_out_packed[pack_index(row, n0 + c, p_N)] = __builtin_bit_cast(uint16_t, o);
            } else if constexpr (EPI == EPI_ROPE) {
                float x1 = y[0], x2 = y[1];
                if (a.bias) {                                              // (x + 0.f would also be exact, but keep the no-bias path add-free)
                    uint32_t b1 = rp_b1[j], b2 = rp_b2[j];
                    asm volatile("" : "+v"(b1), "+v"(b2));                  // the bf16 -> f32 shifts stay here, behind the MFMAs
                    x1 += __builtin_bit_cast(float, b1 << 16);
                    x2 += __builtin_bit_cast(float, b2 << 16);
                }
                x1 = (float)(__bf16)x1;                                    // the projection output is bf16 in the reference
                x2 = (float)(__bf16)x2;
                const int i = hi0 + c;                                     // index inside the half head
                float y1 = x1, y2 = x2;
                if (head < a.h + a.kvh) {                                  // q or k head: rotate (products and sums rounded separately)
                    const float co = rp_co[j], si = rp_si[j];
                    const float p1 = x1 * co, p2 = x2 * si, p3 = x2 * co, p4 = x1 * si;
                    y1 = p1 - p2;
                    y2 = p3 + p4;
                }
                if (head < a.h) {
                    __bf16* q = out + (int64_t)row * e_out_stride + head * p_hd + i;
                    q[0] = (__bf16)y1;
                    q[p_hd / 2] = (__bf16)y2;
                } else {
                    const int slot = rp_slot[j];
                    if (slot >= 0) {
                        const bool is_v = head >= a.h + a.kvh;
                        __bf16* dst = reinterpret_cast<__bf16*>(is_v ? e_v_cache : e_k_cache) +
                                      ((int64_t)slot * a.kvh + (head - a.h - (is_v ? a.kvh : 0))) * p_hd + i;
                        dst[0] = (__bf16)y1;
                        dst[p_hd / 2] = (__bf16)y2;
                    }
                }
            } else {
                const __bf16 o = (__bf16)(y[0] + (a.bias ? (float)__builtin_bit_cast(__bf16, a.bias[n0 + c]) : 0.f));
                if (out) out[(int64_t)row * e_out_stride + n0 + c] = o;
                if (e_out_packed) e_out_packed[pack_index(row, n0 + c, p_N)] = __builtin_bit_cast(uint16_t, o);
                if (a.cand_val && argmax_better((float)o, n0 + c, best_v[j], best_i[j])) {   // torch.argmax order (NaN first, lowest column on ties)
                    best_v[j] = (float)o;
                    best_i[j] = n0 + c;
                }
            }
        }
        if constexpr (MULTI) lds_barrier();                           // reduction tiles read before the next-but-one DMA lands on them
    }
    if constexpr (EPI == EPI_NONE) {
        if (a.cand_val) {
            // element slot j of thread tid is (row = 16*mt + 4*(l>>4) + r, column l&15) with l = (v>>2)&63, v = tid + 256 j: the 16
            // columns of one row sit in one wave at lane stride 4 -> fold lanes 4, 8, 16, 32 apart (value desc, column asc)
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                float bv = best_v[j];
                int bi = best_i[j];
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) {
                    const float ov = __shfl_xor(bv, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                }
                const int v = tid + TPB * j, l = (v >> 2) & 63;
                const int row = 16 * (v >> 8) + 4 * (l >> 4) + (v & 3);
                if ((l & 15) == 0 && row < a.M) {
                    a.cand_val[(int64_t)blockIdx.x * a.cand_stride + row] = bv;
                    a.cand_idx[(int64_t)blockIdx.x * a.cand_stride + row] = bi;
                }
            }
        }
    }
    LS_STAMP(6);
}

// grid of a single-tile-per-workgroup launch: (tiles, ksplit), widened in x by prefetch workgroups for the CUs it leaves idle
static dim3 grid_of(const LinearArgs& a) {
    int extra = 0;
    if ((a.pf_ptr && a.pf_bytes >= 128) || a.pf_kv.k_cache) {
        const int spare = device_cus_ls() - a.tiles * a.ksplit;
        if (spare >= 8) extra = spare / a.ksplit;
    }
    return dim3(a.tiles + extra, a.ksplit);
}

template <int MT, int EPI, int NORM, bool XPACK>
int launch_x(const LinearArgs& a_in, hipStream_t stream) {
    LinearArgs a = a_in;
    const bool multi = EPI == EPI_NONE && a.ksplit == 1 && a.tiles > 1024;
    if (multi) {
        if constexpr (EPI == EPI_NONE) {
            // (eight waves per workgroup measured slower here: 72 vs 49 us for the LM head)
            hipLaunchKernelGGL((linear_stream_kernel<MT, EPI, NORM, XPACK, true>), dim3(kMultiWgs), dim3(SW * 64), 0, stream, LS_FLAT(a));
        }
        return check_launch("linear_stream");
    }
    if constexpr (EPI == EPI_SILU) {
        const int wide_tiles = (a.inter + 23) / 24;                  // 24 + 24 columns per workgroup
        if (a.ksplit == 1 && a.tiles > 256 && wide_tiles <= 256 && a.inter % 8 == 0) {
            a.tiles = wide_tiles;
            hipLaunchKernelGGL((linear_stream_kernel<MT, EPI, NORM, XPACK, false, 2, true, 8>), grid_of(a), dim3(8 * 64), 0, stream, LS_FLAT(a));
            return check_launch("linear_stream");
        }
    }
    if constexpr (EPI == EPI_RESADD) {
        // Every workgroup streams at about the same rate, so the most loaded CU sets the load phase: prefer a split that
        // puts at most one (equal) workgroup on each of the 256 CUs.  down_proj of Qwen2-0.5B: 56 tiles x 5 splits = 280
        // workgroups (24 CUs doubled) with 4 pieces per wave, 56 x 4 = 224 with 5.
        const int pieces = a.K / 64, ks5 = (pieces + SW * 5 - 1) / (SW * 5);
        if (a.ksplit > 1 && a.tiles * a.ksplit > 256 && a.tiles * ks5 <= 256) {
            a.ksplit = ks5;
            hipLaunchKernelGGL((linear_stream_kernel<MT, EPI, NORM, XPACK, false, 3, false, 8>), grid_of(a), dim3(8 * 64), 0, stream, LS_FLAT(a));
            return check_launch("linear_stream");
        }
    }
    // eight waves per workgroup (two per SIMD) by default: measured -4.7 % on the decode step against four
    // (an LM head too deep for the multi-tile form, e.g. K = 3584: tens of thousands of workgroups, where four waves with four
    // pieces each measured 262 us against 326 us for eight with two)
    if (!(EPI == EPI_NONE && a.tiles > 1024)) {
        hipLaunchKernelGGL((linear_stream_kernel<MT, EPI, NORM, XPACK, false, 2, false, 8>), grid_of(a), dim3(8 * 64), 0, stream, LS_FLAT(a));
        return check_launch("linear_stream");
    }
    hipLaunchKernelGGL((linear_stream_kernel<MT, EPI, NORM, XPACK, false>), grid_of(a), dim3(SW * 64), 0, stream, LS_FLAT(a));
    return check_launch("linear_stream");
}

template <int MT, int EPI>
int launch_e(const LinearArgs& a, hipStream_t stream) {
    if (a.norm_mode == 2) return a.x_packed ? launch_x<MT, EPI, 2, true>(a, stream) : launch_x<MT, EPI, 2, false>(a, stream);
    return a.x_packed ? launch_x<MT, EPI, 0, true>(a, stream) : launch_x<MT, EPI, 0, false>(a, stream);
}

template <int MT>
int launch_mt(const LinearArgs& a, hipStream_t stream) {
    switch (a.epi) {
        case EPI_NONE: return launch_e<MT, EPI_NONE>(a, stream);
        case EPI_SILU: return launch_e<MT, EPI_SILU>(a, stream);
        case EPI_RESADD: return launch_e<MT, EPI_RESADD>(a, stream);
        case EPI_ROPE: return launch_e<MT, EPI_ROPE>(a, stream);
    }
    return -100;
}

int tiles_of(int n, int inter, int h, int kvh, int hd, int epi) {
    if (epi == EPI_ROPE) return (h + 2 * kvh) * (hd / 32);
    if (epi == EPI_SILU) return inter / 16;
    return n / 16;
}

}  // namespace

int linear_stream_candidate_groups(int n, int k) {
    if (k % 64 != 0 || k / 64 > SW * PMAX || n % 16 != 0) return 0;    // one workgroup must see the whole K range
    const int tiles = n / 16;
    return tiles > 1024 ? (int)kMultiWgs : tiles;
}

// counters [tiles], ONE PER 128-BYTE LINE (the memory side executes the adds on one line one after the other: dense tickets made every
// tile's last arriver queue behind its neighbours' adds), then partial records [tiles][ksplit][NB*MT*256 + MT*16] fp32; 0 when K needs no split
size_t linear_stream_workspace_bytes(int m, int n, int k, int epi) {
    const int pieces = k / 64, ksplit = (pieces + SW * PMAX - 1) / (SW * PMAX);
    if (ksplit <= 1) return 0;
    const int nb = (epi == EPI_SILU || epi == EPI_ROPE) ? 2 : 1, mt = (m + 15) / 16;
    const int tiles = nb == 2 ? n / 32 : n / 16;
    return (size_t)tiles * kTicketWords * 4 + (size_t)tiles * ksplit * (nb * mt * 256 + mt * 16) * 4;
}

// Returns -100 when the call is not this kernel's (the caller falls back to skinny_gemm.hip's loop kernel).
int launch_linear_stream(const LinearArgs& a_in, hipStream_t stream) {
    LinearArgs a = a_in;
    if (a.M == 0) return 0;
    if (a.norm_mode == 1 || a.K % 64 != 0 || a.M > 64) return -100;
    if (a.cand_val && (a.epi != EPI_NONE || !a.cand_idx || linear_stream_candidate_groups(a.N, a.K) == 0)) return -100;
    a.tiles = tiles_of(a.N, a.inter, a.h, a.kvh, a.hd, a.epi);
    const int pieces = a.K / 64;
    a.ksplit = (pieces + SW * PMAX - 1) / (SW * PMAX);
    if (a.ksplit > 1) {
        const size_t counters_bytes = (size_t)a.tiles * kTicketWords * 4;
        const int nb = (a.epi == EPI_SILU || a.epi == EPI_ROPE) ? 2 : 1, mt = (a.M + 15) / 16;
        if (!a.ws_raw || a.ws_bytes < counters_bytes + (size_t)a.tiles * a.ksplit * (nb * mt * 256 + mt * 16) * 4) return -100;
        a.counters = reinterpret_cast<unsigned*>(a.ws_raw);
        a.ws = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(a.ws_raw) + counters_bytes);
    }
    switch ((a.M + 15) / 16) {
        case 1: return launch_mt<1>(a, stream);
        case 2: return launch_mt<2>(a, stream);
        case 3: return launch_mt<3>(a, stream);
        case 4: return launch_mt<4>(a, stream);
    }
    return -100;
}

}  // namespace nvh
