// Variable-length causal prefill attention on MFMA tiles (gfx950).
// Replaces flash_attn_varlen_func at nanovllm/layers/attention.py:93-96, both modes:
//   PAGED = false : K/V are this step's packed rows            (oracle attention_sdpa.py:65-119)
//   PAGED = true  : K/V live in the paged cache via block_tables (prefix-cache hit, attention.py:90-91;
//                   bottom-right aligned causal mask; parity unpinned by the reference)
//
// Dense contraction -> MFMA (v_mfma_f32_16x16x32_bf16), flash-style online softmax, fp32 accumulate.
// Workgroup = 4 waves = a tile of BM = 64*QT query rows of one (sequence, q head); wave w owns rows 16*QT*w .. +16*QT-1 as QT
// sub-tiles of 16 rows (QT = 1 or 2, picked per launch from head_dim and the longest sequence).
// K/V tiles of BN = 64 keys are staged through LDS by LDS-DMA (global_load_lds_dwordx4: one instruction = 1 KiB of
// whole, coalesced key rows, no VGPR round trip), DOUBLE-BUFFERED: the DMA of tile i+1 is in flight while tile i is
// computed; waves wait with a counted vmcnt and meet at raw s_barriers (a __syncthreads() would drain the prefetch).
// LDS images are row-major [64 keys][D bf16]; the 16-byte chunk order inside a row is XOR-swizzled on the SOURCE
// address so the MFMA operand reads are bank-conflict free (same images as the decode kernel).
//
// Orientation (what makes the softmax and the P operand lane-local):
//   S^T = K Q^T     A = K rows (ds_read_b128), B = Q rows (registers, loaded once).
//                   C layout: lane l, reg r holds S^T[key 4(l>>4)+r][query l&15]
//                   -> a query's scores sit in 4 registers x 4 lane groups: row max/sum are local ops plus a
//                   2-step permlane reduction; the rescale factor is one scalar per lane.
//   O^T = V^T P^T   B = P^T straight from the S^T accumulators of two 16-key tiles (k-slot j<4 -> key 4g+j,
//                   j>=4 -> key 16+4g+j-4, g = l>>4); A = V^T in the SAME k order, produced by ds_read_b64_tr_b16
//                   (hardware transpose of a 4-key x 16-dim block) from row-major V.  P enters as hi + lo bf16
//                   halves (two MFMAs per tile) so it keeps ~16 mantissa bits (1e-3 parity bar at |o| ~ 3).
//                   C layout: lane l, reg r holds O^T[dim 16t+4(l>>4)+r][query l&15] -> 4 contiguous dims per lane.
// Two forms of P V (template parameter PV of the kernel): the exact one above (nvh_prefill_varlen) and, for nvh_prefill_varlen_pv16, P rounded to fp16
// against an fp16 copy of V — ONE v_mfma_f32_16x16x32_f16 per operand pair, a third fewer vector instructions per tile (the loop is bound by vector
// issue) — behind a range guard: the conversion launch flags every 64 rows of V that do not fit fp16 and a flagged sequence runs the exact body.
// Workgroup order: XCD-aware (the heads x q-tiles that stream one (sequence, kv head)'s K/V share one XCD's L2), heaviest q-tiles first.
// Causal structure: a workgroup only walks the key tiles its rows can see; a wave skips the MFMA work of tiles that lie
// entirely above its own 16 rows' diagonal (it still takes part in the staging and the barriers).
// Algorithmic flops per launch: sum_seq 4*D*H*(causal pairs); bytes: Tq*H*D*2*2 + Tk*KVH*D*2*2.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DNVH_STAMPS, tools/probes/stamp_prefill.py): wave 0 of every workgroup records the clock at
// fixed points (slot k of 32 per workgroup).  Never compiled into the shipped library.
#ifdef NVH_STAMPS
#define PF_STAMP(k)                                                                                      \
    do {                                                                                                 \
        unsigned long long t_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (a.stamps && tid == 0 && (k) < 32)                                                            \
            a.stamps[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z + (gridDim.z - 1 - blockIdx.z)) * 32 + (k)] = t_; \
    } while (0)
// short-sequence kernel: 16 slots per WAVE (lane 0), [(b * kvh + kh) * NW + wave]
#define SK_STAMP(k)                                                                                      \
    do {                                                                                                 \
        unsigned long long t_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (a.stamps && lane == 0 && (k) < 16)                                                           \
            a.stamps[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 16 + (k)] = t_;      \
    } while (0)
#else
#define PF_STAMP(k) do {} while (0)
#define SK_STAMP(k) do {} while (0)
#endif

// QT = 16-row query sub-tiles per wave (template parameter): with QT = 2 every K / V fragment read from LDS feeds both
// sub-tiles' MFMAs and the per-tile scalar / DMA / barrier work is halved per flop, at half the occupancy.  Measured (lean
// loop, heaviest-first grid): D=128 +18.5 % (482 -> 571 TFLOP/s at S=4096), D=64 +6.6 % at S=4096, neutral at S=1024.
constexpr int BN = 64;      // keys per LDS tile

// ds_read_b64_tr_b16 through inline asm: the builtin form makes hipcc wait vmcnt(0) before the read (it cannot prove the read
// does not alias the LDS-DMA of the NEXT tile that is still in flight), which serialised the double buffer.  The caller
// issues a batch of these, then `s_waitcnt lgkmcnt(0)` + a scheduling fence before the first use (guide rule 18).
template <int OFF>
__device__ __forceinline__ u32x2 ds_read_tr16_b64_asm(uint32_t lds_addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF) : "memory");
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

template <int LPT>
__device__ __forceinline__ int chunk_swz(int row) {
    return LPT == 8 ? ((row >> 1) & 7) : (row & 15);
}

// The V image has its own swizzle: its transposed reads (ds_read_b64_tr_b16) bank per 32-lane half, where lanes address rows
// r and r+2 (128-byte rows) / r and r+1 (256-byte rows) on the same banks; with the K swizzle those pairs land on the same
// chunk pair (2-way conflict on every V read: a third of all LDS cycles, SQ_LDS_BANK_CONFLICT).  Even XOR values that differ
// for the 4 (8) rows of a bank class make the 32 lanes hit 32 distinct 8-byte slots.  Rows r, r+16, r+32 share a value.
template <int LPT>
__device__ __forceinline__ int chunk_swz_v(int row) {
    return LPT == 8 ? (row & 6) : ((2 * row) & 14);
}

// one v_cvt_pk_bf16_f32 (round to nearest even): {lo half = a, hi half = b}.  A vector conversion, not inline asm: hipcc pads every
// asm statement whose result the next VALU instruction reads with an s_nop, 40 of them per tile here
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
#ifdef NVH_CVT_ASM                                                                 // A/B builds only: the round-1 form
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
#endif
}

// F16V (measurement variant, NVH_PREFILL_TILED_F16V of nvh_prefill_varlen_variant): `v` holds fp16 rows (the caller converted them; same layout
// and strides), P is rounded to fp16 and P V is ONE v_mfma_f32_16x16x32_f16 per operand pair — no lo half.  Everything else is the same code.
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// grid (H, B, q-tiles) -> this workgroup's (q-tile, head, sequence).  Workgroups are dispatched x-fastest, so the q-tile index is the SLOW dimension
// and runs backwards: under the causal mask tile t costs t+1 key tiles, and dispatching the heaviest tiles of every (sequence, head) first leaves
// the light ones for the tail (census in tools/probes/stamp_prefill.py: the old x = q-tile order spent the last 40 % of the launch draining a
// few heavy workgroups that had started late)
__device__ __forceinline__ void prefill_tile_of_block(const PrefillArgs& a, int& qt, int& head, int& b) {
    qt = gridDim.z - 1 - blockIdx.z, head = blockIdx.x, b = blockIdx.y;
#ifndef NVH_PREFILL_NO_XCD_MAP
    {
        // XCD-aware order (speed only; any mapping is correct): workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its
        // own, so in the plain order the G heads x q-tiles that stream the SAME K/V rows of a (sequence, kv head) pair land on all eight
        // and each L2 fetches those rows for itself.  Re-deal: linear workgroup id -> (XCD = id % 8, index on that XCD); pair u lives on
        // XCD u % 8, and on its XCD the order is heads of the group fastest, then the XCD's pairs, then q-tiles heaviest first — so every
        // XCD still works through the same mix of heavy and light tiles.
        const int G = a.h / a.kvh, units = (int)gridDim.y * a.kvh, total = (int)(gridDim.x * gridDim.y * gridDim.z);
        if ((units & 7) == 0 && (total & 7) == 0) {
            const int lin = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
            const int xcd = lin & 7, idx = lin >> 3, per = units >> 3;
#ifdef NVH_PREFILL_PAIR_MAJOR                 // A/B builds: one pair after the other on an XCD (its K/V alone in the L2) instead of all of the XCD's pairs per q-tile rank
            const int g_in = idx % G, r = idx / G, zs = (int)gridDim.z, u = xcd + 8 * (r / zs), z = r % zs;
#else
            const int g_in = idx % G, r = idx / G, u = xcd + 8 * (r % per), z = r / per;
#endif
            b = u / a.kvh;
            head = (u - b * a.kvh) * G + g_in;
            qt = (int)gridDim.z - 1 - z;
        }
    }
#endif
}

// One workgroup's tile: query rows qt * 64 * QT .. of (sequence b, head), all the keys they can see.
// GUARD (nvh_prefill_varlen_pv16): `a.v` is the fp16 copy of V; the range flags of this sequence's rows are fetched FIRST, the first K image's DMA is
// issued behind them, and only then are they looked at — so the check costs no memory latency of its own.  Returns false (nothing but that K image
// touched) when a flag is set: the caller runs the exact body instead.
template <int D, bool PAGED, int QT, bool F16V, bool GUARD = false>
__device__ __forceinline__ bool prefill_varlen_body(const PrefillArgs& a, unsigned char* const lds, const int qt, const int head, const int b) {
    constexpr int BM = 64 * QT;                      // query rows per workgroup (4 waves x QT x 16)
    constexpr int ROWB = D * 2;                      // LDS row, bytes
    constexpr int LPT = D / 8;                       // 16-byte chunks per row
    constexpr int TPI = 64 / LPT;                    // rows per DMA instruction
    constexpr int IMG = BN * ROWB;                   // one K (or V) image: 8 KiB (D=64) / 16 KiB (D=128)
    constexpr int NI = IMG / 1024;                   // DMA instructions per image
    constexpr int NIW = NI / 4;                      // ... per wave
    constexpr int STEPS = D / 32;                    // k-steps of the QK^T contraction
    constexpr int DT = D / 16;                       // 16-dim output tiles
    constexpr int NT = BN / 16;                      // 16-key tiles of S^T per LDS tile
    constexpr int WR = 16 * QT;                      // query rows per wave
    const int q_beg = a.cu_q[b], q_end = a.cu_q[b + 1];
    const int k_beg = a.cu_k[b], k_end = a.cu_k[b + 1];
    const int sq = q_end - q_beg, sk = k_end - k_beg;
    const int q0 = qt * BM;
    if (q0 >= sq || sk <= 0) return true;
    const int kh = head / (a.h / a.kvh);
    const int shift = sk - sq;                       // bottom-right alignment: query r sees keys <= r + shift

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15;                        // query column of this lane
    const int lg = lane >> 4;                        // lane group: k-block of the operands / key rows of C

    f32x4 o[QT][DT];
    float m_run[QT], l_run[QT];                      // per query column; l_run is this lane group's share
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        m_run[qs] = -INFINITY;
        l_run[qs] = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[qs][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // keys any row of this workgroup can see: 0 .. min(sk, q0 + BM + shift) - 1
    int kv_end = q0 + BM + shift;
    if (kv_end > sk) kv_end = sk;
    const int n_tiles = (kv_end + BN - 1) / BN;
    const int wave_first_key = q0 + wave * WR + shift;            // last key the wave's FIRST row may see
    const int wave_last_key = wave_first_key + WR - 1;            // last key ANY row of this wave may see

    // ---- staging: wave w issues DMA instructions w*NIW .. w*NIW+NIW-1 of the K image and of the V image.  Source address =
    // a wave-uniform base per tile (scalar arithmetic) + a loop-invariant 32-bit lane offset (row of the tile x row stride +
    // swizzled chunk), so a full tile costs no vector integer work; only the sequence's last, ragged tile re-derives the
    // lane offsets with its rows clamped to the last key (those rows are masked).
    const int dp = lane % LPT, dr = lane / LPT;
    const int64_t kstride = PAGED ? (int64_t)a.kvh * D : a.k_row_stride;
    const int64_t vstride = PAGED ? (int64_t)a.kvh * D : a.v_row_stride;
    uint32_t koff[NIW], voff[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int R = (wave * NIW + j) * TPI + dr;
        koff[j] = (uint32_t)(R * kstride + (dp ^ chunk_swz<LPT>(R)) * 8);
        voff[j] = (uint32_t)(R * vstride + (dp ^ chunk_swz_v<LPT>(R)) * 8);
    }
    auto stage = [&](int tile, int buf, const bool do_k = true, const bool do_v = true) {
        unsigned char* kimg = lds + buf * 2 * IMG;
        unsigned char* vimg = kimg + IMG;
        const int kv0 = tile * BN;
        const uint16_t *kb, *vb;                     // wave-uniform bases of the tile's first key row, this kv head
        if constexpr (PAGED) {                       // BN divides block_size: one block per tile
            const int blk = kv0 / a.block_size;
            const int bid = a.block_tables[(int64_t)b * a.bt_row_stride + blk];
            const int64_t base = (((int64_t)bid * a.block_size + (kv0 - blk * a.block_size)) * a.kvh + kh) * D;
            kb = a.k + base;
            vb = a.v + base;
        } else {
            kb = a.k + (int64_t)(k_beg + kv0) * kstride + (int64_t)kh * D;
            vb = a.v + (int64_t)(k_beg + kv0) * vstride + (int64_t)kh * D;
        }
        const bool ragged = kv0 + BN > sk;           // wave-uniform
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int ins = wave * NIW + j;
            uint32_t ko = koff[j], vo = voff[j];
            if (ragged) {
                const int R = ins * TPI + dr;
                const int Rc = kv0 + R < sk ? R : sk - 1 - kv0;
                ko = (uint32_t)(Rc * kstride + (dp ^ chunk_swz<LPT>(R)) * 8);
                vo = (uint32_t)(Rc * vstride + (dp ^ chunk_swz_v<LPT>(R)) * 8);
            }
            if (do_k) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko),
                                                       (__attribute__((address_space(3))) void*)(kimg + ins * 1024), 16, 0, 0);
            if (do_v) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo),
                                                       (__attribute__((address_space(3))) void*)(vimg + ins * 1024), 16, 0, 0);
        }
    };

    if constexpr (GUARD) {
        // one flag per kPv16GroupRows rows of V; lane l looks at groups g0 + l, + 64, + 128, + 192 (16384 rows: the largest prefill batch), a plain loop
        // takes longer sequences.  The loads are inline asm so that no compiler-placed wait sits between them and the K image's DMA.
        const int rows_ok = k_end < a.pv16_rows ? k_end : a.pv16_rows;
        const int g0 = k_beg / kPv16GroupRows, g1 = (rows_ok + kPv16GroupRows - 1) / kPv16GroupRows;
        const int last = g1 > g0 ? g1 - 1 : g0;
        int fl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = g0 + lane + 64 * i;
            const int32_t* p = a.pv16_flags + (g < last ? g : last);
            asm volatile("global_load_dword %0, %1, off" : "=v"(fl[i]) : "v"(p) : "memory");
        }
        stage(0, 0, true, false);
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]), "+v"(fl[3]) : "n"(NIW) : "memory");
        bool bad = k_end > a.pv16_rows;              // (a caller whose total_k was short of cu_seqlens_k[batch]: no fp16 copy of those rows)
#pragma unroll
        for (int i = 0; i < 4; ++i) bad |= __builtin_amdgcn_ballot_w64(g0 + lane + 64 * i < g1 && fl[i] != 0) != 0;
        for (int g = g0 + 256 + lane; __builtin_amdgcn_ballot_w64(g < g1) != 0; g += 64)
            bad |= __builtin_amdgcn_ballot_w64(g < g1 && a.pv16_flags[g < g1 ? g : last] != 0) != 0;
        if (bad) return false;
        stage(0, 0, false, true);
    } else {
        stage(0, 0);
    }
    // ---- Q fragments (B operand of S^T): Q[row][32*step + 8*lg .. +8]; plain loads issued BEHIND the first tile's DMA, so that a workgroup
    // pays one memory latency at its start, not two in a row (the counted wait of the first tile body leaves only tile 1's DMA in flight:
    // vector-memory operations complete in order, so tile 0 and these loads have landed by then)
    int my_q[QT];                                    // query index inside the sequence, per sub-tile
    bool q_ok[QT];
    bf16x8 qf[QT][STEPS];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        my_q[qs] = q0 + wave * WR + 16 * qs + lq;
        q_ok[qs] = my_q[qs] < sq;
        const uint16_t* qp = a.q + (int64_t)(q_beg + (q_ok[qs] ? my_q[qs] : sq - 1)) * a.q_row_stride + (int64_t)head * D + lg * 8;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + st * 32);
            qf[qs][st] = *reinterpret_cast<const bf16x8*>(&raw);
        }
    }
    PF_STAMP(0);
    // tr-read: lane 4q+p of its group addresses key row q, dims 4p..4p+3.  The per-lane part of the address (row 4*lg+q,
    // swizzled chunk of dim tile t) is loop invariant: DT registers; buffer, image, 32-key half and the +16-row partner are
    // immediates of the instruction (hence the tile loop unrolled by two: the buffer index must be a compile-time constant).
    const int vq = lq >> 2, vp = lq & 3;
    uint32_t vaddr[DT];
    {
        const int R0 = 4 * lg + vq;                  // chunk_swz_v(R0) == chunk_swz_v(R0 + 16) == chunk_swz_v(R0 + 32)
        const int swz = chunk_swz_v<LPT>(R0);
#pragma unroll
        for (int t = 0; t < DT; ++t) vaddr[t] = lds_offset(lds + R0 * ROWB + (vp & 1) * 8 + (((2 * t + (vp >> 1)) ^ swz) * 16));
    }
    auto tile_body = [&](auto bufc, const int it) {
        constexpr int buf = decltype(bufc)::value;
        if (it + 1 < n_tiles) {
            stage(it + 1, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");       // tile `it` landed, tile it+1 in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PF_STAMP(2 + 3 * it);
        __builtin_amdgcn_s_barrier();                                            // every wave's share of tile `it` is in LDS
        asm volatile("" ::: "memory");
        PF_STAMP(3 + 3 * it);
        const unsigned char* kimg = lds + buf * 2 * IMG;
        const int kv0 = it * BN;
        if (kv0 <= wave_last_key) {                                              // wave-uniform: tile not entirely above the diagonal
            // ---- S^T for the four 16-key tiles of both query sub-tiles: each K fragment is read once, used QT times
            // (skipping the 16-key groups above the wave's diagonal, as the short-sequence kernel below does, measured 2-5 % SLOWER
            // here at every S >= 256, same-box A/B: the interior tiles pay for the extra branches and live ranges)
            f32x4 sT[QT][NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
                for (int qs = 0; qs < QT; ++qs) sT[qs][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int R = 16 * tt + lq;
#pragma unroll
                for (int st = 0; st < STEPS; ++st) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + R * ROWB + (((4 * st + lg) ^ chunk_swz<LPT>(R)) * 16));
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) sT[qs][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qs][st], sT[qs][tt], 0, 0, 0);
                }
            }
            // The max chain reads the accumulators through inline asm, which hipcc does not pad with the wait states a VALU read of
            // a fresh MFMA result needs; one statement that takes every accumulator as an operand (so it follows every MFMA of the
            // tile) carries them (see the same guard in paged_decode.hip, where a 16-key tile exposed the hazard).
            if constexpr (QT == 1) asm volatile("s_nop 7" : "+v"(sT[0][0]), "+v"(sT[0][1]), "+v"(sT[0][2]), "+v"(sT[0][3]));
            else asm volatile("s_nop 7" : "+v"(sT[0][0]), "+v"(sT[0][1]), "+v"(sT[0][2]), "+v"(sT[0][3]), "+v"(sT[1][0]), "+v"(sT[1][1]), "+v"(sT[1][2]), "+v"(sT[1][3]));
            static_assert(QT == 1 || QT == 2, "query sub-tiles per wave");
            // ---- online softmax in the log2 domain.  Masking only where it can matter: tiles that reach past this wave's
            // first row's diagonal or past the sequence end (wave-uniform test); interior tiles skip the per-element work.
            const bool need_mask = kv0 + BN - 1 > wave_first_key || kv0 + BN > sk;
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) {
                if (need_mask) {
                    // element (tt, r) of this lane is key kv0 + 16 tt + 4 lg + r; it is visible iff key <= min(q_pos, sk - 1), i.e. iff the
                    // CONSTANT 16 tt + r is <= a per-lane limit: one compare against an inline constant + one select per element — and only
                    // in the 16-key groups that reach past what the sub-tile's FIRST row may see (wave-uniform test): the groups below the
                    // diagonal block of a diagonal tile are visible to every row
                    const int q_pos = my_q[qs] + shift;                          // last key this lane's query may see
                    const int limit = (q_pos < sk - 1 ? q_pos : sk - 1) - kv0 - 4 * lg;
                    const int first_row_key = wave_first_key + 16 * qs;
                    const int all_see = (first_row_key < sk - 1 ? first_row_key : sk - 1) - kv0;      // keys kv0 .. kv0 + all_see: no mask needed
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        if (16 * tt + 15 > all_see) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) sT[qs][tt][r] = (16 * tt + r <= limit) ? sT[qs][tt][r] : -INFINITY;
                        }
                    }
                }
                float mx = max3(sT[qs][0][0], sT[qs][0][1], sT[qs][0][2]);
                mx = max3(mx, sT[qs][0][3], sT[qs][1][0]);
                mx = max3(mx, sT[qs][1][1], sT[qs][1][2]);
                mx = max3(mx, sT[qs][1][3], sT[qs][2][0]);
                mx = max3(mx, sT[qs][2][1], sT[qs][2][2]);
                mx = max3(mx, sT[qs][2][3], sT[qs][3][0]);
                mx = max3(mx, sT[qs][3][1], sT[qs][3][2]);
                mx = max2(mx, sT[qs][3][3]);
                mx = max_xor16(mx);
                mx = max_xor32(mx);
                mx *= a.scale_log2;                                              // scale > 0: max commutes with the scaling
                const float m_new = max2(m_run[qs], mx);
                // rows that have seen no key yet keep m = -inf; use 0 as the reference point so exp2 stays finite
                const float m_use = m_new == -INFINITY ? 0.f : m_new;
                // scale-and-subtract and the row sum on element PAIRS (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue slot;
                // the loop is VALU-bound), exp2 per element
                const f32x2 sc2 = {a.scale_log2, a.scale_log2}, mm2 = {-m_use, -m_use};
                f32x2 psum2 = {0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 sv = {sT[qs][tt][r], sT[qs][tt][r + 1]};
                        const f32x2 z = __builtin_elementwise_fma(sv, sc2, mm2);
                        const f32x2 e = {fast_exp2(z[0]), fast_exp2(z[1])};
                        sT[qs][tt][r] = e[0];
                        sT[qs][tt][r + 1] = e[1];
                        psum2 += e;
                    }
                const float psum = psum2[0] + psum2[1];
                if (__builtin_amdgcn_ballot_w64(m_new != m_run[qs]) != 0) {      // some row's max moved: rescale (wave-uniform branch)
                    const float alpha = fast_exp2(m_run[qs] - m_use);            // m_run = -inf -> 0
                    l_run[qs] *= alpha;
#pragma unroll
                    for (int t = 0; t < DT; ++t) o[qs][t] *= alpha;
                }
                l_run[qs] += psum;
                m_run[qs] = m_new;
            }
            // ---- O^T += V^T P^T, 32 keys at a time, P as hi + lo bf16; each V fragment is read once, used 2*QT times
#pragma unroll
            for (int hh = 0; hh < BN / 32; ++hh) {
                bf16x8 p_hi[QT], p_lo[QT];
                f16x8 p16[QT];
#pragma unroll
                for (int qs = 0; qs < QT; ++qs) {
                    // five VALU instructions per element PAIR: pack hi, two unpacks, one packed (exact) subtract, pack lo
                    u32x4 hraw, lraw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 pv = {sT[qs][2 * hh + (j >> 1)][2 * (j & 1)], sT[qs][2 * hh + (j >> 1)][2 * (j & 1) + 1]};
                        if constexpr (F16V) {
                            hraw[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pv, f16x2v));   // v_cvt_pk_f16_f32 (round to nearest even)
                        } else {
                            const uint32_t hp = cvt_pk_bf16(pv[0], pv[1]);
                            const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
                            const f32x2 lo = pv - hf;
                            hraw[j] = hp;
                            lraw[j] = cvt_pk_bf16(lo[0], lo[1]);
                        }
                    }
                    if constexpr (F16V) {
                        p16[qs] = __builtin_bit_cast(f16x8, hraw);
                    } else {
                        p_hi[qs] = *reinterpret_cast<const bf16x8*>(&hraw);
                        p_lo[qs] = *reinterpret_cast<const bf16x8*>(&lraw);
                    }
                }
                u32x2 vlo[DT], vhi[DT];
                if (hh == 0) {
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        vlo[t] = ds_read_tr16_b64_asm<buf * 2 * IMG + IMG>(vaddr[t]);
                        vhi[t] = ds_read_tr16_b64_asm<buf * 2 * IMG + IMG + 16 * ROWB>(vaddr[t]);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        vlo[t] = ds_read_tr16_b64_asm<buf * 2 * IMG + IMG + 32 * ROWB>(vaddr[t]);
                        vhi[t] = ds_read_tr16_b64_asm<buf * 2 * IMG + IMG + 48 * ROWB>(vaddr[t]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16V) {
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const u32x4 raw = {vlo[t][0], vlo[t][1], vhi[t][0], vhi[t][1]};
#pragma unroll
                        for (int qs = 0; qs < QT; ++qs) o[qs][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, raw), p16[qs], o[qs][t], 0, 0, 0);
                    }
                } else {
                bf16x8 vf[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const u32x4 raw = {vlo[t][0], vlo[t][1], vhi[t][0], vhi[t][1]};
                    vf[t] = *reinterpret_cast<const bf16x8*>(&raw);
                }
                // all hi products, then all lo: the two MFMAs on one accumulator are QT*DT-1 independent MFMAs apart
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) o[qs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t], p_hi[qs], o[qs][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) o[qs][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t], p_lo[qs], o[qs][t], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // this wave's LDS reads of buffer `buf` are done
        PF_STAMP(4 + 3 * it);
        __builtin_amdgcn_s_barrier();                                            // ... and everyone's: tile it+2 may overwrite it
        asm volatile("" ::: "memory");
    };
    for (int it = 0; it < n_tiles; it += 2) {
        tile_body(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_tiles) tile_body(std::integral_constant<int, 1>{}, it + 1);
    }

    PF_STAMP(31);
    // ---- finalise: total row sum over the 4 lane groups, normalise, store 4 contiguous dims per tile
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        float l = sum_xor16(l_run[qs]);
        l = sum_xor32(l);
        if (!q_ok[qs]) continue;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int64_t orow = ((int64_t)(q_beg + my_q[qs]) * a.h + head) * D;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int d0 = 16 * t + 4 * lg;
            const f32x4 r = o[qs][t] * inv;
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + orow + d0) = r;
            } else {
                u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + orow + d0) = pk;
            }
        }
    }
    return true;
}

// PV: 0 = P as hi + lo bf16 against the caller's bf16 V (exact to 6e-6; the form every other path of this file uses), 1 = fp16 P against fp16 V rows
// handed over in `v` (NVH_PREFILL_TILED_F16V, measurement variant), 2 = nvh_prefill_varlen_pv16: fp16 P against the fp16 copy of V in `v16` UNLESS the
// conversion launch ahead of this one flagged a 64-row group of THIS sequence's V rows (a finite |v| > 65504 does not fit fp16): then the hi + lo form
// on the caller's own rows.  Every conversion workgroup writes its group's flag unconditionally (nothing to clear); a wave ORs the flags that cover its
// sequence behind the first K image's DMA (GUARD above); both bodies are instantiations of the same code.
template <int D, bool PAGED, int QT, int PV = 0>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(PV == 2 && D == 64 && QT == 2 ? 4 : 1)))         // (the guarded kernel at the fp16 body's 128 registers: hipcc gives it 134 otherwise)
void prefill_varlen_kernel(const PrefillArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * BN * D * 2];     // [buffer][K | V]
    int qt, head, b;
    prefill_tile_of_block(a, qt, head, b);
    if constexpr (PV == 2) {
        static_assert(!PAGED, "the fp16 copy of V is made from packed rows");
        PrefillArgs f = a;
        f.v = a.v16;
        f.v_row_stride = a.v16_row_stride;
        if (!prefill_varlen_body<D, false, QT, true, true>(f, lds, qt, head, b)) {
            // the fall-back runs the one-sub-tile-per-wave body QT times over the workgroup's rows: its registers stay below the fp16 body's, so the
            // guard costs the fast path no occupancy (D = 64, QT = 2: 128 registers = 4 waves per SIMD; the QT = 2 hi + lo body needs 136).  The
            // abandoned K image's DMA is drained first (the exact body stages it again).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int qt2 = qt, head2 = head, b2 = b;      // (laundered: the exact body shares no value with the abandoned one, so the fast path's registers
            asm volatile("" : "+s"(qt2), "+s"(head2), "+s"(b2));     //  are not held across the branch for its sake)
            for (int sub = 0; sub < QT; ++sub) prefill_varlen_body<D, false, 1, false>(a, lds, qt2 * QT + sub, head2, b2);
        }
    } else {
        prefill_varlen_body<D, PAGED, QT, PV == 1>(a, lds, qt, head, b);
    }
}

// ---- short sequences: every key of a sequence fits in LDS (sk <= 64 * NKT) ---------------------------------------------------
// The tiled kernel above gives a (head, 64-row q-tile) to each workgroup: at S = 128 (BASELINE config 5) that is 3584 workgroups of
// 3-7 us whose load -> compute phases run in lockstep, four per CU (census: tools/probes/stamp_prefill.py --batch 128 --seq 128), and
// K / V are fetched 10x per kv head.  Here ONE workgroup per (sequence, kv head) stages the sequence's K / V images once (the same
// swizzled images, all tiles resident), then its 8 waves stream the G query heads x 16-row sub-tiles through them: no barrier after
// the staging one, the next task's Q rows are in flight while the current task computes, every HBM byte is read once.
// Task t = (head g, sub-tile); the sub-tile index is rotated by g so that the causal triangle's light and heavy sub-tiles are
// spread evenly over the waves.
#ifndef NVH_SHORT_SPLIT
#define NVH_SHORT_SPLIT 1                    // A/B builds: workgroups per (sequence, kv head) pair (the tasks dealt round-robin to them)
#endif
#ifndef NVH_SHORT_WAVES_PER_EU
#define NVH_SHORT_WAVES_PER_EU 1             // A/B builds: occupancy target of the 8-wave instantiation (register cap)
#endif
template <int D, int NKT, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(D == 64 && NW == 8 ? NVH_SHORT_WAVES_PER_EU : 1)))
void prefill_short_kernel(const PrefillArgs a) {
    constexpr int ROWB = D * 2, LPT = D / 8, TPI = 64 / LPT, IMG = BN * ROWB, NI = IMG / 1024;
    constexpr int STEPS = D / 32, DT = D / 16, NT = BN / 16;
    static_assert((NKT - 1) * 2 * IMG + IMG + 48 * ROWB < 65536, "transposed V reads address tiles through the 16-bit ds offset");
    // PV16 (nvh_prefill_varlen_pv16 on this kernel's shapes, D = 64): a third image per tile holds V as fp16, converted HERE once per workgroup (one
    // 16-byte chunk per lane) with the range check folded into the barrier that publishes it; P V then runs as one fp16 MFMA per operand pair.
    constexpr bool PV16 = D == 64;
    constexpr int V16_BASE = NKT * 2 * IMG;
    static_assert(!PV16 || V16_BASE + (NKT - 1) * IMG + 48 * ROWB < 65536, "fp16 V images within the 16-bit ds offset");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NKT * 2 * IMG + (PV16 ? NKT * IMG : 0)];   // [tile][K | V], then [tile][V as fp16]

    const int kh = blockIdx.x, b = blockIdx.y;
    const int q_beg = a.cu_q[b], q_end = a.cu_q[b + 1];
    const int k_beg = a.cu_k[b], k_end = a.cu_k[b + 1];
    const int sq = q_end - q_beg, sk = k_end - k_beg;
    if (sq <= 0 || sk <= 0) return;                  // whole workgroup, before the barrier
    const int G = a.h / a.kvh;
    const int shift = sk - sq;                       // bottom-right alignment: query r sees keys <= r + shift
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    int n_tiles = (sk + BN - 1) / BN;                // <= NKT by the caller's contract (max_seqlen_k bounds every sequence);
    if (n_tiles > NKT) n_tiles = NKT;                // a caller that understated it must not make the staging loop write past the array

    SK_STAMP(0);
    // ---- stage every K / V tile of the sequence: DMA instruction `ins` = 1 KiB of one image, dealt round-robin to the waves
    {
        const int dp = lane % LPT, dr = lane / LPT;
        for (int ins = wave; ins < n_tiles * NI; ins += NW) {                    // wave-uniform
            const int tile = ins / NI, j = ins - tile * NI;
            const int R = j * TPI + dr;                                          // row inside the tile
            const int key = tile * BN + R;
            const int kc = key < sk ? key : sk - 1;                              // rows past the end repeat the last key (masked)
            const uint16_t* kp = a.k + (int64_t)(k_beg + kc) * a.k_row_stride + (int64_t)kh * D + (dp ^ chunk_swz<LPT>(R)) * 8;
            const uint16_t* vp = a.v + (int64_t)(k_beg + kc) * a.v_row_stride + (int64_t)kh * D + (dp ^ chunk_swz_v<LPT>(R)) * 8;
            unsigned char* const kimg = lds + tile * 2 * IMG + j * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)kp, (__attribute__((address_space(3))) void*)kimg, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vp, (__attribute__((address_space(3))) void*)(kimg + IMG), 16, 0, 0);
        }
    }

    const int n_sub = (sq + 15) / 16, n_tasks = G * n_sub;
    auto task_of = [&](int t, int& head, int& q0) {
        const int g = t / n_sub;
        int sub = t - g * n_sub + g;
        sub -= (sub / n_sub) * n_sub;
        head = kh * G + g;
        q0 = 16 * sub;
    };
    // Q fragments of a task (B operand of S^T): Q[row][32 * step + 8 * lg .. + 8], plain loads to registers
    auto load_q = [&](int t, bf16x8 (&qf)[STEPS]) {
        int head, q0;
        task_of(t, head, q0);
        const int row = q0 + lq < sq ? q0 + lq : sq - 1;
        const uint16_t* qp = a.q + (int64_t)(q_beg + row) * a.q_row_stride + (int64_t)head * D + lg * 8;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + st * 32);
            qf[st] = *reinterpret_cast<const bf16x8*>(&raw);
        }
    };
    bf16x8 qf[STEPS];
    const int t_step = NW * (int)gridDim.z;          // (gridDim.z workgroups share the pair's tasks: A/B builds; 1 in the shipped launch)
    int t = wave * (int)gridDim.z + (int)blockIdx.z;
    if (t < n_tasks) load_q(t, qf);
    SK_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // this wave's share of the images (and its first Q)
    SK_STAMP(2);
    __syncthreads();                                                             // everyone's: K / V are resident from here on
    SK_STAMP(3);
    [[maybe_unused]] int sk_round = 0;
    bool use16 = false;                              // workgroup-uniform
    if constexpr (PV16) {
        if (a.short_pv16) {
            uint32_t big = 0;                        // largest finite |v| of this lane's chunks, as fp32 bits
            for (int c = tid; c < n_tiles * (IMG / 16); c += NW * 64) {
                const int tile = c / (IMG / 16), off = (c - tile * (IMG / 16)) * 16;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(lds + tile * 2 * IMG + IMG + off);
                u32x4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lo = raw[j] << 16, hi = raw[j] & 0xffff0000u;
                    h[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)));   // exact in range
                    const uint32_t alo = lo & 0x7fffffffu, ahi = hi & 0x7fffffffu;
                    if (alo < 0x7f800000u) big = big > alo ? big : alo;          // inf / NaN stay inf / NaN
                    if (ahi < 0x7f800000u) big = big > ahi ? big : ahi;
                }
                *reinterpret_cast<u32x4*>(lds + V16_BASE + tile * IMG + off) = h;
            }
            use16 = !__syncthreads_or(big > 0x477f0000u);                        // (65504 as fp32 is 0x477fe000; largest bf16 below: 0x477f0000)
        }
    }

    // transposed V reads: the per-lane part of the address is invariant; tile, image, 32-key half and the +16-row partner are
    // immediates (see the tiled kernel)
    const int vq = lq >> 2, vp = lq & 3;
    uint32_t vaddr[DT];
    {
        const int R0 = 4 * lg + vq;
        const int swz = chunk_swz_v<LPT>(R0);
#pragma unroll
        for (int tt = 0; tt < DT; ++tt) vaddr[tt] = lds_offset(lds + R0 * ROWB + (vp & 1) * 8 + (((2 * tt + (vp >> 1)) ^ swz) * 16));
    }

    auto run_tasks = [&](auto f16c) {
    constexpr bool F16 = decltype(f16c)::value;
    for (; t < n_tasks; t += t_step) {
        bf16x8 qn[STEPS];
        if (t + t_step < n_tasks) load_q(t + t_step, qn);                        // in flight while this task computes
        int head, q0;
        task_of(t, head, q0);
        const int my_q = q0 + lq;
        const int q_pos = my_q + shift;                                          // last key this lane's query may see
        int last_key = q0 + 15 + shift;                                          // last key any row of the sub-tile may see
        if (last_key > sk - 1) last_key = sk - 1;
        const int n_t = last_key < 0 ? 0 : last_key / BN + 1;                    // wave-uniform

        f32x4 o[DT];
#pragma unroll
        for (int tt = 0; tt < DT; ++tt) o[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;                                    // per query column; l_run is this lane group's share

        auto tile = [&](auto itc) {
            constexpr int it = decltype(itc)::value;
            constexpr int KOFF = it * 2 * IMG, VOFF = KOFF + IMG;
            const unsigned char* const kimg = lds + KOFF;
            const int kv0 = it * BN;
            // groups of 16 keys entirely above the sub-tile's last row's diagonal (or past the end) are not computed (wave-uniform):
            // at S = 128 two thirds of the tile bodies are diagonal ones that need 1..4 of their 4 groups
            int n_tt = (last_key - kv0) / 16 + 1;                                // last_key already <= sk - 1; >= 1 here
            n_tt = n_tt < NT ? n_tt : NT;
#ifdef NVH_PREFILL_NO_GROUP_SKIP                                                  // A/B builds only
            n_tt = NT;
#endif
            f32x4 sT[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (tt < n_tt) {
                    sT[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int R = 16 * tt + lq;
#pragma unroll
                    for (int st = 0; st < STEPS; ++st) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + R * ROWB + (((4 * st + lg) ^ chunk_swz<LPT>(R)) * 16));
                        sT[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[st], sT[tt], 0, 0, 0);
                    }
                } else {
                    sT[tt] = f32x4{0.f, 0.f, 0.f, 0.f};                          // (dead group: never looked at below; defined for the asm operand list)
                }
            }
            asm volatile("s_nop 7" : "+v"(sT[0]), "+v"(sT[1]), "+v"(sT[2]), "+v"(sT[3]));   // wait states: MFMA results -> the asm max chain
            // The softmax work too is done per LIVE 16-key group (wave-uniform guards): at S = 128 a quarter of the groups a tile body
            // touches are dead, and on a sub-tile's diagonal tile mask + max + exp2 of a dead group cost as much as those of a live one.
            const bool masked = kv0 + BN - 1 > q0 + shift || kv0 + BN > sk;      // wave-uniform: the tile reaches the diagonal or the end
            const int limit = (q_pos < sk - 1 ? q_pos : sk - 1) - kv0 - 4 * lg;  // see the tiled kernel: constant <= per-lane limit
            const int all_see = (q0 + shift < sk - 1 ? q0 + shift : sk - 1) - kv0;   // keys every row of the sub-tile sees (wave-uniform)
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (tt < n_tt) {
                    if (masked && 16 * tt + 15 > all_see) {                      // (groups below the sub-tile's diagonal block are visible to every row)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sT[tt][r] = (16 * tt + r <= limit) ? sT[tt][r] : -INFINITY;
                    }
                    mx = max3(mx, sT[tt][0], sT[tt][1]);
                    mx = max3(mx, sT[tt][2], sT[tt][3]);
                }
            }
            mx = max_xor16(mx);
            mx = max_xor32(mx);
            mx *= a.scale_log2;
            const float m_new = max2(m_run, mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;                // rows that have seen no key yet
            const f32x2 sc2 = {a.scale_log2, a.scale_log2}, mm2 = {-m_use, -m_use};
            f32x2 psum2 = {0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (tt < n_tt) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 sv = {sT[tt][r], sT[tt][r + 1]};
                        const f32x2 z = __builtin_elementwise_fma(sv, sc2, mm2);
                        const f32x2 e = {fast_exp2(z[0]), fast_exp2(z[1])};
                        sT[tt][r] = e[0];
                        sT[tt][r + 1] = e[1];
                        psum2 += e;
                    }
                } else {
                    sT[tt] = f32x4{0.f, 0.f, 0.f, 0.f};                          // p = 0 (read only when its 32-key half has a live group)
                }
            }
            if constexpr (it > 0) {                                              // (a task's first tile starts from o = l = 0: nothing to rescale)
                if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {          // some row's max moved: rescale (wave-uniform branch)
                    const float alpha = fast_exp2(m_run - m_use);                // m_run = -inf -> 0
                    l_run *= alpha;
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt) o[tt] *= alpha;
                }
            }
            l_run += psum2[0] + psum2[1];
            m_run = m_new;
#pragma unroll
            for (int hh = 0; hh < BN / 32; ++hh) {
                if (2 * hh >= n_tt) continue;                                    // wave-uniform: a dead 32-key half (p = 0)
                u32x4 hraw, lraw;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 pv = {sT[2 * hh + (j >> 1)][2 * (j & 1)], sT[2 * hh + (j >> 1)][2 * (j & 1) + 1]};
                    if constexpr (F16) {
                        hraw[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pv, f16x2v));       // v_cvt_pk_f16_f32, round to nearest even
                    } else {
                        const uint32_t hp = cvt_pk_bf16(pv[0], pv[1]);
                        const f32x2 hf = {__builtin_bit_cast(float, hp << 16), __builtin_bit_cast(float, hp & 0xffff0000u)};
                        const f32x2 lo = pv - hf;
                        hraw[j] = hp;
                        lraw[j] = cvt_pk_bf16(lo[0], lo[1]);
                    }
                }
                constexpr int VIMG = F16 ? V16_BASE + it * IMG : VOFF;           // same layout, same swizzle: only the element type differs
                u32x2 vlo[DT], vhi[DT];
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    if (hh == 0) {
                        vlo[tt] = ds_read_tr16_b64_asm<VIMG>(vaddr[tt]);
                        vhi[tt] = ds_read_tr16_b64_asm<VIMG + 16 * ROWB>(vaddr[tt]);
                    } else {
                        vlo[tt] = ds_read_tr16_b64_asm<VIMG + 32 * ROWB>(vaddr[tt]);
                        vhi[tt] = ds_read_tr16_b64_asm<VIMG + 48 * ROWB>(vaddr[tt]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) {
                    const f16x8 p16 = __builtin_bit_cast(f16x8, hraw);
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt) {
                        const u32x4 raw = {vlo[tt][0], vlo[tt][1], vhi[tt][0], vhi[tt][1]};
                        o[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, raw), p16, o[tt], 0, 0, 0);
                    }
                } else {
                    const bf16x8 p_hi = *reinterpret_cast<const bf16x8*>(&hraw), p_lo = *reinterpret_cast<const bf16x8*>(&lraw);
                    bf16x8 vf[DT];
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt) {
                        const u32x4 raw = {vlo[tt][0], vlo[tt][1], vhi[tt][0], vhi[tt][1]};
                        vf[tt] = *reinterpret_cast<const bf16x8*>(&raw);
                    }
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[tt], p_hi, o[tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < DT; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[tt], p_lo, o[tt], 0, 0, 0);
                }
            }
        };
        if (0 < n_t) tile(std::integral_constant<int, 0>{});
        if constexpr (NKT > 1) { if (1 < n_t) tile(std::integral_constant<int, 1>{}); }
        if constexpr (NKT > 2) { if (2 < n_t) tile(std::integral_constant<int, 2>{}); }
        if constexpr (NKT > 3) { if (3 < n_t) tile(std::integral_constant<int, 3>{}); }

        SK_STAMP(4 + 2 * sk_round);
        // ---- finalise the task: total row sum over the 4 lane groups, normalise, store 4 contiguous dims per tile
        float l = sum_xor16(l_run);
        l = sum_xor32(l);
        if (my_q < sq) {
            const float inv = l > 0.f ? 1.f / l : 0.f;
            const int64_t orow = ((int64_t)(q_beg + my_q) * a.h + head) * D;
#pragma unroll
            for (int tt = 0; tt < DT; ++tt) {
                const int d0 = 16 * tt + 4 * lg;
                const f32x4 r = o[tt] * inv;
                if (a.out_f32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + orow + d0) = r;
                } else {
                    u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + orow + d0) = pk;
                }
            }
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) qf[st] = qn[st];
        SK_STAMP(4 + 2 * sk_round + 1);                                          // (diagnostic builds: 4 + 2r = tiles of round r done, 5 + 2r = stores issued)
        ++sk_round;
    }
    };
    if constexpr (PV16) {
        if (use16) run_tasks(std::true_type{});
        else run_tasks(std::false_type{});
    } else {
        run_tasks(std::false_type{});
    }
    SK_STAMP(15);
}

template <int D, int QT>
int launch_q(const PrefillArgs& a, hipStream_t stream) {
    dim3 grid(a.h, a.batch, (a.max_seqlen_q + 64 * QT - 1) / (64 * QT));
    if (a.kernel == 3) {                                         // NVH_PREFILL_TILED_F16V: v holds fp16 rows (measurement variant; never paged)
        hipLaunchKernelGGL((prefill_varlen_kernel<D, false, QT, 1>), grid, dim3(256), 0, stream, a);
        return check_launch("prefill_varlen_f16v");
    }
    if (a.v16) {                                                 // nvh_prefill_varlen_pv16: fp16 P V unless the conversion raised the guard
        hipLaunchKernelGGL((prefill_varlen_kernel<D, false, QT, 2>), grid, dim3(256), 0, stream, a);
        return check_launch("prefill_varlen_pv16");
    }
    if (a.block_tables) hipLaunchKernelGGL((prefill_varlen_kernel<D, true, QT>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((prefill_varlen_kernel<D, false, QT>), grid, dim3(256), 0, stream, a);
    return check_launch("prefill_varlen");
}

// a.kernel (nvh_prefill_varlen_variant): 1 = never, 2 = whenever the shape allows (tests), 0 = when it also fills the chip
template <int D>
int launch_short(const PrefillArgs& a, hipStream_t stream, bool& taken) {
    taken = false;
    const int mode = (a.kernel == 1 || a.kernel == 3) ? 0 : a.kernel == 2 ? 2 : 1;
    const int max_keys = 128;                                    // (256 keys with four resident tiles measured slower than the tiled kernel)
    if (mode == 0 || a.block_tables || a.max_seqlen_k > max_keys || a.max_seqlen_q > a.max_seqlen_k) return 0;
    if (mode != 2 && a.batch * a.kvh < 128) return 0;            // few sequences: the tiled kernel spreads heads and q-tiles over the CUs
    taken = true;
    dim3 grid(a.kvh, a.batch, NVH_SHORT_SPLIT);
    // a wave's tile is a long dependent chain (QK -> max -> exp2 -> hi/lo -> PV): four waves per SIMD to fill it where the registers allow
    // (measured, Qwen2-0.5B heads, S = 128: 256 workgroups 22.7 us with 16 waves against 25.2 with 8; 512 workgroups, two per CU,
    // 37.4 us with 8 against 41.9 with 16; the tiled kernel 28.3 / 56)
#ifdef NVH_SHORT_FORCE_WAVES                  // A/B builds
    const int waves = a.short_waves ? a.short_waves : NVH_SHORT_FORCE_WAVES;
#else
    const int waves = a.short_waves ? a.short_waves : (a.batch * a.kvh >= 384 ? 8 : 16);
#endif
    if constexpr (D == 64) {
        if (waves == 16) {
            hipLaunchKernelGGL((prefill_short_kernel<D, 2, 16>), grid, dim3(1024), 0, stream, a);
            return check_launch("prefill_short");
        }
    }
    hipLaunchKernelGGL((prefill_short_kernel<D, 2, 8>), grid, dim3(512), 0, stream, a);
    return check_launch("prefill_short");
}

template <int D>
int launch_d(const PrefillArgs& a, hipStream_t stream) {
    bool taken;
    const int rc = launch_short<D>(a, stream, taken);
    if (taken) return rc;
    if (a.kernel == 2) { set_error("prefill_varlen: the short-sequence kernel needs max_seqlen_q <= max_seqlen_k <= 128 and no block table"); return -2; }
#ifndef NVH_PREFILL_QT2_FROM
#define NVH_PREFILL_QT2_FROM 2048            // D = 64: two 16-row sub-tiles per wave from this many query rows on (A/B builds change it)
#endif
#ifndef NVH_PREFILL_QT2_FROM_PV16
#define NVH_PREFILL_QT2_FROM_PV16 512        // ... in the fp16 P V form: its two-sub-tile body fits 128 registers = 4 waves per SIMD (same-box A/B against 2048:
                                             // S = 512 42.5 -> 40.5 us, S = 1024 62.8 -> 58.8, S = 1536 77.8 -> 74.0; profiles/r03_prefill_qt2_pv16_ab.txt)
#endif
    const bool two = D == 128 ? a.max_seqlen_q > 128 : a.max_seqlen_q >= (a.v16 ? NVH_PREFILL_QT2_FROM_PV16 : NVH_PREFILL_QT2_FROM);
    return two ? launch_q<D, 2>(a, stream) : launch_q<D, 1>(a, stream);
}

}  // namespace

int launch_prefill_varlen(const PrefillArgs& a, hipStream_t stream) {
    if (a.batch == 0 || a.max_seqlen_q == 0) return 0;
    return a.hd == 64 ? launch_d<64>(a, stream) : launch_d<128>(a, stream);
}

}  // namespace nvh
