// One launch for the front of a decode layer: the fused qkv projection (+ folded RMSNorm, bias, RoPE, K/V store) AND the paged
// decode attention that consumes its q and its newest K/V rows.
// Replaces, at decode, the sequence nanovllm/models/qwen3.py:104-117 (qkv_proj -> rotary_emb -> self.attn) with
// nanovllm/layers/attention.py:84-86 (store_kvcache) and :99-101 (flash_attn_with_kvcache) inside it: SURVEY.md section 8f row 2
// taken to its end ("... and optionally store fused into the decode kernel").
//
// Why: as two launches (nvh_linear_small_m_ex with the RoPE/store epilogue, then nvh_paged_decode) the attention call is 7-9 us of
// which only ~4 are its K/V stream: a launch boundary (1.2-1.9 us), the dispatch ramp (0.5-0.8) and the first-byte latency (~1.9)
// stand in front of the stream, and none of them can be shortened from inside a launch of its own.  But the K/V
// stream does not depend on the projection at all; only q and ONE row per (sequence, kv head) do.  So the two run as roles of one grid:
//   producers  (blockIdx.z < gz)   one weight tile of the projection each (32 columns: the rotation pairs i, i + D/2 of 16 dims of
//              one head), exactly linear_stream.hip's single-pass form; the epilogue's results leave as 16-byte WRITE-THROUGH stores
//              (q rows; the K and V rows straight into the paged cache), every storing wave drains them, workgroup barrier, then one
//              agent-scope add to the ready counter of the tile's kv head
//   consumers  (the rest)          decode_chunked_body<FUSED> (decode_chunked.h): issue the LDS-DMA of their first two passes and touch
//              the later ones into the XCD's L2 at once — never the row of token ctx - 1 —, prefetch the output projection's weights,
//              then ONE lane per workgroup polls the head's ready counter (bounded); behind it q is read straight into the MFMA operand
//              registers and the newest K / V row is patched into its LDS image, both by L1-bypassing loads.
// The hand-off is the first row of the guide's table (MI355X_MICROARCH.md, Valid forms): one lane per storing workgroup signals for all
// its stores by an agent-scope atomic add after every storing wave's vmcnt(0) and a barrier; the consumer learns it by an sc1 load poll;
// the other waves load after a workgroup barrier the poller joins; 16-byte sc1 stores and loads; hipMalloc memory.
// Residency: every workgroup of the grid must be resident at once (consumers wait for producers).  Both roles use at most 72 KB of LDS
// and 512 threads, so two fit on a CU and the launcher keeps the grid within 2 x CUs - it refuses (falls back to two launches) beyond.
// Producers are the lowest-numbered workgroups and wait for nothing.  A wait that runs out anyway (a device shared with another
// process's kernels) sets the status word and turns the affected rows into NaN; nothing spins for ever.
#include "common.h"
#include "kernels.h"
#include "decode_chunked.h"

namespace nvh {

namespace {

constexpr int QA_WAVES = 8;                    // both roles: 8 waves, two per SIMD
constexpr int QA_PM = 2;                       // K pieces (64 elements each) per wave at most: K <= 8 * 2 * 64 = 1024, no split-K
constexpr int QA_STAGE = 2 * QA_PM * 2048;     // W staging bytes per wave: 2 row blocks x PM pieces x (16 rows x 128 B)
constexpr int QA_D = 64, QA_PASS = 128;        // consumer shape: head_dim 64, 128-token passes (16-token wave tiles): 67 KB of LDS

template <int MT>
constexpr int qkv_role_lds_bytes() { return QA_WAVES * QA_STAGE + QA_WAVES * MT * 16 * 4 + MT * 16 * 32 * 2 + 16; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

struct QkvAttendArgs {
    LinearArgs lin;              // the projection (epi == EPI_ROPE, norm_mode == 2, x_packed, ksplit == 1)
    DecodeArgs dec;              // the attention call on its output (dec.q == lin.out)
    FusedSync fs;
    int G;                       // query heads per kv head
    int bs_shift;                // log2(block_size) or -1
};

__device__ __forceinline__ void st16_global_sc1(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Diagnostic build only (-DNVH_STAMPS, tools/probes/stamp_qkv_attend.py): clock stamps of the producer role, [tile][wave][8], behind the
// consumers' region of the debug buffer.  Never compiled into the shipped library.
#ifdef NVH_STAMPS
#define QA_STAMP(k)                                                                                                          \
    do {                                                                                                                     \
        unsigned long long t_;                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        if (a.stamps && lane == 0) a.stamps[((int64_t)tile * QA_WAVES + wave) * 8 + (k)] = t_;                               \
    } while (0)
#else
#define QA_STAMP(k) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void qa_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Producer role: weight tile `tile` of the projection for all M rows.  Same arithmetic and rounding points as
// linear_stream_kernel<MT, EPI_ROPE, NORM = 2, XPACK> (and therefore as nvh_rope_store); only the way the results leave differs.
template <int MT>
__device__ __forceinline__ void qkv_tile_body(unsigned char* const lds_raw, const int tile, const uint16_t* __restrict__ p_w,
                                              const uint16_t* __restrict__ p_x, const int p_K, const int p_hd, const int p_pieces,
                                              const int64_t* __restrict__ p_positions, const uint16_t* __restrict__ p_bias,
                                              const LinearArgs& a, const FusedSync& fs, const int G) {
    constexpr int SW = QA_WAVES, TPB = SW * 64, EPT = (MT * 256 + TPB - 1) / TPB, NB = 2, PMAX = QA_PM, STAGE = QA_STAGE;
    static_assert(NB * MT * 1024 <= STAGE, "the wave's reduction tile aliases the staging buffer it has just consumed");
    typedef float red_t[NB][MT][64][4];
    typedef float ss_t[MT][16];
    ss_t* const lds_ss = reinterpret_cast<ss_t*>(lds_raw + SW * STAGE);
    uint16_t* const lds_t = reinterpret_cast<uint16_t*>(lds_raw + SW * STAGE + SW * MT * 16 * 4);   // [MT * 16 rows][32 columns] bf16
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    // this wave's K pieces
    const int p0 = __builtin_amdgcn_readfirstlane(p_pieces * wave / SW);
    const int np = __builtin_amdgcn_readfirstlane(p_pieces * (wave + 1) / SW - p0);          // 0..PMAX (host guarantees)
    const int KS = p_K / 32;
    // the tile's weight rows: 16 rotation pairs (i, i + D/2) of one head
    const int ph_shift = p_hd == 128 ? 2 : 1;
    const int head = tile >> ph_shift, hi0 = 16 * (tile & ((1 << ph_shift) - 1));
    const int n0 = head * p_hd + hi0, n1 = n0 + p_hd / 2;

    QA_STAMP(0);
    // ---- W DMA first: one instruction = 8 rows x 128 B; 16-byte chunk order XOR-swizzled on the source (conflict-free operand reads)
    const int dr = lane >> 3, dp = lane & 7;
    const int rswz = (lq >> 1) & 7;
    unsigned char* const stage = lds_raw + wave * STAGE;
#pragma unroll
    for (int pi = 0; pi < PMAX; ++pi) {
        if (pi < np) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row = 8 * hh + dr;
                    const uint16_t* src = p_w + (int64_t)((nb == 0 ? n0 : n1) + row) * p_K + (p0 + pi) * 64 + (dp ^ ((row >> 1) & 7)) * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage + ((pi * NB + nb) * 2 + hh) * 1024), 16, 0, NVH_DMA_AUX);
                }
        }
    }
    // ---- epilogue operands that hang on nothing but the arguments: position, bias pair (raw), and the cache slot of the row this
    // thread will STORE (store phase: thread t moves 16 bytes of row t >> 2)
    uint16_t rp_b1[EPT], rp_b2[EPT];
    int64_t rp_pos[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int v = tid + TPB * j, l = (v >> 2) & 63, c = l & 15;
        const int row = min(16 * min(v >> 8, MT - 1) + 4 * (l >> 4) + (v & 3), a.M - 1);
        rp_pos[j] = p_positions[row];
        const uint16_t* const bsrc = p_bias ? p_bias : p_w;      // (unconditional: see linear_stream.hip)
        rp_b1[j] = bsrc[n0 + c];
        rp_b2[j] = bsrc[n1 + c];
    }
    const int st_row = min(tid >> 2, a.M - 1);
    const int st_slot = head >= a.h ? a.slots[st_row] : 0;
    // ---- x fragments of this wave's K range (fragment-packed: one coalesced 1 KiB load each)
    u32x4 araw[2 * PMAX][MT];
#pragma unroll
    for (int pi = 0; pi < PMAX; ++pi) {
        if (pi < np) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ks = 2 * (p0 + pi) + j;
#pragma unroll
                for (int m = 0; m < MT; ++m) araw[2 * pi + j][m] = *reinterpret_cast<const u32x4*>(p_x + (((int64_t)m * KS + ks) * 64 + lane) * 8);
            }
        }
    }
    float rp_co[EPT], rp_si[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int c = ((tid + TPB * j) >> 2) & 15;
        const float* cs = a.cos_sin + rp_pos[j] * p_hd;
        rp_co[j] = cs[hi0 + c];
        rp_si[j] = cs[p_hd / 2 + hi0 + c];
    }
    // descriptor fields of the epilogue, fetched under the weight stream
    void* const e_out = a.out;
    const int64_t e_out_stride = a.out_stride;
    const float e_norm_eps = a.norm_eps;
    uint16_t* const e_k_cache = a.k_cache;
    uint16_t* const e_v_cache = a.v_cache;
    unsigned* const e_ready = fs.ready;
    asm volatile("" ::"s"(e_out), "s"(e_out_stride), "s"(e_norm_eps), "s"(e_k_cache), "s"(e_v_cache), "s"(e_ready));

    float ss2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ss2[m] = 0.f;
    QA_STAMP(1);
    qa_wait_vm<2 * EPT>();                                     // everything but cos / sin (the youngest loads) has landed
    QA_STAMP(2);
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
#pragma unroll
                    for (int w = 0; w < 4; ++w) ss2[m] = dot2_bf16(av[w], av[w], ss2[m]);
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(&av);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *reinterpret_cast<const bf16x8*>(&braw[nb]), acc[nb][m], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // staging reads done: the buffer becomes the reduction tile
    QA_STAMP(3);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float t = sum_xor16(ss2[m]);
        t = sum_xor32(t);
        if (lg == 0) lds_ss[wave][m][lq] = t;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>((*reinterpret_cast<red_t*>(stage))[nb][m][lane]) = acc[nb][m];
    __syncthreads();
    // ---- cross-wave reduction + epilogue arithmetic; value v = (m tile, lane, r): product[16 mt + 4 (lane >> 4) + r][column lane & 15]
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int v = tid + TPB * j, mt = min(v >> 8, MT - 1), l = (v >> 2) & 63, r = v & 3;
        float y[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            y[nb] = 0.f;
#pragma unroll
            for (int w = 0; w < SW; ++w) y[nb] += (*reinterpret_cast<const red_t*>(lds_raw + w * STAGE))[nb][mt][l][r];
        }
        float rowss = 0.f;
#pragma unroll
        for (int w = 0; w < SW; ++w) rowss += lds_ss[w][mt][4 * (l >> 4) + r];
        if (v < MT * 256) {
            const int row = 16 * mt + 4 * (l >> 4) + r, c = l & 15;
            const float inv_row = rsqrtf(rowss / p_K + e_norm_eps);
            float x1 = y[0] * inv_row, x2 = y[1] * inv_row;
            if (p_bias) {
                uint32_t b1 = rp_b1[j], b2 = rp_b2[j];
                asm volatile("" : "+v"(b1), "+v"(b2));
                x1 += __builtin_bit_cast(float, b1 << 16);
                x2 += __builtin_bit_cast(float, b2 << 16);
            }
            x1 = (float)(__bf16)x1;                              // the projection output is bf16 in the reference
            x2 = (float)(__bf16)x2;
            float y1 = x1, y2 = x2;
            if (head < a.h + a.kvh) {                            // q or k head: rotate; products and sums rounded separately (the
                const float co = rp_co[j], si = rp_si[j];        // pins keep the backend from contracting them into FMAs)
                float p1 = x1 * co, p2 = x2 * si, p3 = x2 * co, p4 = x1 * si;
                asm volatile("" : "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4));
                y1 = p1 - p2;
                y2 = p3 + p4;
            }
            lds_t[row * 32 + c] = __builtin_bit_cast(uint16_t, (__bf16)y1);
            lds_t[row * 32 + 16 + c] = __builtin_bit_cast(uint16_t, (__bf16)y2);
        }
    }
    __syncthreads();
    QA_STAMP(4);
    // ---- results leave as 16-byte write-through stores: thread t moves columns [8 seg, 8 seg + 8) of row t >> 2 (seg = t & 3;
    // segments 0, 1 are dims hi0 .. hi0 + 15 of the head, 2, 3 the same dims + D/2)
    if (tid < MT * 64 && (tid >> 2) < a.M) {
        const int row = tid >> 2, seg = tid & 3;
        const u32x4 val = *reinterpret_cast<const u32x4*>(lds_t + row * 32 + seg * 8);
        const int dim = (seg >> 1) * (p_hd / 2) + hi0 + (seg & 1) * 8;
        if (head < a.h) {
            st16_global_sc1(reinterpret_cast<uint16_t*>(e_out) + (int64_t)row * e_out_stride + head * p_hd + dim, val);
        } else if (st_slot >= 0) {
            const bool is_v = head >= a.h + a.kvh;
            uint16_t* const dst = (is_v ? e_v_cache : e_k_cache) + ((int64_t)st_slot * a.kvh + (head - a.h - (is_v ? a.kvh : 0))) * p_hd + dim;
            st16_global_sc1(dst, val);
        }
    }
    QA_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave: its stores are at the memory side
    __syncthreads();
    QA_STAMP(6);
    if (tid == 0) {
        const int group = head < a.h ? head / G : (head < a.h + a.kvh ? head - a.h : head - a.h - a.kvh);
        __hip_atomic_fetch_add(e_ready + (int64_t)group * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    QA_STAMP(7);
}

template <int MT>
__global__ __launch_bounds__(QA_WAVES * 64) void qkv_attend_kernel(
    // the producers' way to their first DMA comes first and flat (kernarg preload, build.py): they are the launch's critical path
    const uint16_t* __restrict__ p_w, const uint16_t* __restrict__ p_x, const int64_t* __restrict__ p_positions,
    const uint16_t* __restrict__ p_bias, const int p_K, const int p_tiles, const int p_hd, const int p_pieces, const int p_gz,
    const QkvAttendArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[cmax(chunked_lds_bytes<QA_D, QA_WAVES, QA_PASS>(), qkv_role_lds_bytes<MT>())];
    const int z = (int)blockIdx.z;
    if (z < p_gz) {                                            // workgroup-uniform: producer
        const int tile = (z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (tile >= p_tiles) return;
        qkv_tile_body<MT>(lds, tile, p_w, p_x, p_K, p_hd, p_pieces, p_positions, p_bias, a.lin, a.fs, a.G);
        return;
    }
    decode_chunked_body<QA_D, QA_WAVES, QA_PASS, true>(lds, z - p_gz, (int)blockIdx.x, (int)blockIdx.y, a.dec.context_lens, a.dec.block_tables,
                                                       a.dec.k_cache, a.dec.v_cache, a.dec.kvh, a.dec.block_size, a.dec.max_blocks, a.dec.chunks,
                                                       (int)a.dec.bt_row_stride, a.bs_shift, a.dec, a.G, a.fs);
}

static int qa_device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <int MT>
int launch_mt(const QkvAttendArgs& a, int tiles, int gz, hipStream_t stream) {
    const LinearArgs& l = a.lin;
    dim3 grid(a.dec.kvh, a.dec.batch, gz + a.dec.chunks);
    hipLaunchKernelGGL((qkv_attend_kernel<MT>), grid, dim3(QA_WAVES * 64), 0, stream, l.w, l.x, l.positions, l.bias, l.K, tiles, l.hd, l.K / 64, gz, a);
    return check_launch("qkv_attend");
}

}  // namespace

// bytes of the sync area (behind the ticket header of the decode workspace): 16 ready counters, 16 done counters (one per 128-byte
// line each), then the status word
size_t qkv_attend_sync_bytes(void) { return 8192; }

// Can this (projection, attention) pair run as ONE launch?  Pure function of the static shapes (and the device's CU count).
bool qkv_attend_supported(const LinearArgs& l, const DecodeArgs& d) {
    if (l.epi != EPI_ROPE || l.norm_mode != 2 || !l.x_packed || l.hd != 64 || d.hd != 64) return false;
    if (l.M <= 0 || l.M > 64 || l.M != d.batch || l.K % 64 != 0 || l.K / 64 > QA_WAVES * QA_PM) return false;
    if (l.h != d.h || l.kvh != d.kvh || d.h % d.kvh != 0 || d.h / d.kvh > 16 || d.kvh > 15) return false;
    if (d.out_f32 || l.out_stride % 8 != 0) return false;
    const int tiles = (l.h + 2 * l.kvh) * (l.hd / 32);
    const int chunks = decode_chunks(d.batch, d.kvh, d.num_splits, 0);
    // every workgroup resident at once: two per CU by LDS (72 KB each) and threads (512 each)
    return tiles + d.batch * d.kvh * chunks <= 2 * qa_device_cus() && d.batch * d.kvh <= 65536 / (4 * kTicketStride);
}

// l.out = the q rows [M, h*hd] (row stride l.out_stride) = d.q; sync = qkv_attend_sync_bytes() zero-filled bytes (left zero by every launch)
// missing_producers > 0 (tests): the consumers wait for that many producers more than exist, i.e. every wait runs out
int launch_qkv_attend(const LinearArgs& l, const DecodeArgs& d_in, void* sync, unsigned spin_limit, int missing_producers, const void* pf_ptr,
                      int64_t pf_bytes, hipStream_t stream) {
    QkvAttendArgs a;
    a.lin = l;
    a.dec = d_in;
    a.dec.chunks = decode_chunks(a.dec.batch, a.dec.kvh, a.dec.num_splits, 0);
    a.dec.impl = 0; a.dec.waves = 8; a.dec.pass_tokens = QA_PASS;
    a.G = a.dec.h / a.dec.kvh;
    a.bs_shift = (a.dec.block_size & (a.dec.block_size - 1)) == 0 ? __builtin_ctz(a.dec.block_size) : -1;
    const int tph = l.hd / 32;
    const int tiles = (l.h + 2 * l.kvh) * tph;
    const int per_z = a.dec.kvh * a.dec.batch;
    const int gz = (tiles + per_z - 1) / per_z;
    unsigned* const s = reinterpret_cast<unsigned*>(sync);
    a.fs.ready = s;
    a.fs.done = s + 16 * kTicketStride;
    a.fs.status = s + 32 * kTicketStride;
    a.fs.need = (unsigned)((a.G + 2) * tph + missing_producers);
    a.fs.consumers = (unsigned)(a.dec.batch * a.dec.chunks);
    a.fs.spin_limit = spin_limit ? spin_limit : (1u << 20);
    a.fs.pf_ptr = pf_ptr;
    a.fs.pf_bytes = pf_bytes;
    if (a.lin.stamps) a.lin.stamps += (int64_t)a.dec.batch * a.dec.kvh * a.dec.num_splits * QA_WAVES * 8;   // (diagnostic builds: behind the consumers' region)
    switch ((l.M + 15) / 16) {
        case 1: return launch_mt<1>(a, tiles, gz, stream);
        case 2: return launch_mt<2>(a, tiles, gz, stream);
        case 3: return launch_mt<3>(a, tiles, gz, stream);
        case 4: return launch_mt<4>(a, tiles, gz, stream);
    }
    set_error("qkv_attend: m = %d", l.M);
    return -2;
}

}  // namespace nvh
