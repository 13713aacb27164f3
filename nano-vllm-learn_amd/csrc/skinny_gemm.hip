// Weight-streaming linear layer for decode-sized batches (M <= 64 rows) on gfx950, with the neighbouring row-wise ops
// fused in — engine widening around the attention call, not part of the attention parity bar.
//   out[M, N] = prologue(x)[M, K] . W[N, K]^T  -> epilogue,   W in nn.Linear layout (row n = K contiguous bf16).
// Why: at decode every dependent launch costs ~5 us on MI355X whatever it computes (profiles/r01_bench_kernel_stats_10launch.txt:
// add_rmsnorm, rope_store and the attention combine all sit at 4.8-5.0 us), and hipBLASLt needs 10-13 us for these 2-17 MB
// weight streams.  Folding the norms and the elementwise tails into the GEMMs takes a decoder layer from 10 launches to 6.
//
//   prologue  NORM 1    x := bf16(bf16(x * rsqrt(mean(x^2) + eps)) * g)      RMSNorm.rms_forward, layers/layernorm.py:17-27
//             NORM 2    the same norm folded: W already holds g*W (caller), sum(x^2) is accumulated from the x fragments
//                       in the K loop and the row scale is applied in the epilogue (no extra pass, no bf16 re-rounding of x)
//   epilogue  NONE      + bias                                                 F.linear (layers/linear.py)
//             SILU      SiLU(x gate^T) * (x up^T), gate rows [0,inter), up rows [inter,2 inter)   layers/activation.py:11-14
//             RESADD    out (the residual stream, in place) += product         the add of add_rms_forward, layernorm.py:35-36
//             ROPE      + bias, neox RoPE on q and k heads, q -> out[M, H*D], k/v rows -> paged KV cache at slot_mapping
//                       (layers/rotary_embedding.py:6-16 + layers/attention.py:84-86); rounding order as rope_store.hip
//
// HBM-bound on W (read once): algorithmic bytes = N*K*2 + M*K*2 + M*N*2.
//   * one workgroup per 16 output columns (SILU / ROPE: the 16 columns plus their 16 partner columns); its waves split K into
//     contiguous chunks and reduce through LDS: no cross-workgroup reduction, no atomics, deterministic.
//   * v_mfma_f32_16x16x32_bf16 with B = W^T.  W is staged HBM -> LDS by LDS-DMA in coalesced 64-byte row segments and read
//     back as operand fragments (measured: fetching W directly in operand layout, 16 rows x 16 B per 16 lanes, streamed at
//     ~2 TB/s); x rows come from L2 (shared by all workgroups).
//   * NORM: a first pass over the wave's K chunk accumulates sum(x^2) per row, one LDS exchange gives every lane its row's
//     scale, the main loop normalises A fragments on the fly with the reference's two bf16 roundings.
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DNVH_STAMPS, tools/probes/stamp_gemm.py): per-wave clock stamps, 8 slots per wave, 8 waves per workgroup
#ifdef NVH_STAMPS
#define GM_STAMP(k)                                                                                             \
    do {                                                                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                         \
        if (a.stamps && lane == 0) a.stamps[((int64_t)blockIdx.x * 8 + wave) * 8 + (k)] = t_;                  \
    } while (0)
#else
#define GM_STAMP(k) do {} while (0)
#endif

template <int MT, int WAVES, int EPI, int NORM>
__global__ __launch_bounds__(WAVES * 64) void linear_small_m_kernel(const LinearArgs a) {
    constexpr int NB = (EPI == EPI_SILU || EPI == EPI_ROPE) ? 2 : 1;           // weight row blocks per workgroup
    constexpr int GS = WAVES == 8 ? 4 : 8;                                     // k-steps staged per group (per wave)
    constexpr int WAVE_STAGE = GS * NB * 1024;                                 // W staging bytes per wave (GS/2 pieces of 2 KiB per block)
    static_assert(NB * MT * 1024 <= WAVE_STAGE, "the wave's reduction tile aliases its own staging area");
    // one LDS array: [per wave: W staging, later reused for its reduction tile][row sums]
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WAVES * WAVE_STAGE + WAVES * MT * 16 * 4];
    typedef float red_t[NB][MT][64][4];
    typedef float ss_t[MT][16];
    ss_t* const lds_ss = reinterpret_cast<ss_t*>(lds_raw + WAVES * WAVE_STAGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    GM_STAMP(0);
    // K is split among the waves in PIECES of 2 k-steps (64 elements = one 128-byte line of every weight row)
    const int npieces = a.K / 64;
#ifdef NVH_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // kernel arguments have arrived
    GM_STAMP(6);
#endif
    const int pchunk = (npieces + WAVES - 1) / WAVES;
    const int ks0 = 2 * min(npieces, wave * pchunk);
    const int ks1 = 2 * min(npieces, (wave + 1) * pchunk);

    // ---- which weight rows this workgroup owns
    int n0, n1 = 0, head = 0, hi0 = 0;
    if constexpr (EPI == EPI_ROPE) {
        const int tiles = a.hd / 32;                                 // workgroups per head: columns i and i + D/2 together
        head = blockIdx.x / tiles;
        hi0 = 16 * (blockIdx.x % tiles);                             // index inside the half head
        n0 = head * a.hd + hi0;
        n1 = n0 + a.hd / 2;
    } else {
        n0 = blockIdx.x * 16;
        if constexpr (EPI == EPI_SILU) n1 = a.inter + n0;
    }
    const uint16_t* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = 16 * m + lq;
        xrow[m] = a.x + (int64_t)(r < a.M ? r : a.M - 1) * a.x_stride + lg * 8;   // rows past M repeat the last row; discarded
    }

    // ---- prologue: RMSNorm scale of every row this lane feeds into the MFMA (row 16m + lq)
    float inv[MT];
    if constexpr (NORM == 1) {
        float ss[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) ss[m] = 0.f;
        for (int g0 = ks0; g0 < ks1; g0 += 8) {
            u32x4 xv[8][MT];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = g0 + i < ks1 ? g0 + i : ks1 - 1;
#pragma unroll
                for (int m = 0; m < MT; ++m) xv[i][m] = *reinterpret_cast<const u32x4*>(xrow[m] + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float live = g0 + i < ks1 ? 1.f : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float lo = bf16_lo(xv[i][m][w]), hi = bf16_hi(xv[i][m][w]);
                        ss[m] += live * (lo * lo + hi * hi);
                    }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ss[m] = sum_xor16(ss[m]);                                // fold the 4 lane groups (k sub-blocks) of the row
            ss[m] = sum_xor32(ss[m]);
            if (lg == 0) lds_ss[wave][m][lq] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += lds_ss[w][m][lq];
            inv[m] = rsqrtf(t / a.K + a.norm_eps);
        }
    }

    f32x4 acc[NB][MT];
    float ss2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ss2[m] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[nb][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K loop in groups of GS k-steps.  W goes HBM -> LDS by LDS-DMA with lanes 4r..4r+3 fetching 64 contiguous bytes of
    // row r (one instruction = one k-step of one 16-row block = 1 KiB), i.e. coalesced 64-byte segments; loading W straight
    // into MFMA operand layout (consecutive lanes = different rows) streamed at only ~2 TB/s.  The operand fragments are then
    // ds_read_b128 from the [16 rows][64 B] images.  x fragments come from L2 as plain loads issued ahead of the DMAs; one
    // vmcnt(0) per group covers both.  Steps past the chunk end are clamped to its last step and their W fragment is zeroed.
    unsigned char* const stage = lds_raw + wave * WAVE_STAGE;
    // DMA mapping: one instruction = 8 rows x 128 B (a whole line per row); LDS image of a piece = [16 rows][128 B], the
    // 16-byte chunk order inside a row XOR-swizzled on the SOURCE so the operand reads below are conflict free
    const int dr = lane >> 3, dp = lane & 7;
    const uint16_t* wdma[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = 8 * hh + dr;
            wdma[nb][hh] = a.w + (int64_t)((nb == 0 ? n0 : n1) + row) * a.K + (dp ^ ((row >> 1) & 7)) * 8;
        }
    const int rswz = (lq >> 1) & 7;                                            // swizzle of this lane's operand row
    for (int g0 = ks0; g0 < ks1; g0 += GS) {
        u32x4 araw[GS][MT], graw[GS];
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int ks = g0 + i < ks1 ? g0 + i : ks1 - 1;
#pragma unroll
            for (int m = 0; m < MT; ++m) araw[i][m] = *reinterpret_cast<const u32x4*>(xrow[m] + ks * 32);
            if constexpr (NORM == 1) graw[i] = *reinterpret_cast<const u32x4*>(a.norm_w + ks * 32 + lg * 8);
        }
        if (g0 == ks0) GM_STAMP(7);
#pragma unroll
        for (int pi = 0; pi < GS / 2; ++pi) {                                  // piece pi of the group = k-steps g0+2pi, g0+2pi+1
            const int ks = g0 + 2 * pi < ks1 ? g0 + 2 * pi : ks1 - 2;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wdma[nb][hh] + ks * 32),
                                                     (__attribute__((address_space(3))) void*)(stage + ((pi * NB + nb) * 2 + hh) * 1024), 16, 0, 0);
        }
        if (g0 == ks0) GM_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g0 == ks0) GM_STAMP(2);
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const bool live = g0 + i < ks1;
            u32x4 braw[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                braw[nb] = *reinterpret_cast<const u32x4*>(stage + ((i >> 1) * NB + nb) * 2048 + lq * 128 + (((4 * (i & 1) + lg) ^ rswz) * 16));
#pragma unroll
                for (int w = 0; w < 4; ++w) braw[nb][w] = live ? braw[nb][w] : 0u;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                u32x4 av = araw[i][m];
                if constexpr (NORM == 2) {                         // folded norm: only sum(x^2) is needed; the scale is applied in the epilogue
                    float q = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float lo = bf16_lo(av[w]), hi = bf16_hi(av[w]);
                        q += lo * lo + hi * hi;
                    }
                    ss2[m] += live ? q : 0.f;
                }
                if constexpr (NORM == 1) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float lo = (float)(__bf16)(bf16_lo(av[w]) * inv[m]) * bf16_lo(graw[i][w]);
                        const float hi = (float)(__bf16)(bf16_hi(av[w]) * inv[m]) * bf16_hi(graw[i][w]);
                        av[w] = pack_bf16x2(lo, hi);
                    }
                }
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&av);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *reinterpret_cast<const bf16x8*>(&braw[nb]), acc[nb][m], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // LDS reads done before the next group's DMA overwrites
    }
    GM_STAMP(3);
    // ---- reduce the K chunks of the waves
    if constexpr (NORM == 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ss2[m] = sum_xor16(ss2[m]);
            ss2[m] = sum_xor32(ss2[m]);
            if (lg == 0) lds_ss[wave][m][lq] = ss2[m];
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>((*reinterpret_cast<red_t*>(stage))[nb][m][lane]) = acc[nb][m];
    __syncthreads();
    GM_STAMP(4);
    // value (m tile, lane, r) = product[16*mt + 4*(lane>>4) + r][column (lane&15) of each weight row block]
    for (int v = tid; v < MT * 256; v += WAVES * 64) {
        const int mt = v >> 8, l = (v >> 2) & 63, r = v & 3;
        const int row = 16 * mt + 4 * (l >> 4) + r;
        if (row >= a.M) continue;
        const int c = l & 15;
        float s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb] = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s[nb] += (*reinterpret_cast<const red_t*>(lds_raw + w * WAVE_STAGE))[nb][mt][l][r];
        }
        if constexpr (NORM == 2) {                                 // x.(g*W)^T * rsqrt(mean(x^2)+eps) == RMSNorm(x).W^T without the two bf16 roundings
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += lds_ss[w][mt][row & 15];
            const float inv_row = rsqrtf(t / a.K + a.norm_eps);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s[nb] *= inv_row;
        }
        __bf16* const out = reinterpret_cast<__bf16*>(a.out);
        if constexpr (EPI == EPI_SILU) {
            const float g = (float)(__bf16)s[0], u = (float)(__bf16)s[1];          // the projection output is bf16 in the reference
            out[(int64_t)row * a.out_stride + n0 + c] = (__bf16)((float)(__bf16)(g / (1.f + __expf(-g))) * u);
        } else if constexpr (EPI == EPI_RESADD) {
            __bf16* p = out + (int64_t)row * a.out_stride + n0 + c;
            *p = (__bf16)(s[0] + (float)*p);
        } else if constexpr (EPI == EPI_ROPE) {
            float x1 = s[0], x2 = s[1];
            if (a.bias) {
                x1 += (float)__builtin_bit_cast(__bf16, a.bias[n0 + c]);
                x2 += (float)__builtin_bit_cast(__bf16, a.bias[n1 + c]);
            }
            x1 = (float)(__bf16)x1;                                    // the projection output is bf16 in the reference
            x2 = (float)(__bf16)x2;
            const int i = hi0 + c;                                     // index inside the half head
            float y1 = x1, y2 = x2;
            if (head < a.h + a.kvh) {                                  // q or k head: rotate (products and sums rounded separately)
                const float* cs = a.cos_sin + a.positions[row] * a.hd;
                const float co = cs[i], si = cs[a.hd / 2 + i];
                const float p1 = x1 * co, p2 = x2 * si, p3 = x2 * co, p4 = x1 * si;
                y1 = p1 - p2;
                y2 = p3 + p4;
            }
            if (head < a.h) {
                __bf16* q = out + (int64_t)row * a.out_stride + head * a.hd + i;
                q[0] = (__bf16)y1;
                q[a.hd / 2] = (__bf16)y2;
            } else {
                const int slot = a.slots[row];
                if (slot >= 0) {
                    const bool is_v = head >= a.h + a.kvh;
                    __bf16* dst = reinterpret_cast<__bf16*>(is_v ? a.v_cache : a.k_cache) +
                                  ((int64_t)slot * a.kvh + (head - a.h - (is_v ? a.kvh : 0))) * a.hd + i;
                    dst[0] = (__bf16)y1;
                    dst[a.hd / 2] = (__bf16)y2;
                }
            }
        } else {
            const float y = s[0] + (a.bias ? (float)__builtin_bit_cast(__bf16, a.bias[n0 + c]) : 0.f);
            out[(int64_t)row * a.out_stride + n0 + c] = (__bf16)y;
        }
    }
    GM_STAMP(5);
}

template <int MT, int EPI, int NORM>
int launch_w(const LinearArgs& a, hipStream_t stream) {
    int wgs;
    if (EPI == EPI_ROPE) wgs = (a.h + 2 * a.kvh) * (a.hd / 32);
    else if (EPI == EPI_SILU) wgs = a.inter / 16;
    else wgs = a.N / 16;
    if (a.K >= 2048) hipLaunchKernelGGL((linear_small_m_kernel<MT, 8, EPI, NORM>), dim3(wgs), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((linear_small_m_kernel<MT, 4, EPI, NORM>), dim3(wgs), dim3(256), 0, stream, a);
    return check_launch("linear_small_m");
}

template <int MT>
int launch_mt(const LinearArgs& a, hipStream_t stream) {
    const int norm = a.norm_mode;                                  // 0 none, 1 exact prologue, 2 folded (scale in the epilogue)
#define NVH_EPI_CASE(E)                                                            \
    case E:                                                                        \
        return norm == 2 ? launch_w<MT, E, 2>(a, stream) : norm == 1 ? launch_w<MT, E, 1>(a, stream) : launch_w<MT, E, 0>(a, stream);
    switch (a.epi) {
        NVH_EPI_CASE(EPI_NONE)
        NVH_EPI_CASE(EPI_SILU)
        NVH_EPI_CASE(EPI_RESADD)
        NVH_EPI_CASE(EPI_ROPE)
    }
#undef NVH_EPI_CASE
    set_error("linear_small_m: unknown epilogue %d", a.epi);
    return -2;
}

}  // namespace

int launch_linear_small_m(const LinearArgs& a, hipStream_t stream) {
    if (a.M == 0) return 0;
    switch ((a.M + 15) / 16) {
        case 1: return launch_mt<1>(a, stream);
        case 2: return launch_mt<2>(a, stream);
        case 3: return launch_mt<3>(a, stream);
        case 4: return launch_mt<4>(a, stream);
    }
    set_error("linear_small_m: M = %d > 64", a.M);
    return -2;
}

}  // namespace nvh
