// Weight-streaming linear layer for decode-sized batches (M <= 64 rows) on gfx950 — engine widening, not part of the
// attention parity bar.  out[M, N] = x[M, K] . W[N, K]^T (+ bias), W in nn.Linear layout (row n = K contiguous bf16).
// Replaces the hipBLASLt calls PyTorch makes for nanovllm/layers/linear.py's F.linear at decode: those take 10-13 us for
// 2-17 MB of weights at M = 32 (profiles/r01_*), i.e. they are latency-, not bandwidth-bound.
//
// HBM-bound on W (read once): algorithmic bytes = N*K*2 + M*K*2 + M*N*2.
//   * one workgroup per 16 output columns (SILU mode: 16 gate columns + the 16 matching up columns); its WAVES waves split
//     K into contiguous chunks and reduce through LDS -> no cross-workgroup reduction, no atomics, deterministic.
//   * v_mfma_f32_16x16x32_bf16, B operand = W^T: lane l needs W[n0 + (l&15)][k0 + 8(l>>4) .. +8] = 16 contiguous bytes,
//     so W fragments go HBM -> VGPR in operand layout with no LDS staging (streamed once, no reuse); consecutive k-steps
//     walk along the 16 rows, every fetched 64-byte segment is fully used.
//   * A operand = x rows (M <= 64: 1..4 MFMA row tiles), read from L2 (x is M*K*2 <= 2.4 MB and shared by every workgroup).
//   * epilogue: + bias (fp32, one rounding, as addmm does), or SiLU(gate)*up with the reference's rounding points
//     (layers/activation.py:11-14 applied to the bf16 projection output), then bf16 stores.
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int WAVES, bool SILU>
__global__ __launch_bounds__(WAVES * 64) void linear_small_m_kernel(const LinearArgs a) {
    constexpr int NB = SILU ? 2 : 1;                             // weight row blocks per workgroup
    __shared__ __attribute__((aligned(16))) float red[WAVES][NB][MT][64][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ksteps = a.K / 32;
    const int chunk = (ksteps + WAVES - 1) / WAVES;
    const int ks0 = wave * chunk;
    const int ks1 = min(ksteps, ks0 + chunk);

    const uint16_t* wrow[NB];
    wrow[0] = a.w + (int64_t)(n0 + lq) * a.K + lg * 8;
    if constexpr (SILU) wrow[1] = a.w + (int64_t)(a.inter + n0 + lq) * a.K + lg * 8;
    const uint16_t* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = 16 * m + lq;
        xrow[m] = a.x + (int64_t)(r < a.M ? r : a.M - 1) * a.x_stride + lg * 8;   // rows past M repeat the last row; discarded
    }
    f32x4 acc[NB][MT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[nb][m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 4
    for (int ks = ks0; ks < ks1; ++ks) {
        u32x4 braw[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) braw[nb] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[nb] + ks * 32));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4 araw = *reinterpret_cast<const u32x4*>(xrow[m] + ks * 32);
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&araw);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *reinterpret_cast<const bf16x8*>(&braw[nb]), acc[nb][m], 0, 0, 0);
        }
    }
    // ---- reduce the K chunks of the waves
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(red[wave][nb][m][lane]) = acc[nb][m];
    __syncthreads();
    // value (m tile, lane, r) = out[16*mt + 4*(lane>>4) + r][n0 + (lane&15)]
    for (int v = tid; v < MT * 256; v += WAVES * 64) {
        const int mt = v >> 8, l = (v >> 2) & 63, r = v & 3;
        const int row = 16 * mt + 4 * (l >> 4) + r;
        if (row >= a.M) continue;
        const int col = n0 + (l & 15);
        float s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb] = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s[nb] += red[w][nb][mt][l][r];
        }
        float y;
        if constexpr (SILU) {
            const float g = (float)(__bf16)s[0], u = (float)(__bf16)s[1];          // the projection output is bf16 in the reference
            y = (float)(__bf16)(g / (1.f + __expf(-g))) * u;
        } else {
            y = s[0] + (a.bias ? (float)__builtin_bit_cast(__bf16, a.bias[col]) : 0.f);
        }
        reinterpret_cast<__bf16*>(a.out)[(int64_t)row * a.out_stride + col] = (__bf16)y;
    }
}

template <int MT, bool SILU>
int launch_mt(const LinearArgs& a, hipStream_t stream) {
    const int cols = SILU ? a.inter : a.N;
    dim3 grid(cols / 16);
    if (a.K >= 2048) hipLaunchKernelGGL((linear_small_m_kernel<MT, 8, SILU>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((linear_small_m_kernel<MT, 4, SILU>), grid, dim3(256), 0, stream, a);
    return check_launch("linear_small_m");
}

template <bool SILU>
int launch_s(const LinearArgs& a, hipStream_t stream) {
    switch ((a.M + 15) / 16) {
        case 1: return launch_mt<1, SILU>(a, stream);
        case 2: return launch_mt<2, SILU>(a, stream);
        case 3: return launch_mt<3, SILU>(a, stream);
        case 4: return launch_mt<4, SILU>(a, stream);
    }
    set_error("linear_small_m: M = %d > 64", a.M);
    return -2;
}

}  // namespace

int launch_linear_small_m(const LinearArgs& a, hipStream_t stream) {
    if (a.M == 0) return 0;
    return a.inter > 0 ? launch_s<true>(a, stream) : launch_s<false>(a, stream);
}

}  // namespace nvh
