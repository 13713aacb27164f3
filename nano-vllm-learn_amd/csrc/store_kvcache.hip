// store_kvcache for gfx950: scatter this step's K/V rows into the paged cache.
// Replaces nanovllm/layers/attention.py:19-55 (Triton, one program per token).
//
// HBM-bound copy with 16 B per lane: a row of KVH*D bf16 is ROW_BYTES = KVH*D*2 bytes
// (256 B for Qwen2-0.5B), i.e. ROW_BYTES/16 lanes per tensor per token.  One thread moves one
// 16-byte chunk of K or V; consecutive threads take consecutive chunks of one row so both the
// (strided) source row and the cache row are read/written as whole 128-B lines.
// slot < 0 rows are skipped (attention_triton.py:29-31; graph padding rows carry -1).
// Algorithmic bytes: 2 * N * KVH*D*2 read + the same written + 4*N (slots).
#include "common.h"
#include "kernels.h"

namespace nvh {

__global__ __launch_bounds__(256) void store_kvcache_kernel(
    const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
    const int32_t* __restrict__ slot_mapping, int n_tokens, int chunks_per_row,
    int64_t k_row_stride, int64_t v_row_stride) {
    // work item = (token, tensor in {K,V}, 16-byte chunk)
    const int64_t per_token = 2LL * chunks_per_row;
    const int64_t total = per_token * n_tokens;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int tok = (int)(idx / per_token);
        const int rem = (int)(idx - tok * per_token);
        const int is_v = rem >= chunks_per_row;
        const int chunk = rem - is_v * chunks_per_row;
        const int slot = slot_mapping[tok];
        if (slot < 0) continue;
        const uint16_t* src = is_v ? v + tok * v_row_stride : k + tok * k_row_stride;
        uint16_t* dst = (is_v ? v_cache : k_cache) + (int64_t)slot * chunks_per_row * 8;
        *reinterpret_cast<u32x4*>(dst + chunk * 8) = *reinterpret_cast<const u32x4*>(src + chunk * 8);
    }
}

int launch_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                         const int32_t* slot_mapping, int n_tokens, int kvh, int hd,
                         int64_t k_row_stride, int64_t v_row_stride, hipStream_t stream) {
    if (n_tokens == 0) return 0;
    const int chunks = kvh * hd / 8;
    const int64_t total = 2LL * chunks * n_tokens;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;           // 256 CUs x 8 blocks, grid-stride the rest
    hipLaunchKernelGGL(store_kvcache_kernel, dim3(blocks), dim3(256), 0, stream,
                       (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)k_cache, (uint16_t*)v_cache,
                       slot_mapping, n_tokens, chunks, k_row_stride, v_row_stride);
    return check_launch("store_kvcache");
}

// bf16 rows -> fp16 rows (exact for |x| <= 65504: a bf16 value has 8 significant bits), 16 bytes per lane.  Feeds the opt-in fp16 P V form of the
// prefill kernel (NVH_PREFILL_TILED_F16V) when the caller's V is bf16: T x KVH*D elements, e.g. 4 MB in / 4 MB out at 16 x 1024 tokens.
// One workgroup per group of kPv16GroupRows rows.  `group_flags` (nullable): flags[group] = 1 when a finite value of the group does not fit fp16
// (|x| > 65504; the conversion clamps it), else 0 — written by every workgroup, so nothing has to be cleared first; nvh_prefill_varlen_pv16's attention
// kernel ORs the flags of a sequence's rows and falls back to the caller's bf16 rows for that sequence.  bf16 values below fp16's subnormal range
// (|x| < 2^-24) become 0: an absolute error below 6e-8 per element of V.
__global__ __launch_bounds__(512) void bf16_rows_to_f16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int n_rows, int chunks_per_row,
                                                                int64_t in_row_stride, int64_t out_row_stride, int32_t* __restrict__ group_flags) {
    const int row0 = blockIdx.x * kPv16GroupRows;
    const int rows = n_rows - row0 < kPv16GroupRows ? n_rows - row0 : kPv16GroupRows;
    const int total = rows * chunks_per_row;
    uint32_t big = 0;                    // largest |bits| of a finite value seen by this lane (as fp32 bits, sign dropped)
    constexpr int U = 4;                 // 16-byte loads in flight per lane: a group of 64 rows x 256 B is ONE pass of the 512 lanes' first two loads
    for (int base = threadIdx.x; base < total; base += 512 * U) {
        u32x4 raw[U];
        int64_t dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + 512 * u;
            dst[u] = -1;
            if (idx < total) {
                const int r = idx / chunks_per_row, chunk = idx - r * chunks_per_row;
                raw[u] = *reinterpret_cast<const u32x4*>(in + (int64_t)(row0 + r) * in_row_stride + chunk * 8);
                dst[u] = (int64_t)(row0 + r) * out_row_stride + chunk * 8;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (dst[u] < 0) continue;
            u32x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {    // v_cvt_pkrtz_f16_f32: exact in range, so the rounding mode does not matter
                const uint32_t lo = raw[u][j] << 16, hi = raw[u][j] & 0xffff0000u;
                h[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)));
                const uint32_t alo = lo & 0x7fffffffu, ahi = hi & 0x7fffffffu;
                if (alo < 0x7f800000u) big = big > alo ? big : alo;      // inf / NaN convert to inf / NaN: not a range problem
                if (ahi < 0x7f800000u) big = big > ahi ? big : ahi;
            }
            *reinterpret_cast<u32x4*>(out + dst[u]) = h;
        }
    }
    if (group_flags) {                   // 65504 = 0x477fe000 as fp32; the largest bf16 value not above it is 0x477f0000 (65280)
        const int any = __syncthreads_or(big > 0x477f0000u);
        if (threadIdx.x == 0) group_flags[blockIdx.x] = any ? 1 : 0;
    }
}

int launch_bf16_rows_to_f16(void* out, const void* in, int n_rows, int row_elems, int64_t in_row_stride, int64_t out_row_stride, int32_t* group_flags,
                            hipStream_t stream) {
    if (n_rows == 0 || row_elems == 0) return 0;
    const int groups = (n_rows + kPv16GroupRows - 1) / kPv16GroupRows;
    hipLaunchKernelGGL(bf16_rows_to_f16_kernel, dim3(groups), dim3(512), 0, stream, (uint16_t*)out, (const uint16_t*)in, n_rows, row_elems / 8, in_row_stride,
                       out_row_stride, group_flags);
    return check_launch("bf16_rows_to_f16");
}

}  // namespace nvh
