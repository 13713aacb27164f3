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
__global__ __launch_bounds__(256) void bf16_rows_to_f16_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ in, int n_rows, int chunks_per_row,
                                                                int64_t in_row_stride, int64_t out_row_stride) {
    const int64_t total = (int64_t)n_rows * chunks_per_row;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / chunks_per_row), chunk = (int)(idx - (int64_t)row * chunks_per_row);
        const u32x4 raw = *reinterpret_cast<const u32x4*>(in + row * in_row_stride + chunk * 8);
        u32x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j)      // v_cvt_pkrtz_f16_f32: exact in range, so the rounding mode does not matter; out of range -> +-inf
            h[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_bit_cast(float, raw[j] << 16), __builtin_bit_cast(float, raw[j] & 0xffff0000u)));
        *reinterpret_cast<u32x4*>(out + row * out_row_stride + chunk * 8) = h;
    }
}

int launch_bf16_rows_to_f16(void* out, const void* in, int n_rows, int row_elems, int64_t in_row_stride, int64_t out_row_stride, hipStream_t stream) {
    if (n_rows == 0 || row_elems == 0) return 0;
    const int chunks = row_elems / 8;
    const int64_t total = (int64_t)n_rows * chunks;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bf16_rows_to_f16_kernel, dim3(blocks), dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)in, n_rows, chunks, in_row_stride, out_row_stride);
    return check_launch("bf16_rows_to_f16");
}

}  // namespace nvh
