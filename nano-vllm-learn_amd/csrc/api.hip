// C ABI of libnvh_attn.so (declared in include/nvh_attn.h): argument validation mirroring the
// reference's asserts (nanovllm/layers/attention.py:51-54), error strings, kernel dispatch.
// Nothing here allocates, synchronises or reads device memory: every entry point is capture-safe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nvh_attn.h"
#include "common.h"
#include "kernels.h"

namespace nvh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// diagnostic builds (-DNVH_STAMPS) set this through nvh_debug_set_stamps; always null in the shipped library
static unsigned long long* g_stamps = nullptr;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_heads(const char* fn, int h, int kvh, int hd) {
    if (hd != 64 && hd != 128) {
        set_error("%s: head_dim %d unsupported (64 or 128)", fn, hd);
        return NVH_E_SHAPE;
    }
    if (h <= 0 || kvh <= 0 || h % kvh != 0) {
        set_error("%s: num_heads %d must be a positive multiple of num_kv_heads %d", fn, h, kvh);
        return NVH_E_SHAPE;
    }
    return 0;
}

}  // namespace nvh

using namespace nvh;

extern "C" {

int nvh_version(void) { return NVH_VERSION; }

#ifdef NVH_STAMPS
void nvh_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }
#endif

const char* nvh_last_error(void) { return g_err; }

int nvh_store_kvcache(const void* k, const void* v, void* k_cache, void* v_cache,
                      const int32_t* slot_mapping, int n_tokens, int kvh, int hd,
                      int64_t k_row_stride, int64_t v_row_stride, int dtype, void* stream) {
    if (n_tokens == 0) return 0;
    if (dtype != NVH_BF16) { set_error("store_kvcache: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!k || !v || !k_cache || !v_cache || !slot_mapping) { set_error("store_kvcache: null pointer"); return NVH_E_NULL; }
    if (n_tokens < 0 || kvh <= 0 || hd <= 0 || (kvh * hd) % 8 != 0) {
        set_error("store_kvcache: kvh*hd = %d*%d must be a positive multiple of 8", kvh, hd);
        return NVH_E_SHAPE;
    }
    // reference asserts: inner dims contiguous (attention.py:51-52), cache rows = kvh*hd (:53)
    if (k_row_stride % 8 || v_row_stride % 8 || k_row_stride < (int64_t)kvh * hd || v_row_stride < (int64_t)kvh * hd) {
        set_error("store_kvcache: row strides %lld/%lld must be multiples of 8 and >= kvh*hd",
                  (long long)k_row_stride, (long long)v_row_stride);
        return NVH_E_STRIDE;
    }
    if (!aligned16(k) || !aligned16(v) || !aligned16(k_cache) || !aligned16(v_cache)) {
        set_error("store_kvcache: pointers must be 16-byte aligned");
        return NVH_E_ALIGN;
    }
    return launch_store_kvcache(k, v, k_cache, v_cache, slot_mapping, n_tokens, kvh, hd,
                                k_row_stride, v_row_stride, (hipStream_t)stream);
}

static constexpr size_t kDecodeTicketBytes = 65536;       // 512 tickets, one per 128-byte line (pairs are split over chunks only when there are few of them)
static constexpr size_t kDecodeSyncBytes = 8192;          // behind the tickets: the ready / done counters and the status word of nvh_qkv_rope_attend
static constexpr size_t kDecodeHeaderBytes = kDecodeTicketBytes + kDecodeSyncBytes;
// nvh_prefill_varlen_pv16 scratch: one int32 range flag per kPv16GroupRows rows (padded to 256 bytes), then the fp16 rows
static size_t pv16_header_bytes(int total_k) { return ((size_t)((total_k + nvh::kPv16GroupRows - 1) / nvh::kPv16GroupRows) * 4 + 255) / 256 * 256; }

static int decode_num_splits(int hd, int max_blocks, int block_size) {
    const int split = decode_split_tokens(hd);
    const int64_t cap = (int64_t)max_blocks * block_size;
    int n = (int)((cap + split - 1) / split);
    return n < 1 ? 1 : n;
}

size_t nvh_paged_decode_workspace(int batch, int h, int hd, int max_blocks, int block_size) {
    if (batch <= 0 || h <= 0 || (hd != 64 && hd != 128) || max_blocks <= 0 || block_size <= 0) return 0;
    const size_t parts = (size_t)batch * h * decode_num_splits(hd, max_blocks, block_size);
    // a fixed header of arrival tickets (one per (sequence, kv head); the SAME bytes whatever the shape, so that one
    // workspace serves calls of different shapes), then the partial records
    // (twice the packed size: chunk records are padded to 256-byte boundaries, at most a factor two at one query head per kv head)
    return kDecodeHeaderBytes + 2 * parts * (size_t)(hd + 2) * sizeof(float);
}

// validates the call and fills `a`; returns 1 when there is nothing to do (batch == 0), 0 when `a` is ready, < 0 on rejection
static int paged_decode_args(DecodeArgs& a, void* out, void* out_packed, const void* q, const void* k_cache, const void* v_cache,
                             const int32_t* block_tables, const int32_t* context_lens,
                             int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                             int64_t q_row_stride, int64_t bt_row_stride, float scale,
                             int dtype, int out_dtype, void* workspace, size_t workspace_bytes,
                             int variant = NVH_DECODE_CHUNKED, int waves = 0, int chunks = 0) {
    if (batch == 0) return 1;
    if (variant < NVH_DECODE_CHUNKED || variant > NVH_DECODE_CHUNKED_P64 || (waves != 0 && waves != 4 && waves != 8) || chunks < 0) {
        set_error("paged_decode: variant %d / waves %d / chunks %d not supported", variant, waves, chunks);
        return NVH_E_SHAPE;
    }
    if (variant == NVH_DECODE_SPLIT_VALU && h / (kvh > 0 ? kvh : 1) > 8) { set_error("paged_decode: the VALU variant serves groups of at most 8 query heads"); return NVH_E_SHAPE; }
    if (out_packed && (!aligned16(out_packed) || ((int64_t)h * hd) % 32)) { set_error("paged_decode: out_packed needs 16-byte alignment and h*hd %% 32 == 0"); return NVH_E_ALIGN; }
    if (dtype != NVH_BF16 || (out_dtype != NVH_BF16 && out_dtype != NVH_F32)) {
        set_error("paged_decode: dtype %d / out_dtype %d unsupported", dtype, out_dtype);
        return NVH_E_DTYPE;
    }
    if (!out || !q || !k_cache || !v_cache || !block_tables || !context_lens || !workspace) {
        set_error("paged_decode: null pointer");
        return NVH_E_NULL;
    }
    int rc = check_heads("paged_decode", h, kvh, hd);
    if (rc) return rc;
    if (h / kvh > 16) { set_error("paged_decode: group size %d > 16 unsupported", h / kvh); return NVH_E_SHAPE; }
    if (batch < 0 || max_blocks <= 0 || block_size <= 0 || block_size % 64 != 0) {
        set_error("paged_decode: block_size %d must be a positive multiple of 64, max_blocks %d > 0", block_size, max_blocks);
        return NVH_E_SHAPE;
    }
    if (q_row_stride % 8 || q_row_stride < (int64_t)h * hd || bt_row_stride < max_blocks || bt_row_stride > 0x7fffffff) {
        set_error("paged_decode: bad strides q=%lld bt=%lld", (long long)q_row_stride, (long long)bt_row_stride);
        return NVH_E_STRIDE;
    }
    if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(workspace) || !aligned16(out)) {
        set_error("paged_decode: pointers must be 16-byte aligned");
        return NVH_E_ALIGN;
    }
    const size_t need = nvh_paged_decode_workspace(batch, h, hd, max_blocks, block_size);
    if (workspace_bytes < need) {
        set_error("paged_decode: workspace %zu B < required %zu B", workspace_bytes, need);
        return NVH_E_WORKSPACE;
    }
    a.out = out;
    a.q = (const uint16_t*)q;
    a.k_cache = (const uint16_t*)k_cache;
    a.v_cache = (const uint16_t*)v_cache;
    a.block_tables = block_tables;
    a.context_lens = context_lens;
    a.batch = batch; a.h = h; a.kvh = kvh; a.hd = hd;
    a.block_size = block_size; a.max_blocks = max_blocks;
    a.num_splits = decode_num_splits(hd, max_blocks, block_size);
    // (tickets are only drawn when a (sequence, kv head) pair is split over several workgroups, i.e. when batch * kvh is small)
    a.chunks = decode_chunks(batch, kvh, a.num_splits, chunks);
    // (decode_chunks never splits more pairs than the header has tickets for: 512, one per 128-byte line)
    a.counters = reinterpret_cast<unsigned*>(workspace);
    a.ws_acc = reinterpret_cast<float*>((unsigned char*)workspace + kDecodeHeaderBytes);
    a.ws_ml = a.ws_acc + (size_t)batch * h * a.num_splits * hd;
    // (a pass size the head_dim does not have — 256 at hd 128, 64 at hd 64 — leaves the default choice in place)
    a.impl = variant >= NVH_DECODE_CHUNKED_P128 ? NVH_DECODE_CHUNKED : variant; a.waves = waves;
    a.pass_tokens = variant == NVH_DECODE_CHUNKED_P128 ? 128 : (variant == NVH_DECODE_CHUNKED_P256 ? 256 : (variant == NVH_DECODE_CHUNKED_P64 ? 64 : 0));
    a.out_packed = (uint16_t*)out_packed;
    a.q_row_stride = q_row_stride; a.bt_row_stride = bt_row_stride;
    a.scale_log2 = scale * kLog2e;
    a.out_f32 = out_dtype == NVH_F32;
    a.stamps = g_stamps;
    return 0;
}

static int paged_decode_impl(void* out, void* out_packed, const void* q, const void* k_cache, const void* v_cache,
                             const int32_t* block_tables, const int32_t* context_lens,
                             int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                             int64_t q_row_stride, int64_t bt_row_stride, float scale,
                             int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream,
                             int variant = NVH_DECODE_CHUNKED, int waves = 0, int chunks = 0) {
    DecodeArgs a;
    const int rc = paged_decode_args(a, out, out_packed, q, k_cache, v_cache, block_tables, context_lens, batch, h, kvh, hd, block_size, max_blocks,
                                     q_row_stride, bt_row_stride, scale, dtype, out_dtype, workspace, workspace_bytes, variant, waves, chunks);
    if (rc) return rc > 0 ? 0 : rc;
    return launch_paged_decode(a, (hipStream_t)stream);
}

int nvh_paged_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                     const int32_t* block_tables, const int32_t* context_lens,
                     int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                     int64_t q_row_stride, int64_t bt_row_stride, float scale,
                     int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return paged_decode_impl(out, nullptr, q, k_cache, v_cache, block_tables, context_lens, batch, h, kvh, hd, block_size, max_blocks,
                             q_row_stride, bt_row_stride, scale, dtype, out_dtype, workspace, workspace_bytes, stream);
}

int nvh_paged_decode_variant(int variant, int waves, int chunks, void* out, const void* q, const void* k_cache, const void* v_cache,
                             const int32_t* block_tables, const int32_t* context_lens,
                             int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                             int64_t q_row_stride, int64_t bt_row_stride, float scale,
                             int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return paged_decode_impl(out, nullptr, q, k_cache, v_cache, block_tables, context_lens, batch, h, kvh, hd, block_size, max_blocks,
                             q_row_stride, bt_row_stride, scale, dtype, out_dtype, workspace, workspace_bytes, stream, variant, waves, chunks);
}

int nvh_paged_decode_packed(void* out, void* out_packed, const void* q, const void* k_cache, const void* v_cache,
                            const int32_t* block_tables, const int32_t* context_lens,
                            int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                            int64_t q_row_stride, int64_t bt_row_stride, float scale,
                            int dtype, int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!out_packed) { set_error("paged_decode_packed: out_packed is NULL"); return NVH_E_NULL; }
    return paged_decode_impl(out, out_packed, q, k_cache, v_cache, block_tables, context_lens, batch, h, kvh, hd, block_size, max_blocks,
                             q_row_stride, bt_row_stride, scale, dtype, out_dtype, workspace, workspace_bytes, stream);
}

int nvh_decode_step(void* out, const void* q, const void* k_new, const void* v_new,
                    void* k_cache, void* v_cache, const int32_t* slot_mapping,
                    const int32_t* block_tables, const int32_t* context_lens,
                    int batch, int h, int kvh, int hd, int block_size, int max_blocks,
                    int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                    int64_t bt_row_stride, float scale, int dtype, int out_dtype,
                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = nvh_store_kvcache(k_new, v_new, k_cache, v_cache, slot_mapping, batch, kvh, hd,
                               k_row_stride, v_row_stride, dtype, stream);
    if (rc) return rc;
    return nvh_paged_decode(out, q, k_cache, v_cache, block_tables, context_lens, batch, h, kvh, hd,
                            block_size, max_blocks, q_row_stride, bt_row_stride, scale, dtype, out_dtype,
                            workspace, workspace_bytes, stream);
}

static int prefill_varlen_impl(int kernel, int short_waves, void* out, const void* q, const void* k, const void* v,
                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                       const int32_t* block_tables, int batch, int max_seqlen_q, int max_seqlen_k,
                       int h, int kvh, int hd, int block_size, int max_blocks,
                       int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                       int64_t bt_row_stride, float scale, int dtype, int out_dtype, void* stream,
                       bool pv16 = false, int total_k = 0, void* pv16_scratch = nullptr, size_t pv16_scratch_bytes = 0) {
    if (batch == 0 || max_seqlen_q == 0) return 0;
    if (kernel < 0 || kernel > 3 || (kernel == 3 && block_tables) || (short_waves != 0 && short_waves != 8 && short_waves != 16)) {
        set_error("prefill_varlen: kernel %d / short_waves %d not supported", kernel, short_waves);
        return NVH_E_SHAPE;
    }
    if (dtype != NVH_BF16 || (out_dtype != NVH_BF16 && out_dtype != NVH_F32)) {
        set_error("prefill_varlen: dtype %d / out_dtype %d unsupported", dtype, out_dtype);
        return NVH_E_DTYPE;
    }
    if (!out || !q || !k || !v || !cu_seqlens_q || !cu_seqlens_k) { set_error("prefill_varlen: null pointer"); return NVH_E_NULL; }
    int rc = check_heads("prefill_varlen", h, kvh, hd);
    if (rc) return rc;
    if (batch < 0 || max_seqlen_q < 0 || max_seqlen_k < 0) { set_error("prefill_varlen: negative size"); return NVH_E_SHAPE; }
    if (q_row_stride % 8 || q_row_stride < (int64_t)h * hd) {
        set_error("prefill_varlen: q row stride %lld", (long long)q_row_stride);
        return NVH_E_STRIDE;
    }
    if (block_tables) {
        if (block_size <= 0 || block_size % 64 != 0 || max_blocks <= 0 || bt_row_stride < max_blocks) {
            set_error("prefill_varlen: paged mode needs block_size/max_blocks/bt_row_stride");
            return NVH_E_SHAPE;
        }
    } else if (k_row_stride % 8 || v_row_stride % 8 || k_row_stride < (int64_t)kvh * hd || v_row_stride < (int64_t)kvh * hd) {
        set_error("prefill_varlen: k/v row strides %lld/%lld", (long long)k_row_stride, (long long)v_row_stride);
        return NVH_E_STRIDE;
    }
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) {
        set_error("prefill_varlen: pointers must be 16-byte aligned");
        return NVH_E_ALIGN;
    }
    PrefillArgs a;
    a.out = out;
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v;
    a.cu_q = cu_seqlens_q; a.cu_k = cu_seqlens_k; a.block_tables = block_tables;
    a.batch = batch; a.max_seqlen_q = max_seqlen_q; a.max_seqlen_k = max_seqlen_k;
    a.h = h; a.kvh = kvh; a.hd = hd; a.block_size = block_size; a.max_blocks = max_blocks;
    a.q_row_stride = q_row_stride; a.k_row_stride = k_row_stride; a.v_row_stride = v_row_stride;
    a.bt_row_stride = bt_row_stride;
    a.scale_log2 = scale * kLog2e;
    a.out_f32 = out_dtype == NVH_F32;
    a.stamps = g_stamps;
    a.kernel = kernel; a.short_waves = short_waves;
    if (pv16) {
        if (block_tables) { set_error("prefill_varlen_pv16: not with a block table (V is read from the cache)"); return NVH_E_SHAPE; }
        if (!nvh_prefill_pv16_uses_scratch(batch, max_seqlen_q, max_seqlen_k, kvh, hd)) {
            // the short-sequence kernel's shapes: it converts its resident V images itself (head_dim 64, more than one key tile: below, the conversion
            // costs what it saves; otherwise the kernel keeps hi + lo); scratch untouched
            a.short_pv16 = hd == 64 && max_seqlen_k > 64;
        } else {
            // fp16 P V on an fp16 copy of V made here: [one range flag per 64 rows, padded to 256 bytes][total_k rows of kvh*hd fp16].  Two launches, no
            // host read, nothing to clear: convert (every workgroup writes its group's flag), attend (a sequence with a flagged group falls back to `v`).
            if (!pv16_scratch) { set_error("prefill_varlen_pv16: null scratch"); return NVH_E_NULL; }
            if (total_k <= 0 || !aligned16(pv16_scratch) || pv16_scratch_bytes < nvh_prefill_pv16_scratch_bytes(total_k, kvh, hd)) {
                set_error("prefill_varlen_pv16: scratch of %zu bytes, %zu needed for %d rows (16-byte aligned)", pv16_scratch_bytes,
                          nvh_prefill_pv16_scratch_bytes(total_k, kvh, hd), total_k);
                return NVH_E_WORKSPACE;
            }
            int32_t* flags = (int32_t*)pv16_scratch;
            uint16_t* v16 = (uint16_t*)((char*)pv16_scratch + pv16_header_bytes(total_k));
            rc = launch_bf16_rows_to_f16(v16, v, total_k, kvh * hd, v_row_stride, (int64_t)kvh * hd, flags, (hipStream_t)stream);
            if (rc) return rc;
            a.v16 = v16; a.v16_row_stride = (int64_t)kvh * hd; a.pv16_flags = flags; a.pv16_rows = total_k;
            a.kernel = 1;                                        // the tiled kernel
        }
    }
    return launch_prefill_varlen(a, (hipStream_t)stream);
}

int nvh_prefill_pv16_uses_scratch(int batch, int max_seqlen_q, int max_seqlen_k, int kvh, int hd) {
    (void)hd;
    const bool short_kernel = max_seqlen_k <= 128 && max_seqlen_q <= max_seqlen_k && (int64_t)batch * kvh >= 128;     // launch_short's rule (prefill_mfma.hip)
    return short_kernel ? 0 : 1;
}

size_t nvh_prefill_pv16_scratch_bytes(int total_k, int kvh, int hd) {
    if (total_k <= 0 || kvh <= 0 || hd <= 0) return 0;
    return pv16_header_bytes(total_k) + (size_t)total_k * kvh * hd * 2;
}

int nvh_prefill_varlen_pv16(void* out, const void* q, const void* k, const void* v,
                            const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int batch, int max_seqlen_q, int max_seqlen_k, int total_k,
                            int h, int kvh, int hd, int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                            float scale, int dtype, int out_dtype, void* scratch, size_t scratch_bytes, void* stream) {
    if (batch == 0 || max_seqlen_q == 0) return 0;
    return prefill_varlen_impl(0, 0, out, q, k, v, cu_seqlens_q, cu_seqlens_k, nullptr, batch, max_seqlen_q, max_seqlen_k, h, kvh, hd,
                               0, 0, q_row_stride, k_row_stride, v_row_stride, 0, scale, dtype, out_dtype, stream, true, total_k, scratch, scratch_bytes);
}

int nvh_prefill_varlen(void* out, const void* q, const void* k, const void* v,
                       const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                       const int32_t* block_tables, int batch, int max_seqlen_q, int max_seqlen_k,
                       int h, int kvh, int hd, int block_size, int max_blocks,
                       int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                       int64_t bt_row_stride, float scale, int dtype, int out_dtype, void* stream) {
    return prefill_varlen_impl(0, 0, out, q, k, v, cu_seqlens_q, cu_seqlens_k, block_tables, batch, max_seqlen_q, max_seqlen_k, h, kvh, hd,
                               block_size, max_blocks, q_row_stride, k_row_stride, v_row_stride, bt_row_stride, scale, dtype, out_dtype, stream);
}

int nvh_prefill_varlen_variant(int kernel, int short_waves, void* out, const void* q, const void* k, const void* v,
                               const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                               const int32_t* block_tables, int batch, int max_seqlen_q, int max_seqlen_k,
                               int h, int kvh, int hd, int block_size, int max_blocks,
                               int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride,
                               int64_t bt_row_stride, float scale, int dtype, int out_dtype, void* stream) {
    return prefill_varlen_impl(kernel, short_waves, out, q, k, v, cu_seqlens_q, cu_seqlens_k, block_tables, batch, max_seqlen_q, max_seqlen_k, h, kvh, hd,
                               block_size, max_blocks, q_row_stride, k_row_stride, v_row_stride, bt_row_stride, scale, dtype, out_dtype, stream);
}

int nvh_bf16_rows_to_f16(void* out, const void* in, int n_rows, int row_elems, int64_t in_row_stride, int64_t out_row_stride, void* stream) {
    if (n_rows == 0) return 0;
    if (!out || !in) { set_error("bf16_rows_to_f16: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || row_elems <= 0 || row_elems % 8) { set_error("bf16_rows_to_f16: n_rows %d, row_elems %d (a positive multiple of 8)", n_rows, row_elems); return NVH_E_SHAPE; }
    if (in_row_stride % 8 || out_row_stride % 8 || in_row_stride < row_elems || out_row_stride < row_elems) { set_error("bf16_rows_to_f16: bad row strides"); return NVH_E_STRIDE; }
    if (!aligned16(out) || !aligned16(in)) { set_error("bf16_rows_to_f16: pointers must be 16-byte aligned"); return NVH_E_ALIGN; }
    return launch_bf16_rows_to_f16(out, in, n_rows, row_elems, in_row_stride, out_row_stride, nullptr, (hipStream_t)stream);
}

int nvh_rope_store(void* qkv, const int64_t* positions, const float* cos_sin,
                   const void* q_norm_w, const void* k_norm_w, float eps,
                   void* k_cache, void* v_cache, const int32_t* slot_mapping,
                   int n_tokens, int h, int kvh, int hd, int64_t qkv_row_stride, int dtype, void* stream) {
    if (n_tokens == 0) return 0;
    if (dtype != NVH_BF16) { set_error("rope_store: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!qkv || !positions || !cos_sin) { set_error("rope_store: null pointer"); return NVH_E_NULL; }
    if ((q_norm_w == nullptr) != (k_norm_w == nullptr)) { set_error("rope_store: q_norm_w and k_norm_w must both be set or both be NULL"); return NVH_E_NULL; }
    if ((k_cache == nullptr) != (v_cache == nullptr)) { set_error("rope_store: k_cache and v_cache must both be set or both be NULL"); return NVH_E_NULL; }
    if (n_tokens < 0 || h <= 0 || kvh <= 0 || (hd != 64 && hd != 128)) { set_error("rope_store: bad shape h=%d kvh=%d hd=%d", h, kvh, hd); return NVH_E_SHAPE; }
    if (qkv_row_stride % 8 || qkv_row_stride < (int64_t)(h + 2 * kvh) * hd) {
        set_error("rope_store: qkv row stride %lld", (long long)qkv_row_stride);
        return NVH_E_STRIDE;
    }
    if (!aligned16(qkv) || !aligned16(cos_sin) || (k_cache && (!aligned16(k_cache) || !aligned16(v_cache))) ||
        (q_norm_w && (!aligned16(q_norm_w) || !aligned16(k_norm_w)))) {
        set_error("rope_store: pointers must be 16-byte aligned");
        return NVH_E_ALIGN;
    }
    RopeStoreArgs a;
    a.qkv = (uint16_t*)qkv; a.positions = positions; a.cos_sin = cos_sin;
    a.q_norm_w = (const uint16_t*)q_norm_w; a.k_norm_w = (const uint16_t*)k_norm_w; a.eps = eps;
    a.k_cache = (uint16_t*)k_cache; a.v_cache = (uint16_t*)v_cache; a.slot_mapping = slot_mapping;
    a.n_tokens = n_tokens; a.h = h; a.kvh = kvh; a.hd = hd; a.qkv_row_stride = qkv_row_stride;
    return launch_rope_store(a, (hipStream_t)stream);
}

int nvh_add_rmsnorm(void* out, const void* x, void* residual, const void* weight, float eps, int n_rows, int hidden,
                    int64_t x_row_stride, int64_t out_row_stride, int64_t residual_row_stride, int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("add_rmsnorm: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!out || !x || !weight) { set_error("add_rmsnorm: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || hidden <= 0 || hidden % 8 || hidden > max_rmsnorm_hidden()) { set_error("add_rmsnorm: hidden %d unsupported", hidden); return NVH_E_SHAPE; }
    if (x_row_stride % 8 || out_row_stride % 8 || (residual && residual_row_stride % 8) || x_row_stride < hidden || out_row_stride < hidden) {
        set_error("add_rmsnorm: row strides must be multiples of 8 and >= hidden");
        return NVH_E_STRIDE;
    }
    if (!aligned16(out) || !aligned16(x) || !aligned16(weight) || (residual && !aligned16(residual))) { set_error("add_rmsnorm: pointers must be 16-byte aligned"); return NVH_E_ALIGN; }
    return launch_add_rmsnorm(out, x, residual, weight, eps, n_rows, hidden, x_row_stride, out_row_stride, residual_row_stride, (hipStream_t)stream);
}

int nvh_residual_add_pack(void* residual, const void* y, void* packed, int n_rows, int hidden, int64_t residual_row_stride,
                          int64_t y_row_stride, int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("residual_add_pack: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!residual || !y) { set_error("residual_add_pack: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || hidden <= 0 || hidden % 32) { set_error("residual_add_pack: hidden %d must be a positive multiple of 32", hidden); return NVH_E_SHAPE; }
    if (residual_row_stride % 8 || y_row_stride % 8 || residual_row_stride < hidden || y_row_stride < hidden) { set_error("residual_add_pack: bad row strides"); return NVH_E_STRIDE; }
    if (!aligned16(residual) || !aligned16(y) || (packed && !aligned16(packed))) { set_error("residual_add_pack: pointers must be 16-byte aligned"); return NVH_E_ALIGN; }
    return launch_residual_add_pack(residual, y, packed, n_rows, hidden, residual_row_stride, y_row_stride, (hipStream_t)stream);
}

int nvh_silu_mul(void* out, const void* gate_up, int n_rows, int inter, int64_t gate_up_row_stride, int64_t out_row_stride,
                 int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("silu_mul: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!out || !gate_up) { set_error("silu_mul: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || inter <= 0 || inter % 8) { set_error("silu_mul: inter %d must be a positive multiple of 8", inter); return NVH_E_SHAPE; }
    if (gate_up_row_stride % 8 || out_row_stride % 8 || gate_up_row_stride < 2 * (int64_t)inter || out_row_stride < inter) {
        set_error("silu_mul: bad row strides");
        return NVH_E_STRIDE;
    }
    if (!aligned16(out) || !aligned16(gate_up)) { set_error("silu_mul: pointers must be 16-byte aligned"); return NVH_E_ALIGN; }
    return launch_silu_mul(out, gate_up, n_rows, inter, gate_up_row_stride, out_row_stride, (hipStream_t)stream);
}

int nvh_argmax_rows(int64_t* out, const void* x, int n_rows, int n, int64_t x_row_stride, int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("argmax_rows: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!out || !x) { set_error("argmax_rows: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || n <= 0) { set_error("argmax_rows: bad shape"); return NVH_E_SHAPE; }
    if (x_row_stride % 8 || x_row_stride < n) { set_error("argmax_rows: row stride must be a multiple of 8 and >= n"); return NVH_E_STRIDE; }
    if (!aligned16(x)) { set_error("argmax_rows: x must be 16-byte aligned"); return NVH_E_ALIGN; }
    return launch_argmax_rows(out, x, n_rows, n, x_row_stride, AdvanceArgs{}, (hipStream_t)stream);
}

int nvh_greedy_advance(const void* logits, int n_rows, int n, int64_t logits_row_stride,
                       int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                       const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                       int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps, int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("greedy_advance: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!logits || !input_ids || !positions || !context_lens || !slot_mapping || !block_tables || !tokens_log || !row_steps) { set_error("greedy_advance: null pointer"); return NVH_E_NULL; }
    if (n_rows < 0 || n <= 0 || block_size <= 0 || log_row_stride < n_rows) { set_error("greedy_advance: bad shape"); return NVH_E_SHAPE; }
    if (logits_row_stride % 8 || logits_row_stride < n) { set_error("greedy_advance: logits row stride must be a multiple of 8 and >= n"); return NVH_E_STRIDE; }
    if (!aligned16(logits)) { set_error("greedy_advance: logits must be 16-byte aligned"); return NVH_E_ALIGN; }
    AdvanceArgs adv{input_ids, positions, context_lens, slot_mapping, block_tables, bt_row_stride, block_size, tokens_log, log_row_stride, row_steps};
    return launch_argmax_rows(nullptr, logits, n_rows, n, logits_row_stride, adv, (hipStream_t)stream);
}

int nvh_linear_small_m(void* out, const void* x, const void* w, const void* bias, int m, int n, int k, int silu_inter,
                       int64_t x_row_stride, int64_t out_row_stride, int dtype, void* stream) {
    if (m == 0) return 0;
    if (dtype != NVH_BF16) { set_error("linear_small_m: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!out || !x || !w) { set_error("linear_small_m: null pointer"); return NVH_E_NULL; }
    if (m < 0 || m > 64 || n <= 0 || k <= 0 || k % 32 || n % 16) { set_error("linear_small_m: m=%d (<=64) n=%d (%%16) k=%d (%%32)", m, n, k); return NVH_E_SHAPE; }
    if (silu_inter && (silu_inter * 2 != n || silu_inter % 16 || bias)) { set_error("linear_small_m: silu mode needs n == 2*inter, inter %% 16 == 0, no bias"); return NVH_E_SHAPE; }
    const int out_cols = silu_inter ? silu_inter : n;
    if (x_row_stride % 8 || x_row_stride < k || out_row_stride < out_cols) { set_error("linear_small_m: bad row strides"); return NVH_E_STRIDE; }
    if (!aligned16(x) || !aligned16(w)) { set_error("linear_small_m: x and w must be 16-byte aligned"); return NVH_E_ALIGN; }
    nvh_linear_desc d = {};
    d.out = out; d.x = x; d.w = w; d.bias = bias; d.m = m; d.n = n; d.k = k; d.silu_inter = silu_inter;
    d.x_row_stride = x_row_stride; d.out_row_stride = out_row_stride;
    d.epilogue = silu_inter ? NVH_EPI_SILU_MUL : NVH_EPI_NONE;
    return nvh_linear_small_m_ex(&d, dtype, stream);
}

// validates the descriptor and fills `a`; returns 1 when there is nothing to do (m == 0), 0 when `a` is ready, < 0 on rejection
static int linear_args(LinearArgs& a, const nvh_linear_desc* d, int dtype) {
    if (!d) { set_error("linear_small_m_ex: null descriptor"); return NVH_E_NULL; }
    if (d->m == 0) return 1;
    if (dtype != NVH_BF16) { set_error("linear_small_m_ex: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    const bool out_optional = (d->out_packed && (d->epilogue == NVH_EPI_NONE || d->epilogue == NVH_EPI_SILU_MUL)) ||
                              (d->candidate_val && d->epilogue == NVH_EPI_NONE);
    if ((d->candidate_val != nullptr) != (d->candidate_idx != nullptr) || (d->candidate_val && d->candidate_stride < d->m)) {
        set_error("linear_small_m_ex: candidate_val / candidate_idx come together, candidate_stride >= m");
        return NVH_E_NULL;
    }
    if ((!d->out && !out_optional) || !d->x || !d->w) { set_error("linear_small_m_ex: null pointer"); return NVH_E_NULL; }
    if (d->m < 0 || d->m > 64 || d->n <= 0 || d->k <= 0 || d->k % 64 || d->n % 16) {
        set_error("linear_small_m_ex: m=%d (<=64) n=%d (%%16) k=%d (%%64)", d->m, d->n, d->k);
        return NVH_E_SHAPE;
    }
    int out_cols = d->n;
    switch (d->epilogue) {
        case NVH_EPI_NONE: break;
        case NVH_EPI_SILU_MUL:
            if (d->silu_inter * 2 != d->n || d->silu_inter % 16 || d->bias) { set_error("linear_small_m_ex: silu needs n == 2*inter, inter %% 16 == 0, no bias"); return NVH_E_SHAPE; }
            out_cols = d->silu_inter;
            break;
        case NVH_EPI_RESIDUAL_ADD:
            if (d->bias) { set_error("linear_small_m_ex: residual-add epilogue takes no bias"); return NVH_E_SHAPE; }
            break;
        case NVH_EPI_ROPE_STORE:
            if ((d->hd != 64 && d->hd != 128) || d->h <= 0 || d->kvh <= 0 || (d->h + 2 * d->kvh) * d->hd != d->n) {
                set_error("linear_small_m_ex: rope epilogue needs n == (h + 2*kvh) * hd, hd in {64,128}");
                return NVH_E_SHAPE;
            }
            if (!d->positions || !d->cos_sin || !d->k_cache || !d->v_cache || !d->slot_mapping) { set_error("linear_small_m_ex: rope epilogue pointers"); return NVH_E_NULL; }
            out_cols = d->h * d->hd;
            break;
        default: set_error("linear_small_m_ex: unknown epilogue %d", d->epilogue); return NVH_E_SHAPE;
    }
    if ((!d->x_packed && (d->x_row_stride % 8 || d->x_row_stride < d->k)) || (d->out && d->out_row_stride < out_cols)) { set_error("linear_small_m_ex: bad row strides"); return NVH_E_STRIDE; }
    if (d->out_packed && (!aligned16(d->out_packed) || out_cols % 32)) { set_error("linear_small_m_ex: out_packed needs 16-byte alignment and cols %% 32 == 0"); return NVH_E_ALIGN; }
    if (d->workspace && !aligned16(d->workspace)) { set_error("linear_small_m_ex: workspace must be 16-byte aligned"); return NVH_E_ALIGN; }
    if (!aligned16(d->x) || !aligned16(d->w) || (d->norm_weight && !aligned16(d->norm_weight))) { set_error("linear_small_m_ex: x, w, norm_weight must be 16-byte aligned"); return NVH_E_ALIGN; }
    a.out = d->out; a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.bias = (const uint16_t*)d->bias;
    a.M = d->m; a.N = d->n; a.K = d->k; a.inter = d->silu_inter; a.x_stride = d->x_row_stride; a.out_stride = d->out_row_stride;
    if (d->norm_folded && d->norm_weight) { set_error("linear_small_m_ex: norm_folded excludes norm_weight"); return NVH_E_NULL; }
    a.norm_w = (const uint16_t*)d->norm_weight; a.norm_eps = d->norm_eps; a.epi = d->epilogue;
    a.norm_mode = d->norm_folded ? 2 : (d->norm_weight ? 1 : 0);
    a.positions = d->positions; a.cos_sin = d->cos_sin; a.k_cache = (uint16_t*)d->k_cache; a.v_cache = (uint16_t*)d->v_cache;
    a.slots = d->slot_mapping; a.h = d->h; a.kvh = d->kvh; a.hd = d->hd;
    a.stamps = g_stamps;
    a.x_packed = d->x_packed; a.out_packed = (uint16_t*)d->out_packed; a.ws_raw = d->workspace; a.ws_bytes = d->workspace_bytes;
    a.ws = nullptr; a.counters = nullptr; a.ksplit = 1; a.tiles = 0;
    a.cand_val = d->candidate_val; a.cand_idx = d->candidate_idx; a.cand_stride = d->candidate_stride;
    // prefetch hint: whole 128-byte lines inside [prefetch, prefetch + prefetch_bytes) only (reads never leave the caller's range)
    // Ranges above 4 MiB are ignored: the hint pays for operands whose FIRST-BYTE latency dominates their consumer (a few MB read by a
    // few dozen workgroups); 8.7 and 17.4 MB ranges measured +-0 and +1.3 % on the decode step (the prefetching workgroups outlive the launch)
    a.pf_ptr = nullptr; a.pf_bytes = 0;
    a.pf_kv = {};
#ifndef NVH_PF_MAX_MB
#define NVH_PF_MAX_MB 4                      // (A/B builds lift it)
#endif
    if (d->prefetch && d->prefetch_bytes >= 256 && d->prefetch_bytes <= ((size_t)NVH_PF_MAX_MB << 20)) {
        const uintptr_t p0 = ((uintptr_t)d->prefetch + 127) & ~(uintptr_t)127, p1 = ((uintptr_t)d->prefetch + d->prefetch_bytes) & ~(uintptr_t)127;
        if (p1 > p0) { a.pf_ptr = (const void*)p0; a.pf_bytes = (int64_t)(p1 - p0); }
    }
    return 0;
}

int nvh_linear_small_m_ex(const nvh_linear_desc* d, int dtype, void* stream) {
    LinearArgs a;
    const int rca = linear_args(a, d, dtype);
    if (rca) return rca > 0 ? 0 : rca;
    const int rc = launch_linear_stream(a, (hipStream_t)stream);
    if (rc != -100) return rc;
    if (d->x_packed || d->out_packed || d->candidate_val) {
        set_error("linear_small_m_ex: packed activations / candidates need the streaming kernel (k %% 64 == 0, no exact-norm prologue, workspace when k > 1024)");
        return NVH_E_SHAPE;
    }
    return launch_linear_small_m(a, (hipStream_t)stream);
}

size_t nvh_linear_small_m_workspace(int m, int n, int k, int epilogue) { return linear_stream_workspace_bytes(m, n, k, epilogue); }
int nvh_linear_small_m_candidate_groups(int n, int k) { return linear_stream_candidate_groups(n, k); }

// ---- qkv projection (RoPE / store epilogue) + decode attention
static int qkv_rope_attend_impl(int mode, uint32_t spin_limit, int missing_producers, const nvh_linear_desc* qkv, void* attn_out, void* attn_out_packed,
                                const int32_t* block_tables, const int32_t* context_lens, int block_size, int max_blocks,
                                int64_t bt_row_stride, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream, int* fused_out) {
    if (fused_out) *fused_out = 0;
    if (!qkv) { set_error("qkv_rope_attend: null descriptor"); return NVH_E_NULL; }
    if (qkv->epilogue != NVH_EPI_ROPE_STORE || !qkv->out) { set_error("qkv_rope_attend: the descriptor must be a ROPE_STORE projection with `out` (the q rows)"); return NVH_E_SHAPE; }
    if (mode < 0 || mode > 3 || missing_producers < 0) { set_error("qkv_rope_attend: mode %d / missing_producers %d", mode, missing_producers); return NVH_E_SHAPE; }
    LinearArgs l;
    int rc = linear_args(l, qkv, dtype);
    if (rc) return rc > 0 ? 0 : rc;
    DecodeArgs d;
    rc = paged_decode_args(d, attn_out, attn_out_packed, qkv->out, qkv->k_cache, qkv->v_cache, block_tables, context_lens, qkv->m, qkv->h, qkv->kvh,
                           qkv->hd, block_size, max_blocks, qkv->out_row_stride, bt_row_stride, scale, dtype, NVH_BF16, workspace, workspace_bytes);
    if (rc) return rc > 0 ? 0 : rc;
    const bool can = qkv_attend_supported(l, d);
    if (mode == 2 && !can) { set_error("qkv_rope_attend: this shape cannot run as one launch (hd 64, k <= 1024, packed x, folded norm, grid <= 2 x CUs)"); return NVH_E_SHAPE; }
    // mode 0 (what nvh_qkv_rope_attend runs) takes the measured winner: TWO launches.  On MI355X the one-launch form loses at every
    // decode shape tried (its K/V stream fills the memory system's queues in front of the very loads the hand-off
    // chain waits for); it stays reachable (mode 2) and under the same parity tests.
    if (mode == 2 && can) {
        if (fused_out) *fused_out = 1;
        return launch_qkv_attend(l, d, (unsigned char*)workspace + kDecodeTicketBytes, spin_limit, missing_producers, l.pf_ptr, l.pf_bytes, (hipStream_t)stream);
    }
    if (mode == 3 && d.hd == 64) {
        // two launches, the first one's idle CUs touching the attention launch's first K/V images (its geometry: paged_decode.hip launch_chunked)
        const int chunks = d.chunks;
        l.pf_kv.k_cache = d.k_cache; l.pf_kv.v_cache = d.v_cache; l.pf_kv.block_tables = d.block_tables; l.pf_kv.context_lens = d.context_lens;
        l.pf_kv.bt_stride = d.bt_row_stride; l.pf_kv.batch = d.batch; l.pf_kv.kvh = d.kvh; l.pf_kv.hd = d.hd; l.pf_kv.block_size = d.block_size;
        l.pf_kv.chunks = chunks; l.pf_kv.pass_tokens = (chunks >= 3 && chunks <= 5) ? 128 : 256;
        l.pf_kv.passes = spin_limit ? (int)spin_limit : 1;            // (A/B: the variant's spin_limit argument carries the number of passes here)
        if (d.block_size % l.pf_kv.pass_tokens) l.pf_kv.k_cache = nullptr;
    }
    rc = launch_linear_stream(l, (hipStream_t)stream);
    if (rc == -100) {
        if (qkv->x_packed || qkv->out_packed) { set_error("qkv_rope_attend: packed activations need the streaming kernel"); return NVH_E_SHAPE; }
        rc = launch_linear_small_m(l, (hipStream_t)stream);
    }
    return rc ? rc : launch_paged_decode(d, (hipStream_t)stream);
}

int nvh_qkv_rope_attend(const nvh_linear_desc* qkv, void* attn_out, void* attn_out_packed, const int32_t* block_tables, const int32_t* context_lens,
                        int block_size, int max_blocks, int64_t bt_row_stride, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return qkv_rope_attend_impl(0, 0, 0, qkv, attn_out, attn_out_packed, block_tables, context_lens, block_size, max_blocks, bt_row_stride, scale, dtype,
                                workspace, workspace_bytes, stream, nullptr);
}

int nvh_qkv_rope_attend_variant(int mode, uint32_t spin_limit, int missing_producers, int* fused_out, const nvh_linear_desc* qkv, void* attn_out,
                                void* attn_out_packed, const int32_t* block_tables, const int32_t* context_lens, int block_size, int max_blocks,
                                int64_t bt_row_stride, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return qkv_rope_attend_impl(mode, spin_limit, missing_producers, qkv, attn_out, attn_out_packed, block_tables, context_lens, block_size, max_blocks,
                                bt_row_stride, scale, dtype, workspace, workspace_bytes, stream, fused_out);
}

int nvh_qkv_rope_attend_status(const void* workspace, uint32_t* timed_out) {
    if (!workspace || !timed_out) { set_error("qkv_rope_attend_status: null pointer"); return NVH_E_NULL; }
    const hipError_t e = hipMemcpy(timed_out, (const unsigned char*)workspace + kDecodeTicketBytes + 4096, sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("qkv_rope_attend_status: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

static int greedy_advance_candidates_impl(const float* candidate_val, const int32_t* candidate_idx, int groups, int64_t candidate_stride,
                                  int n_rows, int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                                  const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                                  int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps,
                                  const void* embed, int vocab, int hidden, void* hidden_out, int64_t hidden_row_stride, void* hidden_packed, void* stream) {
    if (n_rows == 0) return 0;
    if (!candidate_val || !candidate_idx || !input_ids || !positions || !context_lens || !slot_mapping || !block_tables || !tokens_log || !row_steps) {
        set_error("greedy_advance_candidates: null pointer");
        return NVH_E_NULL;
    }
    if (n_rows < 0 || groups <= 0 || candidate_stride < n_rows || block_size <= 0 || log_row_stride < n_rows) { set_error("greedy_advance_candidates: bad shape"); return NVH_E_SHAPE; }
    AdvanceArgs adv{input_ids, positions, context_lens, slot_mapping, block_tables, bt_row_stride, block_size, tokens_log, log_row_stride, row_steps};
    if (embed) {
        if (!hidden_out) { set_error("greedy_advance_candidates_embed: null pointer"); return NVH_E_NULL; }
        if (vocab <= 0) { set_error("greedy_advance_candidates_embed: vocab %d", vocab); return NVH_E_SHAPE; }
        if (hidden <= 0 || hidden % 32 != 0 || hidden_row_stride < hidden || hidden_row_stride % 8 != 0) {
            set_error("greedy_advance_candidates_embed: hidden=%d must be a multiple of 32, row stride %lld a multiple of 8 and >= hidden", hidden, (long long)hidden_row_stride);
            return NVH_E_SHAPE;
        }
        if (!aligned16(embed) || !aligned16(hidden_out) || (hidden_packed && !aligned16(hidden_packed))) {
            set_error("greedy_advance_candidates_embed: embed / hidden_out / hidden_packed must be 16-byte aligned");
            return NVH_E_ALIGN;
        }
        adv.embed = (const uint16_t*)embed; adv.hidden = hidden; adv.hidden_out = (uint16_t*)hidden_out;
        adv.hidden_stride = hidden_row_stride; adv.hidden_packed = (uint16_t*)hidden_packed;
    }
    return launch_argmax_candidates(candidate_val, candidate_idx, groups, candidate_stride, n_rows, embed ? vocab : 0x7fffffff, adv, (hipStream_t)stream);
}
int nvh_greedy_advance_candidates(const float* candidate_val, const int32_t* candidate_idx, int groups, int64_t candidate_stride,
                                  int n_rows, int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                                  const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                                  int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps, void* stream) {
    return greedy_advance_candidates_impl(candidate_val, candidate_idx, groups, candidate_stride, n_rows, input_ids, positions, context_lens,
                                          slot_mapping, block_tables, bt_row_stride, block_size, tokens_log, log_row_stride, row_steps,
                                          nullptr, 0, 0, nullptr, 0, nullptr, stream);
}
int nvh_greedy_advance_candidates_embed(const float* candidate_val, const int32_t* candidate_idx, int groups, int64_t candidate_stride,
                                        int n_rows, int64_t* input_ids, int64_t* positions, int32_t* context_lens, int32_t* slot_mapping,
                                        const int32_t* block_tables, int64_t bt_row_stride, int block_size,
                                        int64_t* tokens_log, int64_t log_row_stride, int64_t* row_steps,
                                        const void* embed_weight, int vocab, int hidden, void* hidden_out, int64_t hidden_row_stride, void* hidden_packed,
                                        int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("greedy_advance_candidates_embed: dtype %d not supported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!embed_weight) { set_error("greedy_advance_candidates_embed: null pointer"); return NVH_E_NULL; }
    return greedy_advance_candidates_impl(candidate_val, candidate_idx, groups, candidate_stride, n_rows, input_ids, positions, context_lens,
                                          slot_mapping, block_tables, bt_row_stride, block_size, tokens_log, log_row_stride, row_steps,
                                          embed_weight, vocab, hidden, hidden_out, hidden_row_stride, hidden_packed, stream);
}
int64_t nvh_pack_index(int row, int col, int cols) { return pack_index(row, col, cols); }

// ---- one-shot all-reduce over IPC-mapped peer buffers: set-up helpers (host-synchronous, not capturable) and the launch
static int hip_rc(const char* what, hipError_t e) {
    if (e == hipSuccess) return 0;
    set_error("%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return (int)e;
}

int nvh_comm_alloc(void** ptr, size_t bytes) {
    if (!ptr || bytes == 0) { set_error("comm_alloc: null pointer or zero size"); return NVH_E_NULL; }
    // fine-grained device memory: peers read and write it while kernels of this device run; coherence comes from the
    // system-scope fences and atomics of the all-reduce kernel
    int rc = hip_rc("comm_alloc: hipExtMallocWithFlags", hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
    if (rc) return rc;
    rc = hip_rc("comm_alloc: hipMemset", hipMemset(*ptr, 0, bytes));
    if (!rc) rc = hip_rc("comm_alloc: hipDeviceSynchronize", hipDeviceSynchronize());
    return rc;
}
int nvh_comm_free(void* ptr) { return ptr ? hip_rc("comm_free", hipFree(ptr)) : 0; }
int nvh_comm_ipc_export(void* ptr, void* handle_out) {
    if (!ptr || !handle_out) { set_error("comm_ipc_export: null pointer"); return NVH_E_NULL; }
    static_assert(sizeof(hipIpcMemHandle_t) == NVH_COMM_IPC_HANDLE_BYTES, "handle size of the C ABI");
    return hip_rc("comm_ipc_export: hipIpcGetMemHandle", hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), ptr));
}
int nvh_comm_ipc_open(const void* handle, void** ptr) {
    if (!handle || !ptr) { set_error("comm_ipc_open: null pointer"); return NVH_E_NULL; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return hip_rc("comm_ipc_open: hipIpcOpenMemHandle", hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}
int nvh_comm_ipc_close(void* ptr) { return ptr ? hip_rc("comm_ipc_close", hipIpcCloseMemHandle(ptr)) : 0; }
int nvh_allreduce_status(const void* state, uint32_t* calls_completed, uint32_t* failed_epoch) {
    if (!state) { set_error("allreduce_status: null state"); return NVH_E_NULL; }
    uint32_t host[4] = {0, 0, 0, 0};
    int rc = hip_rc("allreduce_status: hipDeviceSynchronize", hipDeviceSynchronize());
    if (!rc) rc = hip_rc("allreduce_status: hipMemcpy", hipMemcpy(host, state, sizeof(host), hipMemcpyDeviceToHost));
    if (rc) return rc;
    if (calls_completed) *calls_completed = host[0];
    if (failed_epoch) *failed_epoch = host[2];
    return 0;
}

size_t nvh_allreduce_stage_bytes(int max_rows, int hidden) {
    if (max_rows <= 0 || hidden <= 0 || hidden % 8) return 0;
    return 2 * (((size_t)max_rows * hidden * 2 + 255) & ~(size_t)255);           // two slots (epoch parity)
}
size_t nvh_allreduce_flag_bytes(int world) { return world > 0 ? (size_t)world * AR_MAX_BLOCKS * sizeof(uint32_t) : 0; }

int nvh_allreduce_oneshot(void* out, const void* x, void* packed, void* const* stage_ptrs, void* const* flag_ptrs, void* state,
                          int world, int rank, int rows, int hidden, int64_t x_row_stride, int64_t out_row_stride,
                          size_t stage_bytes, int epilogue, int dtype, void* stream) {
    if (rows == 0) return 0;
    if (dtype != NVH_BF16) { set_error("allreduce_oneshot: dtype %d unsupported (bf16 only)", dtype); return NVH_E_DTYPE; }
    if (!out || !x || !stage_ptrs || !flag_ptrs || !state) { set_error("allreduce_oneshot: null pointer"); return NVH_E_NULL; }
    if (world < 2 || world > 64 || rank < 0 || rank >= world || rows < 0 || hidden <= 0 || hidden % 8) {
        set_error("allreduce_oneshot: world %d (2..64) rank %d rows %d hidden %d (%%8)", world, rank, rows, hidden);
        return NVH_E_SHAPE;
    }
    if (epilogue != NVH_AR_EPI_NONE && epilogue != NVH_AR_EPI_RESIDUAL_ADD) { set_error("allreduce_oneshot: unknown epilogue %d", epilogue); return NVH_E_SHAPE; }
    if (packed && (epilogue != NVH_AR_EPI_RESIDUAL_ADD || hidden % 32)) { set_error("allreduce_oneshot: packed needs the residual-add epilogue and hidden %% 32 == 0"); return NVH_E_SHAPE; }
    if (x_row_stride % 8 || out_row_stride % 8 || x_row_stride < hidden || out_row_stride < hidden) { set_error("allreduce_oneshot: row strides must be multiples of 8 and >= hidden"); return NVH_E_STRIDE; }
    if (!aligned16(out) || !aligned16(x) || (packed && !aligned16(packed))) { set_error("allreduce_oneshot: pointers must be 16-byte aligned"); return NVH_E_ALIGN; }
    const size_t need = nvh_allreduce_stage_bytes(rows, hidden);
    if (stage_bytes < need || (stage_bytes / 2) % 16) { set_error("allreduce_oneshot: staging buffers of %zu B < required %zu B", stage_bytes, need); return NVH_E_WORKSPACE; }
    AllReduceArgs a;
    a.x = (const uint16_t*)x; a.out = (uint16_t*)out; a.packed = (uint16_t*)packed;
    a.stage = stage_ptrs; a.flags = reinterpret_cast<uint32_t* const*>(flag_ptrs); a.state = (uint32_t*)state;
    a.slot_bytes = stage_bytes / 2;
    a.world = world; a.rank = rank; a.rows = rows; a.hidden = hidden; a.epi = epilogue;
    a.x_stride = x_row_stride; a.out_stride = out_row_stride;
    return launch_allreduce_oneshot(a, allreduce_blocks(rows, hidden), (hipStream_t)stream);
}

}  // extern "C"
