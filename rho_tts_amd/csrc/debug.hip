// Kernel-level test hooks of the C ABI (include/rho_tts_amd.h, "kernel-level test hooks").
#include <algorithm>
#include <vector>

#include "kernels.h"

namespace {
__global__ void k_touch(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
template <int MODE>
__global__ void k_bench_barrier(unsigned* bar, int n, float* buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (i + blockDim.x) % (gridDim.x * blockDim.x);     // the next workgroup's element
    float acc = buf[i];
    for (int e = 1; e <= n; ++e) {
        if (MODE == 2) st_agent(buf + i, acc + 1.0f); else buf[i] = acc + 1.0f;      // something to publish ...
        if (!grid_barrier<MODE>(bar, gridDim.x, (unsigned)e, 1u << 20)) return;
        acc = MODE == 2 ? ld_agent(buf + j) : buf[j];                                 // ... and to fetch
        if (acc != (float)e && i == 0) bar[2] = 1u;                                   // stale read detector
    }
    buf[i] = acc;
}
// Pair hand-off ping-pong (rt_bench_grid_barrier modes 3 / 4): workgroups b and b ^ P (P = 1: neighbours in launch order, i.e. two
// XCDs; P = 8: the same XCD) pass 2 KB and a flag back and forth, n times - what ONE workgroup-to-workgroup hand-off costs when a
// K-split pair of a decode GEMM would add its halves in a fixed order.  flags[b] counts what b has published.
template <int P>
__global__ void k_bench_pair(unsigned* flags, int n, float* buf, unsigned* bar) {
    const int b = blockIdx.x, mate = b ^ P, t = threadIdx.x;
    float* mine = buf + (size_t)b * 512;
    const float* theirs = buf + (size_t)mate * 512;
    const bool first = (b & P) == 0;
    for (int e = 1; e <= n; ++e) {
        const bool give = first == ((e & 1) == 1);          // odd rounds: the lower workgroup gives, even rounds: the upper one
        if (give) {
            if (t < 512) st_agent(mine + t, (float)e);
            __syncthreads();
            if (t == 0) __hip_atomic_store(flags + b * 32, (unsigned)e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (t == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(flags + mate * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)e)
                    if (++spins > (1u << 22)) { bar[1] = 1u; break; }
            }
            __syncthreads();
            if (bar[1]) return;                              // (spin bound hit somewhere: everybody leaves)
            if (t < 512 && ld_agent(theirs + t) != (float)e) bar[2] = 1u;
        }
    }
}
__global__ void k_iota64(int64_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// row-major <-> fragment-tiled (common.h tile_off) converters for the column-GEMM hook
template <typename T>
__global__ void k_tile_rows(const T* __restrict__ src, int M, int K, int row_off, T* __restrict__ dst) {
    const int64_t n = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), k = (int)(i % K);
        dst[tile_off(row_off + m, k, K)] = src[i];
    }
}
template <typename T>
__global__ void k_untile_rows(const T* __restrict__ src, int M, int K, int row_off, T* __restrict__ dst) {
    const int64_t n = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), k = (int)(i % K);
        dst[i] = src[tile_off(row_off + m, k, K)];
    }
}
}  // namespace

extern "C" {

int rt_debug_gemm(rt_ctx* ctx, const void* d_a, int32_t a_is_f32, int64_t M, int32_t cin, int32_t taps, int32_t tap_stride,
                  int32_t tap_offset, int32_t rows_out, int32_t rows_in, const void* d_w_bf16, int32_t N, const float* d_bias,
                  int32_t act, float* d_out, int32_t mode, int32_t split_k) {
    if (!ctx || !d_a || !d_w_bf16 || !d_out || M < 1 || N < 1 || cin < 8 || taps < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm: bad argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int K = cin * taps;
    bf16_t* packed = nullptr;
    float* slabs = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&packed, packed_bytes(N, K)));
    PackedW pw;
    int rc = launch_pack_weight(ctx, (const bf16_t*)d_w_bf16, N, K, packed, &pw);
    if (!rc) {
        if (mode == 2) {             // the prompt-prefill kernel (k_gemm_mid): plain bf16 A, 65..1024 rows, K a multiple of 64
            if (a_is_f32 || taps != 1 || !gemm_mid_ok((int)M, pw)) rc = rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm: mid mode needs a plain bf16 A, 65..1024 rows, K %% 64 == 0");
            else {
                if (hipMalloc((void**)&slabs, (size_t)M * N * 4) != hipSuccess) rc = RT_ERR_OOM;
                if (!rc) rc = launch_gemm_mid(ctx, (const bf16_t*)d_a, (int)M, pw, slabs, N);
                if (!rc) rc = launch_reduce_slabs(ctx, slabs, 1, M, N, d_bias, act, d_out, nullptr);
            }
        } else if (mode == 1) {
            if (a_is_f32 || taps != 1) rc = rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm: skinny mode needs a plain bf16 A");
            else {
                if (split_k < 1) split_k = skinny_pick_split((int)M, N, K, ctx->n_cu);
                if (hipMalloc((void**)&slabs, (size_t)split_k * M * N * 4) != hipSuccess) rc = RT_ERR_OOM;
                if (!rc) rc = launch_gemm_skinny(ctx, (const bf16_t*)d_a, (int)M, pw, slabs, N, split_k);
                if (!rc) rc = launch_reduce_slabs(ctx, slabs, split_k, M, N, d_bias, act, d_out, nullptr);
            }
        } else {
            GemmA a; a.ptr = d_a; a.is_f32 = a_is_f32 == 1 || a_is_f32 == 2; a.split = a_is_f32 >= 2; a.M = M; a.Cin = cin; a.taps = taps; a.tap_stride = tap_stride; a.tap_offset = tap_offset;
            a.rows_out = rows_out; a.rows_in = rows_in;
            if (a_is_f32 == 3) {     // operand planes: the bf16 hi plane, then the bf16 lo plane, each [input rows][cin]
                const int64_t in_rows = rows_out > 0 ? (M / rows_out) * rows_in : M;
                a.ptr_lo = (const bf16_t*)d_a + in_rows * cin;
            }
            GemmEpi e; e.ldc = N;
            if (split_k <= 1) {
                e.bias = d_bias; e.act = act; e.out_f32 = d_out; e.split_k = 1;
                rc = launch_gemm(ctx, a, pw, e);
            } else {
                if (hipMalloc((void**)&slabs, (size_t)split_k * M * N * 4) != hipSuccess) rc = RT_ERR_OOM;
                e.out_f32 = slabs; e.split_k = split_k;
                if (!rc) rc = launch_gemm(ctx, a, pw, e);
                if (!rc) rc = launch_reduce_slabs(ctx, slabs, split_k, M, N, d_bias, act, d_out, nullptr);
            }
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (packed) (void)hipFree(packed);
    if (slabs) (void)hipFree(slabs);
    return rc;
}

int rt_debug_attention(rt_ctx* ctx, const float* d_q, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim, const int32_t* d_row_slot,
                       const int32_t* d_row_pos, int32_t window, const void* d_k, const void* d_v, int32_t slots, int32_t max_pos,
                       void* d_out_bf16) {
    if (!ctx || !d_q || !d_k || !d_v || !d_out_bf16) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_attention: null argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    KvCache kv;
    kv.k = (bf16_t*)d_k; kv.v = (bf16_t*)d_v; kv.layers = 1; kv.slots = slots; kv.kv_heads = kv_heads; kv.max_pos = max_pos; kv.head_dim = head_dim;
    int rc = launch_attention(ctx, d_q, M, heads, kv_heads, head_dim, d_row_slot, d_row_pos, 0, window, kv, 0, (bf16_t*)d_out_bf16);
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

// The column-owner decode GEMM (k_gemm_col) on its own, launched exactly as model_stack.hip's col_gemm launches it (row blocks of
// 64 / 32, the production sub-tile split unless one is forced).  Row-major operands at the ABI; the hook tiles / un-tiles.
int rt_debug_gemm_col(rt_ctx* ctx, const void* d_a_bf16, int32_t M, int32_t K, const void* d_w_bf16, int32_t N, int32_t epi, int32_t split,
                      int32_t row_off, int32_t nt, const float* d_rowsq, int32_t rowsq_n, float eps, const float* d_bias, const float* d_scale,
                      float* d_x, const float* d_next_norm_w, void* d_next_bf16, float* d_rowsq_out, void* d_act_bf16) {
    if (!ctx || !d_a_bf16 || !d_w_bf16 || M < 1 || M > 64 || K < 32 || K % 32 || N < 1 || row_off < 0 || row_off % 16)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm_col: bad argument (1 <= M <= 64, K %% 32 == 0, row_off %% 16 == 0)");
    if (epi < COL_STORE || epi > COL_SILU) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm_col: epi %d", epi);
    if ((epi != COL_SILU && !d_x) || (epi == COL_SILU && (!d_act_bf16 || N % 32)) || (epi == COL_RESID && !d_rowsq_out))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_gemm_col: missing output for epilogue %d", epi);
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int No = epi == COL_SILU ? N / 2 : N;                  // output width
    if (split <= 0) split = epi == COL_SILU ? col_split_silu(N, ctx->n_cu) : col_split_for(N, ctx->n_cu);
    const int rows = (row_off + M + 31) / 32 * 32;               // tiled buffers hold whole 32-row blocks (alloc_dec_ws)
    bf16_t *w16 = nullptr, *a_t = nullptr, *next_t = nullptr, *act_t = nullptr;
    float *x_t = nullptr, *rowsq_all = nullptr, *rowsq_out_all = nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : {(void*)w16, (void*)a_t, (void*)next_t, (void*)act_t, (void*)x_t, (void*)rowsq_all, (void*)rowsq_out_all}) if (p) (void)hipFree(p);
    };
#define DBG_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return rt_fail(ctx, rt_hip_status(_e), "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    DBG_HIP(hipMalloc((void**)&w16, packed16_bytes(N, K)));
    DBG_HIP(hipMalloc((void**)&a_t, (size_t)rows * K * 2));
    DBG_HIP(hipMemsetAsync(a_t, 0, (size_t)rows * K * 2, ctx->stream));
    PackedW pw;
    pw.N = N; pw.K = K;
    rc = launch_pack_weight16(ctx, (const bf16_t*)d_w_bf16, N, K, w16, &pw);
    const int nb = 256;
    hipLaunchKernelGGL((k_tile_rows<bf16_t>), dim3(nb), dim3(256), 0, ctx->stream, (const bf16_t*)d_a_bf16, M, K, row_off, a_t);
    const int n_part = (N + 15) / 16 * split;                    // RESID: rowsq partials per row
    ColArgs c;
    c.A = a_t; c.M = M; c.K = K; c.epi = epi; c.split = split; c.row_off = row_off; c.nt = nt ? 1 : 0; c.eps = eps;
    c.bias = d_bias; c.scale = d_scale; c.ldc = No;
    if (d_rowsq) {                                               // rowsq rows are addressed by absolute row (row_off + m)
        DBG_HIP(hipMalloc((void**)&rowsq_all, (size_t)rows * rowsq_n * 4));
        DBG_HIP(hipMemsetAsync(rowsq_all, 0, (size_t)rows * rowsq_n * 4, ctx->stream));
        DBG_HIP(hipMemcpyAsync(rowsq_all + (size_t)row_off * rowsq_n, d_rowsq, (size_t)M * rowsq_n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        c.post_scale = 1; c.rowsq = rowsq_all; c.rowsq_n = rowsq_n;
    }
    if (epi == COL_STORE) {
        c.out = d_x - (size_t)row_off * No;                       // as col_head does: output rows are indexed from row_off
    } else if (epi == COL_RESID) {
        DBG_HIP(hipMalloc((void**)&x_t, (size_t)rows * No * 4));
        DBG_HIP(hipMemsetAsync(x_t, 0, (size_t)rows * No * 4, ctx->stream));
        hipLaunchKernelGGL((k_tile_rows<float>), dim3(nb), dim3(256), 0, ctx->stream, (const float*)d_x, M, No, row_off, x_t);
        DBG_HIP(hipMalloc((void**)&rowsq_out_all, (size_t)rows * n_part * 4));
        DBG_HIP(hipMemsetAsync(rowsq_out_all, 0, (size_t)rows * n_part * 4, ctx->stream));
        c.out = x_t; c.rowsq_out = rowsq_out_all; c.rowsq_out_n = n_part;
        if (d_next_bf16) {
            DBG_HIP(hipMalloc((void**)&next_t, (size_t)rows * No * 2));
            DBG_HIP(hipMemsetAsync(next_t, 0, (size_t)rows * No * 2, ctx->stream));
            c.next_bf16 = next_t; c.next_norm_w = d_next_norm_w;
        }
    } else {
        DBG_HIP(hipMalloc((void**)&act_t, (size_t)rows * No * 2));
        DBG_HIP(hipMemsetAsync(act_t, 0, (size_t)rows * No * 2, ctx->stream));
        c.out_bf16 = act_t;
    }
    const int blk = g_col_rows64 ? 64 : 32;
    for (int r0 = 0; r0 < M && !rc; r0 += blk) {                 // model_stack.hip col_gemm
        ColArgs a = c;
        a.M = std::min(blk, M - r0);
        a.row_off = row_off + r0;
        rc = launch_gemm_col(ctx, a, pw);
    }
    if (!rc && epi == COL_RESID) {
        hipLaunchKernelGGL((k_untile_rows<float>), dim3(nb), dim3(256), 0, ctx->stream, (const float*)x_t, M, No, row_off, d_x);
        if (next_t) hipLaunchKernelGGL((k_untile_rows<bf16_t>), dim3(nb), dim3(256), 0, ctx->stream, (const bf16_t*)next_t, M, No, row_off, (bf16_t*)d_next_bf16);
        DBG_HIP(hipMemcpyAsync(d_rowsq_out, rowsq_out_all + (size_t)row_off * n_part, (size_t)M * n_part * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (!rc && epi == COL_SILU)
        hipLaunchKernelGGL((k_untile_rows<bf16_t>), dim3(nb), dim3(256), 0, ctx->stream, (const bf16_t*)act_t, M, No, row_off, (bf16_t*)d_act_bf16);
    if (!rc) DBG_HIP(hipGetLastError());
#undef DBG_HIP
    cleanup();
    return rc;
}

// The decode step's fused attention (q/k-RMSNorm + RoPE + KV append + attention, shared voice-prefix slot) on its own.
int rt_debug_attention_fused(rt_ctx* ctx, const float* d_qkv, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim, const float* d_q_norm_w,
                             const float* d_k_norm_w, float eps, const float* d_cos, const float* d_sin, const int32_t* d_row_slot,
                             const int32_t* d_row_pos, int32_t pos_add, void* d_k, void* d_v, int32_t slots, int32_t max_pos,
                             int32_t prefix_slot, int32_t prefix_len, void* d_out_bf16) {
    // (d_row_slot == NULL: row i uses slot i; d_row_pos == NULL: every row at pos_add - the array-free form the predictor's passes take)
    if (!ctx || !d_qkv || !d_cos || !d_sin || !d_k || !d_v || !d_out_bf16 || M < 1)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_attention_fused: null argument");
    if (prefix_slot >= slots || prefix_len < 0 || prefix_len > max_pos) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_attention_fused: bad prefix");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    KvCache kv;
    kv.k = (bf16_t*)d_k; kv.v = (bf16_t*)d_v; kv.layers = 1; kv.slots = slots; kv.kv_heads = kv_heads; kv.max_pos = max_pos; kv.head_dim = head_dim;
    kv.prefix_slot = prefix_slot; kv.prefix_len = prefix_slot >= 0 ? prefix_len : 0;
    bf16_t* vt = nullptr;
    if (prefix_slot >= 0 && prefix_len > 0 && head_dim == 128) {      // the matrix-core path reads the prefix through fragment-tiled copies
        kv.vt_stride = (prefix_len + 31) / 32 * 4096;
        kv.prefix_slot_alloc = prefix_slot;
        RT_HIP(ctx, hipMalloc((void**)&vt, (size_t)2 * kv_heads * kv.vt_stride * 2));
        kv.kt_prefix = vt;
        kv.vt_prefix = vt + (size_t)kv_heads * kv.vt_stride;
        const int rt = launch_transpose_prefix_v(ctx, kv, prefix_len);
        if (rt) { (void)hipFree(vt); return rt; }
    }
    const int rc = launch_attention_fused(ctx, d_qkv, M, heads, kv_heads, head_dim, d_q_norm_w, d_k_norm_w, eps, d_cos, d_sin, d_row_slot, d_row_pos,
                                          pos_add, 0, kv, 0, (bf16_t*)d_out_bf16, nullptr, 0, 0);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    if (vt) (void)hipFree(vt);
    RT_HIP(ctx, se);
    return rc;
}

// Prompt-prefill attention behind a shared prefix slot (q prepared, K / V rows already in the caches): mode 0 = the vector-unit
// kernel (k_attention), 1 = the matrix-core form the model uses for prompt rows (attention_mfma.hip, head_dim 128, 2 query heads per
// kv head, prefix of >= 64 rows)
int rt_debug_attention_prefill(rt_ctx* ctx, const float* d_q, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim, const int32_t* d_row_slot,
                               const int32_t* d_row_pos, const void* d_k, const void* d_v, int32_t slots, int32_t max_pos, int32_t prefix_slot,
                               int32_t prefix_len, int32_t mode, void* d_out_bf16) {
    if (!ctx || !d_q || !d_row_slot || !d_row_pos || !d_k || !d_v || !d_out_bf16 || M < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_attention_prefill: null argument");
    if (prefix_slot < 0 || prefix_slot >= slots || prefix_len < 1 || prefix_len > max_pos) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_attention_prefill: bad prefix");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    KvCache kv;
    kv.k = (bf16_t*)d_k; kv.v = (bf16_t*)d_v; kv.layers = 1; kv.slots = slots; kv.kv_heads = kv_heads; kv.max_pos = max_pos; kv.head_dim = head_dim;
    kv.prefix_slot = prefix_slot; kv.prefix_len = prefix_len;
    bf16_t* vt = nullptr;
    const int saved = g_prefill_attn_mfma;
    g_prefill_attn_mfma = mode ? 1 : 0;
    if (mode) {
        if (head_dim != 128) { g_prefill_attn_mfma = saved; return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_debug_attention_prefill: the matrix-core form needs head_dim 128"); }
        kv.vt_stride = (prefix_len + 31) / 32 * 4096;
        kv.prefix_slot_alloc = prefix_slot;
        RT_HIP(ctx, hipMalloc((void**)&vt, (size_t)2 * kv_heads * kv.vt_stride * 2));
        kv.kt_prefix = vt;
        kv.vt_prefix = vt + (size_t)kv_heads * kv.vt_stride;
        const int rt = launch_transpose_prefix_v(ctx, kv, prefix_len);
        if (rt) { (void)hipFree(vt); g_prefill_attn_mfma = saved; return rt; }
        if (!attention_prefill_mfma_ok(heads, kv_heads, head_dim, 0, kv)) {
            (void)hipFree(vt); g_prefill_attn_mfma = saved;
            return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_debug_attention_prefill: shape not served by the matrix-core form");
        }
    }
    int rc;
    if (mode == 2) {        // the prefix slot's own prefill: rows = positions 0 .. M - 1 of prefix_slot, causal (tiles made by the call itself)
        kv.prefix_slot = -1; kv.prefix_len = 0;
        kv.vt_stride = std::max(kv.vt_stride, (M + 31) / 32 * 4096);
        if (vt) { (void)hipFree(vt); vt = nullptr; }
        hipError_t me = hipMalloc((void**)&vt, (size_t)2 * kv_heads * kv.vt_stride * 2);
        if (me != hipSuccess) { g_prefill_attn_mfma = saved; return rt_fail(ctx, rt_hip_status(me), "rt_debug_attention_prefill: out of memory"); }
        kv.kt_prefix = vt;
        kv.vt_prefix = vt + (size_t)kv_heads * kv.vt_stride;
        rc = attention_block_prefix_ok(M, heads, kv_heads, head_dim, 0, kv)
                 ? launch_attention_block_prefix(ctx, d_q, M, heads, kv_heads, d_row_slot, d_row_pos, kv, 0, (bf16_t*)d_out_bf16)
                 : rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_debug_attention_prefill: shape not served by the block-prefix form");
    } else {
        rc = launch_attention(ctx, d_q, M, heads, kv_heads, head_dim, d_row_slot, d_row_pos, 0, 0, kv, 0, (bf16_t*)d_out_bf16);
    }
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    g_prefill_attn_mfma = saved;
    if (vt) (void)hipFree(vt);
    RT_HIP(ctx, se);
    return rc;
}

int rt_debug_sample(rt_ctx* ctx, const float* d_logits, int32_t M, int32_t V, const rt_sampling* sp, uint64_t seed, int32_t frame,
                    int32_t group, int32_t suppress_from, int32_t allow_token, uint8_t* d_seen, int32_t* d_out) {
    if (!ctx || !d_logits || !sp || !d_out || M < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_debug_sample: null argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int64_t* items = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&items, (size_t)M * 8));
    hipLaunchKernelGGL(k_iota64, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, items, M);
    SampleArgs a{};
    a.logits = d_logits; a.n_slabs = 1; a.M = M; a.V = V;
    a.do_sample = sp->do_sample; a.temperature = sp->temperature; a.top_k = sp->top_k; a.top_p = sp->top_p; a.rep_penalty = sp->repetition_penalty;
    a.seen = d_seen; a.suppress_from = suppress_from; a.allow_token = allow_token; a.seed = seed; a.item_ids = items; a.frame = frame; a.group = group;
    a.forced = nullptr; a.out = d_out; a.out_stride = 1; a.eos_token = -1; a.eos_flag = nullptr; a.logits_copy = nullptr;
    a.frame_ptr = nullptr; a.seed_ptr = nullptr; a.stamps = nullptr;
    int rc = launch_sample(ctx, a);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(items);
    return rc;
}

int rt_debug_tune(int32_t skinny_variant, int32_t skinny_waves_per_cu) {
    // exclusive: waits until no call is executing on any context (CtxLock holds this lock shared), and a resumable generation in
    // flight keeps the plan it began with - the switch is refused rather than applied under it
    std::unique_lock<std::shared_mutex> all(g_tune_mu);
    if (g_runs_in_flight.load() > 0) return RT_ERR_STATE;
    if (skinny_variant >= 2800) { g_pair_attn = skinny_variant - 2800; return RT_OK; }              // 2800/2801: the predictor's two-position first pass with k_qkv_post + attention / on the fused attention
    if (skinny_variant >= 2700) { g_frame_inc_fold = skinny_variant - 2700; return RT_OK; }         // 2700/2701: frame += 1 as a launch of its own / by the last workgroup of the talker-input launch
    if (skinny_variant >= 2600) { g_conv_unroll = skinny_variant - 2600; return RT_OK; }            // 2600/2601: k = 7 convs on the generic / the tap-unrolled k_conv_win
    if (skinny_variant >= 2400) { g_col_silu_x = skinny_variant - 2400; return RT_OK; }            // 2400/2401: gate/up decode GEMM as pairs in 1.5 rounds / as one round of 1.5-pair workgroups
    if (skinny_variant >= 2300) { g_col_rows16 = skinny_variant - 2300; return RT_OK; }           // 2300/2301: <= 16-row decode GEMMs on the 32-row / the 2-workgroups-per-CU 16-row instantiation
    if (skinny_variant >= 2200) { g_prefill_attn_mfma = skinny_variant - 2200; return RT_OK; }    // 2200/2201: prompt attention behind a shared prefix on the vector unit / matrix cores
    if (skinny_variant >= 2100) { g_fuse_conv = skinny_variant - 2100; return RT_OK; }            // 2100/2101: the 96-channel conv pairs of the codec decoder as two launches / one
    if (skinny_variant >= 2000) { g_col_max_rows = std::min(64, std::max(1, skinny_variant - 2000)); return RT_OK; }   // 20nn: batches up to nn rows take the column decode path
    if (skinny_variant >= 1900) { g_prefill_mid = skinny_variant - 1900; return RT_OK; }         // 1900/1901: prompt-prefill GEMMs on the split-K tiled kernel / on k_gemm_mid
    if (skinny_variant >= 1800) { g_conv_tall = skinny_variant - 1800; return RT_OK; }           // 1800/1801: 128- / 256-row tiles for the narrow-channel k>1 convs
    if (skinny_variant >= 1700) { g_handover_every = std::max(1, skinny_variant - 1700); return RT_OK; }  // 17nn: queued items take over finished rows every nn frames
    if (skinny_variant >= 1600) { g_col_split4 = skinny_variant - 1600; return RT_OK; }           // 1600/1601: quarter-tile split of N <= 1024 decode GEMMs off/on
    if (skinny_variant >= 1500) { g_attn_mfma = skinny_variant - 1500; return RT_OK; }            // 1500/1501/1502: shared-prefix decode attention on the vector unit / matrix cores, 4 rows per workgroup / matrix cores, 1 row
    if (skinny_variant >= 1400) { g_eos_check_every = std::max(1, skinny_variant - 1400); return RT_OK; }   // 14nn: look at the end-of-sequence flags every nn frames
    if (skinny_variant >= 1300) { g_sync_parts = skinny_variant - 1300; return RT_OK; }             // 1300/1301: stream sync after every frame part off/on
    if (skinny_variant >= 1200) { g_conv_win = skinny_variant - 1200; return RT_OK; }               // 1200/1201: conv input window in LDS off/on
    if (skinny_variant >= 1100) { g_final_conv = skinny_variant - 1100; return RT_OK; }             // 1100/1101: last conv as GEMM / own kernel
    if (skinny_variant >= 1000) { g_xcd_order = skinny_variant - 1000; return RT_OK; }              // 1000/1001: XCD-aware tile order off/on
    if (skinny_variant >= 900) { g_prefill_fill = skinny_variant - 900; return RT_OK; }            // 90n: prefill split-K target, n workgroups per CU
    if (skinny_variant >= 800) { g_fuse_sample_embed = skinny_variant - 800; return RT_OK; }      // 800/801: separate / fused sampler + embedding
    if (skinny_variant >= 700) { g_col_rows64 = skinny_variant - 700; return RT_OK; }             // 700/701: 32-row / 64-row decode GEMM launches
    if (skinny_variant >= 600) { g_tile96 = skinny_variant - 600; return RT_OK; }                 // 600/601: 128x96 tiles off/on
    if (skinny_variant >= 500) { g_col_split = skinny_variant - 500; return RT_OK; }             // 500: automatic sub-tile split, 501/502/504: forced
    if (skinny_variant >= 400) { g_decode_lanes = skinny_variant - 400; return RT_OK; }          // 40n: n decode lanes
    if (skinny_variant >= 300) { g_pred_nt = skinny_variant - 300; return RT_OK; }               // 300: predictor weights cacheable, 301: nt
    if (skinny_variant >= 200) { g_use_graph = skinny_variant - 200; return RT_OK; }            // 200: eager frames, 201: graph replay
    if (skinny_variant >= 100) { g_decode_col = skinny_variant - 100; return RT_OK; }   // 100: legacy 9-launch decode, 101: column path
    if (skinny_variant >= 0) g_skinny_variant = skinny_variant;
    if (skinny_waves_per_cu > 0) g_skinny_waves_per_cu = skinny_waves_per_cu;
    return RT_OK;
}

int rt_bench_gemm_skinny(rt_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t split_k, int32_t n_mats, int32_t iters, double* avg_us,
                         int32_t* used_split) {
    if (!ctx || !avg_us || M < 1 || M > 64 || N < 32 || K < 16 || K % 16 || n_mats < 1 || iters < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_gemm_skinny: bad argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pb = packed_bytes(N, K);
    bf16_t* wbuf = nullptr;
    bf16_t* a = nullptr;
    float* out = nullptr;
    if (split_k < 1) split_k = skinny_pick_split(M, N, K, ctx->n_cu);
    if (used_split) *used_split = split_k;
    RT_HIP(ctx, hipMalloc((void**)&wbuf, pb * n_mats));
    RT_HIP(ctx, hipMalloc((void**)&a, (size_t)M * K * 2));
    RT_HIP(ctx, hipMalloc((void**)&out, (size_t)split_k * M * N * 4));
    RT_HIP(ctx, hipMemsetAsync(wbuf, 0x3c, pb * n_mats, ctx->stream));     // bf16 0x3c3c = 0.0115: non-zero operands
    RT_HIP(ctx, hipMemsetAsync(a, 0x3c, (size_t)M * K * 2, ctx->stream));
    PackedW pw;
    pw.N = N; pw.K = K; pw.Np = (N + 31) / 32 * 32; pw.Kp = K;
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    int rc = RT_OK;
    for (int i = 0; i < n_mats && !rc; ++i) { pw.data = wbuf + (pb / 2) * i; rc = launch_gemm_skinny(ctx, a, M, pw, out, N, split_k); }
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters && !rc; ++i) { pw.data = wbuf + (pb / 2) * (i % n_mats); rc = launch_gemm_skinny(ctx, a, M, pw, out, N, split_k); }
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(wbuf); (void)hipFree(a); (void)hipFree(out);
    return rc;
}

// Back-to-back launches of the column-owner GEMM over n_mats weight matrices (> 512 MB in total => HBM-cold).
int rt_bench_gemm_col(rt_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_norm, int32_t epi, int32_t n_mats, int32_t iters,
                      double* avg_us, int64_t* stamps8) {
    if (!ctx || !avg_us || M < 1 || M > 64 || N < 64 || K < 16 || K % 16 || n_mats < 1 || iters < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_gemm_col: bad argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pb = packed_bytes(N, K);
    bf16_t* wbuf = nullptr;
    void* a = nullptr;
    float *out = nullptr, *rowsq = nullptr, *rowsq_out = nullptr, *normw = nullptr;
    bf16_t* act = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&wbuf, pb * n_mats));
    RT_HIP(ctx, hipMalloc(&a, (size_t)64 * K * 4));
    RT_HIP(ctx, hipMalloc((void**)&out, (size_t)64 * N * 4));
    RT_HIP(ctx, hipMalloc((void**)&act, (size_t)64 * N * 2));
    RT_HIP(ctx, hipMalloc((void**)&rowsq, (size_t)64 * 512 * 4));
    RT_HIP(ctx, hipMalloc((void**)&rowsq_out, (size_t)64 * (N / 4 + 4) * 4));
    RT_HIP(ctx, hipMalloc((void**)&normw, (size_t)std::max(K, N) * 4));
    RT_HIP(ctx, hipMemsetAsync(wbuf, 0x3c, pb * n_mats, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(a, 0x3c, (size_t)64 * K * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(out, 0, (size_t)64 * N * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(rowsq, 0x3c, (size_t)64 * 512 * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(normw, 0x3c, (size_t)std::max(K, N) * 4, ctx->stream));
    PackedW pw;
    pw.N = N; pw.K = K; pw.Np = (N + 31) / 32 * 32; pw.Kp = K; pw.Np16 = (N + 15) / 16 * 16;
    ColArgs c;
    c.nt = (a_norm & 2) ? 0 : 1;             // bit 1: cacheable weight loads (hot-cache experiment with n_mats = 1)
    a_norm &= 1;
    c.A = a; c.post_scale = a_norm; c.rowsq = rowsq; c.rowsq_n = K / 16; c.eps = 1e-6f; c.M = M; c.K = K; c.epi = epi;
    c.next_bf16 = epi == COL_RESID ? act : nullptr; c.next_norm_w = normw;
    c.out = out; c.ldc = epi == COL_SILU ? N / 2 : N; c.split = epi == COL_SILU ? col_split_silu(N, ctx->n_cu) : col_split_for(N, ctx->n_cu);
    c.rowsq_out = rowsq_out; c.rowsq_out_n = N / 16 * c.split; c.out_bf16 = act;
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    int rc = RT_OK;
    for (int i = 0; i < n_mats && !rc; ++i) { pw.data16 = wbuf + (pb / 2) * i; rc = launch_gemm_col(ctx, c, pw); }
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters && !rc; ++i) {
        pw.data16 = wbuf + (pb / 2) * (i % n_mats);
        rc = launch_gemm_col(ctx, c, pw);
    }
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    if (stamps8 && !rc) {       // one more launch with the in-kernel phase stamps of workgroup 0
        long long* st = nullptr;
        RT_HIP(ctx, hipMalloc((void**)&st, 64));
        RT_HIP(ctx, hipMemsetAsync(st, 0, 64, ctx->stream));
        c.stamps = st;
        pw.data16 = wbuf + (pb / 2) * (iters % n_mats);
        rc = launch_gemm_col(ctx, c, pw);
        RT_HIP(ctx, hipMemcpy(stamps8, st, 64, hipMemcpyDeviceToHost));
        (void)hipFree(st);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(wbuf); (void)hipFree(a); (void)hipFree(out); (void)hipFree(act); (void)hipFree(rowsq); (void)hipFree(rowsq_out); (void)hipFree(normw);
    return rc;
}

// Launch-floor microbenchmark: n dependent launches of a trivial kernel (grid_wgs x 256 threads), eagerly or as one
// captured hipGraph replayed `reps` times.  Returns microseconds per launch.
int rt_bench_launch(rt_ctx* ctx, int32_t grid_wgs, int32_t n, int32_t use_graph, int32_t reps, double* us_per_launch) {
    if (!ctx || !us_per_launch || grid_wgs < 1 || n < 1 || reps < 1) return RT_ERR_INVALID;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    float* buf = nullptr;
    const int elems = grid_wgs * 256;
    RT_HIP(ctx, hipMalloc((void**)&buf, (size_t)elems * 4));
    RT_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)elems * 4, ctx->stream));
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (use_graph) {
        RT_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_touch, dim3(grid_wgs), dim3(256), 0, ctx->stream, buf, elems);
        RT_HIP(ctx, hipStreamEndCapture(ctx->stream, &graph));
        RT_HIP(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        RT_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_touch, dim3(grid_wgs), dim3(256), 0, ctx->stream, buf, elems);
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int rpt = 0; rpt < reps; ++rpt) {
        if (use_graph) RT_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
        else for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_touch, dim3(grid_wgs), dim3(256), 0, ctx->stream, buf, elems);
    }
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *us_per_launch = (double)ms * 1e3 / ((double)n * reps);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return RT_OK;
}

// Sampler microbenchmark: `iters` back-to-back launches over M rows of V logits; returns the average launch time and
// (row 0 of the last launch) the 100-MHz wall-clock stamps at the kernel's phase boundaries.
int rt_bench_sample(rt_ctx* ctx, const float* d_logits, int32_t M, int32_t V, const rt_sampling* sp, int32_t iters, double* avg_us,
                    int64_t* stamps8) {
    if (!ctx || !d_logits || !sp || !avg_us || M < 1 || iters < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_sample: bad argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int64_t* items = nullptr;
    int32_t* out = nullptr;
    long long* st = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&items, (size_t)M * 8));
    RT_HIP(ctx, hipMalloc((void**)&out, (size_t)M * 4));
    RT_HIP(ctx, hipMalloc((void**)&st, 8 * 8));
    RT_HIP(ctx, hipMemsetAsync(st, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_iota64, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, items, M);
    SampleArgs a{};
    a.logits = d_logits; a.n_slabs = 1; a.M = M; a.V = V;
    a.do_sample = sp->do_sample; a.temperature = sp->temperature; a.top_k = sp->top_k; a.top_p = sp->top_p; a.rep_penalty = sp->repetition_penalty;
    a.suppress_from = V; a.allow_token = -1; a.seed = 1; a.item_ids = items; a.out = out; a.out_stride = 1; a.eos_token = -1;
    a.stamps = st;
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    int rc = launch_sample(ctx, a);
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters && rc == RT_OK; ++i) { a.frame = i; rc = launch_sample(ctx, a); }
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    if (stamps8) RT_HIP(ctx, hipMemcpy(stamps8, st, 64, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(items); (void)hipFree(out); (void)hipFree(st);
    return rc;
}

// Decode-attention microbenchmark: `iters` back-to-back launches of the decode step's fused attention over M rows, each row at
// position prefix_len + own_len - 1 of its own slot, positions < prefix_len read from a shared prefix slot (shared != 0) or from
// the row's own slot (shared == 0: what the launch would cost without the shared voice prefix).  The launches cycle through
// `layers` cache regions (as the model's 28 layers do: a layer's K / V is L2-cold when its turn comes).  Operands are zeros
// (uniform softmax): the timing does not depend on the values.
int rt_bench_attention_fused(rt_ctx* ctx, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t prefix_len, int32_t own_len,
                             int32_t shared, int32_t layers, int32_t iters, double* avg_us) {
    if (!ctx || !avg_us || M < 1 || M > 64 || heads < 1 || kv_heads < 1 || heads % kv_heads || prefix_len < 0 || own_len < 1 || layers < 1 || iters < 1)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_attention_fused: bad argument");
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    // (more than 64 cache rows select the 16-wave form of the talker's decode step; the predictor's <= 16 positions the 4-wave form)
    const int max_pos = std::max((prefix_len + own_len + 8 + 63) / 64 * 64, (prefix_len > 0 || own_len > 16) ? 128 : 64), slots = M + 1, width = (heads + 2 * kv_heads) * head_dim;
    KvCache kv;
    kv.layers = layers; kv.slots = slots; kv.kv_heads = kv_heads; kv.max_pos = max_pos; kv.head_dim = head_dim;
    kv.prefix_slot = shared && prefix_len > 0 ? M : -1;
    kv.prefix_len = kv.prefix_slot >= 0 ? prefix_len : 0;
    const size_t cache_bytes = (size_t)layers * kv.layer_stride() * sizeof(bf16_t);
    float *qkv = nullptr, *cs = nullptr, *nw = nullptr;
    int32_t *pos = nullptr, *slot = nullptr;
    bf16_t* out = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&kv.k, cache_bytes));
    RT_HIP(ctx, hipMalloc((void**)&kv.v, cache_bytes));
    RT_HIP(ctx, hipMalloc((void**)&qkv, (size_t)M * width * 4));
    RT_HIP(ctx, hipMalloc((void**)&cs, (size_t)max_pos * head_dim * 4));
    RT_HIP(ctx, hipMalloc((void**)&nw, (size_t)head_dim * 4));
    RT_HIP(ctx, hipMalloc((void**)&pos, (size_t)M * 4));
    RT_HIP(ctx, hipMalloc((void**)&slot, (size_t)M * 4));
    RT_HIP(ctx, hipMalloc((void**)&out, ((size_t)M + 31) / 32 * 32 * heads * head_dim * 2));
    RT_HIP(ctx, hipMemsetAsync(kv.k, 0, cache_bytes, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(kv.v, 0, cache_bytes, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(qkv, 0, (size_t)M * width * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(cs, 0, (size_t)max_pos * head_dim * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(nw, 0, (size_t)head_dim * 4, ctx->stream));
    std::vector<int32_t> hp(M, prefix_len + own_len - 1), hs(M);
    for (int i = 0; i < M; ++i) hs[i] = i;
    RT_HIP(ctx, hipMemcpyAsync(pos, hp.data(), (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(slot, hs.data(), (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    bf16_t* tiles = nullptr;
    if (kv.prefix_slot >= 0 && prefix_len >= 64 && head_dim == 128) {      // fragment-tiled prefix copies: what the matrix-core forms read
        kv.vt_stride = (prefix_len + 31) / 32 * 4096;
        kv.prefix_slot_alloc = kv.prefix_slot;
        RT_HIP(ctx, hipMalloc((void**)&tiles, (size_t)2 * layers * kv_heads * kv.vt_stride * 2));
        kv.kt_prefix = tiles;
        kv.vt_prefix = tiles + (size_t)layers * kv_heads * kv.vt_stride;
        const int rt = launch_transpose_prefix_v(ctx, kv, prefix_len);
        if (rt) return rt;
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    const float* cosT = cs;
    const float* sinT = cs + (size_t)max_pos * head_dim / 2;
    int rc = RT_OK;
    for (int i = 0; i < layers && rc == RT_OK; ++i)          // warm-up: one pass over every layer
        rc = launch_attention_fused(ctx, qkv, M, heads, kv_heads, head_dim, nw, nw, 1e-6f, cosT, sinT, slot, pos, 0, 0, kv, i, out, nullptr, 1, -1);
    RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters && rc == RT_OK; ++i)
        rc = launch_attention_fused(ctx, qkv, M, heads, kv_heads, head_dim, nw, nw, 1e-6f, cosT, sinT, slot, pos, 0, 0, kv, i % layers, out, nullptr, 1, -1);
    RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    for (void* p : {(void*)kv.k, (void*)kv.v, (void*)qkv, (void*)cs, (void*)nw, (void*)pos, (void*)slot, (void*)out, (void*)tiles}) if (p) (void)hipFree(p);
    return rc;
}

// Grid-barrier microbenchmark: one persistent launch of `wgs` x `threads`, n barriers; microseconds per barrier (and the
// abort flag: non-zero = the grid was not co-resident or a spin bound was hit).
int rt_bench_grid_barrier(rt_ctx* ctx, int32_t wgs, int32_t threads, int32_t n, int32_t mode, double* us_per_barrier, int32_t* aborted) {
    if (!ctx || !us_per_barrier || wgs < 1 || threads < 64 || threads > 1024 || n < 1) return RT_ERR_INVALID;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int per_cu = 0;
    auto kern = mode == 2 ? k_bench_barrier<2> : (mode == 1 ? k_bench_barrier<1> : k_bench_barrier<0>);
    if (mode >= 3 && (threads < 512 || wgs % 16)) return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_grid_barrier: the pair modes take >= 512 threads and a multiple of 16 workgroups");
    RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0));
    if ((int64_t)per_cu * ctx->n_cu < wgs) return rt_fail(ctx, RT_ERR_INVALID, "rt_bench_grid_barrier: %d workgroups cannot be co-resident (%d per CU x %d CUs)", wgs, per_cu, ctx->n_cu);
    unsigned* bar = nullptr;
    float* buf = nullptr;
    unsigned* flags = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&bar, 64));
    RT_HIP(ctx, hipMalloc((void**)&buf, (size_t)wgs * threads * 4));
    RT_HIP(ctx, hipMalloc((void**)&flags, (size_t)wgs * 128));
    RT_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)wgs * threads * 4, ctx->stream));
    hipEvent_t e0, e1;
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        RT_HIP(ctx, hipMemsetAsync(bar, 0, 64, ctx->stream));
        RT_HIP(ctx, hipEventRecord(e0, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)wgs * threads * 4, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)wgs * 128, ctx->stream));
        if (mode == 3) hipLaunchKernelGGL(k_bench_pair<1>, dim3(wgs), dim3(threads), 0, ctx->stream, flags, n, buf, bar);
        else if (mode == 4) hipLaunchKernelGGL(k_bench_pair<8>, dim3(wgs), dim3(threads), 0, ctx->stream, flags, n, buf, bar);
        else hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, ctx->stream, bar, n, buf);
        RT_HIP(ctx, hipEventRecord(e1, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned h[3] = {0, 0, 0};
    RT_HIP(ctx, hipMemcpy(h, bar, 12, hipMemcpyDeviceToHost));
    if (aborted) *aborted = (int32_t)(h[1] | (h[2] << 1));     // bit 0: spin bound hit, bit 1: a stale value was read
    *us_per_barrier = (double)ms * 1e3 / n;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(bar); (void)hipFree(buf); (void)hipFree(flags);
    return RT_OK;
}

}  // extern "C"
