/*
 * rho_tts_amd_debug.h - measurement and test entry points of librho_tts_amd.so.
 *
 * NOT part of the drop-in boundary: nothing here stands behind an interface of the reference (rho-tts has no profiling, tuning or
 * kernel-test surface: SURVEY.md section 5), and a host binding of the generation path (rho_tts_amd.h) never
 * calls it.  Used by bench.py (rt_profile_*: the roofline figure), tests/ (rt_debug_*: single kernels against the CPU oracle)
 * and tools/ (rt_bench_*: microbenchmarks; rt_debug_tune: A/B switches).
 *
 * Threading: rt_debug_tune changes PROCESS-WIDE launch-plan switches.  It takes every context's work to a stop first - it waits
 * until no library call is executing on any context (calls hold a shared lock, the switch an exclusive one) and refuses
 * (RT_ERR_STATE) while a generation is in flight between rt_generate_begin and rt_generate_end on any model - so a thread that
 * tunes can never change the launch plan of a call or of a resumable generation that another thread has under way.
 */
#ifndef RHO_TTS_AMD_DEBUG_H
#define RHO_TTS_AMD_DEBUG_H

#include "rho_tts_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing of the decode step for bench.py's roofline figure: when enabled, every weight-streaming GEMM
 * launch is bracketed by HIP events on the context's stream. */
RT_API int rt_profile_enable(rt_model* m, int32_t on);
RT_API int rt_profile_read(rt_model* m, int64_t* n_launches, double* total_ms, double* total_bytes);
/* The same sums over the recorded launches of ONE class of weight stream: 0 = talker layers, codec head, mtp projection (cross
 * HBM once per frame); 1 = the predictor's first pass over its layers + its heads (each byte's first use in the frame);
 * 2 = predictor passes 2.. over the same layers (re-streamed from the Infinity Cache).  SURVEY.md 8d counts classes 0 + 1. */
RT_API int rt_profile_read_class(rt_model* m, int32_t cls, int64_t* n_launches, double* total_ms, double* total_bytes);

/* ------------------------------------------------------------------ kernel-level test hooks
 * Exercise single kernels against a float32 reference (tests/test_kernels_gpu.py).  All pointers are HBM.
 * rt_debug_gemm: out[M][N] (f32) = A . W^T for a row-major bf16 W[N][K=taps*cin]; A is the implicit-GEMM view
 * of a channels-last activation [batch][rows_in][cin] (bf16; f32 when a_is_f32 = 1; f32 fed as hi+lo bf16 planes
 * when a_is_f32 = 2; a_is_f32 = 3: d_a holds the bf16 hi plane followed by the bf16 lo plane, as a producing epilogue
 * writes them): output row (b, t) reads input
 * rows t + tap_offset + tap*tap_stride, zero outside [0, rows_in).  mode 0: LDS-tiled kernel (split_k slabs are
 * summed on return), mode 1: weight-streaming skinny kernel (plain A only, M <= 64), mode 2: the prompt-prefill kernel
 * (k_gemm_mid: plain bf16 A, 65..1024 rows, K a multiple of 64, final sums from 64 x 64 tiles over the whole K). */
RT_API int rt_debug_gemm(rt_ctx* ctx, const void* d_a, int32_t a_is_f32, int64_t M, int32_t cin, int32_t taps, int32_t tap_stride,
                         int32_t tap_offset, int32_t rows_out, int32_t rows_in, const void* d_w_bf16, int32_t N,
                         const float* d_bias, int32_t act, float* d_out, int32_t mode, int32_t split_k);
/* q [M][heads][d] f32, caches [slots][kv_heads][max_pos][d] bf16 -> out [M][heads*d] bf16 */
RT_API int rt_debug_attention(rt_ctx* ctx, const float* d_q, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim,
                              const int32_t* d_row_slot, const int32_t* d_row_pos, int32_t window, const void* d_k, const void* d_v,
                              int32_t slots, int32_t max_pos, void* d_out_bf16);
/* The dominant decode kernel on its own (gemm_col.hip k_gemm_col; launched as model_stack.hip launches it: row blocks of <= 64,
 * production sub-tile split when split = 0).  Row-major operands; the hook converts to / from the fragment-tiled layout.
 *   A [M][K] bf16, W [N][K] bf16 (epi 2: gate rows [0, N/2) then up rows), M <= 64, K % 32 == 0, row_off % 16 == 0
 *   d_rowsq [M][rowsq_n] or NULL: partial sums of squares of the pre-norm row; acc rows are scaled by rsqrt(sum/K + eps)
 *   epi 0 STORE: d_x [M][N] f32 out = scale_row * (A W^T) + bias
 *   epi 1 RESID: d_x [M][N] f32 in/out: x += scale .* (scale_row * (A W^T) + bias); d_next_bf16 [M][N] = bf16(next_norm_w .* x);
 *                d_rowsq_out [M][ceil(N/16)*split] partial sums of squares of the new x
 *   epi 2 SILU : d_act_bf16 [M][N/2] = bf16(silu(g) * u)
 * nt: 1 = non-temporal weight loads (talker), 0 = cacheable (predictor). */
RT_API int rt_debug_gemm_col(rt_ctx* ctx, const void* d_a_bf16, int32_t M, int32_t K, const void* d_w_bf16, int32_t N, int32_t epi, int32_t split,
                             int32_t row_off, int32_t nt, const float* d_rowsq, int32_t rowsq_n, float eps, const float* d_bias,
                             const float* d_scale, float* d_x, const float* d_next_norm_w, void* d_next_bf16, float* d_rowsq_out,
                             void* d_act_bf16);
/* The decode step's fused attention launch: qkv [M][(heads+2kv)*d] f32 (raw projections), q/k norm weights [d] or NULL,
 * cos/sin [max_pos][d/2]; row r is sequence slot row_slot[r] at position row_pos[r] + pos_add: its K/V row is appended to the
 * caches [slots][kv_heads][max_pos][d] bf16, then it attends to positions [0, pos]; positions < prefix_len are read from
 * prefix_slot (-1: no shared prefix).  out [M][heads*d] bf16 row-major.  d_row_slot NULL: row r is slot r; d_row_pos NULL: every
 * row at pos_add (the array-free form of the residual-code predictor's passes). */
RT_API int rt_debug_attention_fused(rt_ctx* ctx, const float* d_qkv, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim,
                                    const float* d_q_norm_w, const float* d_k_norm_w, float eps, const float* d_cos, const float* d_sin,
                                    const int32_t* d_row_slot, const int32_t* d_row_pos, int32_t pos_add, void* d_k, void* d_v,
                                    int32_t slots, int32_t max_pos, int32_t prefix_slot, int32_t prefix_len, void* d_out_bf16);
/* Prompt-prefill attention behind a shared prefix: q [M][heads][d] f32 (normed, roped), row r = sequence slot row_slot[r] at position
 * row_pos[r], whose K / V rows up to that position are already in the caches; positions < prefix_len are read from prefix_slot.
 * mode 0: the vector-unit kernel, 1: the matrix-core form the model uses for prompt rows (head_dim 128, 2 query heads per kv head,
 * prefix_len >= 64), 2: the prefix slot's OWN prefill - rows must be positions 0 .. M - 1 of prefix_slot (M >= 64), causal, the keys in
 * front of each 8-row block on the matrix cores.  out [M][heads*d] bf16. */
RT_API int rt_debug_attention_prefill(rt_ctx* ctx, const float* d_q, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim,
                                      const int32_t* d_row_slot, const int32_t* d_row_pos, const void* d_k, const void* d_v, int32_t slots,
                                      int32_t max_pos, int32_t prefix_slot, int32_t prefix_len, int32_t mode, void* d_out_bf16);
/* One draw per row: logits [M][V] f32 -> tokens [M].  item ids 0..M-1. */
RT_API int rt_debug_sample(rt_ctx* ctx, const float* d_logits, int32_t M, int32_t V, const rt_sampling* sp, uint64_t seed,
                           int32_t frame, int32_t group, int32_t suppress_from, int32_t allow_token, uint8_t* d_seen, int32_t* d_out);

/* A/B switches for measurements and tests (process-wide; the defaults are the fast path).  First argument:
 *   0..2 legacy skinny-GEMM variant (second argument: its waves per CU) | 100/101 legacy 9-launch / column-owner decode |
 *   200/201 eager / hipGraph frames | 300/301 predictor weights cacheable / non-temporal | 40n n decode lanes |
 *   500 automatic, 501/502/504 forced sub-tile split of narrow decode GEMMs | 600/601 128x96 codec tiles off/on |
 *   700/701 32-row / 64-row decode GEMM launches | 800/801 separate / fused sampler + next-input embedding |
 *   90n prefill split-K target of n workgroups per CU | 1000/1001 XCD-aware tile order of the tiled GEMM off/on |
 *   1300/1301 stream sync after every decode frame part off/on (bounds the dispatches in flight under rocprofv3 --pmc) |
 *   14nn end-of-sequence flags fetched every nn frames (default 8; 1401 = a copy + wait per frame) |
 *   1500/1501/1502 shared-prefix decode attention on the vector unit / on the matrix cores with four rows / one row per workgroup | 1600/1601 quarter-tile split off/on |
 *   17nn queued items (rt_generate with n_items > max_batch) take over finished rows every nn frames (default 4) |
 *   20nn batches of up to nn rows (default 64) decode on the column-owner path, larger ones on the legacy split-K path |
 *   1900/1901/1902/1903 prompt-prefill GEMMs on the split-K tiled kernel / on k_gemm_mid (automatic, 64 x 64, 128 x 128 tiles) |
 *   1800/1801/1802 narrow-channel (96 / 192) k>1 convs on 128-row tiles / 256-row tiles for long inputs / 256-row tiles always |
 *   2100/2101 the codec decoder's 96-channel residual units as two launches (k = 7 conv, 1x1 conv) / one fused launch |
 *   2200/2201 prompt-prefill attention behind a shared voice prefix on the vector unit / on the matrix cores |
 *   2300/2301 decode GEMMs of <= 16 rows on the 32-row / the two-workgroups-per-CU 16-row instantiation |
 *   2400/2401 gate/up decode GEMM whose tile pairs are 1.5x the CUs: one pair per workgroup (1.5 rounds) / 1.5 pairs per workgroup (one round) |
 *   2600/2601 the codec decoder's k = 7 convs on the generic / the tap-unrolled instantiation of k_conv_win |
 *   2700/2701 the decode frame counter advanced by a launch of its own / by the last workgroup of the frame's talker-input launch |
 *   2800/2801 the predictor's two-position first pass: q/k norm + RoPE + append as a launch of its own in front of the attention / inside the fused attention
 * The rt_bench_* entry points are the microbenchmarks behind tools/bench_*.py (for rt_bench_gemm_col choose
 * n_mats * N * K * 2 bytes > 512 MB to stream from HBM, not from cache). */
RT_API int rt_debug_tune(int32_t skinny_variant, int32_t skinny_waves_per_cu);
RT_API int rt_bench_gemm_col(rt_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t a_norm, int32_t epi, int32_t n_mats, int32_t iters,
                             double* avg_us, int64_t* stamps8);
RT_API int rt_bench_launch(rt_ctx* ctx, int32_t grid_wgs, int32_t n, int32_t use_graph, int32_t reps, double* us_per_launch);
RT_API int rt_bench_grid_barrier(rt_ctx* ctx, int32_t wgs, int32_t threads, int32_t n, int32_t mode, double* us_per_barrier, int32_t* aborted);
RT_API int rt_bench_sample(rt_ctx* ctx, const float* d_logits, int32_t M, int32_t V, const rt_sampling* sp, int32_t iters, double* avg_us,
                           int64_t* stamps8);
/* iters back-to-back launches of the decode step's fused attention: M rows at position prefix_len + own_len - 1, the first
 * prefix_len positions read from a shared prefix slot (shared = 1) or from each row's own slot (0), cycling through `layers`
 * cache regions; zeros as operands. */
RT_API int rt_bench_attention_fused(rt_ctx* ctx, int32_t M, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t prefix_len, int32_t own_len,
                                    int32_t shared, int32_t layers, int32_t iters, double* avg_us);
RT_API int rt_bench_gemm_skinny(rt_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t split_k, int32_t n_mats, int32_t iters,
                                double* avg_us, int32_t* used_split);

#ifdef __cplusplus
}
#endif
#endif /* RHO_TTS_AMD_DEBUG_H */
