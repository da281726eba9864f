// Launch-level interface of the gfx950 kernels (internal; the public C ABI is include/rho_tts_amd.h).
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------ GEMM
// Weights are bf16 matrices W[N][K] (torch Linear layout) re-tiled once at load time into
// MFMA B-fragment order for v_mfma_f32_32x32x16_bf16: tile (nt, kt) covers rows nt*32..+32 and
// k kt*16..+16 and is stored as 64 lanes x 16 B, lane (h<<5 | r) holding W[nt*32+r][kt*16+8h .. +8].
// A wave's B operand is then ONE fully coalesced 1-KiB load, and all K of a 32-row block is contiguous.
struct PackedW {
    const bf16_t* data = nullptr;  // [Np/32][Kp/16][64][8]
    int N = 0, K = 0;              // logical
    int Np = 0, Kp = 0;            // padded to 32 / 16 (zero filled)
    size_t bytes() const { return (size_t)Np * Kp * 2; }
    // second copy for the decode path, tiled for v_mfma_f32_16x16x32_bf16: tile (n/16, k/32) = 64 lanes x 16 B,
    // lane (q<<4 | c) holding W[16 nt + c][32 kt + 8 q .. +8]   (K % 32 == 0; N padded to 16)
    const bf16_t* data16 = nullptr;
    int Np16 = 0;
};

// Implicit-GEMM view of the A operand: row m = (b, t) of an activation tensor [B][T_in][Cin]
// (channels-last); column kk = tap*Cin + ci reads A[b][t + tap_offset + tap*tap_stride][ci],
// zero outside [0, T_in).  A plain matrix is taps = 1, rows_out = rows_in = M.
struct GemmA {
    const void* ptr = nullptr;
    const void* ptr_lo = nullptr;   // split with a bf16 source: the low plane (same layout as ptr), written by the producer's epilogue
    int is_f32 = 0;        // 1: float32 source converted to bf16 on load; 0: bf16
    int split = 0;         // 1: feed hi + lo bf16 planes, ~f32 activation precision at 2x MFMA work.  f32 source: split on load
                           //    (per consuming workgroup and tap); bf16 source: ptr / ptr_lo are the planes, split once by the producer
    int64_t M = 0;
    int Cin = 0;           // K = taps * Cin, Cin % 8 == 0
    int taps = 1;
    int tap_stride = 1;
    int tap_offset = 0;
    int rows_out = 0;      // output rows per batch item (0: not batched)
    int rows_in = 0;       // input rows per batch item
};

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_SNAKE = 3, ACT_CLAMP1 = 4, ACT_ELU = 5 };

// out = (act(acc + bias)) * scale + residual, optionally also out2 = snake2(out) in bf16.
// All per-column vectors have length N.  With split_k > 1 only raw f32 partial slabs are written
// (slab s at out_f32 + s*M*ldc) and the consumer sums them.
struct GemmEpi {
    const float* bias = nullptr;
    const float* scale = nullptr;
    const float* residual = nullptr;
    float* out_f32 = nullptr;
    bf16_t* out_bf16 = nullptr;
    int act = ACT_NONE;
    const float* snake_a = nullptr;      // exp(alpha)
    const float* snake_ib = nullptr;     // 1 / (exp(beta) + 1e-9)
    bf16_t* out2_bf16 = nullptr;
    float* out2_f32 = nullptr;
    bf16_t* out_hi = nullptr;            // out as hi + lo bf16 planes (hi = bf16(v), lo = bf16(v - hi)): the split operand of the next GEMM
    bf16_t* out_lo = nullptr;
    bf16_t* out2_hi = nullptr;           // the same for out2
    bf16_t* out2_lo = nullptr;
    const float* snake2_a = nullptr;
    const float* snake2_ib = nullptr;
    int act2 = ACT_SNAKE;                // what out2 applies to out: ACT_SNAKE (snake2_a / snake2_ib) or ACT_ELU
    int64_t ldc = 0;
    int split_k = 1;
};

size_t packed_bytes(int N, int K);
// d_dst must hold packed_bytes(N, K); d_src is row-major bf16 [N][K] in HBM.
int launch_pack_weight(rt_ctx* ctx, const bf16_t* d_src, int N, int K, bf16_t* d_dst, PackedW* out);
size_t packed16_bytes(int N, int K);
int launch_pack_weight16(rt_ctx* ctx, const bf16_t* d_src, int N, int K, bf16_t* d_dst, PackedW* out);
// w2 / e2 (optional): the 1x1 conv behind this conv's SnakeBeta, fused into the same launch (conv_pair_fusable must hold; e's
// out_hi / out_lo planes are then not written)
int launch_gemm(rt_ctx* ctx, const GemmA& a, const PackedW& w, const GemmEpi& e, const PackedW* w2 = nullptr, const GemmEpi* e2 = nullptr);
bool conv_pair_fusable(const GemmA& a, const PackedW& w, const GemmEpi& e, const PackedW& w2);
extern rt_knob g_fuse_conv;
// prompt-prefill GEMM (65..1024 rows): out[M][N] f32 = A[M][K] bf16 . W^T, whole K per 64 x 64 tile, final sums (no slabs)
extern rt_knob g_prefill_mid;
bool gemm_mid_ok(int M, const PackedW& w);
bool gemm_mid_shape_ok(const PackedW& w);     // the weight's shape alone (any row count can then be served in chunks of <= 1024 rows)
int launch_gemm_mid(rt_ctx* ctx, const bf16_t* A, int M, const PackedW& w, float* out, int64_t ldc);
// Weight-streaming form for M <= 64 rows (decode): plain bf16 A [M][K], raw f32 slabs out.
int launch_gemm_skinny(rt_ctx* ctx, const bf16_t* d_a, int M, const PackedW& w, float* d_out, int64_t ldc, int split_k,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int64_t slab_stride = 0);   // slab s at d_out + s * slab_stride (0: M * ldc)
int skinny_pick_split(int M, int N, int K, int n_cu);
extern rt_knob g_decode_col;
extern rt_knob g_use_graph;
extern rt_knob g_pred_nt;
extern rt_knob g_decode_lanes;
extern rt_knob g_sync_parts;
extern rt_knob g_eos_check_every;
extern rt_knob g_handover_every;
extern rt_knob g_conv_tall;
extern rt_knob g_col_max_rows;
extern rt_knob g_tile96;
extern rt_knob g_conv_win;
extern rt_knob g_conv_unroll;         // 1: tap-unrolled k = 7 conv kernels
extern rt_knob g_final_conv;           // 1: dedicated last-conv kernel, 0: one-column GEMM
extern rt_knob g_xcd_order;
extern rt_knob g_prefill_fill;         // prefill GEMMs split K until the grid holds this many workgroups per CU
extern rt_knob g_pair_attn;            // 1: the predictor's two-position first pass on the fused attention (no k_qkv_post launch per layer)
extern rt_knob g_frame_inc_fold;       // 1: the talker-input launch of a frame advances the frame counter (no k_frame_inc launch)
extern rt_knob g_fuse_sample_embed;    // 1: the predictor's sampler writes the next pass's input itself (no gather launch)
extern rt_knob g_col_rows64;          // 1: decode GEMM launches take up to 64 rows (4 sub-blocks), 0: 32-row launches only
extern rt_knob g_col_split;           // 0: automatic sub-tile split of narrow decode GEMMs, 1/2/4: forced
extern rt_knob g_col_split4;
extern rt_knob g_col_silu_x;         // 1: 1.5-pair gate/up workgroups when the pairs are 1.5x the CUs
extern rt_knob g_col_rows16;         // 1: <= 16-row decode GEMM launches on the two-workgroups-per-CU instantiation (decode lanes)          // 1: the automatic split may go to quarter tiles (N <= 1024 on 256 CUs)
int col_split_for(int N, int n_cu);
int col_split_silu(int N, int n_cu);
extern rt_knob g_skinny_variant;        // tuning knobs (rt_debug_tune)
extern rt_knob g_skinny_waves_per_cu;

// Column-owner decode GEMM (gemm_col.hip): whole-K per workgroup, fused RMSNorm prologue and residual / SwiGLU epilogues.
enum { COL_STORE = 0, COL_RESID = 1, COL_SILU = 2 };
struct ColArgs {
    // All activations of the column path are fragment-tiled (common.h tile_off); STORE outputs stay row-major.
    int row_off = 0;                // first row (in the tiled A / x / act buffers) of this 32-row block
    const void* A = nullptr;        // tiled bf16 [rows][K]; behind an RMSNorm it holds bf16(norm_w .* x)
    int nt = 1;                     // 1: stream the weights with non-temporal loads (read once per step); 0: let them stay in the
                                    //    Infinity Cache (the 220 MB predictor is re-read 15 times per frame)
    int post_scale = 0;             // 1: multiply the accumulator rows by rsqrt(sum(rowsq)/K + eps)  (the RMSNorm row scale)
    const float* rowsq = nullptr;   // [rows][rowsq_n] partial sums of squares of x's rows
    int rowsq_n = 0;
    float eps = 0.f;
    int M = 0, K = 0;
    int epi = COL_STORE;
    int split = 1;                  // 1, 2 or 4 workgroups per 16-column tile (each owns 16/split columns): spreads a narrow GEMM
                                    // (N/16 < number of CUs) over more CUs; RESID then emits NT*split rowsq partials per row
    float* out = nullptr;           // STORE: out [M][ldc] f32 row-major;  RESID: tiled f32 residual stream, updated in place
    bf16_t* next_bf16 = nullptr;    // RESID: tiled bf16(next_norm_w .* x_new), the operand of the GEMM behind the next RMSNorm
    const float* next_norm_w = nullptr;
    int64_t ldc = 0;
    const float* bias = nullptr;    // [N] optional
    const float* scale = nullptr;   // [N] optional (layer scale), RESID only
    float* rowsq_out = nullptr;     // RESID: [M][rowsq_out_n] sums of squares of the new x per 32-column tile
    int rowsq_out_n = 0;
    bf16_t* out_bf16 = nullptr;     // SILU: act [M][ldc]
    long long* stamps = nullptr;    // debug: 100-MHz wall-clock stamps of workgroup 0 at the phase boundaries
    // filled by the launcher
    const bf16_t* Wp = nullptr;
    int NT = 0, KT = 0, N = 0, up_tile_offset = 0;
    int x_tile0 = 0;                // gate/up: > 0 = workgroups own 1.5 pairs, the halves come from pairs x_tile0 ...
};
int launch_gemm_col(rt_ctx* ctx, const ColArgs& a, const PackedW& w, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// rowsq[M][0] = sum_k x[m][k]^2  (seed of the first NORM prologue of a stack)
// src_rows / dst_rows (device, [M], optional): block i reads x row src_rows[i] and writes (tiled / rowsq) row dst_rows[i]
// gather (optional): rows first..M-1 are read from table[idx[(row - first) * idx_stride]] (f32 rows of width H) instead of x; the
// index array moves by idx_frame_stride per frame of *frame_ptr - launch_gather_f32 folded into this launch
struct RowsqGather {
    const float* table = nullptr;
    const int32_t* idx = nullptr;
    int idx_stride = 1, first = 0;
    const int32_t* frame_ptr = nullptr;
    int64_t idx_frame_stride = 0;
};
int launch_rowsq(rt_ctx* ctx, const float* x, int M, int H, float* rowsq, int rowsq_n, float* x_tiled = nullptr,
                 bf16_t* a_tiled = nullptr, const float* norm_w = nullptr, const int32_t* src_rows = nullptr,
                 const int32_t* dst_rows = nullptr, const RowsqGather* gather = nullptr);

// decode-step input: x = add_vec + sum of n_src (<= 16) bf16 embedding rows (or one row of f32_table), emitted as tiled x,
// tiled bf16(norm_w .* x) and rowsq - the fusion of launch_gather_sum / launch_gather_f32 with launch_rowsq
struct GatherSrc;
int launch_embed_rowsq(rt_ctx* ctx, const GatherSrc* d_srcs, int n_src, const float* f32_table, const int32_t* d_idx, int idx_stride,
                       const int32_t* frame_ptr, int64_t idx_frame_stride, int M, int H, const float* add_vec, float* rowsq, int rowsq_n,
                       float* x_tiled, bf16_t* a_tiled, const float* norm_w, int32_t* frame_inc = nullptr, unsigned* arrive = nullptr);
// (frame_inc == frame_ptr, arrive = a zeroed device counter: the last workgroup of the launch also advances the frame counter)

// out[M][H] (row-major f32) = norm_w .* x * inv_rms(row) from the column path's tiled x and its rowsq partials
int launch_norm_tiled_rows(rt_ctx* ctx, const float* x_tiled, const float* rowsq, int rowsq_n, const float* w, float eps, int M, int H,
                           float* out);

// ---------------------------------------------------------------------------------- row kernels
// x[M][H] (f32, updated in place when n_slabs > 0 or add != nullptr):
//   x += scale[:] * sum_s slab[s]   (slab stride M*H; scale may be null)
//   xn = rmsnorm(x) * w  -> bf16 and/or f32
int launch_add_rmsnorm(rt_ctx* ctx, float* x, int M, int H, const float* slabs, int n_slabs, const float* slab_bias,
                       const float* scale, const float* w, float eps, bf16_t* out_bf16, float* out_f32);
// act[M][I] = silu(g) * u with g = sum_s slab[s][m][0:I], u = sum_s slab[s][m][I:2I]  -> bf16
int launch_silu_mul(rt_ctx* ctx, const float* slabs, int n_slabs, int M, int I, bf16_t* out, float* out_f32 = nullptr);
// y[M][N] = sum_s slab[s] (+ bias), optionally activation, to f32 and/or bf16
int launch_reduce_slabs(rt_ctx* ctx, const float* slabs, int n_slabs, int64_t M, int N, const float* bias, int act,
                        float* out_f32, bf16_t* out_bf16);

struct KvCache {
    bf16_t* k = nullptr;   // [layers][slots][kv_heads][max_pos][head_dim]
    bf16_t* v = nullptr;
    int layers = 0, slots = 0, kv_heads = 0, max_pos = 0, head_dim = 0;
    // Optional low planes (same layout): the cached value is k + k_lo, ~f32 precision.  Only the codec pre-transformer keeps
    // them - its output feeds the waveform, whose RMSE bar (1e-3) plain bf16 K/V alone would already spend.
    bf16_t* k_lo = nullptr;
    bf16_t* v_lo = nullptr;
    // Shared voice prefix: cache rows [0, prefix_len) of EVERY sequence are read from slot `prefix_slot` (one copy in HBM,
    // served from L2 / Infinity Cache to all the workgroups that re-read it) instead of a per-sequence copy.  -1 = off.
    int prefix_slot = -1, prefix_len = 0;
    // Fragment-tiled copies of the prefix slot's K and V, [layers][kv_heads][vt_stride] (32-key blocks of 8 MFMA operand tiles,
    // zero padded): what the matrix-core decode attention (attention_mfma.hip) loads; rebuilt whenever the prefix changes.
    bf16_t* kt_prefix = nullptr;
    bf16_t* vt_prefix = nullptr;
    int vt_stride = 0;              // elements per (layer, kv head) = ceil(max_pos / 32) * 4096
    int prefix_slot_alloc = -1;     // the slot that holds the prefix (fixed per model; prefix_slot above is switched on / off around its prefill)
    int tiles_len = -1;             // prefix length kt_prefix / vt_prefix were built for (-1: stale)
    size_t layer_stride() const { return (size_t)slots * kv_heads * max_pos * head_dim; }
};
// qkv slabs [S][M][(heads+2*kv_heads)*d] -> q (f32 [M][heads][d], normed + roped), k/v appended to the cache
int launch_qkv_post(rt_ctx* ctx, const float* slabs, int n_slabs, int M, int heads, int kv_heads, int head_dim,
                    const float* q_norm_w, const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin,
                    const int32_t* row_slot, const int32_t* row_pos, int pos_add, float* q_out, const KvCache& kv, int layer,
                    const int32_t* frame_ptr = nullptr);
// o[M][heads*d] (bf16) = softmax(q k^T / sqrt(d)) v over cache rows [max(0,pos-window+1), pos] of the row's slot
int launch_attention(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, int head_dim, const int32_t* row_slot,
                     const int32_t* row_pos, int pos_add, int window, const KvCache& kv, int layer, bf16_t* out,
                     const int32_t* frame_ptr = nullptr, int out_tiled = 0, float* out_f32 = nullptr);

// Every launch that depends on the frame index takes `frame_ptr` (device int, nullptr = 0): positions are
// row_pos + pos_add + *frame_ptr, so one captured hipGraph serves every frame of the decode loop.
// decode-only fusion of q/k-norm + RoPE + KV append + attention: qkv [M][(heads+2kv)*d] f32 (complete dot products)
int launch_attention_fused(rt_ctx* ctx, const float* qkv, int M, int heads, int kv_heads, int head_dim, const float* q_norm_w,
                           const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, const int32_t* row_slot,
                           const int32_t* row_pos, int pos_add, int window, const KvCache& kv, int layer, bf16_t* out,
                           const int32_t* frame_ptr = nullptr, int out_tiled = 0, int slot_base = -1,    // row_slot == nullptr: slot = slot_base + row; row_pos == nullptr: every row at pos_add
                           int pair_n = 0);   // > 0: rows [pair_n, 2 pair_n) sit one position behind rows [0, pair_n) of the same slots, appended by this launch

// decode attention with the shared prefix on the matrix cores (attention_mfma.hip): same contract as launch_attention_fused for
// head_dim 128, 2 query heads per kv head, no window, a shared prefix of >= 64 rows with its transposed V copy in place
extern rt_knob g_attn_mfma;
bool attention_mfma_ok(int M, int heads, int kv_heads, int head_dim, int window, const KvCache& kv);
int launch_attention_prefix_mfma(rt_ctx* ctx, const float* qkv, int M, int heads, int kv_heads, const float* q_norm_w, const float* k_norm_w, float eps,
                                 const float* rope_cos, const float* rope_sin, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                                 const KvCache& kv, int layer, bf16_t* out, const int32_t* frame_ptr, int out_tiled);
int launch_transpose_prefix_v(rt_ctx* ctx, KvCache& kv, int prefix_len);
// prompt rows (q prepared by launch_qkv_post, K / V appended) behind a shared prefix on the matrix cores: bf16 out [M][heads * d]
extern rt_knob g_prefill_attn_mfma;
bool attention_prefill_mfma_ok(int heads, int kv_heads, int head_dim, int window, const KvCache& kv);
// the prefix slot's own prefill: tiles layer `layer` of the prefix, then causal attention of its M consecutive rows on those tiles
bool attention_block_prefix_ok(int M, int heads, int kv_heads, int head_dim, int window, const KvCache& kv);
int launch_attention_block_prefix(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, const int32_t* row_slot, const int32_t* row_pos,
                                  KvCache& kv, int layer, bf16_t* out);
int launch_attention_prefill_mfma(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                                  const KvCache& kv, int layer, bf16_t* out);

// ------------------------------------------------------------------------------ embedding kernels
// out[m][:] = sum_j table_j[idx[m][j]][:]  (+ add_vec) ; tables are bf16 [V_j][H]; idx < 0 skips the term.
struct GatherSrc {
    const bf16_t* table;
    int64_t row_stride;  // elements
};
// idx_stride: elements between the index rows of consecutive output rows (default n_src); idx_frame_stride: added per frame.
int launch_gather_sum(rt_ctx* ctx, const GatherSrc* d_srcs, int n_src, const int32_t* d_idx, int M, int H,
                      const float* add_vec, const float* add_rows, const int32_t* add_row_idx, float* out_f32, bf16_t* out_bf16,
                      int idx_stride = 0, const int32_t* frame_ptr = nullptr, int64_t idx_frame_stride = 0);
int launch_gather_f32(rt_ctx* ctx, const float* table, int H, const int32_t* d_idx, int M, float* out_f32, bf16_t* out_bf16,
                      int idx_stride = 1, const int32_t* frame_ptr = nullptr, int64_t idx_frame_stride = 0);

// ------------------------------------------------------------------------------------- sampling
struct SampleArgs {
    const float* logits;       // [M][V] (sum of n_slabs slabs, slab stride M*V)
    int n_slabs;
    int M, V;
    int do_sample;
    float temperature;
    int top_k;
    float top_p;
    float rep_penalty;
    uint8_t* seen;             // [M][V] or null (repetition history, updated with the drawn token)
    int suppress_from;         // tokens >= suppress_from are forbidden ...
    int allow_token;           // ... except this one (-1: none)
    uint64_t seed;
    const uint64_t* seed_ptr;  // when set, the seed is read from the device (keeps captured graphs seed-independent)
    const int64_t* item_ids;   // [M]
    int frame, group;
    const int32_t* forced;     // [M] or null: teacher forcing (value < 0 = not forced)
    int32_t* out;              // token of row r goes to out[r * out_stride]
    int out_stride;
    int eos_token;             // >= 0: a drawn eos is reported in eos_flag[r] and replaced by 0 in `out`
    int32_t* eos_flag;         // [M] or null
    float* logits_copy;        // optional [M][V] summed logits for tracing
    // frame-indexed addressing for graph replay: when frame_ptr != nullptr the frame index is read from the device and
    // out / eos_flag / forced / logits_copy are advanced by frame * their stride; eos is allowed from min_frames on
    const int32_t* frame_ptr;
    const int32_t* frame_off;  // [M] or null: the frame counter value at which row r's current item started (RNG / min_frames use *frame_ptr - frame_off[r])
    int64_t out_fs, eos_fs, forced_fs, copy_fs;
    int eos_live, min_frames;
    long long* stamps;         // debug: wall-clock (100 MHz) stamps of row 0 at the phase boundaries, or null
    // optional fused next-step input (k_sample_w only): the drawn token's row of emb_table [V][emb_H] (f32) becomes the next
    // pass's input - tiled x, tiled bf16(emb_norm_w .* x) and the rowsq seed - without a separate gather launch
    const float* emb_table;
    int emb_H;
    const float* emb_norm_w;
    float* emb_rowsq;
    int emb_rowsq_n;
    float* emb_x_tiled;
    bf16_t* emb_a_tiled;
};
int launch_sample(rt_ctx* ctx, const SampleArgs& a);

// -------------------------------------------------------------------------------- codec elementwise

int launch_dwconv_ln(rt_ctx* ctx, const float* x, int B, int T, int C, const float* w /*[7][C]*/, const float* b,
                     const float* ln_w, const float* ln_b, float eps, float* out_f32);
int launch_code_embed_mean(rt_ctx* ctx, const bf16_t* table, int codebook, int Q, int H, const int32_t* codes /*[B][T][Q]*/,
                           int64_t rows, float* out_f32);

// last conv of the codec decoder (C -> 1, k = 7, causal) + clamp on hi / lo operand planes [B][T][C]; w = [7][C] f32
bool launch_final_conv_ok(int C);
int launch_final_conv(rt_ctx* ctx, const bf16_t* hi, const bf16_t* lo, int B, int T, int C, const float* w, const float* bias, float* wav);
int launch_f32_to_bf16(rt_ctx* ctx, const float* x, int64_t n, bf16_t* out);

// ------------------------------------------------------------------------------ conditioning front-end (encoder.hip)
// first conv of the audio encoder (1 -> C channels, k taps, causal): pcm [T] -> x [T][C] f32 and ELU(x) as hi / lo bf16 planes
int launch_enc_conv0(rt_ctx* ctx, const float* pcm, int64_t T, int C, int k, const float* w /*[C][k]*/, const float* bias, float* x,
                     bf16_t* hi, bf16_t* lo);
// split residual vector quantiser: sem / aco [T][D] f32 (aco is consumed: it holds the final residual afterwards),
// codebooks transposed [Q][D][K] f32 -> codes [T][Q]; fixed evaluation order (oracle/encoder.py rvq_level), lowest index on ties
int launch_rvq(rt_ctx* ctx, const float* sem, float* aco, int T, int D, int K, int Q, const float* const* d_cbT, int32_t* codes);
// statistics pooling over time: x [T][C] -> out [2C] = (mean, sqrt(var + 1e-5))
int launch_stats_pool(rt_ctx* ctx, const float* x, int T, int C, float* out);
// y [N] = act(W [N][K] x [K] + b): f32 weights, one wave per output
int launch_gemv_f32(rt_ctx* ctx, const float* W, const float* b, const float* x, int N, int K, int relu, float* y);
