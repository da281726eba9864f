// Model group, part 2 of 5: the transformer stacks as launch sequences - prompt-prefill form (stack_forward: talker prompts, codec
// pre-transformer and audio encoder in the float32-faithful form) and decode form (stack_decode: 5 launches per layer on the
// column-owner GEMM + fused attention) - with their workspaces, the LM-head GEMM and the text projection.  Host-side orchestration
// only: every FLOP is in gemm.hip / gemm_col.hip / attention*.hip / rowops.hip.
#include "model_internal.h"

namespace rtm {

// ---- device-side begin/end stamps of weight-streaming GEMM launches (hipExtLaunchKernelGGL events)
bool prof_events(rt_model* m, double bytes, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!m->prof) return false;
    if (m->prof_used >= m->prof_ev.size()) {
        if (m->prof_ev.size() >= 400000) return false;
        hipEvent_t x, y;
        if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return false;
        m->prof_ev.push_back({x, y});
    }
    if (m->prof_tag.size() <= m->prof_used) m->prof_tag.resize(m->prof_used + 1);
    m->prof_tag[m->prof_used] = {(uint8_t)m->prof_class, bytes};
    auto& pr = m->prof_ev[m->prof_used++];
    *a = pr.first;
    *b = pr.second;
    m->prof_bytes += bytes;
    return true;
}

// rows x K (bf16) times W^T -> raw f32 slabs [n_slabs][rows][N]
int gemm_rows(rt_model* m, const bf16_t* A, int rows, const PackedW& W, float* slabs, int* n_slabs) {
    if (rows <= 64) {
        const int S = skinny_pick_split(rows, W.N, W.K, m->ctx->n_cu);
        hipEvent_t e0, e1;
        prof_events(m, (double)W.N * W.K * 2.0, &e0, &e1);
        RT_TRY(launch_gemm_skinny(m->ctx, A, rows, W, slabs, W.N, S, e0, e1));
        *n_slabs = S;
    } else if (gemm_mid_shape_ok(W)) {
        // prompt prefill: final sums from 64 x 64 tiles over the whole K, K added in the skinny kernel's segments - a row gets the
        // same float32 sums among 13 rows (skinny) as among 416 or 3000: more than 1024 rows go down in equal chunks of <= 1024
        // (32 rows x > 30-token texts; the split-K tiled kernel's association would depend on the row count)
        const int n_chunks = (rows + 1023) / 1024, per = (rows + n_chunks - 1) / n_chunks;
        for (int r0 = 0; r0 < rows; r0 += per) {
            const int rc = std::min(per, rows - r0);
            if (rc > 64) RT_TRY(launch_gemm_mid(m->ctx, A + (size_t)r0 * W.K, rc, W, slabs + (size_t)r0 * W.N, W.N));
            else {      // (a tail of <= 64 rows cannot occur with equal chunks of > 512 rows; kept for safety: skinny slabs summed here would differ)
                return rt_fail(m->ctx, RT_ERR_STATE, "gemm_rows: %d-row chunk of a %d-row prefill", rc, rows);
            }
        }
        *n_slabs = 1;
    } else {
        // shapes the prompt-prefill kernel does not take (K % 64 != 0 or K < 128: no preset, the tiny test models): 64-row
        // blocks on the skinny kernel, whose split depends on the weight's shape only - slower than a tiled GEMM, but a row's
        // sums must not depend on how many rows it is prefilled with
        const int S = skinny_pick_split(rows, W.N, W.K, m->ctx->n_cu);
        for (int r0 = 0; r0 < rows; r0 += 64) {
            const int rc = std::min(64, rows - r0);
            RT_TRY(launch_gemm_skinny(m->ctx, A + (size_t)r0 * W.K, rc, W, slabs + (size_t)r0 * W.N, W.N, S, nullptr, nullptr, (int64_t)rows * W.N));
        }
        *n_slabs = S;
    }
    return RT_OK;
}

// float32 rows (fed as hi + lo bf16 planes, split on load) times W^T -> raw f32 slabs: the float32-faithful form of gemm_rows
int gemm_rows_f32(rt_model* m, const float* A, int rows, const PackedW& W, float* slabs, int* n_slabs) {
    // The split depends on the weight's shape ONLY, never on the number of rows: a codec frame must get the same float32 sums
    // whether its item is vocoded alone or in a batch of 32 (the waveform of a text may not depend on what it was batched with).
    // (slab workspace: 8 slabs for > 64 rows, 64 x 32768 floats otherwise - slab_floats)
    (void)rows;
    int S = 1;
    while (S < 8 && W.K / (S * 2) >= 256 && (int64_t)S * 2 * W.N <= 32768) S *= 2;
    GemmA a; a.ptr = A; a.is_f32 = 1; a.split = 1; a.M = rows; a.Cin = W.K; a.taps = 1;
    GemmEpi e; e.out_f32 = slabs; e.ldc = W.N; e.split_k = S;
    RT_TRY(launch_gemm(m->ctx, a, W, e));
    *n_slabs = S;
    return RT_OK;
}

size_t slab_floats(const rt_stack_dims& d, int M, int n_cu) {
    const size_t widest = std::max<size_t>((size_t)2 * d.inter, (size_t)(d.heads + 2 * d.kv_heads) * d.head_dim);
    size_t need = std::max<size_t>((size_t)M * widest * (M > 64 ? 8 : 1), (size_t)64 * 32768);
    if (M > 64) {       // gemm_rows on a shape k_gemm_mid does not take: the skinny kernel's slabs for every row
        const int q = d.heads * d.head_dim, qkv = (d.heads + 2 * d.kv_heads) * d.head_dim;
        const int shapes[4][2] = {{qkv, d.hidden}, {d.hidden, q}, {2 * d.inter, d.hidden}, {d.hidden, d.inter}};
        for (auto& s : shapes)
            if (!(g_prefill_mid && s[1] % 64 == 0 && s[1] >= 128))
                need = std::max<size_t>(need, (size_t)M * s[0] * skinny_pick_split(M, s[0], s[1], n_cu));
    }
    return need;
}
int alloc_stack_ws(rt_model* m, const rt_stack_dims& d, int M, StackWs* w, bool precise) {
    if (precise) {
        RT_TRY(pool_arr(m, (size_t)M * d.hidden, &w->xn32));
        RT_TRY(pool_arr(m, (size_t)M * d.heads * d.head_dim, &w->ao32));
        RT_TRY(pool_arr(m, (size_t)M * d.inter, &w->act32));
        RT_TRY(pool_arr(m, slab_floats(d, M), &w->slabs));
        RT_TRY(pool_arr(m, (size_t)M * d.heads * d.head_dim, &w->q));
        return RT_OK;
    }
    RT_TRY(pool_arr(m, (size_t)M * d.hidden, &w->xn));
    RT_TRY(pool_arr(m, slab_floats(d, M), &w->slabs));
    RT_TRY(pool_arr(m, (size_t)M * d.heads * d.head_dim, &w->q));
    RT_TRY(pool_arr(m, (size_t)M * d.heads * d.head_dim, &w->ao));
    RT_TRY(pool_arr(m, (size_t)M * d.inter, &w->act));
    return RT_OK;
}

// x [M][H] f32 in/out (residual stream); on return out_bf16/out_f32 hold the final-norm output.
// prefix_rows: the rows are the voice prefix itself (consecutive positions of the prefix slot): its attention runs on the
// matrix cores over the layer's freshly tiled K / V (launch_attention_block_prefix) where that form applies
int stack_forward(rt_model* m, StackW& S, StackWs& w, float* x, int M, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                  bf16_t* out_bf16, float* out_f32, const int32_t* frame_ptr, bool prefix_rows) {
    rt_ctx* ctx = m->ctx;
    const rt_stack_dims& d = S.d;
    const int H = d.hidden;
    int ns = 0;
    const float* pending_scale = nullptr;
    if (w.xn32) {
        // float32-faithful form (codec pre-transformer): every GEMM operand stays float32 and is fed to the MFMAs as hi + lo
        // bf16 planes, K/V are cached as hi + lo planes, attention and SwiGLU write float32.  Plain bf16 operands here cost
        // 3.2e-3 of waveform RMSE at the real codec dimensions (8 layers, 1024 wide) - each of the four rounding points
        // alone >= 1e-3 (tests/test_model_shapes_gpu.py, DESIGN.md "Precision policy") - for 0.2 of the decoder's 5.1 GFLOP/frame.
        if (!S.kv.k_lo || !out_f32 || out_bf16) return rt_fail(ctx, RT_ERR_STATE, "stack_forward: precise mode needs hi/lo K/V planes and a float32 output");
        for (int i = 0; i < d.layers; ++i) {
            LayerW& L = S.L[i];
            RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, pending_scale, L.ln1, d.rms_eps, nullptr, w.xn32));
            RT_TRY(gemm_rows_f32(m, w.xn32, M, L.wqkv, w.slabs, &ns));
            RT_TRY(launch_qkv_post(ctx, w.slabs, ns, M, d.heads, d.kv_heads, d.head_dim, L.qn, L.kn, d.rms_eps, S.cos, S.sin, row_slot, row_pos,
                                   pos_add, w.q, S.kv, i, frame_ptr));
            RT_TRY(launch_attention(ctx, w.q, M, d.heads, d.kv_heads, d.head_dim, row_slot, row_pos, pos_add, S.window, S.kv, i, nullptr, frame_ptr, 0, w.ao32));
            RT_TRY(gemm_rows_f32(m, w.ao32, M, L.wo, w.slabs, &ns));
            RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, L.ls1, L.ln2, d.rms_eps, nullptr, w.xn32));
            RT_TRY(gemm_rows_f32(m, w.xn32, M, L.wgu, w.slabs, &ns));
            RT_TRY(launch_silu_mul(ctx, w.slabs, ns, M, d.inter, nullptr, w.act32));
            RT_TRY(gemm_rows_f32(m, w.act32, M, L.wd, w.slabs, &ns));
            pending_scale = L.ls2;
            if (g_sync_parts && !g_use_graph) RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, pending_scale, S.norm, d.rms_eps, nullptr, out_f32));
        return RT_OK;
    }
    for (int i = 0; i < d.layers; ++i) {
        LayerW& L = S.L[i];
        RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, pending_scale, L.ln1, d.rms_eps, w.xn, nullptr));
        RT_TRY(gemm_rows(m, w.xn, M, L.wqkv, w.slabs, &ns));
        RT_TRY(launch_qkv_post(ctx, w.slabs, ns, M, d.heads, d.kv_heads, d.head_dim, L.qn, L.kn, d.rms_eps, S.cos, S.sin, row_slot, row_pos,
                               pos_add, w.q, S.kv, i, frame_ptr));
        if (prefix_rows && pos_add == 0 && !frame_ptr && attention_block_prefix_ok(M, d.heads, d.kv_heads, d.head_dim, S.window, S.kv))
            RT_TRY(launch_attention_block_prefix(ctx, w.q, M, d.heads, d.kv_heads, row_slot, row_pos, S.kv, i, w.ao));
        else
            RT_TRY(launch_attention(ctx, w.q, M, d.heads, d.kv_heads, d.head_dim, row_slot, row_pos, pos_add, S.window, S.kv, i, w.ao, frame_ptr));
        RT_TRY(gemm_rows(m, w.ao, M, L.wo, w.slabs, &ns));
        RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, L.ls1, L.ln2, d.rms_eps, w.xn, nullptr));
        RT_TRY(gemm_rows(m, w.xn, M, L.wgu, w.slabs, &ns));
        RT_TRY(launch_silu_mul(ctx, w.slabs, ns, M, d.inter, w.act));
        RT_TRY(gemm_rows(m, w.act, M, L.wd, w.slabs, &ns));
        pending_scale = L.ls2;
        if (g_sync_parts && !g_use_graph) RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (profiling aid: bounds the dispatches in flight)
    }
    RT_TRY(launch_add_rmsnorm(ctx, x, M, H, w.slabs, ns, nullptr, pending_scale, S.norm, d.rms_eps, out_bf16, out_f32));
    return RT_OK;
}

// ---- decode-time stack (M <= 64): 5 launches per layer with the column-owner GEMM and the fused attention.
// x is the un-normalised residual stream; rowsq [M][H/32] carries the per-tile sums of squares of x that the next
// NORM prologue turns into the RMSNorm row scale.  On return x and rowsq describe the stack's output BEFORE the final
// norm, which the consumer (LM head / mtp projection) applies in its own prologue.
int alloc_dec_ws(rt_model* m, const rt_stack_dims& d, int M, DecWs* w) {
    const size_t Mp = (size_t)(M + 31) / 32 * 32;     // tiled buffers hold whole 32-row blocks
    RT_TRY(pool_arr(m, Mp * d.hidden, &w->xT));
    RT_TRY(pool_arr(m, Mp * d.hidden, &w->xa));
    RT_TRY(pool_arr(m, (size_t)M * d.heads * d.head_dim, &w->q));
    RT_TRY(pool_arr(m, (size_t)M * (d.heads + 2 * d.kv_heads) * d.head_dim, &w->qkv));
    RT_TRY(pool_arr(m, Mp * d.heads * d.head_dim, &w->ao));
    RT_TRY(pool_arr(m, Mp * d.inter, &w->act));
    return RT_OK;
}
// Row blocks of up to 64 (one launch for the predictor's two-position first pass at batch 32).
int col_gemm(rt_model* m, const ColArgs& a0, const PackedW& W, bool is_predictor) {
    const int blk = g_col_rows64 ? 64 : 32;
    for (int r0 = 0; r0 < a0.M; r0 += blk) {
        ColArgs a = a0;
        a.nt = is_predictor ? g_pred_nt.load() : 1;
        a.M = std::min(blk, a0.M - r0);
        a.row_off = a0.row_off + r0;
        hipEvent_t e0, e1;
        prof_events(m, (double)W.N * W.K * 2.0, &e0, &e1);
        RT_TRY(launch_gemm_col(m->ctx, a, W, e0, e1));
    }
    return RT_OK;
}
// one_row_per_slot = false (the predictor's 2-row first pass): a row must see the K/V another row of the same launch
// appends, so q/k-norm + RoPE + append run as their own launch before the attention.
int stack_decode(rt_model* m, StackW& S, DecWs& w, float* x, float* rowsq, int M, const int32_t* row_slot, const int32_t* row_pos,
                 int pos_add, bool one_row_per_slot, const int32_t* frame_ptr, int slot_base, bool zero_pos) {
    rt_ctx* ctx = m->ctx;
    const rt_stack_dims& d = S.d;
    const int H = d.hidden, sp_h = col_split_for(H, ctx->n_cu), NTh = H / 16 * sp_h, qw = (d.heads + 2 * d.kv_heads) * d.head_dim;
    const bool isp = &S == &m->pred;
    for (int i = 0; i < d.layers; ++i) {
        LayerW& L = S.L[i];
        const float* next_w = (i + 1 < d.layers) ? S.L[i + 1].ln1 : S.norm;   // the norm that reads x after this layer
        ColArgs a;      // qkv = rmsnorm(x; ln1) Wqkv^T : operand w.xa = bf16(ln1 .* x), row scale from rowsq
        a.A = w.xa; a.post_scale = 1; a.rowsq = rowsq; a.rowsq_n = NTh; a.eps = d.rms_eps; a.M = M; a.K = H;
        a.epi = COL_STORE; a.out = w.qkv; a.ldc = qw; a.split = col_split_for(qw, ctx->n_cu);
        RT_TRY(col_gemm(m, a, L.wqkv, isp));
        if (one_row_per_slot) {
            // (slot_base >= 0: rows sit in consecutive slots; zero_pos: every row at pos_add - the attention then needs no slot /
            //  position arrays, i.e. no dependent scalar loads in front of its K / V requests)
            RT_TRY(launch_attention_fused(ctx, w.qkv, M, d.heads, d.kv_heads, d.head_dim, L.qn, L.kn, d.rms_eps, S.cos, S.sin, slot_base >= 0 ? nullptr : row_slot,
                                          zero_pos ? nullptr : row_pos, pos_add, S.window, S.kv, i, w.ao, frame_ptr, 1, slot_base));
        } else if (g_pair_attn && M % 2 == 0 && d.heads <= 2 * d.kv_heads && d.heads % d.kv_heads == 0 && row_slot && row_pos) {
            // the two-position pass (rows [0, M/2) at one position, rows [M/2, M) of the same slots at the next): the fused launch,
            // each second-position workgroup working out its partner's K / V itself (attention.hip, pair_n)
            RT_TRY(launch_attention_fused(ctx, w.qkv, M, d.heads, d.kv_heads, d.head_dim, L.qn, L.kn, d.rms_eps, S.cos, S.sin, row_slot, row_pos, pos_add,
                                          S.window, S.kv, i, w.ao, frame_ptr, 1, -1, M / 2));
        } else {
            RT_TRY(launch_qkv_post(ctx, w.qkv, 1, M, d.heads, d.kv_heads, d.head_dim, L.qn, L.kn, d.rms_eps, S.cos, S.sin, row_slot, row_pos,
                                   pos_add, w.q, S.kv, i, frame_ptr));
            RT_TRY(launch_attention(ctx, w.q, M, d.heads, d.kv_heads, d.head_dim, row_slot, row_pos, pos_add, S.window, S.kv, i, w.ao, frame_ptr, 1));
        }
        ColArgs o;      // x += ls1 .* (ao Wo^T); emits rowsq and bf16(ln2 .* x) for the MLP
        o.A = w.ao; o.M = M; o.K = d.heads * d.head_dim; o.epi = COL_RESID; o.out = x; o.ldc = H; o.scale = L.ls1;
        o.rowsq_out = rowsq; o.rowsq_out_n = NTh; o.next_bf16 = w.xa; o.next_norm_w = L.ln2; o.split = sp_h;
        RT_TRY(col_gemm(m, o, L.wo, isp));
        ColArgs gu;     // act = silu(g) * u with [g; u] = rmsnorm(x; ln2) Wgu^T
        gu.A = w.xa; gu.post_scale = 1; gu.rowsq = rowsq; gu.rowsq_n = NTh; gu.eps = d.rms_eps; gu.M = M; gu.K = H;
        gu.epi = COL_SILU; gu.out_bf16 = w.act; gu.ldc = d.inter; gu.split = col_split_silu(2 * d.inter, ctx->n_cu);
        RT_TRY(col_gemm(m, gu, L.wgu, isp));
        ColArgs dn;     // x += ls2 .* (act Wd^T); emits rowsq and bf16(next norm .* x)
        dn.A = w.act; dn.M = M; dn.K = d.inter; dn.epi = COL_RESID; dn.out = x; dn.ldc = H; dn.scale = L.ls2;
        dn.rowsq_out = rowsq; dn.rowsq_out_n = NTh; dn.next_bf16 = w.xa; dn.next_norm_w = next_w; dn.split = sp_h;
        RT_TRY(col_gemm(m, dn, L.wd, isp));
        if (g_sync_parts && !g_use_graph) RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}
// out[M][N] = rmsnorm(x) W^T (+ bias): LM heads and the mtp projection, final norm applied in the GEMM prologue
// xa = tiled bf16(final_norm_w .* x) as left by the stack's last down-projection (or by k_rowsq when the stack has not run yet);
// rows [row_off, row_off + M) are read, out rows 0.. are written row-major
int col_head(rt_model* m, const bf16_t* xa, const float* rowsq, int rowsq_n, int row_off, int M, int K, float eps,
             const PackedW& W, const float* bias, float* out) {
    ColArgs a;
    a.A = xa; a.post_scale = 1; a.rowsq = rowsq; a.rowsq_n = rowsq_n; a.eps = eps; a.M = M; a.K = K; a.row_off = row_off;
    a.epi = COL_STORE; a.out = out - (size_t)row_off * W.N; a.ldc = W.N; a.bias = bias; a.split = col_split_for(W.N, m->ctx->n_cu);
    return col_gemm(m, a, W);
}

// text_proj(text_embedding[ids]) -> f32 [n][H]
int alloc_text_ws(rt_model* m, int n, TextWs* w) {
    RT_TRY(pool_arr(m, (size_t)n * m->cfg.text_hidden, &w->e));
    RT_TRY(pool_arr(m, (size_t)n * m->cfg.text_hidden, &w->h1));
    RT_TRY(pool_arr(m, 1, &w->d_src));
    const GatherSrc src{TBL(m, "talker.text_embedding"), m->cfg.text_hidden};
    RT_HIP(m->ctx, hipMemcpy(w->d_src, &src, sizeof(src), hipMemcpyHostToDevice));
    return RT_OK;
}
int text_project(rt_model* m, const int32_t* d_ids, int n, float* out, const TextWs* ws) {
    rt_ctx* ctx = m->ctx;
    const rt_model_config& c = m->cfg;
    TextWs own;
    if (!ws) { RT_TRY(alloc_text_ws(m, n, &own)); ws = &own; }
    bf16_t *e = ws->e, *h1 = ws->h1;
    GatherSrc* d_src = ws->d_src;
    RT_TRY(launch_gather_sum(ctx, d_src, 1, d_ids, n, c.text_hidden, nullptr, nullptr, nullptr, nullptr, e));
    GemmA a; a.ptr = e; a.M = n; a.Cin = c.text_hidden;
    GemmEpi e1; e1.bias = VEC(m, "talker.tp_fc1_b"); e1.act = ACT_SILU; e1.out_bf16 = h1; e1.ldc = c.text_hidden;
    RT_TRY(launch_gemm(ctx, a, PW(m, "talker.tp_fc1"), e1));
    GemmA a2; a2.ptr = h1; a2.M = n; a2.Cin = c.text_hidden;
    GemmEpi e2; e2.bias = VEC(m, "talker.tp_fc2_b"); e2.out_f32 = out; e2.ldc = c.talker.hidden;
    RT_TRY(launch_gemm(ctx, a2, PW(m, "talker.tp_fc2"), e2));
    return RT_OK;
}

}  // namespace rtm
