// Model group, part 4 of 5: autoregressive generation (rt_generate, and the same in pieces: rt_generate_begin / _step / _peek / _end).
// Stands behind the third-party model call the reference makes at providers/qwen.py:247-258 (generate_custom_voice /
// generate_voice_clone): text ids in, codec frames out.  Host-side orchestration only.
#include "model_internal.h"

using namespace rtm;

namespace {

__global__ void k_frame_inc(int32_t* f) { if (threadIdx.x == 0) *f += 1; }
}  // namespace

// --------------------------------------------------------------------------------------- generate
// One generation in flight: everything rt_generate used to keep on its stack, so that the frame loop can be run in pieces
// (rt_generate_begin / rt_generate_step / rt_generate_end - sub-segment streaming) as well as in one go
// (rt_generate = begin + all frames + end).  Its device buffers come from the model's pool under tag 1, which the other entry
// points (rt_code2wav between two steps) leave alone.
struct rt_gen_run {
    struct Staging { std::vector<int32_t> tid, cid, slot, pos, last, dst; };
    struct Lane {
        int b0 = 0, n = 0;
        hipStream_t stream = nullptr;
        float *xt = nullptr, *hn_f32 = nullptr, *xp = nullptr, *logits = nullptr, *rowsq_t = nullptr, *rowsq_p = nullptr;
        bf16_t *hn = nullptr, *hn_p = nullptr;
        int32_t *d_slot_b = nullptr, *d_pos_b = nullptr, *d_pos_p2 = nullptr, *d_zero_pos = nullptr, *d_frame = nullptr, *d_frame_off = nullptr;
        int64_t* d_items = nullptr;
        uint8_t* d_seen = nullptr;
        DecWs dwt, dwp;
        StackWs wt, wp;
        bool done = false;
    };
    rt_model* m = nullptr;
    rt_ctx* ctx = nullptr;
    rt_generate_args A{};                                  // a copy; its input arrays point into the vectors below
    std::vector<int32_t> text_ids, text_offsets, max_frames, forced_codes, forced_offsets;
    std::vector<int64_t> item_ids;
    int N = 0, G = 0, H = 0, Hp = 0, Vc = 0, Vp = 0, B = 0, every = 1, T_max = 0, F_max = 0, Lp = 0, S_cap = 0, max_suffix = 0, n_suffix = 0;
    bool queued = false, col = false, use_graph = false;
    int NTt = 0, NTp = 0;
    int64_t codes_fs = 0;
    std::vector<std::unique_ptr<Staging>> staging;        // host sources of asynchronous uploads: alive until the run ends
    std::vector<int32_t> P;
    int32_t *d_tid = nullptr, *d_cid = nullptr, *d_slot = nullptr, *d_pos = nullptr, *d_last = nullptr, *d_dst = nullptr;
    float *temb = nullptr, *x = nullptr, *hn_all_f32 = nullptr;
    const float* pad_t = nullptr;
    StackWs w_prefill;
    TextWs w_text;
    int32_t *d_codes = nullptr, *d_eos = nullptr, *d_forced = nullptr;
    uint64_t* d_seed = nullptr;
    const PackedW* head = nullptr;
    std::vector<Lane> lanes;
    int n_lanes = 1;
    hipStream_t main_stream = nullptr;
    // frame-loop state
    std::vector<int32_t> eos_host, codes_host;
    std::vector<int> produced, start, item_row, row_item, lane_frames;
    std::vector<char> finished, parked;
    int next_item = 0, n_finished = 0, frames_run = 0, checked = 0, t = 0, codes_copied = 0;
    int64_t n_swaps = 0;
    double launch_host_us = 0.0;
    bool cancelled = false, all_done = false;
    int eos_every = 0;
    std::vector<int32_t> h_pos_b, h_off;
    std::vector<int64_t> h_items;

    int init(const rt_generate_args* a);
    int prefill(const std::vector<int>& items, const std::vector<int>& rows, bool first);
    int enqueue_a(Lane& ln);
    int enqueue_b(Lane& ln);
    bool apply_frames(int upto);
    int swap_in(int t1);
    int fetch(int upto, bool with_codes);
    int advance(int n_frames);
    int frames_of(int it) const { return item_row[it] < 0 ? 0 : std::max(0, std::min(produced[it], frames_run - start[it])); }
    int finish(int32_t* h_codes, int32_t* h_n_frames);
};

namespace {
void gen_release(rt_model* m) {
    if (!m->run) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    for (auto& ln : m->run->lanes) if (ln.stream && ln.stream != m->ctx->stream) (void)hipStreamSynchronize(ln.stream);
    delete m->run;
    m->run = nullptr;
    g_runs_in_flight.fetch_sub(1);
    for (auto& b : m->pool) if (b.tag == 1) { b.used = false; b.tag = 0; }
}
}  // namespace
void rt_gen_drop(rt_model* m) { gen_release(m); }

int rt_gen_run::init(const rt_generate_args* a) {
    const rt_model_config& c = m->cfg;
    // N items are decoded on B = min(N, max_batch) rows.  With N > B the first B items start on the rows and the others
    // wait in a queue: whenever the host learns (every g_handover_every frames) that rows have finished, the next queued
    // items take them over - their prompt suffixes are prefilled into the rows' KV slots between two frames, the rows'
    // state / position base / RNG stream / repetition history are re-pointed - so that every weight pass keeps serving
    // live rows.  An item's result depends only on (item id, seed): bit for bit the codes it gets in any static batch or
    // alone - the prompt prefill gives a row the same float32 sums whatever it is batched with (k_gemm_mid adds K in the
    // skinny kernel's segments, prompt attention always runs the 4-wave split), decode rows never see each other.
    A = *a;
    N = A.n_items; G = c.n_groups; H = c.talker.hidden; Hp = c.predictor.hidden; Vc = c.codec_vocab; Vp = c.predictor_vocab;
    if (N < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: n_items %d < 1", N);
    B = std::min(N, A.max_rows > 0 ? std::min(A.max_rows, c.max_batch) : c.max_batch);
    queued = N > B;
    if (!A.h_text_ids || !A.h_text_offsets || !A.h_max_frames || !A.h_item_ids) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: null array");
    if (queued && (A.h_forced_codes || A.d_trace_talker || A.d_trace_predictor))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: teacher forcing / logit traces need n_items <= max_batch (%d)", c.max_batch);
    // the caller's arrays need not outlive rt_generate_begin: keep copies
    text_offsets.assign(A.h_text_offsets, A.h_text_offsets + N + 1);
    if (text_offsets[0] != 0 || text_offsets[N] < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: bad text offsets");
    text_ids.assign(A.h_text_ids, A.h_text_ids + text_offsets[N]);
    max_frames.assign(A.h_max_frames, A.h_max_frames + N);
    item_ids.assign(A.h_item_ids, A.h_item_ids + N);
    A.h_text_ids = text_ids.data(); A.h_text_offsets = text_offsets.data(); A.h_max_frames = max_frames.data(); A.h_item_ids = item_ids.data();
    every = std::max(1, queued ? g_handover_every.load() : g_eos_check_every.load());
    int64_t budget_sum = 0;
    for (int b = 0; b < N; ++b) {
        const int nt = A.h_text_offsets[b + 1] - A.h_text_offsets[b];
        if (nt < 0 || A.h_max_frames[b] < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: item %d has bad text or max_frames", b);
        T_max = std::max(T_max, A.h_max_frames[b]);
        if (b < B) n_suffix += nt + 2;
        max_suffix = std::max(max_suffix, nt + 2);
        budget_sum += A.h_max_frames[b] + every;
    }
    for (int i = 0; i < A.h_text_offsets[N]; ++i)
        if (A.h_text_ids[i] < 0 || A.h_text_ids[i] >= c.text_vocab) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: text id out of range");
    Lp = m->prefix_len;
    F_max = T_max;                                         // the longest single item
    // (a finished row keeps stepping until the host has seen its flag: up to `every` - 1 positions past its last frame)
    if (Lp + max_suffix + F_max + (queued ? every : 0) + 1 > c.max_positions)
        return rt_fail(ctx, RT_ERR_LENGTH, "rt_generate: prompt length %d + %d frames exceeds max_positions %d", Lp + max_suffix, F_max, c.max_positions);
    // frame-counter budget: list scheduling finishes within sum / rows + longest (each item charged its wait for the next check)
    if (queued) T_max = (int)std::min<int64_t>(budget_sum / B + F_max + every + 1, (int64_t)1 << 24);
    if (A.talker.do_sample && (A.talker.top_k < 1 || A.talker.top_k > 64 || !(A.talker.temperature > 0)))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: sampling needs 1 <= top_k <= 64 and temperature > 0");
    if (A.predictor.do_sample && (A.predictor.top_k < 1 || A.predictor.top_k > 64 || !(A.predictor.temperature > 0)))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: predictor sampling needs 1 <= top_k <= 64 and temperature > 0");
    if (A.h_forced_codes) {
        if (!A.h_forced_offsets) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: forced codes without offsets");
        forced_offsets.assign(A.h_forced_offsets, A.h_forced_offsets + N + 1);
        forced_codes.assign(A.h_forced_codes, A.h_forced_codes + (size_t)forced_offsets[N] * G);
        A.h_forced_codes = forced_codes.data(); A.h_forced_offsets = forced_offsets.data();
    }

    // ---- the voice prefix KV stays in its own slot: every sequence reads cache rows [0, Lp) from there (KvCache::prefix_slot)
    m->talker.kv.prefix_slot = m->prefix_slot();
    m->talker.kv.prefix_len = Lp;
    if (g_attn_mfma && c.talker.head_dim == 128 && !m->prefix_tiles_valid) {
        RT_TRY(launch_transpose_prefix_v(ctx, m->talker.kv, Lp));
        m->prefix_tiles_valid = true;
    }
    // ---- suffix rows: [text tokens + tts_eos] x codec_pad, then (tts_pad, codec_bos).  Row 0 of the id / embedding buffers
    // is the projected tts_pad that every decode step adds to its input; the suffix rows of the items being prefilled follow.
    S_cap = queued ? B * max_suffix : n_suffix;
    P.assign(N, 0);
    RT_TRY(pool_arr(m, S_cap + 1, &d_tid));
    RT_TRY(pool_arr(m, (size_t)S_cap * G, &d_cid));
    RT_TRY(pool_arr(m, S_cap, &d_slot));
    RT_TRY(pool_arr(m, S_cap, &d_pos));
    RT_TRY(pool_arr(m, B, &d_last));
    RT_TRY(pool_arr(m, B, &d_dst));
    RT_TRY(pool_arr(m, (size_t)(S_cap + 1) * H, &temb));
    RT_TRY(pool_arr(m, (size_t)S_cap * H, &x));
    RT_TRY(pool_arr(m, (size_t)S_cap * H, &hn_all_f32));
    pad_t = temb;
    RT_TRY(alloc_stack_ws(m, c.talker, S_cap, &w_prefill));
    RT_TRY(alloc_text_ws(m, S_cap + 1, &w_text));
    {
        std::vector<int> items(B), rows(B);
        for (int b = 0; b < B; ++b) { items[b] = b; rows[b] = b; }
        RT_TRY(prefill(items, rows, true));
    }
    // ---- decode state.  The batch is cut into `lanes` groups of consecutive items, each decoding on its own stream with
    // its own workspaces: a decode step is a chain of ~600 short dependent kernels whose cost is latency, not bytes, so two
    // chains in flight overlap each other's launch/drain gaps (items are independent: same results for any lane count).
    RT_TRY(pool_arr(m, (size_t)T_max * B * G, &d_codes));      // [frame counter][row][group]
    RT_TRY(pool_arr(m, (size_t)T_max * B, &d_eos));
    RT_HIP(ctx, hipMemsetAsync(d_codes, 0, (size_t)T_max * B * G * 4, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(d_eos, 0, (size_t)T_max * B * 4, ctx->stream));
    std::vector<int32_t> forced_host;
    if (A.h_forced_codes) {
        forced_host.assign((size_t)T_max * G * B, -1);  // layout [t][g][b]
        for (int b = 0; b < B; ++b) {
            const int o = A.h_forced_offsets[b], nf = A.h_forced_offsets[b + 1] - o;
            for (int tt = 0; tt < T_max; ++tt) {
                if (nf <= 0) continue;
                const int ts = tt < nf ? tt : nf - 1;                    // predictor groups reuse the last forced frame
                for (int q = 0; q < G; ++q) {
                    if (q == 0 && tt >= nf) continue;                    // group 0 is only forced while frames remain
                    forced_host[((size_t)tt * G + q) * B + b] = A.h_forced_codes[((size_t)o + ts) * G + q];
                }
            }
        }
        RT_TRY(pool_arr(m, forced_host.size(), &d_forced));
        RT_HIP(ctx, hipMemcpyAsync(d_forced, forced_host.data(), forced_host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    RT_TRY(pool_arr(m, 1, &d_seed));
    RT_HIP(ctx, hipMemcpyAsync(d_seed, &A.seed, 8, hipMemcpyHostToDevice, ctx->stream));
    codes_fs = (int64_t)B * G;
    // Column path (2B <= 64): xt = un-normalised talker residual stream + rowsq_t; legacy path: hn = final-norm output
    col = g_decode_col && B <= g_col_max_rows && H % 32 == 0 && Hp % 32 == 0 && c.talker.inter % 32 == 0 && c.predictor.inter % 32 == 0;
    NTt = H / 16 * col_split_for(H, ctx->n_cu); NTp = Hp / 16 * col_split_for(Hp, ctx->n_cu);   // rowsq partials per row
    head = &PW(m, "talker.codec_head");
    float* x_all = x;

    n_lanes = std::max(1, std::min(g_decode_lanes.load(), 8));
    if (!col || m->prof || queued) n_lanes = 1;            // (per-launch profiling wants undisturbed launches)
    if (queued && !col) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_generate: queued items (n_items %d > %d rows) need the column decode path", N, B);
    while (n_lanes > 1 && B / n_lanes < 8) --n_lanes;      // a lane narrower than 8 rows only multiplies the weight traffic
    lanes.assign(n_lanes, Lane());
    main_stream = ctx->stream;
    if (n_lanes > 1) {
        while ((int)m->lane_streams.size() < n_lanes) {
            hipStream_t st = nullptr;
            RT_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            m->lane_streams.push_back(st);
            hipEvent_t ev = nullptr;
            RT_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            m->lane_events.push_back(ev);
        }
        if (!m->fork_event) RT_HIP(ctx, hipEventCreateWithFlags(&m->fork_event, hipEventDisableTiming));
    }
    for (int l = 0; l < n_lanes; ++l) {
        Lane& ln = lanes[l];
        ln.b0 = (int)((int64_t)B * l / n_lanes);
        ln.n = (int)((int64_t)B * (l + 1) / n_lanes) - ln.b0;
        ln.stream = n_lanes > 1 ? m->lane_streams[l] : main_stream;
        const int n = ln.n, n2 = 2 * n;
        RT_TRY(pool_arr(m, (size_t)n * H, &ln.xt));
        RT_TRY(pool_arr(m, (size_t)n * H, &ln.hn_f32));
        RT_TRY(pool_arr(m, (size_t)n * H, &ln.hn));
        RT_TRY(pool_arr(m, (size_t)n2 * Hp, &ln.xp));
        RT_TRY(pool_arr(m, (size_t)n2 * Hp, &ln.hn_p));
        RT_TRY(pool_arr(m, (size_t)64 * 32768, &ln.logits));
        RT_TRY(pool_arr(m, n2, &ln.d_slot_b));
        RT_TRY(pool_arr(m, n, &ln.d_pos_b));
        RT_TRY(pool_arr(m, n2, &ln.d_pos_p2));
        RT_TRY(pool_arr(m, n2, &ln.d_zero_pos));
        RT_TRY(pool_arr(m, n, &ln.d_items));
        RT_TRY(pool_arr(m, (size_t)n * Vc, &ln.d_seen));
        RT_TRY(pool_arr(m, 2, &ln.d_frame));                      // [0] the frame counter, [1] arrivals of the launch that advances it
        RT_TRY(pool_arr(m, n, &ln.d_frame_off));
        RT_HIP(ctx, hipMemsetAsync(ln.d_frame_off, 0, n * 4, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_seen, 0, (size_t)n * Vc, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_zero_pos, 0, n2 * 4, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_frame, 0, 8, ctx->stream));
        std::vector<int32_t> sl(n2), p2(n2);
        for (int b = 0; b < n; ++b) { sl[b] = ln.b0 + b; sl[n + b] = ln.b0 + b; p2[b] = 0; p2[n + b] = 1; }
        RT_HIP(ctx, hipMemcpyAsync(ln.d_slot_b, sl.data(), n2 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(ln.d_pos_p2, p2.data(), n2 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(ln.d_pos_b, P.data() + ln.b0, n * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(ln.d_items, A.h_item_ids + ln.b0, n * 8, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // (sl / p2 are stack-local staging)
        if (col) {
            RT_TRY(pool_arr(m, (size_t)n * NTt, &ln.rowsq_t));
            RT_TRY(pool_arr(m, (size_t)n2 * NTp, &ln.rowsq_p));
            RT_TRY(alloc_dec_ws(m, c.talker, n, &ln.dwt));
            RT_TRY(alloc_dec_ws(m, c.predictor, n2, &ln.dwp));
            // xt <- residual-stream rows (before the final norm) of each item's last prompt position
            RT_TRY(launch_gather_f32(ctx, x_all, H, d_last + ln.b0, n, ln.xt, nullptr));
            // (the prompt's last rows are already the talker's OUTPUT: their next consumer is the final norm)
            RT_TRY(launch_rowsq(ctx, ln.xt, n, H, ln.rowsq_t, NTt, ln.dwt.xT, ln.dwt.xa, m->talker.norm));
        } else {
            // hn <- final-norm rows of each item's last prompt position
            RT_TRY(launch_gather_f32(ctx, hn_all_f32, H, d_last + ln.b0, n, ln.hn_f32, ln.hn));
            RT_TRY(alloc_stack_ws(m, c.talker, n, &ln.wt));
            RT_TRY(alloc_stack_ws(m, c.predictor, n2, &ln.wp));
        }
    }

    // ---- graphs: capture A and B once per lane and launch signature, replay per frame
    use_graph = g_use_graph && !m->prof;
    if (use_graph) {
        uint64_t sig = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
        for (const void* p : {(const void*)d_codes, (const void*)d_eos, (const void*)d_forced, (const void*)A.d_trace_talker,
                              (const void*)A.d_trace_predictor, (const void*)pad_t, (const void*)d_seed})
            mix((uint64_t)(uintptr_t)p);
        for (const Lane& ln : lanes)
            for (const void* p : {(const void*)ln.xt, (const void*)ln.xp, (const void*)ln.logits, (const void*)ln.d_seen, (const void*)ln.d_frame,
                                  (const void*)ln.d_frame_off, (const void*)ln.d_items, (const void*)ln.d_slot_b, (const void*)ln.d_pos_b, (const void*)ln.d_pos_p2,
                                  (const void*)ln.d_zero_pos, (const void*)ln.rowsq_t, (const void*)ln.rowsq_p, (const void*)ln.dwt.xT,
                                  (const void*)ln.dwp.xT, (const void*)ln.dwt.xa, (const void*)ln.dwp.xa, (const void*)ln.dwt.qkv,
                                  (const void*)ln.dwp.qkv, (const void*)ln.dwt.act, (const void*)ln.dwp.act, (const void*)ln.wt.slabs,
                                  (const void*)ln.wp.slabs, (const void*)ln.hn, (const void*)ln.hn_p, (const void*)ln.hn_f32, (const void*)ln.stream}) {
                mix((uint64_t)(uintptr_t)p);
                mix(ln.b0); mix(ln.n);
            }
        mix(B); mix(col); mix(n_lanes); mix(A.ignore_eos); mix(A.min_frames); mix(g_attn_mfma); mix(g_col_split); mix(g_col_split4); mix(g_col_rows64); mix(g_col_rows16); mix(g_col_silu_x); mix(g_fuse_sample_embed);
        // the attention nodes carry the voice prefix (slot, length) by value: a voice of another length must not replay the
        // old graphs.  The prefix KV *content* is read through pointers, so re-setting a voice of the same length keeps them.
        mix((uint64_t)Lp); mix((uint64_t)(int64_t)m->talker.kv.prefix_slot);
        for (const rt_sampling* sp : {&A.talker, &A.predictor}) {
            mix(sp->do_sample); mix(sp->top_k);
            uint32_t f;
            memcpy(&f, &sp->temperature, 4); mix(f);
            memcpy(&f, &sp->top_p, 4); mix(f);
            memcpy(&f, &sp->repetition_penalty, 4); mix(f);
        }
        if (sig != m->graph_sig || (int)m->graphs.size() != 2 * n_lanes) {
            for (auto ex : m->graphs) if (ex) (void)hipGraphExecDestroy(ex);
            m->graphs.clear();
            m->graph_sig = 0;
            RT_HIP(ctx, hipStreamSynchronize(main_stream));
            for (int l = 0; l < n_lanes; ++l) {
                for (int which = 0; which < 2; ++which) {
                    hipGraph_t gr = nullptr;
                    ctx->stream = lanes[l].stream;
                    RT_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
                    const int rc = which == 0 ? enqueue_a(lanes[l]) : enqueue_b(lanes[l]);
                    const hipError_t ce = hipStreamEndCapture(ctx->stream, &gr);
                    ctx->stream = main_stream;
                    if (rc || ce != hipSuccess) {
                        if (gr) (void)hipGraphDestroy(gr);
                        return rc ? rc : rt_fail(ctx, RT_ERR_HIP, "rt_generate: graph capture failed: %s", hipGetErrorString(ce));
                    }
                    hipGraphExec_t ex = nullptr;
                    const hipError_t ie = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(gr);
                    if (ie != hipSuccess) return rt_fail(ctx, RT_ERR_HIP, "rt_generate: graph instantiate failed: %s", hipGetErrorString(ie));
                    m->graphs.push_back(ex);
                }
            }
            m->graph_sig = sig;
        }
    }

    // ---- fork: the lane streams start after everything enqueued so far (prompt prefill, state set-up)
    if (n_lanes > 1) {
        RT_HIP(ctx, hipEventRecord(m->fork_event, main_stream));
        for (Lane& ln : lanes) RT_HIP(ctx, hipStreamWaitEvent(ln.stream, m->fork_event, 0));
    }
    eos_host.assign((size_t)T_max * B, 0);
    produced.assign(N, 0); start.assign(N, 0); item_row.assign(N, -1);   // per item: frames kept, first frame-counter value, row
    finished.assign(N, 0);
    row_item.assign(B, -1);                                // item on each row, -1 = idle
    for (int b = 0; b < B; ++b) { row_item[b] = b; item_row[b] = b; }
    next_item = B;
    // End-of-sequence is decided on the device (the sampler writes one flag per row and frame); the host only needs the
    // flags to know when to STOP launching (or to hand a row to the next queued item), so it fetches them every
    // `g_eos_check_every` frames instead of stalling the launch queue with a copy + wait per frame.  Frames launched past an
    // item's end are wasted work on a finished row (the results are cut at the flag), at most g_eos_check_every - 1 of them.
    eos_every = A.ignore_eos ? 0 : every;
    lane_frames.assign(n_lanes, 0);                        // frame budget of a lane = its longest item (static batches)
    for (int l = 0; l < n_lanes; ++l) {
        for (int bb = lanes[l].b0; bb < lanes[l].b0 + lanes[l].n; ++bb) lane_frames[l] = std::max(lane_frames[l], A.h_max_frames[bb]);
        if (queued) lane_frames[l] = T_max;
    }
    h_pos_b.assign(P.begin(), P.begin() + B);
    h_off.assign(B, 0);
    h_items.assign(A.h_item_ids, A.h_item_ids + B);
    parked.assign(B, 0);
    return RT_OK;
}

// prefill the suffixes of `items` into the KV slots of `rows`; on return x holds the residual stream, d_last / d_dst
// the (last suffix row, decode row) of each item
int rt_gen_run::prefill(const std::vector<int>& items, const std::vector<int>& rows, bool first) {
    const int k = (int)items.size();
    int n = 0;
    for (int it : items) n += A.h_text_offsets[it + 1] - A.h_text_offsets[it] + 2;
    if (n > S_cap) return rt_fail(ctx, RT_ERR_STATE, "rt_generate: prefill of %d rows exceeds its workspace (%d)", n, S_cap);
    staging.emplace_back(new Staging());
    Staging& st = *staging.back();
    std::vector<int32_t>&s_tid = st.tid, &s_cid = st.cid, &s_slot = st.slot, &s_pos = st.pos, &last_row = st.last, &dst_row = st.dst;
    s_tid.assign(n + 1, 0); s_cid.assign((size_t)n * G, -1); s_slot.assign(n, 0); s_pos.assign(n, 0); last_row.assign(k, 0); dst_row.assign(k, 0);
    s_tid[0] = A.tts_pad_id;
    int r = 0;
    for (int i = 0; i < k; ++i) {
        const int it = items[i], o = A.h_text_offsets[it], nt = A.h_text_offsets[it + 1] - o;
        for (int j = 0; j < nt + 2; ++j, ++r) {
            s_tid[1 + r] = j < nt ? A.h_text_ids[o + j] : (j == nt ? A.tts_eos_id : A.tts_pad_id);
            s_cid[(size_t)r * G] = j <= nt ? A.codec_pad_id : A.codec_bos_id;
            s_slot[r] = rows[i];
            s_pos[r] = Lp + j;
        }
        last_row[i] = r - 1;
        dst_row[i] = rows[i];
        P[it] = Lp + nt + 2;
    }
    RT_HIP(ctx, hipMemcpyAsync(d_tid, s_tid.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_cid, s_cid.data(), (size_t)n * G * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_slot, s_slot.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_pos, s_pos.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_last, last_row.data(), k * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_dst, dst_row.data(), k * 4, hipMemcpyHostToDevice, ctx->stream));
    if (first) RT_TRY(text_project(m, d_tid, n + 1, temb, &w_text));  // (the tts_pad row is projected once)
    else RT_TRY(text_project(m, d_tid + 1, n, temb + H, &w_text));
    RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, G, d_cid, n, H, nullptr, temb + H, nullptr, x, nullptr));
    RT_TRY(stack_forward(m, m->talker, w_prefill, x, n, d_slot, d_pos, 0, nullptr, hn_all_f32));
    return RT_OK;
}

// ---- frame part A: group 0 from the talker state, then the residual-code predictor.  Every frame-dependent address
// is base + *d_frame * stride resolved on the device, so the same launches (or one captured graph) serve every frame.
int rt_gen_run::enqueue_a(Lane& ln) {
    const rt_model_config& c = m->cfg;
    const int n = ln.n, n2 = 2 * n;
    int32_t* codes = d_codes + (size_t)ln.b0 * G;
    int ns = 0;
    m->prof_class = 0;
    if (col) { RT_TRY(col_head(m, ln.dwt.xa, ln.rowsq_t, NTt, 0, n, H, c.talker.rms_eps, *head, nullptr, ln.logits)); ns = 1; }
    else RT_TRY(gemm_rows(m, ln.hn, n, *head, ln.logits, &ns));
    SampleArgs sa{};
    sa.logits = ln.logits; sa.n_slabs = ns; sa.M = n; sa.V = Vc;
    sa.do_sample = A.talker.do_sample; sa.temperature = A.talker.temperature; sa.top_k = A.talker.top_k; sa.top_p = A.talker.top_p;
    sa.rep_penalty = A.talker.repetition_penalty; sa.seen = ln.d_seen;
    sa.suppress_from = c.codebook_size; sa.allow_token = -1;
    sa.seed_ptr = d_seed; sa.item_ids = ln.d_items; sa.frame = 0; sa.group = 0;
    sa.forced = d_forced ? d_forced + ln.b0 : nullptr; sa.forced_fs = (int64_t)G * B;
    sa.out = codes; sa.out_stride = G; sa.out_fs = codes_fs; sa.eos_token = c.codec_eos_id; sa.eos_flag = d_eos + ln.b0; sa.eos_fs = B;
    sa.logits_copy = A.d_trace_talker ? A.d_trace_talker + (size_t)ln.b0 * Vc : nullptr; sa.copy_fs = (int64_t)B * Vc;
    sa.frame_ptr = ln.d_frame; sa.frame_off = ln.d_frame_off; sa.eos_live = A.ignore_eos ? 0 : 1; sa.min_frames = A.min_frames;
    RT_TRY(launch_sample(ctx, sa));
    // predictor: rows [0,n) = past hidden (pos 0), rows [n,2n) = embedding of code 0 (pos 1)
    if (m->has_mtp()) {
        if (col) RT_TRY(col_head(m, ln.dwt.xa, ln.rowsq_t, NTt, 0, n, H, c.talker.rms_eps, PW(m, "pred.mtp"), VEC(m, "pred.mtp_b"), ln.xp));
        else {
            RT_TRY(gemm_rows(m, ln.hn, n, PW(m, "pred.mtp"), ln.logits, &ns));
            RT_TRY(launch_reduce_slabs(ctx, ln.logits, ns, n, Hp, VEC(m, "pred.mtp_b"), ACT_NONE, ln.xp, nullptr));
        }
        // (column path: the embedding rows of code 0 are gathered by the rowsq launch below - one launch less per frame)
        if (!col) RT_TRY(launch_gather_f32(ctx, m->proj_c0, Hp, codes, n, ln.xp + (size_t)n * Hp, nullptr, G, ln.d_frame, codes_fs));
    } else {
        // equal-width predictor: its first input row is the talker's normalised hidden state itself
        if (col) RT_TRY(launch_norm_tiled_rows(ctx, ln.dwt.xT, ln.rowsq_t, NTt, m->talker.norm, c.talker.rms_eps, n, H, ln.xp));
        else RT_HIP(ctx, hipMemcpyAsync(ln.xp, ln.hn_f32, (size_t)n * H * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, 1, codes, n, H, nullptr, nullptr, nullptr, ln.xp + (size_t)n * Hp, nullptr, G, ln.d_frame, codes_fs));
    }
    m->prof_class = 1;
    if (col) {
        RowsqGather gt;
        if (m->has_mtp()) { gt.table = m->proj_c0; gt.idx = codes; gt.idx_stride = G; gt.first = n; gt.frame_ptr = ln.d_frame; gt.idx_frame_stride = codes_fs; }
        RT_TRY(launch_rowsq(ctx, ln.xp, n2, Hp, ln.rowsq_p, NTp, ln.dwp.xT, ln.dwp.xa, m->pred.L[0].ln1, nullptr, nullptr, &gt));
        RT_TRY(stack_decode(m, m->pred, ln.dwp, ln.dwp.xT, ln.rowsq_p, n2, ln.d_slot_b, ln.d_pos_p2, 0, false));
    } else {
        RT_TRY(stack_forward(m, m->pred, ln.wp, ln.xp, n2, ln.d_slot_b, ln.d_pos_p2, 0, ln.hn_p, nullptr));
    }
    for (int q = 0; q < G - 1; ++q) {
        const size_t roff = (q == 0) ? (size_t)n : 0;      // the first head reads the rows of position 1
        m->prof_class = 1;
        if (col) {
            RT_TRY(col_head(m, ln.dwp.xa, ln.rowsq_p, NTp, (int)roff, n, Hp, c.predictor.rms_eps,
                            PW(m, "pred.head" + std::to_string(q)), nullptr, ln.logits));
            ns = 1;
        } else {
            RT_TRY(gemm_rows(m, ln.hn_p + roff * Hp, n, PW(m, "pred.head" + std::to_string(q)), ln.logits, &ns));
        }
        SampleArgs sp{};
        sp.logits = ln.logits; sp.n_slabs = ns; sp.M = n; sp.V = Vp;
        sp.do_sample = A.predictor.do_sample; sp.temperature = A.predictor.temperature; sp.top_k = A.predictor.top_k;
        sp.top_p = A.predictor.top_p; sp.rep_penalty = 1.0f; sp.seen = nullptr; sp.suppress_from = Vp; sp.allow_token = -1;
        sp.seed_ptr = d_seed; sp.item_ids = ln.d_items; sp.frame = 0; sp.group = q + 1;
        sp.forced = d_forced ? d_forced + (size_t)(q + 1) * B + ln.b0 : nullptr; sp.forced_fs = (int64_t)G * B;
        sp.out = codes + q + 1; sp.out_stride = G; sp.out_fs = codes_fs; sp.eos_token = -1; sp.eos_flag = nullptr; sp.eos_fs = 0;
        sp.logits_copy = A.d_trace_predictor ? A.d_trace_predictor + ((size_t)q * B + ln.b0) * Vp : nullptr;
        sp.copy_fs = (int64_t)(G - 1) * B * Vp;
        sp.frame_ptr = ln.d_frame; sp.frame_off = ln.d_frame_off; sp.eos_live = 0; sp.min_frames = 0;
        const bool fuse_emb = col && m->has_mtp() && g_fuse_sample_embed && q < G - 2 && Vp <= 4096 && Hp % 8 == 0;
        if (fuse_emb) {     // the sampler itself turns the drawn code into the next pass's input
            sp.emb_table = m->proj_emb[q]; sp.emb_H = Hp; sp.emb_norm_w = m->pred.L[0].ln1; sp.emb_rowsq = ln.rowsq_p; sp.emb_rowsq_n = NTp;
            sp.emb_x_tiled = ln.dwp.xT; sp.emb_a_tiled = ln.dwp.xa;
        }
        RT_TRY(launch_sample(ctx, sp));
        if (q < G - 2) {
            if (fuse_emb) {
            } else if (col && m->has_mtp()) {
                RT_TRY(launch_embed_rowsq(ctx, nullptr, 0, m->proj_emb[q], codes + q + 1, G, ln.d_frame, codes_fs, n, Hp, nullptr, ln.rowsq_p,
                                          NTp, ln.dwp.xT, ln.dwp.xa, m->pred.L[0].ln1));
            } else if (col) {      // equal-width predictor: the group's own bf16 embedding table
                RT_TRY(launch_embed_rowsq(ctx, m->d_frame_srcs + q + 1, 1, nullptr, codes + q + 1, G, ln.d_frame, codes_fs, n, Hp, nullptr,
                                          ln.rowsq_p, NTp, ln.dwp.xT, ln.dwp.xa, m->pred.L[0].ln1));
            } else if (m->has_mtp()) RT_TRY(launch_gather_f32(ctx, m->proj_emb[q], Hp, codes + q + 1, n, ln.xp, nullptr, G, ln.d_frame, codes_fs));
            else RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs + q + 1, 1, codes + q + 1, n, H, nullptr, nullptr, nullptr, ln.xp, nullptr, G, ln.d_frame, codes_fs));
            m->prof_class = 2;
            if (col) {
                RT_TRY(stack_decode(m, m->pred, ln.dwp, ln.dwp.xT, ln.rowsq_p, n, ln.d_slot_b, ln.d_zero_pos, q + 2, true, nullptr, ln.b0, true));
            } else {
                RT_TRY(stack_forward(m, m->pred, ln.wp, ln.xp, n, ln.d_slot_b, ln.d_zero_pos, q + 2, ln.hn_p, nullptr));
            }
        }
    }
    m->prof_class = 0;
    return RT_OK;
}

// ---- frame part B: next talker input (sum of the frame's G code embeddings + projected tts_pad), talker step, frame += 1
int rt_gen_run::enqueue_b(Lane& ln) {
    const int n = ln.n;
    int32_t* codes = d_codes + (size_t)ln.b0 * G;
    // (rt_debug_tune 2701, off by default - see the end of this function) frame += 1 by the last workgroup of the talker-input
    // launch once every workgroup has read the counter (rowops.hip); the talker step behind it then sees frame + 1 and takes its
    // positions with pos_add = -1
    const bool inc_early = col && G <= 16 && g_frame_inc_fold;
    if (col && G <= 16) {
        RT_TRY(launch_embed_rowsq(ctx, m->d_frame_srcs, G, nullptr, codes, G, ln.d_frame, codes_fs, n, H, pad_t, ln.rowsq_t, NTt, ln.dwt.xT,
                                  ln.dwt.xa, m->talker.L[0].ln1, inc_early ? ln.d_frame : nullptr,
                                  inc_early ? reinterpret_cast<unsigned*>(ln.d_frame + 1) : nullptr));
    } else {
        RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, G, codes, n, H, pad_t, nullptr, nullptr, ln.xt, nullptr, G, ln.d_frame, codes_fs));
        if (col) RT_TRY(launch_rowsq(ctx, ln.xt, n, H, ln.rowsq_t, NTt, ln.dwt.xT, ln.dwt.xa, m->talker.L[0].ln1));
    }
    if (col) {
        RT_TRY(stack_decode(m, m->talker, ln.dwt, ln.dwt.xT, ln.rowsq_t, n, ln.d_slot_b, ln.d_pos_b, inc_early ? -1 : 0, true, ln.d_frame));
    } else {
        RT_TRY(stack_forward(m, m->talker, ln.wt, ln.xt, n, ln.d_slot_b, ln.d_pos_b, 0, ln.hn, ln.hn_f32, ln.d_frame));
    }
    // frame += 1 stays a launch of its own: BOTH ways of folding it into a neighbour measured slower, A/B on one box each (round 4,
    // DESIGN.md section 6) - the talker step's last GEMM launch advancing it (204.56 / 203.23 ms per step against 203.98 / 202.65),
    // and the talker-input launch advancing it by its last workgroup (rt_debug_tune 2701: 198.04 / 197.36 against 196.20 / 196.35).
    if (!inc_early)     hipLaunchKernelGGL(k_frame_inc, dim3(1), dim3(64), 0, ctx->stream, ln.d_frame);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

bool rt_gen_run::apply_frames(int upto) {                  // bookkeeping of frames [checked, upto), in order
    for (int tt = checked; tt < upto; ++tt)
        for (int r = 0; r < B; ++r) {
            const int it = row_item[r];
            if (it < 0 || tt < start[it]) continue;
            bool fin = false;
            if (eos_host[(size_t)tt * B + r]) fin = true;
            else if (++produced[it] >= A.h_max_frames[it]) fin = true;
            if (fin) { finished[it] = 1; row_item[r] = -1; ++n_finished; }
        }
    checked = std::max(checked, upto);
    for (Lane& ln : lanes) {
        bool lane_done = true;
        for (int r = ln.b0; r < ln.b0 + ln.n; ++r) lane_done = lane_done && row_item[r] < 0;
        ln.done = lane_done && next_item >= N;
    }
    return n_finished == N;
}

// idle rows take the next queued items; the caller has launched part B of frame t1 - 1, so the new items' first frame is t1
int rt_gen_run::swap_in(int t1) {
    Lane& ln = lanes[0];
    std::vector<int> items, rows;
    bool dirty = false;
    for (int r = 0; r < B; ++r) {
        if (row_item[r] >= 0) continue;
        if (next_item < N) {
            const int it = next_item++;
            items.push_back(it); rows.push_back(r);
            row_item[r] = it; item_row[it] = r; start[it] = t1; parked[r] = 0;
        } else if (!parked[r]) {                       // nothing left for this row: restart its positions so that they stay in range
            parked[r] = 1; h_pos_b[r] = Lp - t1; dirty = true;
        }
    }
    if (!items.empty()) {
        RT_TRY(prefill(items, rows, false));
        const int k = (int)items.size();
        // rows' state <- residual stream of each new item's last prompt position (what the initial hand-over does for all rows)
        RT_TRY(launch_rowsq(ctx, x, k, H, ln.rowsq_t, NTt, ln.dwt.xT, ln.dwt.xa, m->talker.norm, d_last, d_dst));
        for (int i = 0; i < k; ++i) {
            const int r = rows[i], it = items[i];
            h_pos_b[r] = P[it] - t1; h_off[r] = t1; h_items[r] = A.h_item_ids[it];
            RT_HIP(ctx, hipMemsetAsync(ln.d_seen + (size_t)r * Vc, 0, Vc, ctx->stream));
        }
        dirty = true;
        ++n_swaps;
    }
    if (dirty) {
        RT_HIP(ctx, hipMemcpyAsync(ln.d_pos_b, h_pos_b.data(), B * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(ln.d_frame_off, h_off.data(), B * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(ln.d_items, h_items.data(), B * 8, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the three host arrays are rewritten by the next hand-over)
    }
    return RT_OK;
}

// host copies of the end-of-sequence flags of frames [checked, upto) and, with_codes, of the codes of frames [codes_copied, upto):
// enqueued behind part A of frame upto - 1 on every lane, then waited for
int rt_gen_run::fetch(int upto, bool with_codes) {
    if (with_codes && codes_host.size() < (size_t)T_max * B * G) codes_host.resize((size_t)T_max * B * G);
    for (Lane& ln : lanes) {
        if (n_lanes > 1) { RT_HIP(ctx, hipStreamSynchronize(ln.stream)); continue; }
        if (eos_every && upto > checked)
            RT_HIP(ctx, hipMemcpyAsync(eos_host.data() + (size_t)checked * B, d_eos + (size_t)checked * B, (size_t)(upto - checked) * B * 4,
                                       hipMemcpyDeviceToHost, ln.stream));
        if (with_codes && upto > codes_copied)
            RT_HIP(ctx, hipMemcpyAsync(codes_host.data() + (size_t)codes_copied * B * G, d_codes + (size_t)codes_copied * B * G,
                                       (size_t)(upto - codes_copied) * B * G * 4, hipMemcpyDeviceToHost, ln.stream));
        RT_HIP(ctx, hipStreamSynchronize(ln.stream));
    }
    if (n_lanes > 1) {                                     // (every lane has been waited for: plain copies)
        if (eos_every && upto > checked)
            RT_HIP(ctx, hipMemcpy(eos_host.data() + (size_t)checked * B, d_eos + (size_t)checked * B, (size_t)(upto - checked) * B * 4, hipMemcpyDeviceToHost));
        if (with_codes && upto > codes_copied)
            RT_HIP(ctx, hipMemcpy(codes_host.data() + (size_t)codes_copied * B * G, d_codes + (size_t)codes_copied * B * G,
                                  (size_t)(upto - codes_copied) * B * G * 4, hipMemcpyDeviceToHost));
    }
    if (with_codes) codes_copied = std::max(codes_copied, upto);
    return RT_OK;
}

// Run up to n_frames more frames.  On return the host knows every end-of-sequence flag and every code of the frames run so far
// (the last frame of a step is a check point), and part B of the last frame - the next talker step - is already in flight.
int rt_gen_run::advance(int n_frames) {
    struct StreamGuard { rt_ctx* c; hipStream_t s; ~StreamGuard() { c->stream = s; } } stream_guard{ctx, main_stream};
    int done_here = 0;
    while (!all_done && !cancelled && t < T_max && done_here < n_frames) {
        if (A.h_cancel_flag && *A.h_cancel_flag) { cancelled = true; break; }
        for (int l = 0; l < n_lanes; ++l) {
            Lane& ln = lanes[l];
            if (ln.done || t >= lane_frames[l]) continue;
            ctx->stream = ln.stream;
            const auto h0 = std::chrono::steady_clock::now();
            if (use_graph) RT_HIP(ctx, hipGraphLaunch(m->graphs[2 * l], ln.stream));
            else RT_TRY(enqueue_a(ln));
            launch_host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
            if (g_sync_parts) RT_HIP(ctx, hipStreamSynchronize(ln.stream));
        }
        ctx->stream = main_stream;
        frames_run = t + 1;
        const bool last_of_step = done_here + 1 == n_frames;
        const bool check = (t + 1) % every == 0 || t + 1 == T_max || last_of_step;
        if (eos_every == 0 && !last_of_step) {
            all_done = apply_frames(t + 1);                // no flags to wait for: the frame budgets alone decide
        } else if (check) {
            RT_TRY(fetch(t + 1, last_of_step));
            all_done = apply_frames(t + 1);
        }
        ++done_here;
        if (all_done || t + 1 == T_max) { ++t; break; }
        for (int l = 0; l < n_lanes; ++l) {
            Lane& ln = lanes[l];
            if (ln.done || t + 1 >= lane_frames[l]) continue;
            ctx->stream = ln.stream;
            const auto h0 = std::chrono::steady_clock::now();
            if (use_graph) RT_HIP(ctx, hipGraphLaunch(m->graphs[2 * l + 1], ln.stream));
            else RT_TRY(enqueue_b(ln));
            launch_host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
            if (g_sync_parts) RT_HIP(ctx, hipStreamSynchronize(ln.stream));
        }
        ctx->stream = main_stream;
        if (queued && check) RT_TRY(swap_in(t + 1));
        ++t;
    }
    if (cancelled) return rt_fail(ctx, RT_ERR_CANCELLED, "rt_generate: cancelled after %d frames", frames_run);
    // a step ends with the host up to date: every flag applied and every code of the frames run so far copied (the step's last
    // frame was a check point unless the run ended on its frame budget or on its last item first)
    if (checked < frames_run || codes_copied < frames_run) {
        RT_TRY(fetch(frames_run, true));
        all_done = apply_frames(frames_run) || all_done;
    }
    if (t >= T_max) all_done = true;                       // the frame budget is spent: nothing more can run
    return RT_OK;
}

int rt_gen_run::finish(int32_t* h_codes, int32_t* h_n_frames) {
    // ---- join
    if (n_lanes > 1)
        for (int l = 0; l < n_lanes; ++l) {
            RT_HIP(ctx, hipEventRecord(m->lane_events[l], lanes[l].stream));
            RT_HIP(ctx, hipStreamWaitEvent(main_stream, m->lane_events[l], 0));
        }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (checked < frames_run || codes_copied < frames_run) {
        RT_TRY(fetch(frames_run, true));
        (void)apply_frames(frames_run);
    }
    if (n_finished != N) return rt_fail(ctx, RT_ERR_STATE, "rt_generate: %d of %d items unfinished after %d frames", N - n_finished, N, frames_run);
    size_t off = 0;
    int64_t kept = 0;
    for (int it = 0; it < N; ++it) {
        const int n = std::min(produced[it], frames_run - start[it]), r = item_row[it];
        if (h_n_frames) h_n_frames[it] = n;
        kept += n;
        if (h_codes)
            for (int tt = 0; tt < n; ++tt)
                for (int q = 0; q < G; ++q) h_codes[(off + tt) * G + q] = codes_host[((size_t)(start[it] + tt) * B + r) * G + q];
        off += A.h_max_frames[it];
    }
    m->last_frames_run = frames_run; m->last_rows = B; m->last_kept = kept; m->last_swaps = n_swaps; m->last_launch_host_us = launch_host_us;
    if (getenv("RHO_TTS_AMD_TRACE_HOST")) fprintf(stderr, "rt_generate: %d frames, host time inside frame launches %.1f us (%.1f us per frame)\n", frames_run, launch_host_us, launch_host_us / std::max(1, frames_run));
    return RT_OK;
}

extern "C" {

static int gen_begin_locked(rt_model* m, const rt_generate_args* A) {
    rt_ctx* ctx = m->ctx;
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_generate: model not finalized");
    if (m->prefix_len < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: no voice set (rt_model_set_voice)");
    gen_release(m);                                        // (a run that was never ended: dropped)
    pool_release_all(m);
    m->run = new rt_gen_run();
    g_runs_in_flight.fetch_add(1);
    m->run->m = m; m->run->ctx = ctx;
    m->pool_tag = 1;
    const int rc = m->run->init(A);
    m->pool_tag = 0;
    if (rc) { ctx->stream = m->run->main_stream ? m->run->main_stream : ctx->stream; const std::string keep = ctx->last_error; gen_release(m); ctx->last_error = keep; }
    return rc;
}

int rt_generate(rt_model* m, const rt_generate_args* A) {
    if (!m || !A) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_generate: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!A->h_codes || !A->h_n_frames) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate: null array");
    RT_TRY(gen_begin_locked(m, A));
    m->pool_tag = 1;
    int rc = m->run->advance(0x7fffffff);
    if (!rc) rc = m->run->finish(A->h_codes, A->h_n_frames);
    m->pool_tag = 0;
    const std::string keep = ctx->last_error;
    gen_release(m);
    if (rc) ctx->last_error = keep;
    return rc;
}

int rt_generate_begin(rt_model* m, const rt_generate_args* A) {
    if (!m || !A) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_generate_begin: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return gen_begin_locked(m, A);
}

int rt_generate_step(rt_model* m, int32_t n_frames, int32_t* h_frames_run, int32_t* h_all_done) {
    if (!m || n_frames < 1) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_generate_step: bad argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->run) return rt_fail(ctx, RT_ERR_STATE, "rt_generate_step: no generation in flight (rt_generate_begin)");
    m->pool_tag = 1;
    const int rc = m->run->advance(n_frames);
    m->pool_tag = 0;
    if (rc) { const std::string keep = ctx->last_error; gen_release(m); ctx->last_error = keep; return rc; }
    if (h_frames_run) *h_frames_run = m->run->frames_run;
    if (h_all_done) *h_all_done = m->run->all_done ? 1 : 0;
    return RT_OK;
}

int rt_generate_peek(rt_model* m, int32_t item, int32_t first_frame, int32_t max_frames, int32_t* h_codes, int32_t* h_n_frames, int32_t* h_finished) {
    if (!m || !h_n_frames) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_generate_peek: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    rt_gen_run* R = m->run;
    if (!R) return rt_fail(ctx, RT_ERR_STATE, "rt_generate_peek: no generation in flight");
    if (item < 0 || item >= R->N || first_frame < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_generate_peek: item %d / frame %d out of range", item, first_frame);
    const int have = std::min(R->frames_of(item), R->item_row[item] < 0 ? 0 : std::max(0, R->codes_copied - R->start[item]));
    const int n = std::max(0, std::min(have - first_frame, max_frames));
    if (h_codes && n > 0) {
        const int r = R->item_row[item];
        for (int tt = 0; tt < n; ++tt)
            for (int q = 0; q < R->G; ++q)
                h_codes[(size_t)tt * R->G + q] = R->codes_host[((size_t)(R->start[item] + first_frame + tt) * R->B + r) * R->G + q];
    }
    *h_n_frames = n;
    if (h_finished) *h_finished = R->finished[item] ? 1 : 0;
    return RT_OK;
}

int rt_generate_end(rt_model* m, int32_t* h_codes, int32_t* h_n_frames) {
    if (!m) return RT_ERR_INVALID;
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->run) return RT_OK;                              // nothing in flight
    int rc = RT_OK;
    if (m->run->all_done && !m->run->cancelled) rc = m->run->finish(h_codes, h_n_frames);      // (an abandoned run reports nothing)
    else if (h_codes || h_n_frames) rc = rt_fail(ctx, RT_ERR_STATE, "rt_generate_end: the generation was ended before every item finished");
    const std::string keep = ctx->last_error;
    gen_release(m);
    if (rc) ctx->last_error = keep;
    return rc;
}

int rt_generate_stats(rt_model* m, int64_t* frames_run, int64_t* rows, int64_t* frames_kept, int64_t* hand_overs) {
    if (!m) return RT_ERR_INVALID;
    if (frames_run) *frames_run = m->last_frames_run;
    if (rows) *rows = m->last_rows;
    if (frames_kept) *frames_kept = m->last_kept;
    if (hand_overs) *hand_overs = m->last_swaps;
    return RT_OK;
}

}  // extern "C"
