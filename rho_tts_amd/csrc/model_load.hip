// Model group of the C ABI, part 1 of 5: tensor slots, weight binding, finalize, workspace pool, profiling taps.
// (model_stack.hip: transformer stacks; voice.hip: conditioning; generate.hip: autoregressive decode; code2wav.hip: codec decoder.)
// Host-side orchestration only — every FLOP and byte moves in the kernels of gemm.hip, rowops.hip,
// attention.hip and sampling.hip.  Stands behind the third-party model object the reference drives at
// providers/qwen.py:160-165 (load), :247-258 (generate_custom_voice / generate_voice_clone).
#include "model_internal.h"

namespace {

__global__ void k_bf16_to_f32(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = bf16_to_f32(x[i]);
}
__global__ void k_fill_i32(int32_t* p, int n, int v, int step_every, int step) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v + (step_every > 0 ? (i / step_every) * step : 0);
}
}  // namespace

namespace rtm {

int pool_get(rt_model* m, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    int best = -1;
    for (size_t i = 0; i < m->pool.size(); ++i)
        if (!m->pool[i].used && m->pool[i].size >= bytes && (best < 0 || m->pool[i].size < m->pool[best].size)) best = (int)i;
    if (best >= 0 && m->pool[best].size <= bytes * 2 + (1 << 20)) {
        m->pool[best].used = true;
        m->pool[best].tag = m->pool_tag;
        *out = m->pool[best].p;
        return RT_OK;
    }
    void* p = nullptr;
    const size_t sz = (bytes + 255) & ~(size_t)255;
    RT_HIP(m->ctx, hipMalloc(&p, sz));
    m->pool.push_back({p, sz, true, m->pool_tag});
    *out = p;
    return RT_OK;
}
void pool_release_all(rt_model* m) {      // (the blocks of a generation in flight stay: gen_release frees them)
    for (auto& b : m->pool) if (b.tag != 1) b.used = false;
}

int launch_fill_i32(rt_ctx* ctx, int32_t* p, int n, int v, int step_every, int step) {
    hipLaunchKernelGGL(k_fill_i32, dim3(8), dim3(256), 0, ctx->stream, p, n, v, step_every, step);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

Slot* find_slot(rt_model* m, const std::string& name) {
    auto it = m->by_name.find(name);
    return it == m->by_name.end() ? nullptr : &m->slots[it->second];
}
const PackedW& PW(rt_model* m, const std::string& n) { return find_slot(m, n)->pw; }
float* VEC(rt_model* m, const std::string& n) { Slot* s = find_slot(m, n); return s ? s->vec : nullptr; }
bf16_t* TBL(rt_model* m, const std::string& n) { return find_slot(m, n)->tbl; }

}  // namespace rtm

using namespace rtm;

namespace {

void add_slot(rt_model* m, const std::string& name, int kind, int64_t rows, int64_t cols) {
    Slot s;
    s.name = name; s.kind = kind; s.rows = rows; s.cols = cols;
    m->by_name[name] = (int)m->slots.size();
    m->slots.push_back(s);
}

void add_stack_slots(rt_model* m, const char* p, const rt_stack_dims& d, bool qk_norm, bool layer_scale) {
    const int64_t qd = (int64_t)d.heads * d.head_dim, kvd = (int64_t)d.kv_heads * d.head_dim;
    for (int i = 0; i < d.layers; ++i) {
        const std::string b = std::string(p) + ".l" + std::to_string(i);
        add_slot(m, b + ".wqkv", K_GEMM, qd + 2 * kvd, d.hidden);
        add_slot(m, b + ".wo", K_GEMM, d.hidden, qd);
        add_slot(m, b + ".wgu", K_GEMM, 2 * (int64_t)d.inter, d.hidden);
        add_slot(m, b + ".wd", K_GEMM, d.hidden, d.inter);
        add_slot(m, b + ".ln1", K_VEC, d.hidden, 1);
        add_slot(m, b + ".ln2", K_VEC, d.hidden, 1);
        if (qk_norm) { add_slot(m, b + ".qn", K_VEC, d.head_dim, 1); add_slot(m, b + ".kn", K_VEC, d.head_dim, 1); }
        if (layer_scale) { add_slot(m, b + ".ls1", K_VEC, d.hidden, 1); add_slot(m, b + ".ls2", K_VEC, d.hidden, 1); }
    }
    add_slot(m, std::string(p) + ".norm", K_VEC, d.hidden, 1);
}

void declare_slots(rt_model* m) {
    const rt_model_config& c = m->cfg;
    const int H = c.talker.hidden, Hp = c.predictor.hidden, Hc = c.codec_tf.hidden;
    add_slot(m, "talker.text_embedding", K_TABLE, c.text_vocab, c.text_hidden);
    add_slot(m, "talker.tp_fc1", K_GEMM, c.text_hidden, c.text_hidden);
    add_slot(m, "talker.tp_fc1_b", K_VEC, c.text_hidden, 1);
    add_slot(m, "talker.tp_fc2", K_GEMM, H, c.text_hidden);
    add_slot(m, "talker.tp_fc2_b", K_VEC, H, 1);
    add_slot(m, "talker.codec_embedding", K_TABLE, c.codec_vocab, H);
    add_slot(m, "talker.codec_head", K_GEMM, c.codec_vocab, H);
    add_stack_slots(m, "talker", c.talker, true, false);
    if (m->has_mtp()) { add_slot(m, "pred.mtp", K_GEMM, Hp, H); add_slot(m, "pred.mtp_b", K_VEC, Hp, 1); }
    for (int g = 0; g < c.n_groups - 1; ++g) {
        add_slot(m, "pred.emb" + std::to_string(g), K_TABLE, c.predictor_vocab, H);
        add_slot(m, "pred.head" + std::to_string(g), K_GEMM, c.predictor_vocab, Hp);
    }
    add_stack_slots(m, "pred", c.predictor, true, false);
    add_slot(m, "codec.code_embedding", K_TABLE, (int64_t)c.codebook_size * c.num_quantizers, Hc);
    add_stack_slots(m, "ctf", c.codec_tf, false, true);
    for (int i = 0; i < c.n_upsampling; ++i) {
        const std::string u = "codec.up" + std::to_string(i);
        const int r = c.upsampling_ratios[i];
        add_slot(m, u + ".tconv", K_GEMM, (int64_t)r * Hc, Hc);
        add_slot(m, u + ".tconv_b", K_VEC, (int64_t)r * Hc, 1);
        add_slot(m, u + ".dw_w", K_VEC, 7 * (int64_t)Hc, 1);
        add_slot(m, u + ".dw_b", K_VEC, Hc, 1);
        add_slot(m, u + ".ln_w", K_VEC, Hc, 1);
        add_slot(m, u + ".ln_b", K_VEC, Hc, 1);
        add_slot(m, u + ".pw1", K_GEMM, 4 * (int64_t)Hc, Hc);
        add_slot(m, u + ".pw1_b", K_VEC, 4 * (int64_t)Hc, 1);
        add_slot(m, u + ".pw2", K_GEMM, Hc, 4 * (int64_t)Hc);
        add_slot(m, u + ".pw2_b", K_VEC, Hc, 1);
        add_slot(m, u + ".gamma", K_VEC, Hc, 1);
    }
    m->dec_ch.clear();
    for (int i = 0; i <= c.n_upsample_rates; ++i) m->dec_ch.push_back(c.decoder_dim >> i);
    add_slot(m, "codec.dec0", K_GEMM, m->dec_ch[0], 7 * (int64_t)Hc);
    add_slot(m, "codec.dec0_b", K_VEC, m->dec_ch[0], 1);
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        const std::string b = "codec.b" + std::to_string(i);
        const int cin = m->dec_ch[i], cout = m->dec_ch[i + 1], r = c.upsample_rates[i];
        add_slot(m, b + ".sa", K_VEC, cin, 1);
        add_slot(m, b + ".sib", K_VEC, cin, 1);
        add_slot(m, b + ".tconv", K_GEMM, (int64_t)r * cout, 2 * (int64_t)cin);
        add_slot(m, b + ".tconv_b", K_VEC, (int64_t)r * cout, 1);
        for (int j = 0; j < 3; ++j) {
            const std::string u = b + ".u" + std::to_string(j);
            add_slot(m, u + ".a1", K_VEC, cout, 1);
            add_slot(m, u + ".ib1", K_VEC, cout, 1);
            add_slot(m, u + ".c1", K_GEMM, cout, 7 * (int64_t)cout);
            add_slot(m, u + ".c1_b", K_VEC, cout, 1);
            add_slot(m, u + ".a2", K_VEC, cout, 1);
            add_slot(m, u + ".ib2", K_VEC, cout, 1);
            add_slot(m, u + ".c2", K_GEMM, cout, cout);
            add_slot(m, u + ".c2_b", K_VEC, cout, 1);
        }
    }
    const int cl = m->dec_ch.back();
    add_slot(m, "codec.fin_a", K_VEC, cl, 1);
    add_slot(m, "codec.fin_ib", K_VEC, cl, 1);
    add_slot(m, "codec.fin_w", K_GEMM, 1, 7 * (int64_t)cl);   // last conv (C -> 1, k = 7) as a one-column GEMM
    add_slot(m, "codec.fin_wv", K_VEC, 7 * (int64_t)cl, 1);  // ... and as a plain f32 vector for the dedicated last-conv kernel
    add_slot(m, "codec.fin_b", K_VEC, 1, 1);
    // ---- conditioning front-end (optional)
    const rt_encoder_config& e = c.enc;
    if (e.filters > 0) {
        m->enc_ch.clear();
        for (int i = 0; i <= e.n_ratios; ++i) m->enc_ch.push_back(e.filters << i);
        add_slot(m, "enc.conv0_w", K_VEC, (int64_t)e.filters * e.kernel, 1);
        add_slot(m, "enc.conv0_b", K_VEC, e.filters, 1);
        int ci = 1;
        auto conv = [&](int co, int cin, int k) {
            add_slot(m, "enc.c" + std::to_string(ci), K_GEMM, co, (int64_t)k * cin);
            add_slot(m, "enc.c" + std::to_string(ci) + "_b", K_VEC, co, 1);
            ++ci;
        };
        for (int st = 0; st < e.n_ratios; ++st) {
            const int d = m->enc_ch[st];
            conv(d / 2, d, e.res_kernel);
            conv(d, d / 2, 1);
            conv(2 * d, d, 2 * e.ratios[st]);
        }
        conv(e.tf.hidden, m->enc_ch.back(), e.last_kernel);
        add_stack_slots(m, "etf", e.tf, false, true);
        add_slot(m, "enc.down", K_GEMM, e.tf.hidden, 4 * (int64_t)e.tf.hidden);
        add_slot(m, "enc.vq_sem", K_GEMM, e.vq_dim, e.tf.hidden);
        add_slot(m, "enc.vq_aco", K_GEMM, e.vq_dim, e.tf.hidden);
        for (int q = 0; q < c.num_quantizers; ++q) add_slot(m, "enc.cbT" + std::to_string(q), K_VEC, (int64_t)e.vq_dim * c.codebook_size, 1);
        add_slot(m, "enc.spk_fc1", K_VEC, (int64_t)e.spk_hidden * 2 * e.tf.hidden, 1);
        add_slot(m, "enc.spk_fc1_b", K_VEC, e.spk_hidden, 1);
        add_slot(m, "enc.spk_fc2", K_VEC, (int64_t)c.talker.hidden * e.spk_hidden, 1);
        add_slot(m, "enc.spk_fc2_b", K_VEC, c.talker.hidden, 1);
    }
}

int bind_stack(rt_model* m, StackW& S, const char* p, const rt_stack_dims& d, int slots, int max_pos, int window, bool lo_planes = false) {
    S.d = d;
    S.window = window;
    S.L.resize(d.layers);
    for (int i = 0; i < d.layers; ++i) {
        const std::string b = std::string(p) + ".l" + std::to_string(i);
        LayerW& L = S.L[i];
        L.wqkv = PW(m, b + ".wqkv"); L.wo = PW(m, b + ".wo"); L.wgu = PW(m, b + ".wgu"); L.wd = PW(m, b + ".wd");
        L.ln1 = VEC(m, b + ".ln1"); L.ln2 = VEC(m, b + ".ln2");
        L.qn = VEC(m, b + ".qn"); L.kn = VEC(m, b + ".kn");
        L.ls1 = VEC(m, b + ".ls1"); L.ls2 = VEC(m, b + ".ls2");
    }
    S.norm = VEC(m, std::string(p) + ".norm");
    S.kv.layers = d.layers; S.kv.slots = slots; S.kv.kv_heads = d.kv_heads; S.kv.max_pos = max_pos; S.kv.head_dim = d.head_dim;
    const size_t bytes = (size_t)d.layers * S.kv.layer_stride() * sizeof(bf16_t);
    RT_HIP(m->ctx, hipMalloc((void**)&S.kv.k, bytes));
    RT_HIP(m->ctx, hipMalloc((void**)&S.kv.v, bytes));
    RT_HIP(m->ctx, hipMemsetAsync(S.kv.k, 0, bytes, m->ctx->stream));
    RT_HIP(m->ctx, hipMemsetAsync(S.kv.v, 0, bytes, m->ctx->stream));
    if (lo_planes) {
        RT_HIP(m->ctx, hipMalloc((void**)&S.kv.k_lo, bytes));
        RT_HIP(m->ctx, hipMalloc((void**)&S.kv.v_lo, bytes));
        RT_HIP(m->ctx, hipMemsetAsync(S.kv.k_lo, 0, bytes, m->ctx->stream));
        RT_HIP(m->ctx, hipMemsetAsync(S.kv.v_lo, 0, bytes, m->ctx->stream));
    }
    return RT_OK;
}

int expand_vec(rt_model* m, const float* src, int n, int reps, float** out) {
    float* p = nullptr;
    RT_HIP(m->ctx, hipMalloc((void**)&p, (size_t)n * reps * sizeof(float)));
    for (int r = 0; r < reps; ++r)
        RT_HIP(m->ctx, hipMemcpyAsync(p + (size_t)r * n, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, m->ctx->stream));
    m->exp_vecs.push_back(p);
    *out = p;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_model_create(rt_ctx* ctx, const rt_model_config* cfg, rt_model** out_model) {
    if (!ctx || !cfg || !out_model) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_create: null argument");
    *out_model = nullptr;
    const rt_model_config& c = *cfg;
    auto bad_stack = [](const rt_stack_dims& d) {
        return d.hidden < 16 || d.hidden % 16 || d.layers < 1 || d.heads < 1 || d.kv_heads < 1 || d.heads % d.kv_heads ||
               (d.head_dim != 32 && d.head_dim != 64 && d.head_dim != 128) || d.inter % 16 || (d.heads * d.head_dim) % 32 ||
               (d.kv_heads * d.head_dim) % 16;
    };
    if (bad_stack(c.talker) || bad_stack(c.predictor) || bad_stack(c.codec_tf))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_model_create: unsupported stack dimensions (hidden/inter %% 16, head_dim in {32,64,128})");
    if (c.n_groups < 2 || c.n_groups > 32 || c.num_quantizers < 1 || c.num_quantizers > c.n_groups || c.max_batch < 1 || c.max_batch > 64 ||
        c.n_upsampling < 0 || c.n_upsampling > 4 || c.n_upsample_rates < 1 || c.n_upsample_rates > 8 || c.text_hidden % 16 ||
        c.max_positions < 8 || c.max_codec_frames < 1 || (c.decoder_dim >> c.n_upsample_rates) < 8 || (c.decoder_dim >> c.n_upsample_rates) % 8)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_model_create: unsupported configuration (n_groups 2..32, max_batch 1..64, channels %% 8)");
    if (c.enc.filters > 0) {
        const rt_encoder_config& e = c.enc;
        bool bad = e.n_ratios < 1 || e.n_ratios > 8 || e.filters % 16 || 256 % e.filters || e.kernel < 1 || e.kernel > 15 || e.res_kernel < 1 ||
                   e.last_kernel < 1 || bad_stack(e.tf) || e.vq_dim % 8 || e.vq_dim > 4096 || c.codebook_size > 4096 || e.spk_hidden < 1 ||
                   e.max_ref_frames < 1 || e.window < 1;
        for (int i = 0; i < e.n_ratios && !bad; ++i) bad = e.ratios[i] < 1;
        if (bad) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_create: unsupported encoder configuration (filters %% 16, filters | 256, vq_dim %% 8)");
    }
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rt_model* m = new rt_model();
    m->ctx = ctx;
    m->cfg = c;
    declare_slots(m);
    *out_model = m;
    return RT_OK;
}

int rt_model_destroy(rt_model* m) {
    if (!m) return RT_OK;
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    rt_gen_drop(m);
    for (auto& s : m->slots) { if (s.raw) (void)hipFree(s.raw); if (s.raw16) (void)hipFree(s.raw16); }
    if (m->d_cbT) (void)hipFree((void*)m->d_cbT);
    for (StackW* S : {&m->talker, &m->pred, &m->ctf, &m->etf}) {
        if (S->kv.k) (void)hipFree(S->kv.k);
        if (S->kv.v) (void)hipFree(S->kv.v);
        if (S->kv.k_lo) (void)hipFree(S->kv.k_lo);
        if (S->kv.v_lo) (void)hipFree(S->kv.v_lo);
        if (S->kv.vt_prefix) (void)hipFree(S->kv.vt_prefix);
        if (S->kv.kt_prefix) (void)hipFree(S->kv.kt_prefix);
        if (S->cos) (void)hipFree(S->cos);
        if (S->sin) (void)hipFree(S->sin);
    }
    for (auto& b : m->pool) (void)hipFree(b.p);
    for (auto p : m->exp_vecs) (void)hipFree(p);
    for (auto p : m->proj_emb) (void)hipFree(p);
    if (m->proj_c0) (void)hipFree(m->proj_c0);
    if (m->pad_t) (void)hipFree(m->pad_t);
    if (m->d_frame_srcs) (void)hipFree(m->d_frame_srcs);
    for (auto& e : m->prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto ex : m->graphs) if (ex) (void)hipGraphExecDestroy(ex);
    for (auto st : m->lane_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto ev : m->lane_events) (void)hipEventDestroy(ev);
    if (m->fork_event) (void)hipEventDestroy(m->fork_event);
    delete m;
    return RT_OK;
}

int rt_model_tensor_count(rt_model* m) { return m ? (int)m->slots.size() : -1; }

int rt_model_tensor_info(rt_model* m, int32_t index, char* name, size_t name_cap, int64_t* shape2, int32_t* kind) {
    if (!m || index < 0 || index >= (int)m->slots.size()) return RT_ERR_INVALID;
    const Slot& s = m->slots[index];
    if (name && name_cap) snprintf(name, name_cap, "%s", s.name.c_str());
    if (shape2) { shape2[0] = s.rows; shape2[1] = s.cols; }
    if (kind) *kind = s.kind;
    return RT_OK;
}

int rt_model_set_tensor(rt_model* m, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols, int32_t on_device) {
    if (!m || !name || !data) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_model_set_tensor: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    Slot* s = find_slot(m, name);
    if (!s) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_tensor: unknown tensor '%s'", name);
    if (s->rows * s->cols != rows * cols || (s->kind != K_VEC && (s->rows != rows || s->cols != cols)))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_tensor: '%s' expects [%lld, %lld], got [%lld, %lld]", name, (long long)s->rows,
                       (long long)s->cols, (long long)rows, (long long)cols);
    if (s->kind != K_VEC && dtype != RT_DTYPE_BF16) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_tensor: '%s' must be bf16", name);
    const int64_t n = rows * cols;
    const size_t esz = dtype == RT_DTYPE_BF16 ? 2 : 4;
    const void* d_src = data;
    if (!on_device) {
        void* stage = nullptr;
        RT_TRY(rt_ctx_scratch(ctx, (size_t)n * esz, &stage));
        RT_HIP(ctx, hipMemcpyAsync(stage, data, (size_t)n * esz, hipMemcpyHostToDevice, ctx->stream));
        d_src = stage;
    }
    if (s->raw) { RT_HIP(ctx, hipStreamSynchronize(ctx->stream)); RT_HIP(ctx, hipFree(s->raw)); s->raw = nullptr; }
    if (s->kind == K_GEMM) {
        const size_t pb = packed_bytes((int)rows, (int)cols);
        RT_HIP(ctx, hipMalloc(&s->raw, pb));
        RT_TRY(launch_pack_weight(ctx, (const bf16_t*)d_src, (int)rows, (int)cols, (bf16_t*)s->raw, &s->pw));
        m->weight_bytes += (int64_t)pb;
        // weights the decode step streams get a second copy tiled for the 16-column GEMM (talker / predictor layers, heads, mtp)
        const std::string nm(name);
        const bool decode_w = nm.rfind("talker.l", 0) == 0 || nm.rfind("pred.l", 0) == 0 || nm.rfind("pred.head", 0) == 0 ||
                              nm == "pred.mtp" || nm == "talker.codec_head";
        if (decode_w && cols % 32 == 0) {
            if (s->raw16) { RT_HIP(ctx, hipFree(s->raw16)); s->raw16 = nullptr; }
            RT_HIP(ctx, hipMalloc(&s->raw16, packed16_bytes((int)rows, (int)cols)));
            RT_TRY(launch_pack_weight16(ctx, (const bf16_t*)d_src, (int)rows, (int)cols, (bf16_t*)s->raw16, &s->pw));
        }
    } else if (s->kind == K_TABLE) {
        RT_HIP(ctx, hipMalloc(&s->raw, (size_t)n * 2));
        RT_HIP(ctx, hipMemcpyAsync(s->raw, d_src, (size_t)n * 2, hipMemcpyDeviceToDevice, ctx->stream));
        s->tbl = (bf16_t*)s->raw;
        m->weight_bytes += n * 2;
    } else {
        RT_HIP(ctx, hipMalloc(&s->raw, (size_t)n * 4));
        s->vec = (float*)s->raw;
        if (dtype == RT_DTYPE_F32) RT_HIP(ctx, hipMemcpyAsync(s->raw, d_src, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        else hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream,
                                (const bf16_t*)d_src, n, s->vec);
        RT_HIP(ctx, hipGetLastError());
    }
    if (!on_device) RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next call
    s->set = true;
    return RT_OK;
}

int rt_model_finalize(rt_model* m, const float* h_rope_cos[3], const float* h_rope_sin[3]) {
    if (!m || !h_rope_cos || !h_rope_sin) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_model_finalize: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_model_finalize: already finalized");
    for (auto& s : m->slots)
        if (!s.set) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_finalize: tensor '%s' was never set", s.name.c_str());
    const rt_model_config& c = m->cfg;
    RT_TRY(bind_stack(m, m->talker, "talker", c.talker, c.max_batch + 1, c.max_positions, 0));
    if (c.talker.head_dim == 128) {   // transposed copy of the voice prefix's V for the matrix-core decode attention (attention_mfma.hip)
        KvCache& kv = m->talker.kv;
        kv.vt_stride = (c.max_positions + 31) / 32 * 4096;
        kv.prefix_slot_alloc = m->prefix_slot();
        RT_HIP(ctx, hipMalloc((void**)&kv.kt_prefix, (size_t)kv.layers * kv.kv_heads * kv.vt_stride * sizeof(bf16_t)));
        RT_HIP(ctx, hipMalloc((void**)&kv.vt_prefix, (size_t)kv.layers * kv.kv_heads * kv.vt_stride * sizeof(bf16_t)));
    }
    RT_TRY(bind_stack(m, m->pred, "pred", c.predictor, c.max_batch, c.n_groups + 1, 0));
    RT_TRY(bind_stack(m, m->ctf, "ctf", c.codec_tf, c.max_batch, c.max_codec_frames, c.codec_sliding_window, true));
    StackW* stacks[3] = {&m->talker, &m->pred, &m->ctf};
    for (int i = 0; i < 3; ++i) {
        StackW& S = *stacks[i];
        const size_t n = (size_t)S.kv.max_pos * (S.d.head_dim / 2);
        RT_HIP(ctx, hipMalloc((void**)&S.cos, n * 4));
        RT_HIP(ctx, hipMalloc((void**)&S.sin, n * 4));
        RT_HIP(ctx, hipMemcpy(S.cos, h_rope_cos[i], n * 4, hipMemcpyHostToDevice));
        RT_HIP(ctx, hipMemcpy(S.sin, h_rope_sin[i], n * 4, hipMemcpyHostToDevice));
    }
    if (c.enc.filters > 0) {
        // the encoder's transformer runs at twice the frame rate; its RoPE table is computed here, in float32 and in the order
        // the host-side tables use (inv = 1 / theta^(2i/d); angle = pos * inv)
        const rt_encoder_config& e = c.enc;
        RT_TRY(bind_stack(m, m->etf, "etf", e.tf, 1, 2 * e.max_ref_frames, e.window, true));
        const int half = e.tf.head_dim / 2, npos = 2 * e.max_ref_frames;
        std::vector<float> hc((size_t)npos * half), hs((size_t)npos * half);
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / powf(e.tf.rope_theta, (float)(2 * i) / (float)e.tf.head_dim);
            for (int p = 0; p < npos; ++p) { const float a = (float)p * inv; hc[(size_t)p * half + i] = cosf(a); hs[(size_t)p * half + i] = sinf(a); }
        }
        RT_HIP(ctx, hipMalloc((void**)&m->etf.cos, hc.size() * 4));
        RT_HIP(ctx, hipMalloc((void**)&m->etf.sin, hs.size() * 4));
        RT_HIP(ctx, hipMemcpy(m->etf.cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        RT_HIP(ctx, hipMemcpy(m->etf.sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        std::vector<const float*> cbs(c.num_quantizers);
        for (int q = 0; q < c.num_quantizers; ++q) cbs[q] = VEC(m, "enc.cbT" + std::to_string(q));
        RT_HIP(ctx, hipMalloc((void**)&m->d_cbT, sizeof(float*) * c.num_quantizers));
        RT_HIP(ctx, hipMemcpy((void*)m->d_cbT, cbs.data(), sizeof(float*) * c.num_quantizers, hipMemcpyHostToDevice));
    }
    // frame-embedding sources: group 0 = talker codec table, group g = predictor table g-1
    std::vector<GatherSrc> srcs(c.n_groups);
    srcs[0] = {TBL(m, "talker.codec_embedding"), c.talker.hidden};
    for (int gq = 1; gq < c.n_groups; ++gq) srcs[gq] = {TBL(m, "pred.emb" + std::to_string(gq - 1)), c.talker.hidden};
    RT_HIP(ctx, hipMalloc((void**)&m->d_frame_srcs, sizeof(GatherSrc) * c.n_groups));
    RT_HIP(ctx, hipMemcpy(m->d_frame_srcs, srcs.data(), sizeof(GatherSrc) * c.n_groups, hipMemcpyHostToDevice));
    // projected predictor input tables: mtp(emb) for every code, so the per-frame loop is gathers only
    if (m->has_mtp()) {
        const int Hp = c.predictor.hidden, H = c.talker.hidden;
        auto project = [&](const bf16_t* tbl, int rows, float** out) -> int {
            RT_HIP(ctx, hipMalloc((void**)out, (size_t)rows * Hp * 4));
            GemmA a; a.ptr = tbl; a.M = rows; a.Cin = H;
            GemmEpi e; e.bias = VEC(m, "pred.mtp_b"); e.out_f32 = *out; e.ldc = Hp;
            return launch_gemm(ctx, a, PW(m, "pred.mtp"), e);
        };
        RT_TRY(project(TBL(m, "talker.codec_embedding"), c.codec_vocab, &m->proj_c0));
        m->proj_emb.resize(c.n_groups - 1, nullptr);
        for (int gq = 0; gq < c.n_groups - 1; ++gq) RT_TRY(project(TBL(m, "pred.emb" + std::to_string(gq)), c.predictor_vocab, &m->proj_emb[gq]));
    }
    // SnakeBeta parameters of each block's first residual unit, tiled over the r output phases of the transposed conv
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        const std::string bn = "codec.b" + std::to_string(i);
        for (const char* v : {".u0.a1", ".u0.ib1"}) {
            float* x = nullptr;
            RT_TRY(expand_vec(m, VEC(m, bn + v), m->dec_ch[i + 1], c.upsample_rates[i], &x));
            m->xvec[bn + v] = x;
        }
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->finalized = true;
    return RT_OK;
}

int64_t rt_model_weight_bytes(rt_model* m) { return m ? m->weight_bytes : -1; }

int rt_profile_enable(rt_model* m, int32_t on) {
    if (!m) return RT_ERR_INVALID;
    CtxLock g(m->ctx);
    m->prof = on != 0;
    m->prof_used = 0;
    m->prof_bytes = 0;
    return RT_OK;
}

int rt_profile_read(rt_model* m, int64_t* n_launches, double* total_ms, double* total_bytes) {
    if (!m) return RT_ERR_INVALID;
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0;
    for (size_t i = 0; i < m->prof_used; ++i) {
        float t = 0;
        if (hipEventElapsedTime(&t, m->prof_ev[i].first, m->prof_ev[i].second) == hipSuccess) ms += t;
    }
    if (n_launches) *n_launches = (int64_t)m->prof_used;
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = m->prof_bytes;
    return RT_OK;
}

int rt_profile_read_class(rt_model* m, int32_t cls, int64_t* n_launches, double* total_ms, double* total_bytes) {
    if (!m) return RT_ERR_INVALID;
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0, by = 0;
    int64_t n = 0;
    for (size_t i = 0; i < m->prof_used && i < m->prof_tag.size(); ++i) {
        if (m->prof_tag[i].first != cls) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, m->prof_ev[i].first, m->prof_ev[i].second) == hipSuccess) ms += t;
        by += m->prof_tag[i].second;
        ++n;
    }
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = by;
    return RT_OK;
}

}  // extern "C"
