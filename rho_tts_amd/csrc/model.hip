// Model group of the C ABI: weights, voice prefix, autoregressive decode, codec decoder.
// Host-side orchestration only — every FLOP and byte moves in the kernels of gemm.hip, rowops.hip,
// attention.hip and sampling.hip.  Stands behind the third-party model object the reference drives at
// providers/qwen.py:160-165 (load), :247-258 (generate_custom_voice / generate_voice_clone).
#include <algorithm>
#include <chrono>
#include <map>
#include <memory>

#include "kernels.h"

namespace {

enum SlotKind { K_GEMM = 0, K_TABLE = 1, K_VEC = 2 };

struct Slot {
    std::string name;
    int kind = 0;
    int64_t rows = 0, cols = 0;
    bool set = false;
    PackedW pw;
    bf16_t* tbl = nullptr;
    float* vec = nullptr;
    void* raw = nullptr;  // owning pointer
    void* raw16 = nullptr;  // owning pointer of the 16-column decode copy
};

struct LayerW {
    PackedW wqkv, wo, wgu, wd;
    float *ln1 = nullptr, *ln2 = nullptr, *qn = nullptr, *kn = nullptr, *ls1 = nullptr, *ls2 = nullptr;
};

struct StackW {
    rt_stack_dims d{};
    std::vector<LayerW> L;
    float* norm = nullptr;
    int window = 0;
    KvCache kv;
    float *cos = nullptr, *sin = nullptr;
    int q_dim() const { return d.heads * d.head_dim; }
    int kv_dim() const { return d.kv_heads * d.head_dim; }
};

struct PoolBlock { void* p; size_t size; bool used; int tag; };   // tag 1: owned by the generation in flight (rt_gen_run)

__global__ void k_bf16_to_f32(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = bf16_to_f32(x[i]);
}
// blob [2][layers][kv_heads][prefix_len][d] <-> cache slot
__global__ void k_kv_blob(bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int64_t layer_stride, int layers, int kv_heads, int max_pos,
                          int d, int slot, int prefix_len, bf16_t* __restrict__ blob, int to_blob) {
    const int lh = blockIdx.x, which = blockIdx.y;
    const int layer = lh / kv_heads, kh = lh % kv_heads;
    bf16_t* c = (which ? vc : kc) + layer * layer_stride + ((int64_t)slot * kv_heads + kh) * max_pos * d;
    bf16_t* b = blob + (((int64_t)which * layers + layer) * kv_heads + kh) * prefix_len * d;
    const int64_t n16 = (int64_t)prefix_len * d / 8;
    for (int64_t i = threadIdx.x; i < n16; i += blockDim.x) {
        if (to_blob) reinterpret_cast<uint4*>(b)[i] = reinterpret_cast<const uint4*>(c)[i];
        else reinterpret_cast<uint4*>(c)[i] = reinterpret_cast<const uint4*>(b)[i];
    }
}
__global__ void k_add_vec(float* __restrict__ dst, const float* __restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
__global__ void k_frame_inc(int32_t* f) { if (threadIdx.x == 0) *f += 1; }
__global__ void k_fill_i32(int32_t* p, int n, int v, int step_every, int step) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v + (step_every > 0 ? (i / step_every) * step : 0);
}

}  // namespace

struct rt_gen_run;
void rt_gen_drop(struct rt_model* m);    // ends a generation in flight (defined with rt_gen_run)

struct rt_model {
    rt_gen_run* run = nullptr;         // generation in flight (rt_generate_begin .. rt_generate_end), else null
    int pool_tag = 0;                  // tag of the pool blocks handed out right now (1 while the generation in flight allocates)
    bool prefix_tiles_valid = false;   // kt_prefix / vt_prefix hold the current voice prefix (attention_mfma.hip)
    int64_t last_frames_run = 0, last_rows = 0, last_kept = 0, last_swaps = 0;   // rt_generate_stats
    double last_launch_host_us = 0.0;  // host time spent inside the frame-part launches of the last rt_generate
    rt_ctx* ctx = nullptr;
    rt_model_config cfg{};
    std::vector<Slot> slots;
    std::map<std::string, int> by_name;
    bool finalized = false;
    StackW talker, pred, ctf, etf;     // etf: the audio encoder's transformer (conditioning front-end)
    std::vector<int> enc_ch;           // encoder channel ladder
    const float** d_cbT = nullptr;     // device array of the transposed codebooks
    std::vector<int> dec_ch;  // decoder channel ladder
    // derived tables
    float* pad_t = nullptr;   // text_proj(tts_pad)  [H]  (computed on first use)
    int pad_t_id = -1;
    float* proj_c0 = nullptr;              // [codec_vocab][Hp] f32 (mtp only)
    std::vector<float*> proj_emb;          // [G-1] x [Vp][Hp] f32 (mtp only)
    GatherSrc* d_frame_srcs = nullptr;     // n_groups sources for frame embedding
    std::vector<float*> exp_vecs;          // expanded per-column vectors (owned)
    std::map<std::string, float*> xvec;    // name -> expanded vector (SnakeBeta parameters tiled over a transposed conv's r phases)
    // voice
    int prefix_len = 0;
    // pool
    std::vector<PoolBlock> pool;
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    size_t prof_used = 0;
    double prof_bytes = 0;
    // decode-frame graphs (A: LM head + sample + residual-code predictor, B: next input + talker step), reused while the
    // launch signature (every pointer and parameter baked into the nodes) stays the same
    uint64_t graph_sig = 0;
    std::vector<hipGraphExec_t> graphs;            // [lane][A, B]
    std::vector<hipStream_t> lane_streams;         // decode lanes (created on first use)
    std::vector<hipEvent_t> lane_events;
    hipEvent_t fork_event = nullptr;
    int64_t weight_bytes = 0;

    bool has_mtp() const { return cfg.talker.hidden != cfg.predictor.hidden; }
    int prefix_slot() const { return cfg.max_batch; }
};

namespace {

#define RT_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

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
template <typename T>
int pool_arr(rt_model* m, size_t n, T** out) {
    void* p = nullptr;
    RT_TRY(pool_get(m, n * sizeof(T), &p));
    *out = (T*)p;
    return RT_OK;
}

void add_slot(rt_model* m, const std::string& name, int kind, int64_t rows, int64_t cols) {
    Slot s;
    s.name = name; s.kind = kind; s.rows = rows; s.cols = cols;
    m->by_name[name] = (int)m->slots.size();
    m->slots.push_back(s);
}
Slot* find_slot(rt_model* m, const std::string& name) {
    auto it = m->by_name.find(name);
    return it == m->by_name.end() ? nullptr : &m->slots[it->second];
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

const PackedW& PW(rt_model* m, const std::string& n) { return find_slot(m, n)->pw; }
float* VEC(rt_model* m, const std::string& n) { Slot* s = find_slot(m, n); return s ? s->vec : nullptr; }
bf16_t* TBL(rt_model* m, const std::string& n) { return find_slot(m, n)->tbl; }

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

struct StackWs {
    float* xn32 = nullptr;    // precise stacks: float32 operands [M][H], [M][q_dim], [M][I]
    float* ao32 = nullptr;
    float* act32 = nullptr;
    bf16_t* xn = nullptr;     // [M][H]
    float* slabs = nullptr;   // max over GEMMs
    float* q = nullptr;       // [M][q_dim]
    bf16_t* ao = nullptr;     // [M][q_dim]
    bf16_t* act = nullptr;    // [M][I]
};
size_t slab_floats(const rt_stack_dims& d, int M, int n_cu = 256) {
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
int alloc_stack_ws(rt_model* m, const rt_stack_dims& d, int M, StackWs* w, bool precise = false) {
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
                  bf16_t* out_bf16, float* out_f32, const int32_t* frame_ptr = nullptr, bool prefix_rows = false) {
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
struct DecWs {
    float* qkv = nullptr;   // [M][(heads + 2 kv) * d]
    bf16_t* ao = nullptr;   // [M][q_dim]
    bf16_t* act = nullptr;  // [M][inter]
    float* q = nullptr;     // [M][q_dim] (only for passes with several rows per slot)
    float* xT = nullptr;    // fragment-tiled residual stream [ceil(M/32)*32][H]
    bf16_t* xa = nullptr;   // fragment-tiled bf16(norm_w .* x): operand of the GEMM behind the next RMSNorm
};
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
int col_gemm(rt_model* m, const ColArgs& a0, const PackedW& W, bool is_predictor = false) {
    const int blk = g_col_rows64 ? 64 : 32;
    for (int r0 = 0; r0 < a0.M; r0 += blk) {
        ColArgs a = a0;
        a.nt = is_predictor ? g_pred_nt : 1;
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
                 int pos_add, bool one_row_per_slot = true, const int32_t* frame_ptr = nullptr, int slot_base = -1, bool zero_pos = false) {
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
struct TextWs { bf16_t *e = nullptr, *h1 = nullptr; GatherSrc* d_src = nullptr; };   // optional caller-owned workspace (repeated calls)
int alloc_text_ws(rt_model* m, int n, TextWs* w) {
    RT_TRY(pool_arr(m, (size_t)n * m->cfg.text_hidden, &w->e));
    RT_TRY(pool_arr(m, (size_t)n * m->cfg.text_hidden, &w->h1));
    RT_TRY(pool_arr(m, 1, &w->d_src));
    const GatherSrc src{TBL(m, "talker.text_embedding"), m->cfg.text_hidden};
    RT_HIP(m->ctx, hipMemcpy(w->d_src, &src, sizeof(src), hipMemcpyHostToDevice));
    return RT_OK;
}
int text_project(rt_model* m, const int32_t* d_ids, int n, float* out, const TextWs* ws = nullptr) {
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(m->ctx->mu);
    m->prof = on != 0;
    m->prof_used = 0;
    m->prof_bytes = 0;
    return RT_OK;
}

int rt_profile_read(rt_model* m, int64_t* n_launches, double* total_ms, double* total_bytes) {
    if (!m) return RT_ERR_INVALID;
    rt_ctx* ctx = m->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
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

// ------------------------------------------------------------------------------------------ voice
// (the context mutex is held by the caller)
static int set_voice_impl(rt_model* m, int32_t n_rows, const int32_t* h_text_ids, const int32_t* h_codec_ids, int32_t h_speaker_row,
                          const float* h_speaker_embed) {
    rt_ctx* ctx = m->ctx;
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_model_set_voice: model not finalized");
    if (m->run) return rt_fail(ctx, RT_ERR_STATE, "rt_model_set_voice: a generation is in flight (rt_generate_end first)");
    const rt_model_config& c = m->cfg;
    if (n_rows + 8 > c.max_positions) return rt_fail(ctx, RT_ERR_LENGTH, "voice prefix length %d exceeds max_positions %d", n_rows, c.max_positions);
    if (h_speaker_row >= n_rows || (h_speaker_row >= 0 && !h_speaker_embed)) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_voice: bad speaker row");
    const int H = c.talker.hidden, G = c.n_groups;
    for (int r = 0; r < n_rows; ++r) {
        if (h_text_ids[r] < 0 || h_text_ids[r] >= c.text_vocab) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_voice: text id %d out of range", h_text_ids[r]);
        for (int q = 0; q < G; ++q) {
            const int id = h_codec_ids[r * G + q];
            if (id >= (q == 0 ? c.codec_vocab : c.predictor_vocab)) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_voice: codec id %d out of range", id);
        }
    }
    m->talker.kv.prefix_slot = -1;    // the prefix slot attends to itself while it is being computed
    pool_release_all(m);
    // text side: project every row that has a text id (rows without one get -1 -> no text term... they get tts_pad by contract)
    int32_t *d_tid = nullptr, *d_cid = nullptr, *d_slot = nullptr, *d_pos = nullptr;
    RT_TRY(pool_arr(m, n_rows, &d_tid));
    RT_TRY(pool_arr(m, (size_t)n_rows * G, &d_cid));
    RT_TRY(pool_arr(m, n_rows, &d_slot));
    RT_TRY(pool_arr(m, n_rows, &d_pos));
    std::vector<int32_t> tid(h_text_ids, h_text_ids + n_rows);
    RT_HIP(ctx, hipMemcpyAsync(d_tid, tid.data(), n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(d_cid, h_codec_ids, (size_t)n_rows * G * 4, hipMemcpyHostToDevice, ctx->stream));
    float* temb = nullptr;  // [n_rows + 1][H]: projected text rows, then the speaker vector
    RT_TRY(pool_arr(m, (size_t)(n_rows + 1) * H, &temb));
    RT_TRY(text_project(m, d_tid, n_rows, temb));
    float* x = nullptr;
    RT_TRY(pool_arr(m, (size_t)n_rows * H, &x));
    RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, G, d_cid, n_rows, H, nullptr, temb, nullptr, x, nullptr));
    if (h_speaker_row >= 0) {
        float* spk = temb + (size_t)n_rows * H;
        RT_HIP(ctx, hipMemcpyAsync(spk, h_speaker_embed, H * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_add_vec, dim3((H + 255) / 256), dim3(256), 0, ctx->stream, x + (size_t)h_speaker_row * H, spk, H);
        RT_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(k_fill_i32, dim3(8), dim3(256), 0, ctx->stream, d_slot, n_rows, m->prefix_slot(), 0, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3(8), dim3(256), 0, ctx->stream, d_pos, n_rows, 0, 1, 1);
    RT_HIP(ctx, hipGetLastError());
    StackWs w;
    RT_TRY(alloc_stack_ws(m, c.talker, n_rows, &w));
    bf16_t* hn = nullptr;
    RT_TRY(pool_arr(m, (size_t)n_rows * H, &hn));
    m->talker.kv.tiles_len = -1;
    RT_TRY(stack_forward(m, m->talker, w, x, n_rows, d_slot, d_pos, 0, hn, nullptr, nullptr, true));
    // fragment-tiled copies of the prefix K / V for the matrix-core attention of the prompt prefills (and of the decode step when
    // that form is switched on): once per voice - made layer by layer inside the prefill above where its attention used them
    m->prefix_tiles_valid = m->talker.kv.tiles_len == n_rows;
    if (!m->prefix_tiles_valid && c.talker.head_dim == 128 && m->talker.kv.kt_prefix) {
        RT_TRY(launch_transpose_prefix_v(ctx, m->talker.kv, n_rows));
        m->prefix_tiles_valid = true;
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->prefix_len = n_rows;
    pool_release_all(m);
    return RT_OK;
}

int rt_model_set_voice(rt_model* m, int32_t n_rows, const int32_t* h_text_ids, const int32_t* h_codec_ids, int32_t h_speaker_row,
                       const float* h_speaker_embed) {
    if (!m || n_rows < 1 || !h_text_ids || !h_codec_ids) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_model_set_voice: null argument");
    std::lock_guard<std::mutex> g(m->ctx->mu);
    RT_HIP(m->ctx, hipSetDevice(m->ctx->device));
    return set_voice_impl(m, n_rows, h_text_ids, h_codec_ids, h_speaker_row, h_speaker_embed);
}

// ---- conditioning front-end: reference audio -> codes [frames][num_quantizers] + speaker embedding (mutex held by the caller).
// Channels-last activations; every convolution is an implicit GEMM in split precision (float32 activations fed as hi + lo bf16
// planes): a k = 2r, stride r conv is the 2-tap GEMM over the clip viewed as [T / r][r * C] rows (causal: taps at t - 1 and t).
static int voice_encode_impl(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t* h_codes, int32_t max_frames, int32_t* h_n_frames,
                             float* h_speaker_embed) {
    rt_ctx* ctx = m->ctx;
    const rt_model_config& c = m->cfg;
    const rt_encoder_config& e = c.enc;
    if (e.filters <= 0) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "this model was created without an audio encoder (rt_model_config.enc)");
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_voice_encode: model not finalized");
    int64_t hop = 2;
    for (int i = 0; i < e.n_ratios; ++i) hop *= e.ratios[i];
    const int64_t n_frames = std::min<int64_t>(std::min<int64_t>(n_samples / hop, max_frames), e.max_ref_frames);
    if (n_frames < 1) return rt_fail(ctx, RT_ERR_INVALID, "reference audio is shorter than one codec frame (%lld samples per frame)", (long long)hop);
    const int64_t T = n_frames * hop;
    pool_release_all(m);
    struct Planes { bf16_t* hi = nullptr; bf16_t* lo = nullptr; };
    auto planes = [&](size_t n, Planes* p) -> int {
        RT_TRY(pool_arr(m, n, &p->hi));
        RT_TRY(pool_arr(m, n, &p->lo));
        return RT_OK;
    };
    float* pcm = nullptr;
    RT_TRY(pool_arr(m, (size_t)T, &pcm));
    RT_HIP(ctx, hipMemcpyAsync(pcm, h_pcm, (size_t)T * 4, hipMemcpyHostToDevice, ctx->stream));
    // ---- conv encoder
    int64_t Tc = T;
    float* x = nullptr;
    Planes pa;
    RT_TRY(pool_arr(m, (size_t)Tc * m->enc_ch[0], &x));
    RT_TRY(planes((size_t)Tc * m->enc_ch[0], &pa));
    RT_TRY(launch_enc_conv0(ctx, pcm, Tc, m->enc_ch[0], e.kernel, VEC(m, "enc.conv0_w"), VEC(m, "enc.conv0_b"), x, pa.hi, pa.lo));
    int ci = 1;
    auto W = [&](int i) -> const PackedW& { return PW(m, "enc.c" + std::to_string(i)); };
    auto Bv = [&](int i) { return VEC(m, "enc.c" + std::to_string(i) + "_b"); };
    for (int st = 0; st < e.n_ratios; ++st) {
        const int d = m->enc_ch[st], r = e.ratios[st];
        Planes pb, pn;
        RT_TRY(planes((size_t)Tc * (d / 2), &pb));
        {   // residual branch: ELU -> conv k (dilation 1) -> ELU
            GemmA a; a.ptr = pa.hi; a.ptr_lo = pa.lo; a.split = 1; a.M = Tc; a.Cin = d; a.taps = e.res_kernel; a.tap_stride = 1; a.tap_offset = -(e.res_kernel - 1);
            a.rows_out = (int)Tc; a.rows_in = (int)Tc;
            GemmEpi ep; ep.bias = Bv(ci); ep.act = ACT_ELU; ep.out_hi = pb.hi; ep.out_lo = pb.lo; ep.ldc = d / 2;
            RT_TRY(launch_gemm(ctx, a, W(ci), ep));
        }
        {   // -> conv k1, + skip; ELU of the sum is the strided conv's operand
            GemmA a; a.ptr = pb.hi; a.ptr_lo = pb.lo; a.split = 1; a.M = Tc; a.Cin = d / 2; a.taps = 1;
            GemmEpi ep; ep.bias = Bv(ci + 1); ep.residual = x; ep.out_f32 = x; ep.out2_hi = pa.hi; ep.out2_lo = pa.lo; ep.act2 = ACT_ELU; ep.ldc = d;
            RT_TRY(launch_gemm(ctx, a, W(ci + 1), ep));
        }
        const int64_t To = Tc / r;
        float* xn = nullptr;
        RT_TRY(pool_arr(m, (size_t)To * 2 * d, &xn));
        RT_TRY(planes((size_t)To * 2 * d, &pn));
        {   // down-sampling conv k = 2r, stride r
            GemmA a; a.ptr = pa.hi; a.ptr_lo = pa.lo; a.split = 1; a.M = To; a.Cin = r * d; a.taps = 2; a.tap_stride = 1; a.tap_offset = -1;
            a.rows_out = (int)To; a.rows_in = (int)To;
            GemmEpi ep; ep.bias = Bv(ci + 2); ep.out_f32 = xn; ep.out2_hi = pn.hi; ep.out2_lo = pn.lo; ep.act2 = ACT_ELU; ep.ldc = 2 * d;
            RT_TRY(launch_gemm(ctx, a, W(ci + 2), ep));
        }
        x = xn; pa = pn; Tc = To; ci += 3;
    }
    const int He = e.tf.hidden;
    float* feats = nullptr;                        // [Te][He]: conv features at twice the frame rate
    RT_TRY(pool_arr(m, (size_t)Tc * He, &feats));
    {
        GemmA a; a.ptr = pa.hi; a.ptr_lo = pa.lo; a.split = 1; a.M = Tc; a.Cin = m->enc_ch.back(); a.taps = e.last_kernel; a.tap_stride = 1;
        a.tap_offset = -(e.last_kernel - 1); a.rows_out = (int)Tc; a.rows_in = (int)Tc;
        GemmEpi ep; ep.bias = Bv(ci); ep.out_f32 = feats; ep.ldc = He;
        RT_TRY(launch_gemm(ctx, a, W(ci), ep));
    }
    const int Te = (int)Tc;                        // = 2 * n_frames
    // ---- speaker head on the conv features
    float *stats = nullptr, *sh1 = nullptr, *spk = nullptr;
    RT_TRY(pool_arr(m, (size_t)2 * He, &stats));
    RT_TRY(pool_arr(m, (size_t)e.spk_hidden, &sh1));
    RT_TRY(pool_arr(m, (size_t)c.talker.hidden, &spk));
    RT_TRY(launch_stats_pool(ctx, feats, Te, He, stats));
    RT_TRY(launch_gemv_f32(ctx, VEC(m, "enc.spk_fc1"), VEC(m, "enc.spk_fc1_b"), stats, e.spk_hidden, 2 * He, 1, sh1));
    RT_TRY(launch_gemv_f32(ctx, VEC(m, "enc.spk_fc2"), VEC(m, "enc.spk_fc2_b"), sh1, c.talker.hidden, e.spk_hidden, 0, spk));
    // ---- transformer (float32-faithful form, sliding window), input = a copy of the features (the stack updates in place)
    float *h = nullptr, *hn = nullptr;
    int32_t *d_slot = nullptr, *d_pos = nullptr;
    RT_TRY(pool_arr(m, (size_t)Te * He, &h));
    RT_TRY(pool_arr(m, (size_t)(Te + 2) * He, &hn));           // two extra rows in front: the replicate padding of the next conv
    RT_TRY(pool_arr(m, Te, &d_slot));
    RT_TRY(pool_arr(m, Te, &d_pos));
    RT_HIP(ctx, hipMemcpyAsync(h, feats, (size_t)Te * He * 4, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fill_i32, dim3(8), dim3(256), 0, ctx->stream, d_slot, Te, 0, 0, 0);
    hipLaunchKernelGGL(k_fill_i32, dim3(8), dim3(256), 0, ctx->stream, d_pos, Te, 0, 1, 1);
    RT_HIP(ctx, hipGetLastError());
    {
        StackWs w;
        RT_TRY(alloc_stack_ws(m, e.tf, Te, &w, true));
        RT_TRY(stack_forward(m, m->etf, w, h, Te, d_slot, d_pos, 0, nullptr, hn + 2 * He));
    }
    // ---- stride-2 conv k = 4 with REPLICATE left padding (2 samples = the first row twice), no bias
    RT_HIP(ctx, hipMemcpyAsync(hn, hn + 2 * He, (size_t)He * 4, hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(hn + He, hn + 2 * He, (size_t)He * 4, hipMemcpyDeviceToDevice, ctx->stream));
    const int Tf = Te / 2;
    float* emb = nullptr;
    RT_TRY(pool_arr(m, (size_t)Tf * He, &emb));
    {
        GemmA a; a.ptr = hn; a.is_f32 = 1; a.split = 1; a.M = Tf; a.Cin = 2 * He; a.taps = 2; a.tap_stride = 1; a.tap_offset = 0;
        a.rows_out = Tf; a.rows_in = Tf + 1;
        GemmEpi ep; ep.out_f32 = emb; ep.ldc = He;
        RT_TRY(launch_gemm(ctx, a, PW(m, "enc.down"), ep));
    }
    // ---- split residual vector quantiser
    float *sem = nullptr, *aco = nullptr;
    int32_t* d_codes = nullptr;
    RT_TRY(pool_arr(m, (size_t)Tf * e.vq_dim, &sem));
    RT_TRY(pool_arr(m, (size_t)Tf * e.vq_dim, &aco));
    RT_TRY(pool_arr(m, (size_t)Tf * c.num_quantizers, &d_codes));
    for (int which = 0; which < 2; ++which) {
        GemmA a; a.ptr = emb; a.is_f32 = 1; a.split = 1; a.M = Tf; a.Cin = He; a.taps = 1;
        GemmEpi ep; ep.out_f32 = which ? aco : sem; ep.ldc = e.vq_dim;
        RT_TRY(launch_gemm(ctx, a, PW(m, which ? "enc.vq_aco" : "enc.vq_sem"), ep));
    }
    RT_TRY(launch_rvq(ctx, sem, aco, Tf, e.vq_dim, c.codebook_size, c.num_quantizers, m->d_cbT, d_codes));
    RT_HIP(ctx, hipMemcpyAsync(h_codes, d_codes, (size_t)Tf * c.num_quantizers * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (h_speaker_embed) RT_HIP(ctx, hipMemcpyAsync(h_speaker_embed, spk, (size_t)c.talker.hidden * 4, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *h_n_frames = Tf;
    pool_release_all(m);
    return RT_OK;
}

int rt_voice_encode(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t* h_codes, int32_t max_frames, int32_t* h_n_frames,
                    float* h_speaker_embed) {
    if (!m || !h_pcm || !h_codes || !h_n_frames || max_frames < 1) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_voice_encode: null argument");
    std::lock_guard<std::mutex> g(m->ctx->mu);
    RT_HIP(m->ctx, hipSetDevice(m->ctx->device));
    return voice_encode_impl(m, h_pcm, n_samples, h_codes, max_frames, h_n_frames, h_speaker_embed);
}

int rt_model_set_voice_pcm(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t n_head_rows, const int32_t* h_text_ids,
                           const int32_t* h_codec_ids, int32_t h_speaker_row, int32_t frame_text_id, int32_t max_ref_frames, int32_t* h_codes,
                           int32_t* h_n_frames) {
    if (!m || !h_pcm || n_head_rows < 1 || !h_text_ids || !h_codec_ids || max_ref_frames < 1)
        return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_model_set_voice_pcm: null argument");
    rt_ctx* ctx = m->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const rt_model_config& c = m->cfg;
    const int G = c.n_groups, Q = c.num_quantizers;
    if (Q != G) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_model_set_voice_pcm: the codec has %d codebooks but a prompt row takes %d codes", Q, G);
    if (h_speaker_row >= n_head_rows) return rt_fail(ctx, RT_ERR_INVALID, "rt_model_set_voice_pcm: bad speaker row");
    std::vector<int32_t> codes((size_t)max_ref_frames * Q);
    std::vector<float> spk(c.talker.hidden);
    int32_t nf = 0;
    RT_TRY(voice_encode_impl(m, h_pcm, n_samples, codes.data(), max_ref_frames, &nf, spk.data()));
    const int n_rows = n_head_rows + nf;
    std::vector<int32_t> tid(h_text_ids, h_text_ids + n_head_rows), cid(h_codec_ids, h_codec_ids + (size_t)n_head_rows * G);
    tid.resize(n_rows, frame_text_id);
    cid.insert(cid.end(), codes.begin(), codes.begin() + (size_t)nf * Q);
    if (h_codes) memcpy(h_codes, codes.data(), (size_t)nf * Q * 4);
    if (h_n_frames) *h_n_frames = nf;
    return set_voice_impl(m, n_rows, tid.data(), cid.data(), h_speaker_row, h_speaker_row >= 0 ? spk.data() : nullptr);
}

int32_t rt_voice_prefix_len(rt_model* m) { return m ? m->prefix_len : -1; }

int64_t rt_voice_blob_bytes(rt_model* m) {
    if (!m) return -1;
    const rt_stack_dims& d = m->cfg.talker;
    return (int64_t)2 * d.layers * d.kv_heads * m->prefix_len * d.head_dim * 2;
}

static int voice_blob(rt_model* m, void* d_blob, int64_t bytes, int to_blob, int prefix_len) {
    rt_ctx* ctx = m->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "voice blob: model not finalized");
    if (m->run && !to_blob) return rt_fail(ctx, RT_ERR_STATE, "rt_voice_import: a generation is in flight (rt_generate_end first)");
    const rt_stack_dims& d = m->cfg.talker;
    if (prefix_len < 1 || prefix_len + 8 > m->cfg.max_positions) return rt_fail(ctx, RT_ERR_LENGTH, "voice blob: prefix length %d out of range", prefix_len);
    const int64_t need = (int64_t)2 * d.layers * d.kv_heads * prefix_len * d.head_dim * 2;
    if (!d_blob || bytes != need) return rt_fail(ctx, RT_ERR_INVALID, "voice blob: expected %lld bytes, got %lld", (long long)need, (long long)bytes);
    KvCache& kv = m->talker.kv;
    hipLaunchKernelGGL(k_kv_blob, dim3(d.layers * d.kv_heads, 2), dim3(256), 0, ctx->stream, kv.k, kv.v, (int64_t)kv.layer_stride(), d.layers,
                       d.kv_heads, kv.max_pos, d.head_dim, m->prefix_slot(), prefix_len, (bf16_t*)d_blob, to_blob);
    RT_HIP(ctx, hipGetLastError());
    if (!to_blob) {
        m->prefix_tiles_valid = false;
        m->talker.kv.tiles_len = -1;
        if (d.head_dim == 128 && kv.kt_prefix) {
            RT_TRY(launch_transpose_prefix_v(ctx, kv, prefix_len));
            m->prefix_tiles_valid = true;
        }
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!to_blob) m->prefix_len = prefix_len;
    return RT_OK;
}
int rt_voice_export(rt_model* m, void* d_blob, int64_t bytes) {
    if (!m) return RT_ERR_INVALID;
    return voice_blob(m, d_blob, bytes, 1, m->prefix_len);
}
int rt_voice_import(rt_model* m, int32_t prefix_len, const void* d_blob, int64_t bytes) {
    if (!m) return RT_ERR_INVALID;
    return voice_blob(m, const_cast<void*>(d_blob), bytes, 0, prefix_len);
}

}  // extern "C"

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
    every = std::max(1, queued ? g_handover_every : g_eos_check_every);
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

    n_lanes = std::max(1, std::min(g_decode_lanes, 8));
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
        RT_TRY(pool_arr(m, 1, &ln.d_frame));
        RT_TRY(pool_arr(m, n, &ln.d_frame_off));
        RT_HIP(ctx, hipMemsetAsync(ln.d_frame_off, 0, n * 4, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_seen, 0, (size_t)n * Vc, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_zero_pos, 0, n2 * 4, ctx->stream));
        RT_HIP(ctx, hipMemsetAsync(ln.d_frame, 0, 4, ctx->stream));
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
        RT_TRY(launch_gather_f32(ctx, m->proj_c0, Hp, codes, n, ln.xp + (size_t)n * Hp, nullptr, G, ln.d_frame, codes_fs));
    } else {
        // equal-width predictor: its first input row is the talker's normalised hidden state itself
        if (col) RT_TRY(launch_norm_tiled_rows(ctx, ln.dwt.xT, ln.rowsq_t, NTt, m->talker.norm, c.talker.rms_eps, n, H, ln.xp));
        else RT_HIP(ctx, hipMemcpyAsync(ln.xp, ln.hn_f32, (size_t)n * H * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, 1, codes, n, H, nullptr, nullptr, nullptr, ln.xp + (size_t)n * Hp, nullptr, G, ln.d_frame, codes_fs));
    }
    if (col) {
        RT_TRY(launch_rowsq(ctx, ln.xp, n2, Hp, ln.rowsq_p, NTp, ln.dwp.xT, ln.dwp.xa, m->pred.L[0].ln1));
        RT_TRY(stack_decode(m, m->pred, ln.dwp, ln.dwp.xT, ln.rowsq_p, n2, ln.d_slot_b, ln.d_pos_p2, 0, false));
    } else {
        RT_TRY(stack_forward(m, m->pred, ln.wp, ln.xp, n2, ln.d_slot_b, ln.d_pos_p2, 0, ln.hn_p, nullptr));
    }
    for (int q = 0; q < G - 1; ++q) {
        const size_t roff = (q == 0) ? (size_t)n : 0;      // the first head reads the rows of position 1
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
            if (col) {
                RT_TRY(stack_decode(m, m->pred, ln.dwp, ln.dwp.xT, ln.rowsq_p, n, ln.d_slot_b, ln.d_zero_pos, q + 2, true, nullptr, ln.b0, true));
            } else {
                RT_TRY(stack_forward(m, m->pred, ln.wp, ln.xp, n, ln.d_slot_b, ln.d_zero_pos, q + 2, ln.hn_p, nullptr));
            }
        }
    }
    return RT_OK;
}

// ---- frame part B: next talker input (sum of the frame's G code embeddings + projected tts_pad), talker step, frame += 1
int rt_gen_run::enqueue_b(Lane& ln) {
    const int n = ln.n;
    int32_t* codes = d_codes + (size_t)ln.b0 * G;
    if (col && G <= 16) {
        RT_TRY(launch_embed_rowsq(ctx, m->d_frame_srcs, G, nullptr, codes, G, ln.d_frame, codes_fs, n, H, pad_t, ln.rowsq_t, NTt, ln.dwt.xT,
                                  ln.dwt.xa, m->talker.L[0].ln1));
    } else {
        RT_TRY(launch_gather_sum(ctx, m->d_frame_srcs, G, codes, n, H, pad_t, nullptr, nullptr, ln.xt, nullptr, G, ln.d_frame, codes_fs));
        if (col) RT_TRY(launch_rowsq(ctx, ln.xt, n, H, ln.rowsq_t, NTt, ln.dwt.xT, ln.dwt.xa, m->talker.L[0].ln1));
    }
    if (col) {
        RT_TRY(stack_decode(m, m->talker, ln.dwt, ln.dwt.xT, ln.rowsq_t, n, ln.d_slot_b, ln.d_pos_b, 0, true, ln.d_frame));
    } else {
        RT_TRY(stack_forward(m, m->talker, ln.wt, ln.xt, n, ln.d_slot_b, ln.d_pos_b, 0, ln.hn, ln.hn_f32, ln.d_frame));
    }
    hipLaunchKernelGGL(k_frame_inc, dim3(1), dim3(64), 0, ctx->stream, ln.d_frame);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return gen_begin_locked(m, A);
}

int rt_generate_step(rt_model* m, int32_t n_frames, int32_t* h_frames_run, int32_t* h_all_done) {
    if (!m || n_frames < 1) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_generate_step: bad argument");
    rt_ctx* ctx = m->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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
    std::lock_guard<std::mutex> g(ctx->mu);
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

// --------------------------------------------------------------------------------------- code2wav
int64_t rt_wav_length(rt_model* m, int32_t n_frames) {
    if (!m || n_frames < 0) return -1;
    int64_t L = n_frames;
    for (int i = 0; i < m->cfg.n_upsampling; ++i) L *= m->cfg.upsampling_ratios[i];
    for (int i = 0; i < m->cfg.n_upsample_rates; ++i) L = (L - 1) * m->cfg.upsample_rates[i];
    return L > 0 ? L : 0;
}

int rt_code2wav(rt_model* m, int32_t n_items, int32_t t_max, const int32_t* h_codes, const int32_t* h_n_frames, float* d_wav,
                int64_t wav_stride, int64_t* h_wav_len) {
    if (!m || !h_codes || !h_n_frames || !d_wav || !h_wav_len) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_code2wav: null argument");
    rt_ctx* ctx = m->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_code2wav: model not finalized");
    const rt_model_config& c = m->cfg;
    const int B = n_items, T = t_max, Q = c.num_quantizers, Hc = c.codec_tf.hidden;
    if (B < 1 || B > c.max_batch || T < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_code2wav: n_items %d / t_max %d out of range", B, T);
    if (T > c.max_codec_frames) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length %d frames exceeds max_codec_frames %d", T, c.max_codec_frames);
    const int64_t L_out = rt_wav_length(m, T);
    if (L_out < 1) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length %d frames too short for the decoder", T);
    if (wav_stride < L_out) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: wav_stride %lld < length %lld", (long long)wav_stride, (long long)L_out);
    for (int b = 0; b < B; ++b)
        if (h_n_frames[b] < 0 || h_n_frames[b] > T) return rt_fail(ctx, RT_ERR_INVALID, "rt_code2wav: n_frames[%d] out of range", b);
    pool_release_all(m);
    const int64_t rows0 = (int64_t)B * T;
    int32_t *d_codes, *d_slot, *d_pos;
    RT_TRY(pool_arr(m, (size_t)rows0 * Q, &d_codes));
    RT_TRY(pool_arr(m, rows0, &d_slot));
    RT_TRY(pool_arr(m, rows0, &d_pos));
    RT_HIP(ctx, hipMemcpyAsync(d_codes, h_codes, (size_t)rows0 * Q * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        std::vector<int32_t> sl(rows0), ps(rows0);
        for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) { sl[(size_t)b * T + t] = b; ps[(size_t)b * T + t] = t; }
        RT_HIP(ctx, hipMemcpyAsync(d_slot, sl.data(), rows0 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(d_pos, ps.data(), rows0 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    }
    float* h = nullptr;
    RT_TRY(pool_arr(m, (size_t)rows0 * Hc, &h));
    RT_TRY(launch_code_embed_mean(ctx, TBL(m, "codec.code_embedding"), c.codebook_size, Q, Hc, d_codes, rows0, h));
    float* hn = nullptr;
    RT_TRY(pool_arr(m, (size_t)rows0 * Hc, &hn));
    {
        StackWs w;
        RT_TRY(alloc_stack_ws(m, c.codec_tf, (int)rows0, &w, true));
        RT_TRY(stack_forward(m, m->ctf, w, h, (int)rows0, d_slot, d_pos, 0, nullptr, hn));
    }
    // ---- ConvNeXt upsampling stages (transposed conv k = stride = r is a plain GEMM in channels-last).
    // Every conv-as-GEMM below runs with split (hi + lo) activations: see k_gemm_tiled.
    float* cur = hn;
    int64_t Tc = T;
    for (int i = 0; i < c.n_upsampling; ++i) {
        const std::string u = "codec.up" + std::to_string(i);
        const int r = c.upsampling_ratios[i];
        float* up = nullptr;
        RT_TRY(pool_arr(m, (size_t)B * Tc * r * Hc, &up));
        GemmA a; a.ptr = cur; a.is_f32 = 1; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = Hc;
        GemmEpi e; e.bias = VEC(m, u + ".tconv_b"); e.out_f32 = up; e.ldc = (int64_t)r * Hc;
        RT_TRY(launch_gemm(ctx, a, PW(m, u + ".tconv"), e));
        Tc *= r;
        const int64_t rows = (int64_t)B * Tc;
        float *ln = nullptr, *mid = nullptr;
        RT_TRY(pool_arr(m, (size_t)rows * Hc, &ln));
        RT_TRY(pool_arr(m, (size_t)rows * 4 * Hc, &mid));
        RT_TRY(launch_dwconv_ln(ctx, up, B, (int)Tc, Hc, VEC(m, u + ".dw_w"), VEC(m, u + ".dw_b"), VEC(m, u + ".ln_w"), VEC(m, u + ".ln_b"), 1e-6f, ln));
        GemmA a1; a1.ptr = ln; a1.is_f32 = 1; a1.split = 1; a1.M = rows; a1.Cin = Hc;
        GemmEpi e1; e1.bias = VEC(m, u + ".pw1_b"); e1.act = ACT_GELU; e1.out_f32 = mid; e1.ldc = 4 * (int64_t)Hc;
        RT_TRY(launch_gemm(ctx, a1, PW(m, u + ".pw1"), e1));
        GemmA a2; a2.ptr = mid; a2.is_f32 = 1; a2.split = 1; a2.M = rows; a2.Cin = 4 * Hc;
        GemmEpi e2; e2.bias = VEC(m, u + ".pw2_b"); e2.scale = VEC(m, u + ".gamma"); e2.residual = up; e2.out_f32 = up; e2.ldc = Hc;
        RT_TRY(launch_gemm(ctx, a2, PW(m, u + ".pw2"), e2));
        cur = up;
    }
    // ---- decoder: conv k7 -> [SnakeBeta, transposed conv, 3 residual units] x n -> SnakeBeta -> conv k7 -> clamp
    // The snake-activated operands (s_in, s1, s2) are kept as hi + lo bf16 planes written by the producing epilogue: the same
    // 4 bytes per element as f32, but the split is done once per element instead of once per consuming workgroup and tap.
    struct Planes { bf16_t* hi = nullptr; bf16_t* lo = nullptr; };
    auto planes = [&](size_t n, Planes* p) -> int {
        RT_TRY(pool_arr(m, n, &p->hi));
        RT_TRY(pool_arr(m, n, &p->lo));
        return RT_OK;
    };
    Planes s_in;  // snake-activated input of the next transposed conv
    {
        RT_TRY(planes((size_t)B * Tc * m->dec_ch[0], &s_in));
        GemmA a; a.ptr = cur; a.is_f32 = 1; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = Hc; a.taps = 7; a.tap_stride = 1; a.tap_offset = -6;
        a.rows_out = (int)Tc; a.rows_in = (int)Tc;
        GemmEpi e; e.bias = VEC(m, "codec.dec0_b"); e.out2_hi = s_in.hi; e.out2_lo = s_in.lo;
        e.snake2_a = VEC(m, "codec.b0.sa"); e.snake2_ib = VEC(m, "codec.b0.sib");
        e.ldc = m->dec_ch[0];
        RT_TRY(launch_gemm(ctx, a, PW(m, "codec.dec0"), e));
    }
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        const std::string bn = "codec.b" + std::to_string(i);
        const int cin = m->dec_ch[i], cout = m->dec_ch[i + 1], r = c.upsample_rates[i];
        const int64_t To = (Tc - 1) * r;
        if (To < 1) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length collapsed in decoder block %d", i);
        const int64_t rows = (int64_t)B * To;
        float* xr = nullptr;
        Planes s1, s2;
        RT_TRY(pool_arr(m, (size_t)rows * cout, &xr));
        RT_TRY(planes((size_t)rows * cout, &s1));
        RT_TRY(planes((size_t)rows * cout, &s2));
        {
            // transposed conv k = 2r, stride r, r samples trimmed on both sides: out[m*r + j] = x[m+1] W[j] + x[m] W[j + r]
            GemmA a; a.ptr = s_in.hi; a.ptr_lo = s_in.lo; a.split = 1; a.M = (int64_t)B * (Tc - 1); a.Cin = cin; a.taps = 2; a.tap_stride = 1; a.tap_offset = 0;
            a.rows_out = (int)(Tc - 1); a.rows_in = (int)Tc;
            GemmEpi e; e.bias = VEC(m, bn + ".tconv_b"); e.out_f32 = xr; e.out2_hi = s1.hi; e.out2_lo = s1.lo;
            e.snake2_a = m->xvec[bn + ".u0.a1"]; e.snake2_ib = m->xvec[bn + ".u0.ib1"];
            e.ldc = (int64_t)r * cout;
            RT_TRY(launch_gemm(ctx, a, PW(m, bn + ".tconv"), e));
        }
        for (int j = 0; j < 3; ++j) {
            const std::string u = bn + ".u" + std::to_string(j);
            const int dil = j == 0 ? 1 : (j == 1 ? 3 : 9);
            GemmA a; a.ptr = s1.hi; a.ptr_lo = s1.lo; a.split = 1; a.M = rows; a.Cin = cout; a.taps = 7; a.tap_stride = dil; a.tap_offset = -6 * dil;
            a.rows_out = (int)To; a.rows_in = (int)To;
            GemmEpi e; e.bias = VEC(m, u + ".c1_b"); e.act = ACT_SNAKE; e.snake_a = VEC(m, u + ".a2"); e.snake_ib = VEC(m, u + ".ib2");
            e.out_hi = s2.hi; e.out_lo = s2.lo; e.ldc = cout;
            GemmA a2; a2.ptr = s2.hi; a2.ptr_lo = s2.lo; a2.split = 1; a2.M = rows; a2.Cin = cout;
            GemmEpi e2; e2.bias = VEC(m, u + ".c2_b"); e2.residual = xr; e2.out_f32 = xr; e2.out2_hi = s1.hi; e2.out2_lo = s1.lo; e2.ldc = cout;
            if (j < 2) { e2.snake2_a = VEC(m, bn + ".u" + std::to_string(j + 1) + ".a1"); e2.snake2_ib = VEC(m, bn + ".u" + std::to_string(j + 1) + ".ib1"); }
            else if (i + 1 < c.n_upsample_rates) { e2.snake2_a = VEC(m, "codec.b" + std::to_string(i + 1) + ".sa"); e2.snake2_ib = VEC(m, "codec.b" + std::to_string(i + 1) + ".sib"); }
            else { e2.snake2_a = VEC(m, "codec.fin_a"); e2.snake2_ib = VEC(m, "codec.fin_ib"); }
            // NOTE: the fused form writes s1 (the NEXT unit's operand planes) while other workgroups still read s1 as THIS unit's
            // input window, so it needs a second pair of planes to write to: s1 and s2 swap roles from unit to unit
            if (conv_pair_fusable(a, PW(m, u + ".c1"), e, PW(m, u + ".c2"))) {
                e2.out2_hi = s2.hi; e2.out2_lo = s2.lo;
                RT_TRY(launch_gemm(ctx, a, PW(m, u + ".c1"), e, &PW(m, u + ".c2"), &e2));
                std::swap(s1, s2);
            } else {
                RT_TRY(launch_gemm(ctx, a, PW(m, u + ".c1"), e));
                RT_TRY(launch_gemm(ctx, a2, PW(m, u + ".c2"), e2));
            }
        }
        s_in = s1;
        Tc = To;
    }
    float* wav_tmp = nullptr;
    RT_TRY(pool_arr(m, (size_t)B * Tc, &wav_tmp));
    {
        // last conv: channels -> 1, k = 7, causal, then clamp(-1, 1): the 7 x C taps of one output sample are contiguous
        // in the channels-last buffer, so it is the same implicit GEMM with a single output column
        const int cl = m->dec_ch.back();
        GemmA a; a.ptr = s_in.hi; a.ptr_lo = s_in.lo; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = cl; a.taps = 7; a.tap_stride = 1; a.tap_offset = -6;
        a.rows_out = (int)Tc; a.rows_in = (int)Tc;
        if (launch_final_conv_ok(cl) && g_final_conv) {
            RT_TRY(launch_final_conv(ctx, s_in.hi, s_in.lo, B, (int)Tc, cl, VEC(m, "codec.fin_wv"), VEC(m, "codec.fin_b"), wav_tmp));
        } else {
            GemmEpi e; e.bias = VEC(m, "codec.fin_b"); e.act = ACT_CLAMP1; e.out_f32 = wav_tmp; e.ldc = 1;
            RT_TRY(launch_gemm(ctx, a, PW(m, "codec.fin_w"), e));
        }
    }
    RT_HIP(ctx, hipMemcpy2DAsync(d_wav, (size_t)wav_stride * 4, wav_tmp, (size_t)Tc * 4, (size_t)Tc * 4, B, hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) h_wav_len[b] = rt_wav_length(m, h_n_frames[b]);
    pool_release_all(m);
    return RT_OK;
}

}  // extern "C"
