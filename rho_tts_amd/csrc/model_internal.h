// Internal interface of the model group (weights, voice, generation, codec decoder): shared by model_load.hip, model_stack.hip,
// voice.hip, generate.hip and code2wav.hip.  Not part of the C ABI.
#pragma once

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>

#include "kernels.h"


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
    // what the launches being recorded stream (rt_profile_read_class): 0 = talker layers / codec head / mtp projection (crosses HBM
    // once per frame), 1 = the predictor's first pass over its layers + its 15 heads (each byte's first use in the frame),
    // 2 = predictor passes 2..15 over the same layers (re-streamed from the Infinity Cache)
    int prof_class = 0;
    std::vector<std::pair<uint8_t, double>> prof_tag;      // (class, algorithmic bytes) per recorded launch
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

#define RT_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)


namespace rtm {

// ---- pool of device workspaces (model_load.hip)
int pool_get(rt_model* m, size_t bytes, void** out);
void pool_release_all(rt_model* m);      // (the blocks of a generation in flight stay: gen_release frees them)
template <typename T>
int pool_arr(rt_model* m, size_t n, T** out) {
    void* p = nullptr;
    RT_TRY(pool_get(m, n * sizeof(T), &p));
    *out = (T*)p;
    return RT_OK;
}
// ---- tensors by name (model_load.hip)
Slot* find_slot(rt_model* m, const std::string& name);
const PackedW& PW(rt_model* m, const std::string& n);
float* VEC(rt_model* m, const std::string& n);
bf16_t* TBL(rt_model* m, const std::string& n);
int launch_fill_i32(rt_ctx* ctx, int32_t* p, int n, int v, int step_every, int step);   // p[i] = v + (i / step_every) * step

// ---- transformer stacks (model_stack.hip)
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
struct DecWs {
    float* qkv = nullptr;   // [M][(heads + 2 kv) * d]
    bf16_t* ao = nullptr;   // [M][q_dim]
    bf16_t* act = nullptr;  // [M][inter]
    float* q = nullptr;     // [M][q_dim] (only for passes with several rows per slot)
    float* xT = nullptr;    // fragment-tiled residual stream [ceil(M/32)*32][H]
    bf16_t* xa = nullptr;   // fragment-tiled bf16(norm_w .* x): operand of the GEMM behind the next RMSNorm
};
struct TextWs { bf16_t *e = nullptr, *h1 = nullptr; GatherSrc* d_src = nullptr; };   // optional caller-owned workspace (repeated calls)
bool prof_events(rt_model* m, double bytes, hipEvent_t* a, hipEvent_t* b);
int gemm_rows(rt_model* m, const bf16_t* A, int rows, const PackedW& W, float* slabs, int* n_slabs);
int gemm_rows_f32(rt_model* m, const float* A, int rows, const PackedW& W, float* slabs, int* n_slabs);
size_t slab_floats(const rt_stack_dims& d, int M, int n_cu = 256);
int alloc_stack_ws(rt_model* m, const rt_stack_dims& d, int M, StackWs* w, bool precise = false);
int stack_forward(rt_model* m, StackW& S, StackWs& w, float* x, int M, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                  bf16_t* out_bf16, float* out_f32, const int32_t* frame_ptr = nullptr, bool prefix_rows = false);
int alloc_dec_ws(rt_model* m, const rt_stack_dims& d, int M, DecWs* w);
int col_gemm(rt_model* m, const ColArgs& a0, const PackedW& W, bool is_predictor = false);
int stack_decode(rt_model* m, StackW& S, DecWs& w, float* x, float* rowsq, int M, const int32_t* row_slot, const int32_t* row_pos,
                 int pos_add, bool one_row_per_slot = true, const int32_t* frame_ptr = nullptr, int slot_base = -1, bool zero_pos = false);
int col_head(rt_model* m, const bf16_t* xa, const float* rowsq, int rowsq_n, int row_off, int M, int K, float eps,
             const PackedW& W, const float* bias, float* out);
int alloc_text_ws(rt_model* m, int n, TextWs* w);
int text_project(rt_model* m, const int32_t* d_ids, int n, float* out, const TextWs* ws = nullptr);

}  // namespace rtm
