// Model group, part 3 of 5: the voice.  Conditioning front-end (reference clip -> codec frames + speaker embedding: rt_voice_encode),
// the voice prefix through the talker (rt_model_set_voice*: its K/V, computed once per voice and shared by every sequence) and its
// export / import as a blob (what a data-parallel job broadcasts).  Stands behind the `ref_audio=path` argument the reference hands
// its model on EVERY call (providers/qwen.py:253-258).
#include "model_internal.h"

using namespace rtm;

namespace {

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
}  // namespace

extern "C" {

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
    RT_TRY(launch_fill_i32(ctx, d_slot, n_rows, m->prefix_slot(), 0, 0));
    RT_TRY(launch_fill_i32(ctx, d_pos, n_rows, 0, 1, 1));
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
    CtxLock g(m->ctx);
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
    RT_TRY(launch_fill_i32(ctx, d_slot, Te, 0, 0, 0));
    RT_TRY(launch_fill_i32(ctx, d_pos, Te, 0, 1, 1));
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
    CtxLock g(m->ctx);
    RT_HIP(m->ctx, hipSetDevice(m->ctx->device));
    return voice_encode_impl(m, h_pcm, n_samples, h_codes, max_frames, h_n_frames, h_speaker_embed);
}

int rt_model_set_voice_pcm(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t n_head_rows, const int32_t* h_text_ids,
                           const int32_t* h_codec_ids, int32_t h_speaker_row, int32_t frame_text_id, int32_t max_ref_frames, int32_t* h_codes,
                           int32_t* h_n_frames) {
    if (!m || !h_pcm || n_head_rows < 1 || !h_text_ids || !h_codec_ids || max_ref_frames < 1)
        return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_model_set_voice_pcm: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
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
    CtxLock g(ctx);
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
