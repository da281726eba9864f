// On-GPU speech-to-text for the validation loop: a Whisper-shaped encoder-decoder behind the C ABI, so that a
// generated segment is transcribed where it already lives - in HBM - instead of going through a temporary WAV file and a CPU
// model.  Stands behind validation/stt/stt_validator.py:42-148 (model load, transcribe_audio; the reference's own fallback is
// transformers' Whisper, :85-107) and the temp-WAV round trip of base_tts.py:821-830.
//
//   PCM at the TTS rate -> windowed-sinc resampler (16 kHz) -> log-mel front-end (400-point DFT per frame in float64, 80 slaney mel
//   filters, log10, (max - 8) floor, (x + 4) / 4: WhisperFeatureExtractor) -> 2 convs (GELU) + sinusoidal positions -> pre-LN
//   encoder (bidirectional attention over 1500 positions) -> decoder (causal self-attention + cross-attention, KV caches) ->
//   tied LM head -> greedy token ids after the forced prefix.
//
// Every GEMM is the tiled MFMA kernel of gemm.hip with float32 activations fed as hi + lo bf16 planes (weights bf16), K/V caches
// keep hi + lo planes and attention / LayerNorm run in float32: the token ids must equal the float32 oracle's (greedy argmax over
// 51865 logits), which plain bf16 activations would not guarantee.  The model is tiny (39 M parameters); nothing here is on the
// hot path of generation - it runs once per validated segment - so the kernels are the simple forms.
#include <algorithm>
#include <cmath>
#include <map>

#include "kernels.h"

namespace {

#define ST_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

enum SttKind { S_GEMM = 0, S_VEC = 1, S_TABLE = 2 };
struct SttSlot {
    std::string name;
    int kind = 0;
    int64_t rows = 0, cols = 0;
    bool set = false;
    PackedW pw;
    float* vec = nullptr;
    bf16_t* tbl = nullptr;
    void* raw = nullptr;
    void* raw2 = nullptr;    // TABLE slots that are also multiplied (tied LM head): the packed copy
};

__global__ void k_stt_bf16_to_f32(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = bf16_to_f32(x[i]);
}

// y[n] = sum_j x[base(n) + j - half] h[phase(n)][j]: polyphase windowed-sinc resampler, sr_in / sr_out = M / L in lowest terms,
// base = floor(n M / L), phase = (n M) mod L; samples outside [0, n_in) are zero.  float64 accumulation in tap order.
__global__ void k_resample(const float* __restrict__ x, int64_t n_in, float* __restrict__ y, int64_t n_out, int L, int M, int taps, int half,
                           const float* __restrict__ h) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (int64_t)gridDim.x * blockDim.x) {
        const int64_t num = n * M;
        const int64_t base = num / L;
        const int phase = (int)(num % L);
        const float* hp = h + (int64_t)phase * taps;
        double acc = 0.0;
        for (int j = 0; j < taps; ++j) {
            const int64_t t = base + j - half;
            if (t >= 0 && t < n_in) acc += (double)x[t] * (double)hp[j];
        }
        y[n] = (float)acc;
    }
}

// One frame per workgroup: the frame's 400 samples of the (zero-padded to 30 s, reflect-padded by n_fft / 2 at both ends) signal
// times the periodic Hann window, a direct DFT in float64 (bin k on thread k, twiddles from a 400-entry table: index k n mod 400),
// power spectrum, mel filters, log10(max(., 1e-10)).  logspec is [frames][n_mels]; gmax receives the maximum (ordered-int atomic).
__device__ __forceinline__ int f32_ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_f32(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__global__ __launch_bounds__(256) void k_logmel_frames(const float* __restrict__ pcm, int64_t n_valid, int64_t n_padded, int n_fft, int hop, int n_bins,
                                                       int n_mels, const double* __restrict__ twc, const double* __restrict__ tws,
                                                       const float* __restrict__ window, const float* __restrict__ melT /*[n_bins][n_mels]*/,
                                                       float* __restrict__ logspec, int* __restrict__ gmax) {
    extern __shared__ double sh[];                       // xw[n_fft] | c[n_fft] | s[n_fft] | power[n_bins]
    double* xw = sh;
    double* tc = sh + n_fft;
    double* ts = tc + n_fft;
    double* pw = ts + n_fft;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int64_t start = (int64_t)f * hop - n_fft / 2;
    for (int n = tid; n < n_fft; n += blockDim.x) {
        int64_t t = start + n;
        if (t < 0) t = -t;                                // reflect (no edge repeat) about sample 0 ...
        if (t >= n_padded) t = 2 * (n_padded - 1) - t;    // ... and about the last sample of the padded signal
        const float v = (t >= 0 && t < n_valid) ? pcm[t] : 0.f;
        xw[n] = (double)(v * window[n]);                  // float32 product, as torch.stft applies its float32 window
        tc[n] = twc[n];
        ts[n] = tws[n];
    }
    __syncthreads();
    for (int k = tid; k < n_bins; k += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
            re += xw[n] * tc[idx];
            im -= xw[n] * ts[idx];
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        pw[k] = (double)((float)re * (float)re + (float)im * (float)im);   // |X|^2 from the float32 spectrum
    }
    __syncthreads();
    float best = -INFINITY;
    for (int m = tid; m < n_mels; m += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < n_bins; ++k) acc += (double)melT[(int64_t)k * n_mels + m] * pw[k];
        const float v = log10f(fmaxf((float)acc, 1e-10f));
        logspec[(int64_t)f * n_mels + m] = v;
        best = fmaxf(best, v);
    }
    best = wave_max_f32(best);
    if ((tid & 63) == 0 && best > -INFINITY) atomicMax(gmax, f32_ordered(best));
}
// logspec[f][m] <- (max(v, gmax - 8) + 4) / 4; frames >= n_frames_audio hold the value of silence, log10(1e-10) = -10
__global__ void k_logmel_finish(float* __restrict__ logspec, int64_t n_total, int64_t n_computed, const int* __restrict__ gmax) {
    const float mx = fmaxf(ordered_f32(*gmax), n_computed < n_total ? -10.f : -INFINITY);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = i < n_computed ? logspec[i] : -10.f;
        logspec[i] = (fmaxf(v, mx - 8.0f) + 4.0f) / 4.0f;
    }
}

// LayerNorm with bias over the last dimension, float32 in and out, one workgroup per row (two-pass: mean, then variance)
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int D, const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                   float* __restrict__ out) {
    __shared__ float sh[4];
    const float* r = x + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) s += r[i];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)D;
    __syncthreads();
    float q = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) { const float d = r[i] - mean; q += d * d; }
    q = wave_sum_f32(q);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    const float inv = rsqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)D + eps);
    float* o = out + (int64_t)blockIdx.x * D;
    for (int i = threadIdx.x; i < D; i += 256) o[i] = (r[i] - mean) * inv * w[i] + b[i];
}

// x[r][:] = token_embedding[tok[r]] (bf16) + position_embedding[pos0 + r] (f32)
__global__ void k_stt_embed(const bf16_t* __restrict__ tok_emb, const float* __restrict__ pos_emb, const int32_t* __restrict__ tok, int pos0, int D,
                            float* __restrict__ x) {
    const int r = blockIdx.x;
    const int64_t t = tok[r];
    for (int i = threadIdx.x; i < D; i += blockDim.x) x[(int64_t)r * D + i] = bf16_to_f32(tok_emb[t * D + i]) + pos_emb[(int64_t)(pos0 + r) * D + i];
}

// greedy choice over one row of logits: the largest value among the tokens the mask allows (bit 0: never, bit 1: not as the first
// generated token), lowest index on ties; the winner is written to out[0] and appended to seq[*n_seq]
__global__ __launch_bounds__(1024) void k_stt_argmax(const float* __restrict__ logits, int V, const uint8_t* __restrict__ mask, int first_step,
                                                     int32_t* __restrict__ out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const uint8_t bad = first_step ? 3 : 1;
    for (int i = threadIdx.x; i < V; i += 1024) {
        if (mask[i] & bad) continue;
        const float v = logits[i];
        if (v > best || (v == best && i < bi) || bi == 0x7fffffff) { if (!(v != v)) { best = v; bi = i; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[0] = bi == 0x7fffffff ? 0 : bi;
    }
}

__global__ void k_stt_fill_i32(int32_t* p, int n, int v, int step) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v + i * step;
}

}  // namespace

struct rt_stt {
    rt_ctx* ctx = nullptr;
    rt_stt_config cfg{};
    std::vector<SttSlot> slots;
    std::map<std::string, int> by_name;
    bool finalized = false;
    KvCache enc_kv, dec_kv, cross_kv;
    // front-end constants
    double *d_twc = nullptr, *d_tws = nullptr;
    float *d_window = nullptr, *d_melT = nullptr;
    float* d_resamp = nullptr;        // polyphase filter of the last (sr_in -> cfg.sample_rate) pair
    int rs_in = 0, rs_L = 0, rs_M = 0, rs_taps = 0, rs_half = 0;
    // workspaces (sized for n_ctx rows)
    float *pcm16k = nullptr, *mel = nullptr, *c1 = nullptr, *x = nullptr, *xn = nullptr, *qkv = nullptr, *q = nullptr, *ao = nullptr, *ff = nullptr,
          *enc_out = nullptr, *logits = nullptr;
    float *dx = nullptr, *dxn = nullptr, *dqkv = nullptr, *dq = nullptr, *dao = nullptr, *dff = nullptr;
    int32_t *pos_seq = nullptr, *pos_last = nullptr, *slot0 = nullptr, *d_tok = nullptr, *d_gmax = nullptr;
    uint8_t* d_mask = nullptr;
    std::vector<int32_t> suppress_ids;   // rt_stt_set_suppress: ids never produced (a generation config's `suppress_tokens`)
    std::vector<void*> owned;
};

namespace {

void stt_slot(rt_stt* s, const std::string& name, int kind, int64_t rows, int64_t cols) {
    SttSlot sl;
    sl.name = name; sl.kind = kind; sl.rows = rows; sl.cols = cols;
    s->by_name[name] = (int)s->slots.size();
    s->slots.push_back(sl);
}
SttSlot* stt_find(rt_stt* s, const std::string& n) {
    auto it = s->by_name.find(n);
    return it == s->by_name.end() ? nullptr : &s->slots[it->second];
}
const PackedW& SPW(rt_stt* s, const std::string& n) { return stt_find(s, n)->pw; }
float* SVEC(rt_stt* s, const std::string& n) { return stt_find(s, n)->vec; }

void stt_declare(rt_stt* s) {
    const rt_stt_config& c = s->cfg;
    const int D = c.d_model, F = c.ffn;
    stt_slot(s, "enc.conv1", S_GEMM, D, 3 * (int64_t)c.n_mels);          // [Co][tap * Ci + ci]
    stt_slot(s, "enc.conv1_b", S_VEC, D, 1);
    stt_slot(s, "enc.conv2", S_GEMM, D, 4 * (int64_t)D);                  // stride-2 k3 conv as a 2-tap GEMM over [T/2][2 D] rows
    stt_slot(s, "enc.conv2_b", S_VEC, D, 1);
    stt_slot(s, "enc.pos", S_VEC, (int64_t)c.n_ctx * D, 1);
    auto layer = [&](const std::string& p, bool cross) {
        stt_slot(s, p + ".ln1_w", S_VEC, D, 1); stt_slot(s, p + ".ln1_b", S_VEC, D, 1);
        stt_slot(s, p + ".wqkv", S_GEMM, 3 * (int64_t)D, D); stt_slot(s, p + ".bqkv", S_VEC, 3 * (int64_t)D, 1);
        stt_slot(s, p + ".wo", S_GEMM, D, D); stt_slot(s, p + ".bo", S_VEC, D, 1);
        if (cross) {
            stt_slot(s, p + ".lnc_w", S_VEC, D, 1); stt_slot(s, p + ".lnc_b", S_VEC, D, 1);
            stt_slot(s, p + ".cwq", S_GEMM, D, D); stt_slot(s, p + ".cbq", S_VEC, D, 1);
            stt_slot(s, p + ".cwkv", S_GEMM, 2 * (int64_t)D, D); stt_slot(s, p + ".cbkv", S_VEC, 2 * (int64_t)D, 1);
            stt_slot(s, p + ".cwo", S_GEMM, D, D); stt_slot(s, p + ".cbo", S_VEC, D, 1);
        }
        stt_slot(s, p + ".ln2_w", S_VEC, D, 1); stt_slot(s, p + ".ln2_b", S_VEC, D, 1);
        stt_slot(s, p + ".fc1", S_GEMM, F, D); stt_slot(s, p + ".fc1_b", S_VEC, F, 1);
        stt_slot(s, p + ".fc2", S_GEMM, D, F); stt_slot(s, p + ".fc2_b", S_VEC, D, 1);
    };
    for (int i = 0; i < c.enc_layers; ++i) layer("enc.l" + std::to_string(i), false);
    stt_slot(s, "enc.ln_w", S_VEC, D, 1); stt_slot(s, "enc.ln_b", S_VEC, D, 1);
    stt_slot(s, "dec.tok", S_TABLE, c.vocab, D);                          // token embedding = tied LM head
    stt_slot(s, "dec.pos", S_VEC, (int64_t)c.n_text_ctx * D, 1);
    for (int i = 0; i < c.dec_layers; ++i) layer("dec.l" + std::to_string(i), true);
    stt_slot(s, "dec.ln_w", S_VEC, D, 1); stt_slot(s, "dec.ln_b", S_VEC, D, 1);
    stt_slot(s, "fe.window", S_VEC, c.n_fft, 1);
    stt_slot(s, "fe.melT", S_VEC, (int64_t)(c.n_fft / 2 + 1) * c.n_mels, 1);   // [bins][mels]
}

template <typename T>
int stt_alloc(rt_stt* s, size_t n, T** out) {
    void* p = nullptr;
    RT_HIP(s->ctx, hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    s->owned.push_back(p);
    *out = (T*)p;
    return RT_OK;
}

int stt_kv(rt_stt* s, KvCache& kv, int layers, int max_pos) {
    const rt_stt_config& c = s->cfg;
    kv.layers = layers; kv.slots = 1; kv.kv_heads = c.heads; kv.max_pos = max_pos; kv.head_dim = c.d_model / c.heads;
    const size_t n = (size_t)layers * kv.layer_stride();
    ST_TRY(stt_alloc(s, n, &kv.k)); ST_TRY(stt_alloc(s, n, &kv.v)); ST_TRY(stt_alloc(s, n, &kv.k_lo)); ST_TRY(stt_alloc(s, n, &kv.v_lo));
    for (bf16_t* p : {kv.k, kv.v, kv.k_lo, kv.v_lo}) RT_HIP(s->ctx, hipMemsetAsync(p, 0, n * sizeof(bf16_t), s->ctx->stream));
    return RT_OK;
}

// out[M][N] = act(A[M][K] W^T + bias) (+ residual), float32 activations fed as hi + lo planes
int stt_gemm(rt_stt* s, const float* A, int M, const PackedW& W, const float* bias, int act, const float* residual, float* out) {
    GemmA a; a.ptr = A; a.is_f32 = 1; a.split = 1; a.M = M; a.Cin = W.K; a.taps = 1;
    GemmEpi e; e.bias = bias; e.act = act; e.residual = residual; e.out_f32 = out; e.ldc = W.N;
    return launch_gemm(s->ctx, a, W, e);
}
int stt_ln(rt_stt* s, const float* x, int M, const float* w, const float* b, float* out) {
    hipLaunchKernelGGL(k_layernorm, dim3(M), dim3(256), 0, s->ctx->stream, x, s->cfg.d_model, w, b, 1e-5f, out);
    RT_HIP(s->ctx, hipGetLastError());
    return RT_OK;
}

// windowed-sinc resampling filter (Hann window over `width` zero crossings of the low-pass at rolloff x the lower Nyquist): the
// definition rho_tts_amd/stt.py restates in NumPy for the tests (oracle: parity unpinned - the reference's pipeline decodes its
// temporary WAV through ffmpeg)
int stt_resampler(rt_stt* s, int sr_in) {
    if (s->rs_in == sr_in && s->d_resamp) return RT_OK;
    const int sr_out = s->cfg.sample_rate;
    int a = sr_in, b = sr_out;
    while (b) { const int t = a % b; a = b; b = t; }
    const int L = sr_out / a, M = sr_in / a;
    const double width = 6.0, rolloff = 0.99;
    const double base = std::min(sr_in, sr_out) * rolloff;          // cut-off (both sides) in Hz x 2
    const int half = (int)std::ceil(width * sr_in / base);
    const int taps = 2 * half + 1;
    std::vector<float> h((size_t)L * taps);
    for (int p = 0; p < L; ++p)
        for (int j = 0; j < taps; ++j) {
            // time (in input samples) from tap j of phase p to the output instant: t = (j - half) - p / L
            const double t = ((double)(j - half) - (double)p / L) * base / sr_in;
            double v = 0.0;
            if (std::fabs(t) < width) {
                const double w = std::cos(t * M_PI / width / 2.0);
                const double sinc = t == 0.0 ? 1.0 : std::sin(M_PI * t) / (M_PI * t);
                v = sinc * w * w * base / sr_in;
            }
            h[(size_t)p * taps + j] = (float)v;
        }
    if (s->d_resamp) (void)hipFree(s->d_resamp);
    RT_HIP(s->ctx, hipMalloc((void**)&s->d_resamp, h.size() * 4));
    RT_HIP(s->ctx, hipMemcpy(s->d_resamp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    s->rs_in = sr_in; s->rs_L = L; s->rs_M = M; s->rs_taps = taps; s->rs_half = half;
    return RT_OK;
}

// PCM (device, any rate) -> log-mel [frames][n_mels] in s->mel (the mutex is held)
int stt_features(rt_stt* s, const float* d_pcm, int64_t n, int sr) {
    rt_ctx* ctx = s->ctx;
    const rt_stt_config& c = s->cfg;
    const int64_t n_pad = (int64_t)c.chunk_seconds * c.sample_rate;
    const float* src = d_pcm;
    int64_t n16 = n;
    if (sr != c.sample_rate) {
        ST_TRY(stt_resampler(s, sr));
        n16 = std::min<int64_t>((n * s->rs_L + s->rs_M - 1) / s->rs_M, n_pad);        // ceil(n L / M), at most 30 s
        if (n16 > 0)
            hipLaunchKernelGGL(k_resample, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, d_pcm, n, s->pcm16k, n16,
                               s->rs_L, s->rs_M, s->rs_taps, s->rs_half, s->d_resamp);
        RT_HIP(ctx, hipGetLastError());
        src = s->pcm16k;
    }
    n16 = std::min(n16, n_pad);
    const int n_frames = (int)(n_pad / c.hop);                      // 3000 (the last of the 3001 STFT frames is dropped)
    // frames that can see audio: [f hop - n_fft/2, f hop + n_fft/2) meets [0, n16); the rest are the constant of silence
    int n_comp = (int)std::min<int64_t>(n_frames, (n16 + c.n_fft / 2 + c.hop - 1) / c.hop + 1);
    if (n16 <= 0) n_comp = 0;
    const int n_bins = c.n_fft / 2 + 1;
    const int init = (int)0x80000000;                               // below every ordered float
    RT_HIP(ctx, hipMemcpyAsync(s->d_gmax, &init, 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_comp > 0) {
        const size_t lds = (size_t)(3 * c.n_fft + n_bins) * sizeof(double);
        hipLaunchKernelGGL(k_logmel_frames, dim3(n_comp), dim3(256), lds, ctx->stream, src, n16, n_pad, c.n_fft, c.hop, n_bins, c.n_mels, s->d_twc,
                           s->d_tws, s->d_window, s->d_melT, s->mel, s->d_gmax);
        RT_HIP(ctx, hipGetLastError());
    }
    const int64_t tot = (int64_t)n_frames * c.n_mels;
    hipLaunchKernelGGL(k_logmel_finish, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 2048)), dim3(256), 0, ctx->stream, s->mel, tot,
                       (int64_t)n_comp * c.n_mels, s->d_gmax);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// one pre-LN layer over M rows of x (in place).  self-attention over cache `kv` (rows are written at wpos, attend up to apos);
// cross = the decoder's encoder-attention block between the two
int stt_layer(rt_stt* s, const std::string& p, float* x, float* xn, float* qkv, float* q, float* ao, float* ff, int M, KvCache& kv, int layer,
              const int32_t* wpos, const int32_t* apos, int pos_add, bool cross) {
    rt_ctx* ctx = s->ctx;
    const rt_stt_config& c = s->cfg;
    const int H = c.heads, d = c.d_model / c.heads;
    ST_TRY(stt_ln(s, x, M, SVEC(s, p + ".ln1_w"), SVEC(s, p + ".ln1_b"), xn));
    ST_TRY(stt_gemm(s, xn, M, SPW(s, p + ".wqkv"), SVEC(s, p + ".bqkv"), ACT_NONE, nullptr, qkv));
    ST_TRY(launch_qkv_post(ctx, qkv, 1, M, H, H, d, nullptr, nullptr, 0.f, nullptr, nullptr, s->slot0, wpos, pos_add, q, kv, layer));
    ST_TRY(launch_attention(ctx, q, M, H, H, d, s->slot0, apos, apos == wpos ? pos_add : 0, 0, kv, layer, nullptr, nullptr, 0, ao));
    ST_TRY(stt_gemm(s, ao, M, SPW(s, p + ".wo"), SVEC(s, p + ".bo"), ACT_NONE, x, x));
    if (cross) {
        ST_TRY(stt_ln(s, x, M, SVEC(s, p + ".lnc_w"), SVEC(s, p + ".lnc_b"), xn));
        ST_TRY(stt_gemm(s, xn, M, SPW(s, p + ".cwq"), SVEC(s, p + ".cbq"), ACT_NONE, nullptr, q));
        ST_TRY(launch_attention(ctx, q, M, H, H, d, s->slot0, s->pos_last, 0, 0, s->cross_kv, layer, nullptr, nullptr, 0, ao));
        ST_TRY(stt_gemm(s, ao, M, SPW(s, p + ".cwo"), SVEC(s, p + ".cbo"), ACT_NONE, x, x));
    }
    ST_TRY(stt_ln(s, x, M, SVEC(s, p + ".ln2_w"), SVEC(s, p + ".ln2_b"), xn));
    ST_TRY(stt_gemm(s, xn, M, SPW(s, p + ".fc1"), SVEC(s, p + ".fc1_b"), ACT_GELU, nullptr, ff));
    ST_TRY(stt_gemm(s, ff, M, SPW(s, p + ".fc2"), SVEC(s, p + ".fc2_b"), ACT_NONE, x, x));
    return RT_OK;
}

// log-mel in s->mel -> encoder states in s->enc_out [n_ctx][D], and the decoder's cross-attention K/V caches
int stt_encode(rt_stt* s) {
    rt_ctx* ctx = s->ctx;
    const rt_stt_config& c = s->cfg;
    const int D = c.d_model, T2 = 2 * c.n_ctx, T = c.n_ctx;
    {   // conv1: k = 3, pad 1, GELU, on [T2][n_mels]
        GemmA a; a.ptr = s->mel; a.is_f32 = 1; a.split = 1; a.M = T2; a.Cin = c.n_mels; a.taps = 3; a.tap_stride = 1; a.tap_offset = -1; a.rows_out = T2; a.rows_in = T2;
        GemmEpi e; e.bias = SVEC(s, "enc.conv1_b"); e.act = ACT_GELU; e.out_f32 = s->c1; e.ldc = D;
        ST_TRY(launch_gemm(ctx, a, SPW(s, "enc.conv1"), e));
    }
    {   // conv2: k = 3, stride 2, pad 1, GELU, + positions.  Over rows [x[2t], x[2t+1]] it is the 2-tap GEMM (row t-1, row t) with the
        // weight laid out as [0 | W0 | W1 | W2]
        GemmA a; a.ptr = s->c1; a.is_f32 = 1; a.split = 1; a.M = T; a.Cin = 2 * D; a.taps = 2; a.tap_stride = 1; a.tap_offset = -1; a.rows_out = T; a.rows_in = T;
        GemmEpi e; e.bias = SVEC(s, "enc.conv2_b"); e.act = ACT_GELU; e.residual = SVEC(s, "enc.pos"); e.out_f32 = s->x; e.ldc = D;
        ST_TRY(launch_gemm(ctx, a, SPW(s, "enc.conv2"), e));
    }
    for (int i = 0; i < c.enc_layers; ++i)
        ST_TRY(stt_layer(s, "enc.l" + std::to_string(i), s->x, s->xn, s->qkv, s->q, s->ao, s->ff, T, s->enc_kv, i, s->pos_seq, s->pos_last, 0, false));
    ST_TRY(stt_ln(s, s->x, T, SVEC(s, "enc.ln_w"), SVEC(s, "enc.ln_b"), s->enc_out));
    // cross-attention K / V of every decoder layer (k_proj has no bias: its half of cbkv is zero)
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "dec.l" + std::to_string(i);
        ST_TRY(stt_gemm(s, s->enc_out, T, SPW(s, p + ".cwkv"), SVEC(s, p + ".cbkv"), ACT_NONE, nullptr, s->qkv));
        ST_TRY(launch_qkv_post(ctx, s->qkv, 1, T, 0, c.heads, D / c.heads, nullptr, nullptr, 0.f, nullptr, nullptr, s->slot0, s->pos_seq, 0, s->q, s->cross_kv, i));
    }
    return RT_OK;
}

// M decoder rows (tokens d_tok[0..M) at positions pos0 ..) -> logits of the LAST row in s->logits
int stt_decode_rows(rt_stt* s, int M, int pos0) {
    rt_ctx* ctx = s->ctx;
    const rt_stt_config& c = s->cfg;
    const int D = c.d_model;
    SttSlot* tok = stt_find(s, "dec.tok");
    hipLaunchKernelGGL(k_stt_embed, dim3(M), dim3(128), 0, ctx->stream, tok->tbl, SVEC(s, "dec.pos"), s->d_tok, pos0, D, s->dx);
    RT_HIP(ctx, hipGetLastError());
    for (int i = 0; i < c.dec_layers; ++i)
        ST_TRY(stt_layer(s, "dec.l" + std::to_string(i), s->dx, s->dxn, s->dqkv, s->dq, s->dao, s->dff, M, s->dec_kv, i, s->pos_seq, s->pos_seq, pos0, true));
    ST_TRY(stt_ln(s, s->dx + (size_t)(M - 1) * D, 1, SVEC(s, "dec.ln_w"), SVEC(s, "dec.ln_b"), s->dxn));
    ST_TRY(stt_gemm(s, s->dxn, 1, tok->pw, nullptr, ACT_NONE, nullptr, s->logits));
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_stt_create(rt_ctx* ctx, const rt_stt_config* cfg, rt_stt** out) {
    if (!ctx || !cfg || !out) return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_create: null argument");
    *out = nullptr;
    const rt_stt_config& c = *cfg;
    const int d = c.heads > 0 ? c.d_model / c.heads : 0;
    if (c.d_model < 16 || c.d_model % 8 || c.heads < 1 || c.d_model % c.heads || (d != 32 && d != 64 && d != 128) || c.ffn % 8 || c.n_mels % 8 ||
        c.enc_layers < 1 || c.dec_layers < 1 || c.n_ctx < 2 || c.n_text_ctx < 2 || c.vocab < 2 || c.n_fft < 16 || c.n_fft % 2 || c.hop < 1 ||
        c.sample_rate < 1000 || c.chunk_seconds < 1 || (int64_t)c.chunk_seconds * c.sample_rate / c.hop != 2 * (int64_t)c.n_ctx || c.n_prefix < 1 ||
        c.n_prefix > 8 || c.n_begin_suppress < 0 || c.n_begin_suppress > 4 || c.eos_id < 0 || c.eos_id >= c.vocab || c.max_new_tokens < 1 ||
        c.n_prefix + c.max_new_tokens > c.n_text_ctx || c.n_fft > 2048)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_create: unsupported configuration (head_dim in {32,64,128}, widths %% 8, 2 n_ctx = frames of one chunk)");
    for (int i = 0; i < c.n_prefix; ++i)      // (forced ids index the embedding table)
        if (c.prefix[i] < 0 || c.prefix[i] >= c.vocab) return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_create: forced prefix id %d outside the vocabulary of %d", c.prefix[i], c.vocab);
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rt_stt* s = new rt_stt();
    s->ctx = ctx;
    s->cfg = c;
    stt_declare(s);
    *out = s;
    return RT_OK;
}

int rt_stt_destroy(rt_stt* s) {
    if (!s) return RT_OK;
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& sl : s->slots) { if (sl.raw) (void)hipFree(sl.raw); if (sl.raw2) (void)hipFree(sl.raw2); }
    for (void* p : s->owned) (void)hipFree(p);
    if (s->d_resamp) (void)hipFree(s->d_resamp);
    delete s;
    return RT_OK;
}

int rt_stt_tensor_count(rt_stt* s) { return s ? (int)s->slots.size() : -1; }

int rt_stt_tensor_info(rt_stt* s, int32_t index, char* name, size_t name_cap, int64_t* shape2, int32_t* kind) {
    if (!s || index < 0 || index >= (int)s->slots.size()) return RT_ERR_INVALID;
    const SttSlot& sl = s->slots[index];
    if (name && name_cap) snprintf(name, name_cap, "%s", sl.name.c_str());
    if (shape2) { shape2[0] = sl.rows; shape2[1] = sl.cols; }
    if (kind) *kind = sl.kind;
    return RT_OK;
}

int rt_stt_set_tensor(rt_stt* s, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols, int32_t on_device) {
    if (!s || !name || !data) return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_stt_set_tensor: null argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    SttSlot* sl = stt_find(s, name);
    if (!sl) return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_set_tensor: unknown tensor '%s'", name);
    if (sl->rows * sl->cols != rows * cols || (sl->kind != S_VEC && (sl->rows != rows || sl->cols != cols)))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_set_tensor: '%s' expects [%lld, %lld], got [%lld, %lld]", name, (long long)sl->rows, (long long)sl->cols,
                       (long long)rows, (long long)cols);
    if (sl->kind != S_VEC && dtype != RT_DTYPE_BF16) return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_set_tensor: '%s' must be bf16", name);
    const int64_t n = rows * cols;
    const size_t esz = dtype == RT_DTYPE_BF16 ? 2 : 4;
    const void* d_src = data;
    if (!on_device) {
        void* stage = nullptr;
        ST_TRY(rt_ctx_scratch(ctx, (size_t)n * esz, &stage));
        RT_HIP(ctx, hipMemcpyAsync(stage, data, (size_t)n * esz, hipMemcpyHostToDevice, ctx->stream));
        d_src = stage;
    }
    if (sl->raw) { RT_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(sl->raw); sl->raw = nullptr; }
    if (sl->raw2) { (void)hipFree(sl->raw2); sl->raw2 = nullptr; }
    if (sl->kind == S_GEMM || sl->kind == S_TABLE) {
        void** packed = sl->kind == S_GEMM ? &sl->raw : &sl->raw2;
        RT_HIP(ctx, hipMalloc(packed, packed_bytes((int)rows, (int)cols)));
        ST_TRY(launch_pack_weight(ctx, (const bf16_t*)d_src, (int)rows, (int)cols, (bf16_t*)*packed, &sl->pw));
        if (sl->kind == S_TABLE) {
            RT_HIP(ctx, hipMalloc(&sl->raw, (size_t)n * 2));
            RT_HIP(ctx, hipMemcpyAsync(sl->raw, d_src, (size_t)n * 2, hipMemcpyDeviceToDevice, ctx->stream));
            sl->tbl = (bf16_t*)sl->raw;
        }
    } else {
        RT_HIP(ctx, hipMalloc(&sl->raw, (size_t)n * 4));
        sl->vec = (float*)sl->raw;
        if (dtype == RT_DTYPE_F32) RT_HIP(ctx, hipMemcpyAsync(sl->raw, d_src, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        else hipLaunchKernelGGL(k_stt_bf16_to_f32, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (const bf16_t*)d_src, n, sl->vec);
        RT_HIP(ctx, hipGetLastError());
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sl->set = true;
    return RT_OK;
}

int rt_stt_finalize(rt_stt* s) {
    if (!s) return RT_ERR_INVALID;
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (s->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_stt_finalize: already finalized");
    for (auto& sl : s->slots)
        if (!sl.set) return rt_fail(ctx, RT_ERR_INVALID, "rt_stt_finalize: tensor '%s' was never set", sl.name.c_str());
    const rt_stt_config& c = s->cfg;
    const int D = c.d_model, T = c.n_ctx, T2 = 2 * T, Tt = c.n_text_ctx;
    ST_TRY(stt_kv(s, s->enc_kv, c.enc_layers, T));
    ST_TRY(stt_kv(s, s->dec_kv, c.dec_layers, Tt));
    ST_TRY(stt_kv(s, s->cross_kv, c.dec_layers, T));
    std::vector<double> tc(c.n_fft), ts(c.n_fft);
    for (int n = 0; n < c.n_fft; ++n) { tc[n] = std::cos(2.0 * M_PI * n / c.n_fft); ts[n] = std::sin(2.0 * M_PI * n / c.n_fft); }
    ST_TRY(stt_alloc(s, (size_t)c.n_fft, &s->d_twc)); ST_TRY(stt_alloc(s, (size_t)c.n_fft, &s->d_tws));
    RT_HIP(ctx, hipMemcpy(s->d_twc, tc.data(), c.n_fft * 8, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(s->d_tws, ts.data(), c.n_fft * 8, hipMemcpyHostToDevice));
    s->d_window = SVEC(s, "fe.window");
    s->d_melT = SVEC(s, "fe.melT");
    ST_TRY(stt_alloc(s, (size_t)c.chunk_seconds * c.sample_rate, &s->pcm16k));
    ST_TRY(stt_alloc(s, (size_t)T2 * c.n_mels, &s->mel));
    ST_TRY(stt_alloc(s, (size_t)T2 * D, &s->c1));
    ST_TRY(stt_alloc(s, (size_t)T * D, &s->x)); ST_TRY(stt_alloc(s, (size_t)T * D, &s->xn)); ST_TRY(stt_alloc(s, (size_t)T * 3 * D, &s->qkv));
    ST_TRY(stt_alloc(s, (size_t)T * D, &s->q)); ST_TRY(stt_alloc(s, (size_t)T * D, &s->ao)); ST_TRY(stt_alloc(s, (size_t)T * c.ffn, &s->ff));
    ST_TRY(stt_alloc(s, (size_t)T * D, &s->enc_out));
    ST_TRY(stt_alloc(s, (size_t)c.vocab, &s->logits));
    const int R = 8;                                                  // decoder rows per pass: the forced prefix, then one
    ST_TRY(stt_alloc(s, (size_t)R * D, &s->dx)); ST_TRY(stt_alloc(s, (size_t)R * D, &s->dxn)); ST_TRY(stt_alloc(s, (size_t)R * 3 * D, &s->dqkv));
    ST_TRY(stt_alloc(s, (size_t)R * D, &s->dq)); ST_TRY(stt_alloc(s, (size_t)R * D, &s->dao)); ST_TRY(stt_alloc(s, (size_t)R * c.ffn, &s->dff));
    ST_TRY(stt_alloc(s, (size_t)T, &s->pos_seq)); ST_TRY(stt_alloc(s, (size_t)T, &s->pos_last)); ST_TRY(stt_alloc(s, (size_t)T, &s->slot0));
    ST_TRY(stt_alloc(s, (size_t)R, &s->d_tok)); ST_TRY(stt_alloc(s, (size_t)1, &s->d_gmax));
    hipLaunchKernelGGL(k_stt_fill_i32, dim3(8), dim3(256), 0, ctx->stream, s->pos_seq, T, 0, 1);
    hipLaunchKernelGGL(k_stt_fill_i32, dim3(8), dim3(256), 0, ctx->stream, s->pos_last, T, T - 1, 0);
    hipLaunchKernelGGL(k_stt_fill_i32, dim3(8), dim3(256), 0, ctx->stream, s->slot0, T, 0, 0);
    RT_HIP(ctx, hipGetLastError());
    // suppression mask: bit 0 = never (ids >= suppress_from except end-of-sequence), bit 1 = not as the first generated token
    std::vector<uint8_t> mask(c.vocab, 0);
    for (int i = 0; i < c.vocab; ++i)
        if (c.suppress_from > 0 && i >= c.suppress_from && i != c.eos_id) mask[i] |= 1;
    for (int i = 0; i < c.n_begin_suppress; ++i)
        if (c.begin_suppress[i] >= 0 && c.begin_suppress[i] < c.vocab) mask[c.begin_suppress[i]] |= 2;
    for (int32_t id : s->suppress_ids)
        if (id >= 0 && id < c.vocab && id != c.eos_id) mask[id] |= 1;
    ST_TRY(stt_alloc(s, (size_t)c.vocab, &s->d_mask));
    RT_HIP(ctx, hipMemcpy(s->d_mask, mask.data(), c.vocab, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->finalized = true;
    return RT_OK;
}

int rt_stt_log_mel(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate, float* d_mel) {
    if (!s || !d_mel || n_samples < 0 || (n_samples > 0 && !d_pcm) || sample_rate < 1000) return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_stt_log_mel: bad argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!s->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_stt_log_mel: not finalized");
    ST_TRY(stt_features(s, d_pcm, n_samples, sample_rate));
    RT_HIP(ctx, hipMemcpyAsync(d_mel, s->mel, (size_t)2 * s->cfg.n_ctx * s->cfg.n_mels * 4, hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_stt_encode(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate, float* d_states) {
    if (!s || !d_states || n_samples < 0 || (n_samples > 0 && !d_pcm) || sample_rate < 1000) return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_stt_encode: bad argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!s->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_stt_encode: not finalized");
    ST_TRY(stt_features(s, d_pcm, n_samples, sample_rate));
    ST_TRY(stt_encode(s));
    RT_HIP(ctx, hipMemcpyAsync(d_states, s->enc_out, (size_t)s->cfg.n_ctx * s->cfg.d_model * 4, hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_stt_set_suppress(rt_stt* s, const int32_t* h_ids, int32_t n) {
    if (!s || n < 0 || (n > 0 && !h_ids)) return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_stt_set_suppress: bad argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    if (s->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_stt_set_suppress: call before rt_stt_finalize (the mask is built there)");
    s->suppress_ids.assign(h_ids, h_ids + n);
    return RT_OK;
}

// Audio longer than one chunk is transcribed window by window (consecutive chunk_seconds windows of the INPUT, each through the
// resampler, the log-mel front-end, the encoder and its own greedy decode behind the forced prefix) and the ids are concatenated:
// the whole clip is heard, as with the reference's transcribers (faster-whisper walks 30-s windows, stt_validator.py:133-141),
// though not at their seek positions - those follow timestamp tokens, which the forced <|notimestamps|> prefix rules out.
int rt_stt_transcribe(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate, int32_t* h_tokens, int32_t max_tokens, int32_t* h_n_tokens,
                      float* d_first_logits) {
    if (!s || !h_tokens || !h_n_tokens || max_tokens < 1 || n_samples < 0 || (n_samples > 0 && !d_pcm) || sample_rate < 1000)
        return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_stt_transcribe: bad argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!s->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_stt_transcribe: not finalized");
    const rt_stt_config& c = s->cfg;
    *h_n_tokens = 0;
    const int64_t win = (int64_t)c.chunk_seconds * sample_rate;     // one window, in input samples
    int n = 0;
    for (int64_t off = 0; off == 0 || off < n_samples; off += win) {
        if (n >= max_tokens) break;
        const int64_t n_w = std::min<int64_t>(win, n_samples - off);
        ST_TRY(stt_features(s, n_w > 0 ? d_pcm + off : d_pcm, n_w, sample_rate));
        ST_TRY(stt_encode(s));
        // forced prefix in one pass, then one token per pass: the host reads each token (end-of-sequence decides when to stop)
        RT_HIP(ctx, hipMemcpyAsync(s->d_tok, c.prefix, c.n_prefix * 4, hipMemcpyHostToDevice, ctx->stream));
        ST_TRY(stt_decode_rows(s, c.n_prefix, 0));
        if (d_first_logits && off == 0) RT_HIP(ctx, hipMemcpyAsync(d_first_logits, s->logits, (size_t)c.vocab * 4, hipMemcpyDeviceToDevice, ctx->stream));
        const int budget = std::min(std::min(max_tokens - n, c.max_new_tokens), c.n_text_ctx - c.n_prefix);
        for (int step = 0; step < budget; ++step) {
            hipLaunchKernelGGL(k_stt_argmax, dim3(1), dim3(1024), 0, ctx->stream, s->logits, c.vocab, s->d_mask, step == 0 ? 1 : 0, s->d_tok);
            RT_HIP(ctx, hipGetLastError());
            int32_t tok = 0;
            RT_HIP(ctx, hipMemcpyAsync(&tok, s->d_tok, 4, hipMemcpyDeviceToHost, ctx->stream));
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (tok == c.eos_id) break;
            h_tokens[n++] = tok;
            if (step + 1 < budget) ST_TRY(stt_decode_rows(s, 1, c.n_prefix + step));
        }
    }
    *h_n_tokens = n;
    return RT_OK;
}

}  // extern "C"
