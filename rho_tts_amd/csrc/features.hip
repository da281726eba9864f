// Hand-crafted drift-classifier features from a waveform that already lives in HBM.
// Stands behind the front half of validation/classifier/trainer.py:23-96 (extract_features / _estimate_formants): what the
// reference computes with librosa on a temporary WAV - 13 MFCCs per frame, the pYIN difference function, a Burg LPC of the
// mid-file frame - is computed where the generated segment is, and only a few kilobytes leave the GPU:
//
//   PCM at the TTS rate -> windowed-sinc resampler (16 kHz; the resampler of the speech-to-text front-end)
//     -> MFCC: frames of 2048 (hop 512, zero-padded centre), periodic Hann, 2048-point DFT in float64, power spectrum,
//        128 slaney mel filters, 10 log10 (floor 1e-10), max - 80 dB floor, DCT-II (orthonormal) rows 0..12 -> mean and
//        standard deviation of every coefficient over the frames (26 numbers)
//     -> pYIN front half: per frame d(tau) = sum_{j=1..1024} (x[j] - x[j + tau])^2, cumulative-mean-normalised, for the lags
//        min_period .. max_period ([frames][lags] float64; troughs, thresholds and the Viterbi pass over 1202 states are a
//        few hundred kilobytes of host arithmetic: rho_tts_amd/features.py)
//     -> LPC: pre-emphasis 0.97 in float32, 400-sample symmetric-Hann frame about the middle sample, Burg's recursion in
//        float64 -> order + 1 coefficients (their roots - an 18 x 18 eigenproblem - are taken on the host)
//
// Nothing here is on the hot path of generation (one call per validated segment, ~0.3 ms of GPU time for 3.5 s of audio), so
// the kernels are the simple forms: direct DFT and direct difference sums in float64, one workgroup per frame.
// PARITY UNPINNED (librosa absent): the definitions are those of oracle/features.py, which restates librosa 0.10's defaults.
#include <algorithm>
#include <cmath>

#include "kernels.h"

struct rt_features {
    rt_ctx* ctx = nullptr;
    // tables
    double *d_twc = nullptr, *d_tws = nullptr;     // cos / sin(2 pi n / 2048)
    float* d_melT = nullptr;                       // [1025][128]
    double* d_dct = nullptr;                       // [13][128]
    float* d_resamp = nullptr;
    int rs_in = 0, rs_L = 0, rs_M = 0, rs_taps = 0, rs_half = 0;
    // workspaces (grown on demand)
    float* pcm16k = nullptr;
    size_t pcm_cap = 0;
    float* logmel = nullptr;                       // [frames][128]
    double* mfcc = nullptr;                        // [frames][13]
    double* cmnd = nullptr;                        // [frames][lags]
    size_t frame_cap = 0;
    int* d_gmax = nullptr;
    double *d_stats = nullptr, *d_lpc = nullptr;   // [26], [order + 1]
};

namespace {

constexpr int F_SR = 16000, F_NFFT = 2048, F_HOP = 512, F_BINS = F_NFFT / 2 + 1, F_MELS = 128, F_MFCC = 13;
constexpr int P_FRAME = 2048, P_WIN = 1024, P_HOP = 512;
constexpr int LPC_MAX = 32, LPC_FRAME = 400;

#define FT_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

__device__ __forceinline__ int ft_ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ft_unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// y[n] = sum_j x[base(n) + j - half] h[phase(n)][j] (csrc/stt.hip k_resample: the same polyphase filter, float64 accumulation)
__global__ void k_feat_resample(const float* __restrict__ x, int64_t n_in, float* __restrict__ y, int64_t n_out, int L, int M, int taps, int half,
                                const float* __restrict__ h) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (int64_t)gridDim.x * blockDim.x) {
        const int64_t num = n * M, base = num / L;
        const float* hp = h + (num % L) * taps;
        double acc = 0.0;
        for (int j = 0; j < taps; ++j) {
            const int64_t t = base + j - half;
            if (t >= 0 && t < n_in) acc += (double)x[t] * (double)hp[j];
        }
        y[n] = (float)acc;
    }
}

// One MFCC frame per workgroup: frame f covers samples [f hop - 1024, f hop + 1024) of the signal (zeros outside), times the periodic
// Hann window; bin k on thread k (+ 256 i): direct DFT with the twiddle index k n mod 2048; mel filters; 10 log10.
__global__ __launch_bounds__(256) void k_feat_logmel(const float* __restrict__ pcm, int64_t n, const double* __restrict__ twc, const double* __restrict__ tws,
                                                     const float* __restrict__ melT, float* __restrict__ logmel, int* __restrict__ gmax) {
    extern __shared__ double fsh[];               // xw[2048] | c[2048] | s[2048] | power[1025]
    double* xw = fsh;
    double* tc = fsh + F_NFFT;
    double* ts = tc + F_NFFT;
    double* pw = ts + F_NFFT;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int64_t start = (int64_t)f * F_HOP - F_NFFT / 2;
    for (int i = tid; i < F_NFFT; i += 256) {
        const int64_t t = start + i;
        const double v = (t >= 0 && t < n) ? (double)pcm[t] : 0.0;
        tc[i] = twc[i];
        ts[i] = tws[i];
        xw[i] = v * (0.5 - 0.5 * tc[i]);          // periodic Hann: 0.5 - 0.5 cos(2 pi i / 2048)
    }
    __syncthreads();
    for (int k = tid; k < F_BINS; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int i = 0; i < F_NFFT; ++i) {
            re += xw[i] * tc[idx];
            im -= xw[i] * ts[idx];
            idx = (idx + k) & (F_NFFT - 1);
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    float best = -INFINITY;
    for (int m = tid; m < F_MELS; m += 256) {
        double acc = 0.0;
        for (int k = 0; k < F_BINS; ++k) acc += (double)melT[(int64_t)k * F_MELS + m] * pw[k];
        const float v = (float)(10.0 * log10(fmax(acc, 1e-10)));
        logmel[(int64_t)f * F_MELS + m] = v;
        best = fmaxf(best, v);
    }
    best = wave_max_f32(best);
    if ((tid & 63) == 0 && best > -INFINITY) atomicMax(gmax, ft_ordered(best));
}

// mfcc[f][c] = sum_m dct[c][m] max(logmel[f][m], gmax - 80): one frame per workgroup of 64 threads (13 of them busy: tiny)
__global__ __launch_bounds__(64) void k_feat_dct(const float* __restrict__ logmel, const int* __restrict__ gmax, const double* __restrict__ dct,
                                                 double* __restrict__ mfcc) {
    const int f = blockIdx.x, c = threadIdx.x;
    if (c >= F_MFCC) return;
    const float floor_db = ft_unordered(*gmax) - 80.0f;
    double acc = 0.0;
    for (int m = 0; m < F_MELS; ++m) acc += dct[c * F_MELS + m] * (double)fmaxf(logmel[(int64_t)f * F_MELS + m], floor_db);
    mfcc[(int64_t)f * F_MFCC + c] = acc;
}

// stats[c] = mean over the frames, stats[13 + c] = population standard deviation (np.std, ddof 0); one wave per coefficient
__global__ __launch_bounds__(64) void k_feat_stats(const double* __restrict__ mfcc, int n_frames, double* __restrict__ stats) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int f = lane; f < n_frames; f += 64) s += mfcc[(int64_t)f * F_MFCC + c];
    const double mean = wave_sum_f64(s) / (double)n_frames;
    double q = 0.0;
    for (int f = lane; f < n_frames; f += 64) { const double d = mfcc[(int64_t)f * F_MFCC + c] - mean; q += d * d; }
    q = wave_sum_f64(q);
    if (lane == 0) { stats[c] = mean; stats[F_MFCC + c] = sqrt(q / (double)n_frames); }
}

// One pitch frame per workgroup: x = samples [f hop - 1024, f hop + 1024) (zeros outside); d(tau) = sum_{j=1..1024} (x[j] - x[j+tau])^2
// for tau = 1 .. max_p (thread tau), prefix mean of d over 1 .. tau, out[f][tau - min_p] = d(tau) / (mean + tiny).
__global__ __launch_bounds__(512) void k_feat_cmnd(const float* __restrict__ pcm, int64_t n, int min_p, int max_p, double* __restrict__ out) {
    __shared__ double x[P_FRAME];
    __shared__ double d[1024 + 1];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int64_t start = (int64_t)f * P_HOP - P_FRAME / 2;
    for (int i = tid; i < P_FRAME; i += 512) {
        const int64_t t = start + i;
        x[i] = (t >= 0 && t < n) ? (double)pcm[t] : 0.0;
    }
    __syncthreads();
    for (int tau = 1 + tid; tau <= max_p; tau += 512) {
        double acc = 0.0;
        for (int j = 1; j <= P_WIN; ++j) { const double e = x[j] - x[j + tau]; acc += e * e; }
        d[tau] = acc;
    }
    __syncthreads();
    if (tid == 0) {                                 // running sums in lag order (338 additions: the order the oracle's cumsum takes)
        double run = 0.0;
        for (int tau = 1; tau <= max_p; ++tau) {
            run += d[tau];
            if (tau >= min_p) out[(int64_t)f * (max_p - min_p + 1) + (tau - min_p)] = d[tau] / (run / (double)tau + 2.2250738585072014e-308);
        }
    }
}

// Burg's recursion (Marple) on the pre-emphasised, symmetric-Hann-windowed 25-ms frame about the middle sample; one wave.
__global__ __launch_bounds__(64) void k_feat_lpc(const float* __restrict__ pcm, int64_t n, int order, double* __restrict__ a_out) {
    __shared__ double fwd[LPC_FRAME], bwd[LPC_FRAME], a[LPC_MAX + 1], prev[LPC_MAX + 1];
    const int lane = threadIdx.x;
    const int64_t c = n / 2;
    const int64_t lo = c - LPC_FRAME / 2 > 0 ? c - LPC_FRAME / 2 : 0, hi = c + LPC_FRAME / 2 < n ? c + LPC_FRAME / 2 : n;
    const int len = (int)(hi - lo);
    if (len < 2) { if (lane <= order) a_out[lane] = lane == 0 ? 1.0 : 0.0; return; }
    // frame[i] = y_pre[lo + i] * hanning(len)[i], y_pre[0] = y[0], y_pre[t] = y[t] - 0.97f y[t-1] in float32
    for (int i = lane; i < len; i += 64) {
        const int64_t t = lo + i;
        const float yp = t == 0 ? pcm[0] : pcm[t] - 0.97f * pcm[t - 1];
        const double w = len > 1 ? 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(len - 1)) : 1.0;
        const double v = (double)yp * w;
        if (i >= 1) fwd[i - 1] = v;
        if (i < len - 1) bwd[i] = v;
    }
    for (int i = lane; i <= LPC_MAX; i += 64) { a[i] = i == 0 ? 1.0 : 0.0; prev[i] = a[i]; }
    __syncthreads();
    int m = len - 1;                                // live length of fwd / bwd
    double den;
    {
        double s = 0.0;
        for (int i = lane; i < m; i += 64) s += fwd[i] * fwd[i] + bwd[i] * bwd[i];
        den = wave_sum_f64(s);
    }
    double* pa = a;
    double* pp = prev;
    for (int it = 0; it < order && m > 0; ++it) {
        double s = 0.0;
        for (int i = lane; i < m; i += 64) s += bwd[i] * fwd[i];
        const double k = -2.0 * wave_sum_f64(s) / (den + 2.2250738585072014e-308);
        double* t = pa; pa = pp; pp = t;            // a <-> prev
        for (int j = 1 + lane; j <= it + 1; j += 64) pa[j] = pp[j] + k * pp[it - j + 1];
        if (lane == 0) pa[0] = 1.0;
        // fwd <- fwd + k bwd, bwd <- bwd + k fwd(old); then drop fwd[0] and bwd[last]
        double f0v = 0.0, blast = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double fo = fwd[i], bo = bwd[i];
            const double fn = fo + k * bo, bn = bo + k * fo;
            if (i == 0) f0v = fn;
            if (i == m - 1) blast = bn;
            bwd[i] = bn;
            fwd[i] = fn;
        }
        f0v = wave_sum_f64(f0v);
        blast = wave_sum_f64(blast);
        den = (1.0 - k * k) * den - blast * blast - f0v * f0v;
        __syncthreads();
        // shift fwd down by one (fwd = fwd[1:]); bwd = bwd[:-1] is a shorter live length only
        double keep[(LPC_FRAME + 63) / 64];
        int q = 0;
        for (int i = lane; i + 1 < m; i += 64) keep[q++] = fwd[i + 1];
        __syncthreads();
        q = 0;
        for (int i = lane; i + 1 < m; i += 64) fwd[i] = keep[q++];
        __syncthreads();
        --m;
    }
    __syncthreads();
    for (int i = lane; i <= order; i += 64) a_out[i] = pa[i];
}

int feat_tables(rt_features* s) {
    rt_ctx* ctx = s->ctx;
    std::vector<double> c(F_NFFT), sn(F_NFFT);
    for (int i = 0; i < F_NFFT; ++i) { c[i] = std::cos(2.0 * M_PI * i / F_NFFT); sn[i] = std::sin(2.0 * M_PI * i / F_NFFT); }
    // slaney mel scale, slaney normalisation, 0 .. sr / 2 (librosa.filters.mel defaults)
    auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + std::log(f / 1000.0) / (std::log(6.4) / 27.0) : f / (200.0 / 3.0); };
    auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0)) : (200.0 / 3.0) * m; };
    std::vector<double> mel_f(F_MELS + 2);
    const double m_hi = hz2mel(F_SR / 2.0);
    for (int i = 0; i < F_MELS + 2; ++i) mel_f[i] = mel2hz(m_hi * i / (F_MELS + 1));
    std::vector<float> melT((size_t)F_BINS * F_MELS);
    for (int k = 0; k < F_BINS; ++k) {
        const double fk = (F_SR / 2.0) * k / (F_BINS - 1);
        for (int m = 0; m < F_MELS; ++m) {
            const double lower = (fk - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - fk) / (mel_f[m + 2] - mel_f[m + 1]);
            melT[(size_t)k * F_MELS + m] = (float)(std::max(0.0, std::min(lower, upper)) * (2.0 / (mel_f[m + 2] - mel_f[m])));
        }
    }
    std::vector<double> dct((size_t)F_MFCC * F_MELS);
    for (int k = 0; k < F_MFCC; ++k)
        for (int n = 0; n < F_MELS; ++n)
            dct[(size_t)k * F_MELS + n] = std::cos(M_PI * k * (2 * n + 1) / (2.0 * F_MELS)) * std::sqrt(2.0 / F_MELS) * (k == 0 ? std::sqrt(0.5) : 1.0);
    RT_HIP(ctx, hipMalloc((void**)&s->d_twc, F_NFFT * 8));
    RT_HIP(ctx, hipMalloc((void**)&s->d_tws, F_NFFT * 8));
    RT_HIP(ctx, hipMalloc((void**)&s->d_melT, melT.size() * 4));
    RT_HIP(ctx, hipMalloc((void**)&s->d_dct, dct.size() * 8));
    RT_HIP(ctx, hipMalloc((void**)&s->d_gmax, 4));
    RT_HIP(ctx, hipMalloc((void**)&s->d_stats, 2 * F_MFCC * 8));
    RT_HIP(ctx, hipMalloc((void**)&s->d_lpc, (LPC_MAX + 1) * 8));
    RT_HIP(ctx, hipMemcpy(s->d_twc, c.data(), F_NFFT * 8, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(s->d_tws, sn.data(), F_NFFT * 8, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(s->d_melT, melT.data(), melT.size() * 4, hipMemcpyHostToDevice));
    RT_HIP(ctx, hipMemcpy(s->d_dct, dct.data(), dct.size() * 8, hipMemcpyHostToDevice));
    return RT_OK;
}

// the speech-to-text front-end's resampler (csrc/stt.hip stt_resampler: same taps, same float32 table)
int feat_resampler(rt_features* s, int sr_in) {
    if (s->rs_in == sr_in && s->d_resamp) return RT_OK;
    int a = sr_in, b = F_SR;
    while (b) { const int t = a % b; a = b; b = t; }
    const int L = F_SR / a, M = sr_in / a;
    const double width = 6.0, rolloff = 0.99, base = std::min(sr_in, F_SR) * rolloff;
    const int half = (int)std::ceil(width * sr_in / base), taps = 2 * half + 1;
    std::vector<float> h((size_t)L * taps);
    for (int p = 0; p < L; ++p)
        for (int j = 0; j < taps; ++j) {
            const double t = ((double)(j - half) - (double)p / L) * base / sr_in;
            double v = 0.0;
            if (std::fabs(t) < width) {
                const double w = std::cos(t * M_PI / width / 2.0);
                v = (t == 0.0 ? 1.0 : std::sin(M_PI * t) / (M_PI * t)) * w * w * base / sr_in;
            }
            h[(size_t)p * taps + j] = (float)v;
        }
    if (s->d_resamp) (void)hipFree(s->d_resamp);
    s->d_resamp = nullptr;
    RT_HIP(s->ctx, hipMalloc((void**)&s->d_resamp, h.size() * 4));
    RT_HIP(s->ctx, hipMemcpy(s->d_resamp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    s->rs_in = sr_in; s->rs_L = L; s->rs_M = M; s->rs_taps = taps; s->rs_half = half;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_features_create(rt_ctx* ctx, rt_features** out) {
    if (!ctx || !out) return rt_fail(ctx, RT_ERR_INVALID, "rt_features_create: null argument");
    *out = nullptr;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rt_features* s = new rt_features();
    s->ctx = ctx;
    const int rc = feat_tables(s);
    if (rc) { delete s; return rc; }
    *out = s;
    return RT_OK;
}

int rt_features_destroy(rt_features* s) {
    if (!s) return RT_OK;
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : {(void*)s->d_twc, (void*)s->d_tws, (void*)s->d_melT, (void*)s->d_dct, (void*)s->d_resamp, (void*)s->pcm16k, (void*)s->logmel,
                    (void*)s->mfcc, (void*)s->cmnd, (void*)s->d_gmax, (void*)s->d_stats, (void*)s->d_lpc})
        if (p) (void)hipFree(p);
    delete s;
    return RT_OK;
}

int rt_features_geometry(int32_t pitch_sr, double fmin, double fmax, int32_t* min_period, int32_t* max_period) {
    if (pitch_sr < 1000 || !(fmin > 0) || !(fmax > fmin) || !min_period || !max_period) return RT_ERR_INVALID;
    *min_period = std::max((int)std::floor(pitch_sr / fmax), 1);
    *max_period = std::min((int)std::ceil(pitch_sr / fmin), P_FRAME - P_WIN - 1);
    return RT_OK;
}

int rt_features_extract(rt_features* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate_in, int32_t min_period, int32_t max_period,
                        int32_t lpc_order, double* h_mfcc_stats26, int32_t* h_n_mfcc_frames, double* h_cmnd, int32_t cmnd_cap_frames,
                        int32_t* h_n_pitch_frames, double* h_lpc) {
    if (!s || !d_pcm || n_samples < 2 || sample_rate_in < 1000 || !h_mfcc_stats26 || !h_n_pitch_frames || !h_lpc || lpc_order < 1 || lpc_order > LPC_MAX ||
        min_period < 1 || max_period <= min_period || max_period > P_FRAME - P_WIN - 1)
        return rt_fail(s ? s->ctx : nullptr, RT_ERR_INVALID, "rt_features_extract: bad argument");
    rt_ctx* ctx = s->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const float* src = d_pcm;
    int64_t n16 = n_samples;
    if (sample_rate_in != F_SR) {
        FT_TRY(feat_resampler(s, sample_rate_in));
        n16 = (n_samples * s->rs_L + s->rs_M - 1) / s->rs_M;
        if ((size_t)n16 > s->pcm_cap) {
            if (s->pcm16k) (void)hipFree(s->pcm16k);
            s->pcm16k = nullptr; s->pcm_cap = 0;
            RT_HIP(ctx, hipMalloc((void**)&s->pcm16k, (size_t)n16 * 4));
            s->pcm_cap = (size_t)n16;
        }
        hipLaunchKernelGGL(k_feat_resample, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, d_pcm, n_samples, s->pcm16k,
                           n16, s->rs_L, s->rs_M, s->rs_taps, s->rs_half, s->d_resamp);
        RT_HIP(ctx, hipGetLastError());
        src = s->pcm16k;
    }
    if (n16 < 2) return rt_fail(ctx, RT_ERR_INVALID, "rt_features_extract: fewer than two samples at 16 kHz");
    // centre-padded framing: 1 + n // hop frames for both the MFCC and the pitch front end (same frame and hop lengths)
    const int n_frames = 1 + (int)(n16 / F_HOP);
    const int n_lags = max_period - min_period + 1;
    if (h_cmnd && cmnd_cap_frames < n_frames) return rt_fail(ctx, RT_ERR_LENGTH, "rt_features_extract: %d pitch frames, room for %d", n_frames, cmnd_cap_frames);
    if ((size_t)n_frames > s->frame_cap) {
        for (void* p : {(void*)s->logmel, (void*)s->mfcc, (void*)s->cmnd}) if (p) (void)hipFree(p);
        s->logmel = nullptr; s->mfcc = nullptr; s->cmnd = nullptr; s->frame_cap = 0;
        RT_HIP(ctx, hipMalloc((void**)&s->logmel, (size_t)n_frames * F_MELS * 4));
        RT_HIP(ctx, hipMalloc((void**)&s->mfcc, (size_t)n_frames * F_MFCC * 8));
        RT_HIP(ctx, hipMalloc((void**)&s->cmnd, (size_t)n_frames * (P_FRAME - P_WIN) * 8));
        s->frame_cap = (size_t)n_frames;
    }
    const int init = (int)0x80000000;
    RT_HIP(ctx, hipMemcpyAsync(s->d_gmax, &init, 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_feat_logmel, dim3(n_frames), dim3(256), (size_t)(3 * F_NFFT + F_BINS) * sizeof(double), ctx->stream, src, n16, s->d_twc, s->d_tws,
                       s->d_melT, s->logmel, s->d_gmax);
    hipLaunchKernelGGL(k_feat_dct, dim3(n_frames), dim3(64), 0, ctx->stream, s->logmel, s->d_gmax, s->d_dct, s->mfcc);
    hipLaunchKernelGGL(k_feat_stats, dim3(F_MFCC), dim3(64), 0, ctx->stream, s->mfcc, n_frames, s->d_stats);
    if (h_cmnd) hipLaunchKernelGGL(k_feat_cmnd, dim3(n_frames), dim3(512), 0, ctx->stream, src, n16, min_period, max_period, s->cmnd);
    hipLaunchKernelGGL(k_feat_lpc, dim3(1), dim3(64), 0, ctx->stream, src, n16, lpc_order, s->d_lpc);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(h_mfcc_stats26, s->d_stats, 2 * F_MFCC * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (h_cmnd) RT_HIP(ctx, hipMemcpyAsync(h_cmnd, s->cmnd, (size_t)n_frames * n_lags * 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(h_lpc, s->d_lpc, (size_t)(lpc_order + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_n_mfcc_frames) *h_n_mfcc_frames = n_frames;
    *h_n_pitch_frames = n_frames;
    return RT_OK;
}

}  // extern "C"
