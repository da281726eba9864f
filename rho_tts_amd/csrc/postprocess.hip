// Fused audio post-processing for gfx950: one 1024-thread workgroup per text item runs the
// whole per-item tail of the reference pipeline in ONE launch for the whole batch
//   trim bounds -> DC removal -> (equal-power crossfade join, pauses) -> end fades
//   -> 2-s windowed loudness correction -> -23 dBFS -> tanh soft clip -> decay ratio
// Reference behaviour followed (paths relative to /root/reference/src/rho_tts/):
//   base_tts.py:348-392, 394-399, 401-433, 435-536, 297-323; providers/qwen.py:268-378.
//
// Roofline: HBM streaming, 12 B/sample algorithmic (two reads + one write); the serial
// dependency chain of five global reductions is kept inside the workgroup (barriers instead of
// launches), samples are re-read from L2.  Arithmetic type: f32 samples, f64 reductions.
//
// Exactness contract (tests/test_post_gpu.py): integer outputs (trim bounds, lengths, flags)
// are exact — the frame mean-square is accumulated in f32 in sample order with contraction off,
// which is bit-identical to ATen's avg_pool1d on CPU — samples agree to 2e-6 absolute.
#include "common.h"

#pragma clang fp contract(off)

#define RT_TRY_POST(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

struct SegDesc {
    const float* x;
    int64_t n;
    uint32_t trim;  // RT_POST_TRIM_* for this segment (leaf calls / single-segment items)
    uint32_t pad;
};
struct ItemDesc {
    int32_t first_seg;
    int32_t n_seg;
    float* out;
    int64_t out_cap;
    double* gains;  // scratch for per-window gains, capacity out_cap / loud_window + 2
};
struct SegWork {
    int64_t start, end;
    float dc;
    int32_t silent;
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum_f64(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) t += sh[i];
    return t;
}

// torch.linspace(a, b, n)[i] in float32 (ATen RangeFactories: symmetric two-sided evaluation)
__device__ __forceinline__ float linspace_f32(float a, float b, int n, int i) {
    if (n <= 1) return a;
    const float step = (b - a) / (float)(n - 1);
    return (i < n / 2) ? __fadd_rn(a, __fmul_rn(step, (float)i)) : __fsub_rn(b, __fmul_rn(step, (float)(n - 1 - i)));
}

__device__ __forceinline__ double sumsq_range(const float* p, int64_t lo, int64_t hi) {
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
        const double v = (double)p[i];
        acc += v * v;
    }
    return acc;
}


// The reference decides "frame is audible" by sqrt(mean square) > threshold in float32 (base_tts.py:369-377).  A frame sitting
// within an ulp of the threshold (the edge_thresh fixture: a constant signal AT -50 dB) then depends on the last bit of the
// square root, and the device's sqrt is not the host's.  With a CORRECTLY ROUNDED sqrt the test is monotone in the mean square,
// so it is evaluated on the mean square against the largest float m with sqrt_rn(m) <= threshold - computed here on the host,
// exactly - and no square root runs on the device at all.
static float energy_threshold(float thr) {
    if (!(thr > 0.f)) return thr == 0.f ? 0.f : (thr < 0.f ? -1.f : thr);   // 0: any energy at all; < 0: every frame; NaN: none
    auto sqrt_rn = [](float m) { return (float)sqrt((double)m); };
    float m = thr * thr;
    while (sqrt_rn(m) <= thr) m = nextafterf(m, INFINITY);
    while (sqrt_rn(m) > thr) m = nextafterf(m, 0.f);
    return m;                                                // largest m with sqrt_rn(m) <= thr:  e > thr  <=>  mean square > m
}

// thr_ms: the frame test "sqrt(mean square) > threshold" as a test on the mean square itself (energy_threshold above)
__global__ __launch_bounds__(kThreads) void k_post_item(rt_post_params P, float thr_ms, const ItemDesc* __restrict__ items,
                                                        const SegDesc* __restrict__ segs, SegWork* __restrict__ work,
                                                        rt_post_stats* __restrict__ stats) {
    __shared__ double sh_red[kWaves];
    __shared__ int sh_first, sh_last, sh_flag;
    __shared__ double sh_bcast[4];

    const ItemDesc it = items[blockIdx.x];
    const int k = it.n_seg;
    const SegDesc* sg = segs + it.first_seg;
    SegWork* wk = work + it.first_seg;
    float* out = it.out;
    const uint32_t S = P.stages;
    const bool join = (S & RT_POST_JOIN) && k > 1;
    const int tid = threadIdx.x;

    rt_post_stats st;
    st.out_len = 0;
    st.first_trim_start = 0;
    st.first_trim_end = 0;
    st.decay_ratio = 1.0;
    st.rms_out = 0.0;
    st.decay_ok = 1;
    st.all_silent = 0;
    st.fallback_concat = 0;
    st.windowed_applied = 0;

    // ------------------------------------------------------------ A: trim bounds
    int n_silent = 0;
    for (int s = 0; s < k; ++s) {
        const float* x = sg[s].x;
        const int64_t n = sg[s].n;
        uint32_t tf;
        if (join) tf = (s == 0) ? RT_POST_TRIM_END : (s == k - 1) ? RT_POST_TRIM_START : (RT_POST_TRIM_START | RT_POST_TRIM_END);
        else tf = sg[s].trim & (RT_POST_TRIM_START | RT_POST_TRIM_END);
        const bool do_trim = P.trim_enabled && n > 0 && tf != 0;
        if (tid == 0) { sh_first = 0x7fffffff; sh_last = -1; }
        __syncthreads();
        if (do_trim) {
            const int W = P.window, H = P.window / 2;
            int64_t nf = (n + 2 * (int64_t)H - W) / H + 1;
            if (nf < 1) nf = 1;
            const float fw = (float)W;
            int my_first = 0x7fffffff, my_last = -1;
            for (int64_t f = tid; f < nf; f += kThreads) {
                int64_t lo = f * H - H, hi = lo + W;
                if (lo < 0) lo = 0;
                if (hi > n) hi = n;
                float acc = 0.0f;
                for (int64_t i = lo; i < hi; ++i) {
                    const float v = x[i];
                    acc = __fadd_rn(acc, __fmul_rn(v, v));  // same order and rounding as ATen avg_pool1d (CPU)
                }
                if (__fdiv_rn(acc, fw) > thr_ms) {        // == correctly rounded sqrt(mean square) > threshold, without a device sqrt
                    if ((int)f < my_first) my_first = (int)f;
                    if ((int)f > my_last) my_last = (int)f;
                }
            }
            if (my_last >= 0) { atomicMin(&sh_first, my_first); atomicMax(&sh_last, my_last); }
        }
        __syncthreads();
        if (tid == 0) {
            int64_t a = 0, b = n;
            int silent = 0;
            if (do_trim) {
                if (sh_last < 0) {  // nothing above threshold: keep one window (base_tts.py:379-380)
                    a = 0; b = n < P.window ? n : P.window; silent = 1;
                } else {
                    if (tf & RT_POST_TRIM_START) a = (int64_t)sh_first * P.window / 2;
                    if (tf & RT_POST_TRIM_END) b = ((int64_t)sh_last + 2) * P.window / 2;
                    if (a > n) a = n;
                    if (b > n) b = n;
                    if (b < a) b = a;
                }
            }
            wk[s].start = a; wk[s].end = b; wk[s].dc = 0.0f; wk[s].silent = silent;
        }
        __syncthreads();
        n_silent += wk[s].silent;
    }
    st.all_silent = (k > 0 && n_silent == k) ? 1 : 0;
    // Reference quirk: all-silent segment mixed with others => torch.cat of 1-D and 2-D raises and the
    // reference concatenates the ORIGINAL segments (base_tts.py:530-533).
    const bool fallback = join && n_silent > 0 && n_silent < k;
    st.fallback_concat = fallback ? 1 : 0;
    if (fallback) {
        if (tid == 0) for (int s = 0; s < k; ++s) { wk[s].start = 0; wk[s].end = sg[s].n; wk[s].silent = 0; }
        __syncthreads();
    }
    if (k > 0) { st.first_trim_start = wk[0].start; st.first_trim_end = wk[0].end; }

    // ------------------------------------------------------------ B: DC offset per segment
    if ((S & RT_POST_DC) && !fallback) {
        for (int s = 0; s < k; ++s) {
            const int64_t a = wk[s].start, b = wk[s].end;
            double acc = 0.0;
            for (int64_t i = a + tid; i < b; i += kThreads) acc += (double)sg[s].x[i];
            const double tot = block_sum(acc, sh_red);
            if (tid == 0) wk[s].dc = (b > a) ? (float)(tot / (double)(b - a)) : 0.0f;
            __syncthreads();
        }
    }

    // ------------------------------------------------------------ C: assemble into out
    int64_t L = 0;
    if (!join) {
        // leaf call / single segment: out = x[start:end] - dc
        if (k >= 1) {
            const int64_t a = wk[0].start, b = wk[0].end;
            const float dc = wk[0].dc;
            const float* x = sg[0].x;
            for (int64_t i = tid; i < b - a; i += kThreads) out[i] = __fsub_rn(x[a + i], dc);
            L = b - a;
        }
    } else if (fallback) {
        for (int s = 0; s < k; ++s) {
            const float* x = sg[s].x;
            for (int64_t i = tid; i < sg[s].n; i += kThreads) out[L + i] = x[i];
            L += sg[s].n;
        }
    } else {
        const int64_t XF = P.crossfade;
        const float half_pi = (float)1.5707963267948966;
        for (int s = 0; s < k; ++s) {
            const float* x = sg[s].x + wk[s].start;
            const int64_t Ls = wk[s].end - wk[s].start;
            const float dc = wk[s].dc;
            if (s == 0) {
                const int64_t body = (Ls > XF) ? Ls - XF : Ls;  // base_tts.py:485-488
                for (int64_t i = tid; i < body; i += kThreads) out[L + i] = __fsub_rn(x[i], dc);
                L += body;
                continue;
            }
            const float* xp = sg[s - 1].x + wk[s - 1].start;
            const int64_t Lp = wk[s - 1].end - wk[s - 1].start;
            const float dcp = wk[s - 1].dc;
            int64_t ov = XF < Lp ? XF : Lp;
            if (Ls < ov) ov = Ls;
            if (ov > 10) {
                for (int64_t j = tid; j < ov; j += kThreads) {
                    const float down = cosf(linspace_f32(0.0f, half_pi, (int)ov, (int)j));
                    const float up = cosf(linspace_f32(half_pi, 0.0f, (int)ov, (int)j));
                    const float a = __fmul_rn(__fsub_rn(xp[Lp - ov + j], dcp), down);
                    const float b = __fmul_rn(__fsub_rn(x[j], dc), up);
                    out[L + j] = __fadd_rn(a, b);
                }
                L += ov;
                int64_t rest_end = Ls;
                if (s < k - 1 && Ls > ov + XF) rest_end = Ls - XF;  // base_tts.py:507-513
                for (int64_t i = ov + tid; i < rest_end; i += kThreads) out[L + (i - ov)] = __fsub_rn(x[i], dc);
                L += rest_end - ov;
                if (P.pause > 0 && s < k - 1) {
                    for (int64_t i = tid; i < P.pause; i += kThreads) out[L + i] = 0.0f;
                    L += P.pause;
                }
            } else {
                for (int64_t i = tid; i < Ls; i += kThreads) out[L + i] = __fsub_rn(x[i], dc);
                L += Ls;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------ C2: end fades (base_tts.py:420-431)
    {
        const int F = P.fade;
        if (L > 0 && L >= 2 * (int64_t)F && F > 0) {
            const float pi_f = (float)3.141592653589793;
            if (S & RT_POST_FADE_IN)
                for (int i = tid; i < F; i += kThreads) {
                    const float c = __fmul_rn(0.5f, __fsub_rn(1.0f, cosf(linspace_f32(0.0f, pi_f, F, i))));
                    out[i] = __fmul_rn(out[i], c);
                }
            __syncthreads();  // L == 2F: the two ramps touch disjoint halves, but keep the order fixed
            if (S & RT_POST_FADE_OUT)
                for (int i = tid; i < F; i += kThreads) {
                    const float c = __fmul_rn(0.5f, __fadd_rn(1.0f, cosf(linspace_f32(0.0f, pi_f, F, i))));
                    out[L - F + i] = __fmul_rn(out[L - F + i], c);
                }
            __syncthreads();
        }
    }

    // ------------------------------------------------------------ D: loudness (qwen.py:268-378)
    if ((S & RT_POST_LOUDNESS) && L > 0) {
        const double tot0 = block_sum(sumsq_range(out, 0, L), sh_red);
        const float rms0 = __fsqrt_rn((float)(tot0 / (double)L));
        if (!(rms0 < 1e-8f)) {
            const int64_t W = P.loud_window;
            int apply_env = 0;
            int64_t nw = 0;
            if (W > 0 && L > 2 * W) {
                nw = L / W;
                double* g = it.gains;
                // per-window RMS: one wave per window, round-robin
                const int wv = tid >> 6, ln = tid & 63;
                for (int64_t w = wv; w < nw; w += kWaves) {
                    double acc = 0.0;
                    const float* p = out + w * W;
                    for (int64_t i = ln; i < W; i += 64) { const double v = (double)p[i]; acc += v * v; }
                    acc = wave_sum_f64(acc);
                    if (ln == 0) g[w] = (double)__fsqrt_rn((float)(acc / (double)W));
                }
                __syncthreads();
                if (tid == 0) {
                    int ok = 0;
                    const double ref = g[0];
                    if (nw >= 2 && !(ref < 1e-8)) {
                        const double cap = pow(10.0, P.max_gain_db / 20.0);
                        double mx = -1e300, mn = 1e300;
                        for (int64_t w = 0; w < nw; ++w) {
                            const double r = g[w];
                            double gn = 1.0;
                            if (!(r < 1e-8)) { gn = ref / r; if (gn > cap) gn = cap; }
                            g[w] = gn;
                            mx = gn > mx ? gn : mx;
                            mn = gn < mn ? gn : mn;
                        }
                        if (!(mx - mn < 0.05)) {
                            ok = 1;
                            // two passes of a 3-tap mean over interior points (qwen.py:365-370), in place with carries
                            for (int pass = 0; pass < 2; ++pass) {
                                double prev = g[0];
                                for (int64_t w = 1; w + 1 < nw; ++w) {
                                    const double cur = g[w];
                                    g[w] = (prev + cur + g[w + 1]) / 3;
                                    prev = cur;
                                }
                            }
                        }
                    }
                    sh_flag = ok;
                }
                __syncthreads();
                apply_env = sh_flag;
            }
            st.windowed_applied = apply_env;

            // envelope (np.interp in f64 over window centres) fused with the global sum of squares
            double acc = 0.0;
            if (apply_env) {
                const double* g = it.gains;
                const double Wd = (double)W;
                const double c0 = 0.5 * Wd, cN = ((double)(nw - 1) + 0.5) * Wd;
                for (int64_t i = tid; i < L; i += kThreads) {
                    const double xi = (double)i;
                    double e;
                    if (xi <= c0) e = g[0];
                    else if (xi >= cN) e = g[nw - 1];
                    else {
                        int64_t j = (int64_t)floor(xi / Wd - 0.5);
                        if (j < 0) j = 0;
                        if (j > nw - 2) j = nw - 2;
                        const double xj = ((double)j + 0.5) * Wd;
                        const double slope = (g[j + 1] - g[j]) / Wd;
                        e = slope * (xi - xj) + g[j];
                    }
                    const float v = __fmul_rn(out[i], (float)e);
                    out[i] = v;
                    acc += (double)v * (double)v;
                }
            } else {
                acc = sumsq_range(out, 0, L);
            }
            const double tot1 = block_sum(acc, sh_red);
            const float rms1 = __fsqrt_rn((float)(tot1 / (double)L));
            float gain = 1.0f;
            if (rms1 > 1e-8f) {
                const double cur_db = (double)__fmul_rn(20.0f, log10f(rms1));
                gain = (float)pow(10.0, (P.target_rms_db - cur_db) / 20.0);
            }
            const float amp = (float)P.max_amplitude;
            for (int64_t i = tid; i < L; i += kThreads) {
                float v = out[i];
                if (rms1 > 1e-8f) v = __fmul_rn(v, gain);
                out[i] = __fmul_rn(tanhf(__fdiv_rn(v, amp)), amp);
            }
            __syncthreads();
        }
    }

    // ------------------------------------------------------------ E: statistics
    {
        const double tot = block_sum(sumsq_range(out, 0, L), sh_red);
        st.rms_out = L > 0 ? sqrt(tot / (double)L) : 0.0;
    }
    if (S & RT_POST_DECAY) {
        const int64_t third = L / 3;
        if (L > 0 && third >= 1) {
            const double s1 = block_sum(sumsq_range(out, 0, third), sh_red);
            const double s2 = block_sum(sumsq_range(out, L - third, L), sh_red);
            const double r1 = (double)__fsqrt_rn((float)(s1 / (double)third));
            const double r2 = (double)__fsqrt_rn((float)(s2 / (double)third));
            if (!(r1 < 1e-8)) {
                st.decay_ratio = r2 / r1;
                st.decay_ok = st.decay_ratio >= P.decay_threshold ? 1 : 0;
            }
        }
    }
    st.out_len = L;
    if (tid == 0) stats[blockIdx.x] = st;
    (void)sh_bcast;
}

__global__ void k_pcm16(const float* __restrict__ in, int64_t n, int16_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = in[i];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        out[i] = (int16_t)__fmul_rn(v, 32767.0f);  // C cast truncates toward zero, like numpy astype(int16)
    }
}


// ---- sub-segment streaming: the segment-level leaves for ONE chunk of a segment that is still being decoded (an extension -
// the reference's stream() yields whole segments, base_tts.py:1132-1190 - in the ORDER the reference applies them per segment,
// base_tts.py:1170-1176: loudness (global gain to the target RMS + tanh soft clip, qwen.py:296-310), silence trim, DC removal,
// fades), with the two segment-wide quantities carried from the segment's first chunk instead of being measured on audio that
// does not exist yet:
//   gain  MEASURED on the first chunk that holds audible audio (target RMS / RMS of the raw chunk's audible span, held to +-30 dB;
//         a chunk with no audible frame measures nothing and reports gain 0: the caller passes RT_STREAM_MEASURE again with the
//         next one), applied unchanged to the later ones - no gain step at a chunk boundary; the 2-s windowed decay correction
//         needs the whole segment and is not part of streaming
//   dc    MEASURED on that chunk after its trim, subtracted from every chunk
// Leading silence is trimmed from the first chunk, trailing silence from the last; fade-in / fade-out only at those two edges.
// One 1024-thread workgroup per chunk, float64 reductions; frame energies (10-ms windows, hop 5 ms, count_include_pad as
// avg_pool1d: base_tts.py:366-377) in float64.
struct StreamState { double dc, gain; int64_t out_len, start; };
__device__ __forceinline__ float stream_y(const float* x, int64_t i, float gain, float amp) { return amp * tanhf(__fdiv_rn(__fmul_rn(x[i], gain), amp)); }
__global__ __launch_bounds__(kThreads) void k_stream_chunk(rt_post_params P, const float* __restrict__ x, int64_t n, uint32_t flags, StreamState* st,
                                                           float* __restrict__ out) {
    __shared__ double sh_red[kWaves];
    __shared__ long long sh_first, sh_last;
    double gain_d = st->gain, dc_d = st->dc;
    const float amp = (float)P.max_amplitude;
    bool measured = true;
    if (flags & RT_STREAM_MEASURE) {
        // The segment-wide gain comes from the AUDIBLE span of this chunk - from the first to the last 10-ms frame of the RAW chunk
        // above the silence threshold - not from the whole chunk: a first chunk is mostly lead-in silence and onset, and the RMS of
        // all of it over-amplifies everything that follows into the soft clip.  No audible frame: nothing is measured, the state
        // says so (gain 0) and the caller measures on the next chunk.  The gain is held to +-30 dB.
        const int W = P.window, hop = W / 2;
        const int64_t n_frames = n / hop + 1;
        const double thr2 = (double)P.silence_threshold * (double)P.silence_threshold;
        if (threadIdx.x == 0) { sh_first = n_frames; sh_last = -1; }
        __syncthreads();
        long long my_first = n_frames, my_last = -1;
        for (int64_t f = threadIdx.x; f < n_frames; f += kThreads) {
            int64_t a = f * hop - W / 2, b = a + W;
            if (a < 0) a = 0;
            if (b > n) b = n;
            double acc = 0.0;
            for (int64_t i = a; i < b; ++i) { const double v = (double)x[i]; acc += v * v; }
            if (acc / (double)W > thr2) { if (f < my_first) my_first = f; if (f > my_last) my_last = f; }
        }
        atomicMin(&sh_first, my_first);
        atomicMax(&sh_last, my_last);
        __syncthreads();
        const long long fa = sh_first, fb = sh_last;
        __syncthreads();                                             // (sh_first / sh_last are reused by the trim below)
        const bool whole = (flags & RT_STREAM_FADE_OUT) != 0;        // first AND last chunk: the whole segment is here - level it as the
        if (fb < 0 && !whole) {                                      // reference levels a segment, over all of it (qwen.py:296-306)
            measured = false;
            gain_d = 1.0;
        } else {
            int64_t a = (int64_t)fa * hop;
            int64_t b = ((int64_t)fb + 2) * hop;
            if (b > n) b = n;
            if (whole) { a = 0; b = n; }
            const double ss = block_sum(sumsq_range(x, a, b), sh_red);
            const double rms = sqrt(ss / (double)(b > a ? b - a : 1));
            gain_d = rms > 1e-8 ? pow(10.0, (P.target_rms_db - 20.0 * log10(rms)) / 20.0) : 1.0;
            if (!whole) gain_d = fmin(fmax(gain_d, 0.03162277660168379), 31.622776601683793);
        }
    } else if (!(gain_d > 0.0)) {
        gain_d = 1.0;                                                // (a caller that never measured: unity)
    }
    const float gain = (float)gain_d;
    // trim bounds on the loudness-corrected chunk
    int64_t lo = 0, hi = n;
    if ((flags & (RT_STREAM_TRIM_START | RT_STREAM_TRIM_END)) && P.trim_enabled && n > 0) {
        const int W = P.window, hop = W / 2;
        const int64_t n_frames = n / hop + 1;
        const double thr2 = (double)P.silence_threshold * (double)P.silence_threshold;
        if (threadIdx.x == 0) { sh_first = n_frames; sh_last = -1; }
        __syncthreads();
        long long my_first = n_frames, my_last = -1;
        for (int64_t f = threadIdx.x; f < n_frames; f += kThreads) {
            int64_t a = f * hop - W / 2, b = a + W;
            if (a < 0) a = 0;
            if (b > n) b = n;
            double acc = 0.0;
            for (int64_t i = a; i < b; ++i) { const double v = (double)stream_y(x, i, gain, amp); acc += v * v; }
            if (acc / (double)W > thr2) { if (f < my_first) my_first = f; if (f > my_last) my_last = f; }
        }
        atomicMin(&sh_first, my_first);
        atomicMax(&sh_last, my_last);
        __syncthreads();
        if (sh_last < 0) { lo = 0; hi = n < W ? n : W; }            // all silent: a window's worth stays (base_tts.py:379-380)
        else {
            if (flags & RT_STREAM_TRIM_START) lo = (int64_t)sh_first * hop;
            if (flags & RT_STREAM_TRIM_END) { hi = ((int64_t)sh_last + 2) * hop; if (hi > n) hi = n; }
            if (lo > hi) lo = hi;
        }
    }
    const int64_t m = hi - lo;
    if (flags & RT_STREAM_MEASURE) {
        double acc = 0.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) acc += (double)stream_y(x, i, gain, amp);
        dc_d = m > 0 ? block_sum(acc, sh_red) / (double)m : 0.0;
    }
    const float dc = (float)dc_d;
    const int F = P.fade;
    const bool fades = m >= 2 * (int64_t)F && F > 0;
    for (int64_t i = threadIdx.x; i < m; i += kThreads) {
        float v = __fsub_rn(stream_y(x, lo + i, gain, amp), dc);
        if (fades && (flags & RT_STREAM_FADE_IN) && i < F) v = __fmul_rn(v, 0.5f * (1.0f - cosf(linspace_f32(0.f, 3.14159265358979323846f, F, (int)i))));
        if (fades && (flags & RT_STREAM_FADE_OUT) && i >= m - F) v = __fmul_rn(v, 0.5f * (1.0f + cosf(linspace_f32(0.f, 3.14159265358979323846f, F, (int)(i - (m - F))))));
        out[i] = v;
    }
    if (threadIdx.x == 0) { st->dc = measured ? dc_d : 0.0; st->gain = measured ? gain_d : 0.0; st->out_len = m; st->start = lo; }
}

int post_impl(rt_ctx* ctx, const rt_post_params* p, int32_t n_items, const int32_t* first, const float* const* seg_ptr,
              const int64_t* seg_len, const uint8_t* seg_trim, float* const* out_ptr, const int64_t* out_cap,
              rt_post_stats* h_stats, bool host_buffers) {
    if (!ctx || !p || n_items < 0 || (n_items > 0 && (!first || !seg_ptr || !seg_len || !out_ptr || !out_cap || !h_stats)))
        return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: null argument");
    if (n_items == 0) return RT_OK;
    if (p->window < 2 || p->fade < 0 || p->crossfade < 0 || p->pause < 0 || p->sample_rate <= 0)
        return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: bad geometry (window=%d fade=%d)", p->window, p->fade);
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int n_seg = first[n_items];
    if (first[0] != 0 || n_seg < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: item_first_seg must start at 0");
    int64_t tot_in = 0, tot_out = 0, tot_gain = 0;
    for (int i = 0; i < n_items; ++i) {
        const int k = first[i + 1] - first[i];
        if (k < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: item_first_seg not monotone");
        if (k > 1 && !(p->stages & RT_POST_JOIN))
            return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: item %d has %d segments but RT_POST_JOIN is not set", i, k);
        const int64_t need = rt_post_capacity(p, k, seg_len + first[i]);
        if (out_cap[i] < need)
            return rt_fail(ctx, RT_ERR_LENGTH, "rt_post_process: output length capacity %lld < %lld for item %d",
                           (long long)out_cap[i], (long long)need, i);
        tot_out += out_cap[i];
        tot_gain += (p->loud_window > 0 ? out_cap[i] / p->loud_window : 0) + 2;
    }
    for (int s = 0; s < n_seg; ++s) {
        if (seg_len[s] < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_post_process: negative segment length");
        tot_in += seg_len[s];
    }

    // device scratch layout: [items][segs][work][stats][gains][(host mode) in samples][(host mode) out samples]
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_items = 0;
    const size_t o_segs = al(o_items + sizeof(ItemDesc) * n_items);
    const size_t o_work = al(o_segs + sizeof(SegDesc) * (n_seg > 0 ? n_seg : 1));
    const size_t o_stats = al(o_work + sizeof(SegWork) * (n_seg > 0 ? n_seg : 1));
    const size_t o_gains = al(o_stats + sizeof(rt_post_stats) * n_items);
    const size_t o_in = al(o_gains + sizeof(double) * tot_gain);
    const size_t o_out = al(o_in + (host_buffers ? sizeof(float) * tot_in : 0));
    const size_t total = al(o_out + (host_buffers ? sizeof(float) * tot_out : 0));
    void* dv = nullptr;
    int rc = rt_ctx_scratch(ctx, total, &dv);
    if (rc) return rc;
    char* d = (char*)dv;
    void* hv = nullptr;
    const size_t h_desc = o_work;  // items + segs
    rc = rt_ctx_pinned(ctx, h_desc + sizeof(rt_post_stats) * n_items, &hv);
    if (rc) return rc;
    char* h = (char*)hv;
    ItemDesc* hi = (ItemDesc*)(h + o_items);
    SegDesc* hs = (SegDesc*)(h + o_segs);
    const uint32_t leaf_trim = p->stages & (RT_POST_TRIM_START | RT_POST_TRIM_END);
    int64_t in_off = 0, out_off = 0, gain_off = 0;
    for (int s = 0; s < n_seg; ++s) {
        hs[s].n = seg_len[s];
        hs[s].trim = seg_trim ? (uint32_t)seg_trim[s] : leaf_trim;
        hs[s].pad = 0;
        if (host_buffers) {
            hs[s].x = (const float*)(d + o_in) + in_off;
            if (seg_len[s] > 0)
                RT_HIP(ctx, hipMemcpyAsync((void*)hs[s].x, seg_ptr[s], sizeof(float) * seg_len[s], hipMemcpyHostToDevice, ctx->stream));
            in_off += seg_len[s];
        } else {
            hs[s].x = seg_ptr[s];
        }
    }
    for (int i = 0; i < n_items; ++i) {
        hi[i].first_seg = first[i];
        hi[i].n_seg = first[i + 1] - first[i];
        hi[i].out_cap = out_cap[i];
        hi[i].out = host_buffers ? (float*)(d + o_out) + out_off : out_ptr[i];
        hi[i].gains = (double*)(d + o_gains) + gain_off;
        out_off += out_cap[i];
        gain_off += (p->loud_window > 0 ? out_cap[i] / p->loud_window : 0) + 2;
    }
    RT_HIP(ctx, hipMemcpyAsync(d, h, h_desc, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_post_item, dim3(n_items), dim3(kThreads), 0, ctx->stream, *p, energy_threshold(p->silence_threshold), (const ItemDesc*)(d + o_items),
                       (const SegDesc*)(d + o_segs), (SegWork*)(d + o_work), (rt_post_stats*)(d + o_stats));
    RT_HIP(ctx, hipGetLastError());
    rt_post_stats* hst = (rt_post_stats*)(h + h_desc);
    RT_HIP(ctx, hipMemcpyAsync(hst, d + o_stats, sizeof(rt_post_stats) * n_items, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(h_stats, hst, sizeof(rt_post_stats) * n_items);
    if (host_buffers) {
        for (int i = 0; i < n_items; ++i)
            if (h_stats[i].out_len > 0)
                RT_HIP(ctx, hipMemcpyAsync(out_ptr[i], hi[i].out, sizeof(float) * h_stats[i].out_len, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int64_t rt_post_capacity(const rt_post_params* p, int32_t n_segments, const int64_t* h_seg_len) {
    if (!p || n_segments < 0 || (n_segments > 0 && !h_seg_len)) return -1;
    int64_t t = 0;
    for (int i = 0; i < n_segments; ++i) t += h_seg_len[i] > 0 ? h_seg_len[i] : 0;
    if (n_segments > 2) t += (int64_t)(n_segments - 2) * (p->pause > 0 ? p->pause : 0);
    return t > 0 ? t : 1;
}

int rt_post_process(rt_ctx* ctx, const rt_post_params* p, int32_t n_items, const int32_t* h_item_first_seg,
                    const float* const* h_seg_ptr, const int64_t* h_seg_len, const uint8_t* h_seg_trim,
                    float* const* h_out_ptr, const int64_t* h_out_cap, rt_post_stats* h_stats) {
    return post_impl(ctx, p, n_items, h_item_first_seg, h_seg_ptr, h_seg_len, h_seg_trim, h_out_ptr, h_out_cap, h_stats, false);
}

int rt_post_process_host(rt_ctx* ctx, const rt_post_params* p, int32_t n_items, const int32_t* h_item_first_seg,
                         const float* const* h_seg_ptr, const int64_t* h_seg_len, const uint8_t* h_seg_trim,
                         float* const* h_out_ptr, const int64_t* h_out_cap, rt_post_stats* h_stats) {
    return post_impl(ctx, p, n_items, h_item_first_seg, h_seg_ptr, h_seg_len, h_seg_trim, h_out_ptr, h_out_cap, h_stats, true);
}

int rt_pcm16(rt_ctx* ctx, const float* d_in, int64_t n, int16_t* d_out) {
    if (!ctx || n < 0 || (n > 0 && (!d_in || !d_out))) return rt_fail(ctx, RT_ERR_INVALID, "rt_pcm16: null argument");
    if (n == 0) return RT_OK;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pcm16, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_in, n, d_out);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_stream_chunk(rt_ctx* ctx, const rt_post_params* p, const float* d_in, int64_t n, uint32_t flags, double* h_dc_gain, float* d_out,
                    int64_t* h_out_len) {
    if (!ctx || !p || !h_dc_gain || !h_out_len || n < 0 || (n > 0 && (!d_in || !d_out))) return rt_fail(ctx, RT_ERR_INVALID, "rt_stream_chunk: null argument");
    if (p->window < 2 || p->fade < 0) return rt_fail(ctx, RT_ERR_INVALID, "rt_stream_chunk: bad geometry (window=%d fade=%d)", p->window, p->fade);
    *h_out_len = 0;
    if (n == 0) return RT_OK;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    void *dv = nullptr, *hv = nullptr;
    RT_TRY_POST(rt_ctx_scratch(ctx, sizeof(StreamState), &dv));
    RT_TRY_POST(rt_ctx_pinned(ctx, sizeof(StreamState), &hv));
    StreamState* hs = (StreamState*)hv;
    hs->dc = h_dc_gain[0]; hs->gain = h_dc_gain[1]; hs->out_len = 0; hs->start = 0;
    RT_HIP(ctx, hipMemcpyAsync(dv, hs, sizeof(StreamState), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_stream_chunk, dim3(1), dim3(kThreads), 0, ctx->stream, *p, d_in, n, flags, (StreamState*)dv, d_out);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(hs, dv, sizeof(StreamState), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h_dc_gain[0] = hs->dc; h_dc_gain[1] = hs->gain;
    *h_out_len = hs->out_len;
    return RT_OK;
}

}  // extern "C"
