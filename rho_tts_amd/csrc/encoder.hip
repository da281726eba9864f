// Kernels of the conditioning front-end (reference audio -> codec codes + speaker embedding).  The convolutions
// and projections of the encoder run on the tiled MFMA GEMM (gemm.hip) in split precision; this file holds what is not a GEMM:
// the 1-channel input conv, the residual vector quantiser, the statistics pooling and the speaker head's matrix-vector products.
// Built with -ffp-contract=off: the quantiser's distance has a DEFINED float32 evaluation order (oracle/encoder.py rvq_level)
// and its argmin is compared bit for bit.
#include "kernels.h"

namespace {

// x[t][c] = b[c] + sum_k w[c][k] * pcm[t - (K - 1) + k]   (zeros before the clip); 256 threads = 256 / C time steps x C channels
__global__ __launch_bounds__(256) void k_enc_conv0(const float* __restrict__ pcm, int64_t T, int C, int K, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ x, bf16_t* __restrict__ hi,
                                                   bf16_t* __restrict__ lo) {
    const int c = threadIdx.x % C, tl = threadIdx.x / C, per = blockDim.x / C;
    // the channel's taps live in registers; the K samples of an output are requested together, clamped and masked instead of
    // guarded (a guarded load is waited for on its own: K serial round trips per output).  Same products, same order.
    constexpr int KMAX = 8;
    float wr[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wr[k] = w[c * K + (k < K ? k : K - 1)];
    const float bias_c = bias[c];
    for (int64_t t = (int64_t)blockIdx.x * per + tl; t < T; t += (int64_t)gridDim.x * per) {
        float acc = bias_c;
        if (K <= KMAX) {
            float p[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int64_t ti = t - (K - 1) + (k < K ? k : K - 1);
                p[k] = pcm[ti >= 0 ? ti : 0];
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const float term = wr[k] * p[k];
                if (k < K && t - (K - 1) + k >= 0) acc = acc + term;
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const int64_t ti = t - (K - 1) + k;
                if (ti >= 0) acc += w[c * K + k] * pcm[ti];
            }
        }
        const int64_t o = t * C + c;
        x[o] = acc;
        const float e = acc > 0.f ? acc : expm1f(acc);
        const bf16_t h = f32_to_bf16(e);
        hi[o] = h;
        lo[o] = f32_to_bf16(e - bf16_to_f32(h));
    }
}

// One workgroup per RVQ_FT frames walks the quantiser levels: distances of the frames' current vectors to all K entries of the
// level's codebook (transposed [D][K]: at step k the threads read consecutive entries, and every value fetched is used for
// all RVQ_FT frames - one frame per workgroup re-read the 2-MB codebook per frame and level and was L2-bandwidth-bound),
// argmin with the lowest index on ties, residual update.  Level 0 quantises the semantic projection, levels 1.. the
// acoustic projection residually.  1024 threads: K / 1024 entries per thread.
constexpr int RVQ_T = 1024, RVQ_MAXE = 4, RVQ_FT = 2;        // K <= 4096
template <int NE>                                            // codebook entries per thread = ceil(K / 1024)
__global__ __launch_bounds__(RVQ_T) void k_rvq(const float* __restrict__ sem, float* __restrict__ aco, int T, int D, int K, int Q,
                                               const float* const* __restrict__ cbT, int32_t* __restrict__ codes) {
    extern __shared__ float sh_x[];              // [RVQ_FT][D]
    __shared__ float sh_d[RVQ_FT][RVQ_T / 64];
    __shared__ int sh_j[RVQ_FT][RVQ_T / 64];
    __shared__ int sh_win[RVQ_FT];
    const int t0 = blockIdx.x * RVQ_FT, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int q = 0; q < Q; ++q) {
        if (q <= 1) {                             // level 0: semantic vectors; level 1: (fresh) acoustic vectors; later: residuals already in LDS
            __syncthreads();
            for (int i = tid; i < RVQ_FT * D; i += RVQ_T) {
                const int f = i / D, k = i - f * D;
                const int t = t0 + f < T ? t0 + f : T - 1;
                sh_x[i] = (q == 0 ? sem : aco)[(int64_t)t * D + k];
            }
        }
        __syncthreads();
        const float* cb = cbT[q];
        float acc[RVQ_FT][NE];
#pragma unroll
        for (int f = 0; f < RVQ_FT; ++f)
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[f][i] = 0.f;
        // KU code dimensions per step: their codebook values are requested together, so the L2 round trip is paid once per step;
        // the sums themselves run in ascending k, one rounding per operation (oracle/encoder.py rvq_level)
        constexpr int KU = 8;
        // ... and the NEXT step's values are requested before this step's sums run (two register sets, the loop unrolled by two):
        // a level is D / KU dependent L2 round trips otherwise - 512 of them for 16 levels, which is what this kernel's time was
        auto fetch = [&](int k0, float (&cv)[KU][NE]) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float* row = cb + (int64_t)(k0 + u < D ? k0 + u : D - 1) * K;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int j = tid + i * RVQ_T;
                    cv[u][i] = row[j < K ? j : K - 1];          // (clamped, unconditional: the entry is ignored in the argmin)
                }
            }
        };
        auto sums = [&](int k0, const float (&cv)[KU][NE]) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                if (k0 + u < D) {
#pragma unroll
                    for (int f = 0; f < RVQ_FT; ++f) {
                        const float xv = sh_x[f * D + k0 + u];
#pragma unroll
                        for (int i = 0; i < NE; ++i) {
                            const float diff = xv - cv[u][i];
                            acc[f][i] = acc[f][i] + diff * diff;             // (no contraction: see the file header)
                        }
                    }
                }
            }
        };
        float cva[KU][NE], cvb[KU][NE];
        fetch(0, cva);
        for (int k0 = 0; k0 < D; k0 += 2 * KU) {
            fetch(k0 + KU < D ? k0 + KU : k0, cvb);          // (past the end: a repeat request, never summed)
            sums(k0, cva);
            fetch(k0 + 2 * KU < D ? k0 + 2 * KU : k0, cva);
            if (k0 + KU < D) sums(k0 + KU, cvb);
        }
#pragma unroll
        for (int f = 0; f < RVQ_FT; ++f) {
            float best = 3.4e38f;
            int bj = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int j = tid + i * RVQ_T;
                if (j < K && acc[f][i] < best) { best = acc[f][i]; bj = j; }      // ascending j inside a thread: strict <
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {         // (distance, index) minimum: lowest index on equal distances
                const float d2 = __shfl_xor(best, o, 64);
                const int j2 = __shfl_xor(bj, o, 64);
                if (d2 < best || (d2 == best && j2 < bj)) { best = d2; bj = j2; }
            }
            if (lane == 0) { sh_d[f][w] = best; sh_j[f][w] = bj; }
        }
        __syncthreads();
        if (tid < RVQ_FT) {
            float best = sh_d[tid][0];
            int bj = sh_j[tid][0];
            for (int ww = 1; ww < RVQ_T / 64; ++ww) {
                const float d2 = sh_d[tid][ww];
                const int j2 = sh_j[tid][ww];
                if (d2 < best || (d2 == best && j2 < bj)) { best = d2; bj = j2; }
            }
            sh_win[tid] = bj;
            if (t0 + tid < T) codes[(int64_t)(t0 + tid) * Q + q] = bj;
        }
        __syncthreads();
        if (q >= 1) {                             // residuals for the next acoustic level
            for (int i = tid; i < RVQ_FT * D; i += RVQ_T) {
                const int f = i / D, k = i - f * D;
                sh_x[i] = sh_x[i] - cb[(int64_t)k * K + sh_win[f]];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < RVQ_FT * D; i += RVQ_T) {
        const int f = i / D, k = i - f * D;
        if (t0 + f < T) aco[(int64_t)(t0 + f) * D + k] = sh_x[i];
    }
}

// one workgroup per 64 channels; threads = 4 time lanes x 64 channels, two passes (mean, then variance about it)
__global__ __launch_bounds__(256) void k_stats_pool(const float* __restrict__ x, int T, int C, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C) for (int t = tl; t < T; t += 4) s += x[(int64_t)t * C + c];
    sh[tl][cl] = s;
    __syncthreads();
    const float mean = (sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)T;
    __syncthreads();
    float v = 0.f;
    if (c < C) for (int t = tl; t < T; t += 4) { const float d = x[(int64_t)t * C + c] - mean; v += d * d; }
    sh[tl][cl] = v;
    __syncthreads();
    if (tl == 0 && c < C) {
        out[c] = mean;
        out[C + c] = sqrtf((sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) / (float)T + 1e-5f);
    }
}

__global__ __launch_bounds__(256) void k_gemv_f32(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ x, int N,
                                                  int K, int relu, float* __restrict__ y) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += W[(int64_t)n * K + k] * x[k];
    s = wave_sum_f32(s);
    if (lane == 0) {
        s += b ? b[n] : 0.f;
        y[n] = relu ? fmaxf(s, 0.f) : s;
    }
}

}  // namespace

int launch_enc_conv0(rt_ctx* ctx, const float* pcm, int64_t T, int C, int k, const float* w, const float* bias, float* x, bf16_t* hi, bf16_t* lo) {
    if (T <= 0) return RT_OK;
    if (C < 1 || C > 256 || 256 % C) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "encoder: first-conv width %d must divide 256", C);
    const int per = 256 / C;
    int64_t blocks = (T + per - 1) / per;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    hipLaunchKernelGGL(k_enc_conv0, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, pcm, T, C, k, w, bias, x, hi, lo);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_rvq(rt_ctx* ctx, const float* sem, float* aco, int T, int D, int K, int Q, const float* const* d_cbT, int32_t* codes) {
    if (T <= 0) return RT_OK;
    if (K > RVQ_T * RVQ_MAXE || D > 4096 || Q < 1) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rvq: codebook of %d entries x %d (max %d x 4096)", K, D, RVQ_T * RVQ_MAXE);
    const dim3 grid((T + RVQ_FT - 1) / RVQ_FT);
    const size_t lds = (size_t)RVQ_FT * D * 4;
    switch ((K + RVQ_T - 1) / RVQ_T) {
        case 1: hipLaunchKernelGGL(k_rvq<1>, grid, dim3(RVQ_T), lds, ctx->stream, sem, aco, T, D, K, Q, d_cbT, codes); break;
        case 2: hipLaunchKernelGGL(k_rvq<2>, grid, dim3(RVQ_T), lds, ctx->stream, sem, aco, T, D, K, Q, d_cbT, codes); break;
        case 3: hipLaunchKernelGGL(k_rvq<3>, grid, dim3(RVQ_T), lds, ctx->stream, sem, aco, T, D, K, Q, d_cbT, codes); break;
        default: hipLaunchKernelGGL(k_rvq<4>, grid, dim3(RVQ_T), lds, ctx->stream, sem, aco, T, D, K, Q, d_cbT, codes); break;
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_stats_pool(rt_ctx* ctx, const float* x, int T, int C, float* out) {
    if (T <= 0) return rt_fail(ctx, RT_ERR_INVALID, "stats_pool: empty input");
    hipLaunchKernelGGL(k_stats_pool, dim3((C + 63) / 64), dim3(256), 0, ctx->stream, x, T, C, out);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_gemv_f32(rt_ctx* ctx, const float* W, const float* b, const float* x, int N, int K, int relu, float* y) {
    hipLaunchKernelGGL(k_gemv_f32, dim3((N + 3) / 4), dim3(256), 0, ctx->stream, W, b, x, N, K, relu, y);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
