// Column-owner decode GEMM for gfx950 (M <= 64 rows): one workgroup owns a 32-column tile of the output for the
// WHOLE K, its 8 waves split K, every wave streams its own contiguous run of pre-tiled weights from HBM with
// non-temporal 1-KiB loads (each weight byte is read exactly once, by exactly one wave), partial accumulators are
// combined through LDS in a fixed order (deterministic, no float atomics, no partial slabs in HBM), and because a
// workgroup sees complete dot products the layer's elementwise work is fused into the GEMM:
//   prologue  NORM  : A is the f32 residual stream; RMSNorm (row scale from the producer's sum-of-squares partials,
//                     times the norm weight) is applied while the fragment is loaded, then rounded to bf16
//   epilogue  STORE : out = acc (+ bias)
//             RESID : x += scale * acc (residual stream updated in place) and per-(row, tile) sums of squares of the
//                     NEW x are emitted for the next GEMM's NORM prologue
//             SILU  : the workgroup owns gate tile j and up tile j: act = silu(gate) * up -> bf16
// This takes a decoder layer at decode time from 9 launches to 5 (qkv, attention, o, gate/up, down).
// HBM-bound: N*K*2 bytes per launch; MFMA v_mfma_f32_32x32x16_bf16 at a few % utilisation keeps it that way.
#include <hip/hip_ext.h>

#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(4))) int i4_t;

__device__ __forceinline__ f16_t mfma32(s8_t a, s8_t b, f16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk2(float lo, float hi) { return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16); }

constexpr int WAVES = 16;   // 1024 threads: 4 waves per SIMD, every one with its own weight run in flight

template <int MT, bool NORM, int EPI>
__global__ __launch_bounds__(1024) void k_gemm_col(ColArgs g) {
    __shared__ float red[8][16][64];       // 32 KiB: accumulator tiles of 8 waves at a time
    __shared__ float sh_inv[64];           // RMSNorm row scales
    constexpr int NB = (EPI == COL_SILU) ? 2 : 1;
    constexpr int U = NORM ? (NB == 1 ? 4 : 2) : (NB == 1 ? 8 : 4);   // k-tiles in flight per wave (128-VGPR budget, no spills)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nt = blockIdx.x;
    const int kchunk = (g.KT + WAVES - 1) / WAVES;
    const int kt_lo = w * kchunk;
    int kt_hi = kt_lo + kchunk;
    if (kt_hi > g.KT) kt_hi = g.KT;
    const int n_k = kt_hi > kt_lo ? kt_hi - kt_lo : 0;

    const s8_t* wp[NB];
    wp[0] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)nt * g.KT + kt_lo) * 64 + lane;
    if (NB == 2) wp[1] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)(nt + g.up_tile_offset) * g.KT + kt_lo) * 64 + lane;

    // first weight loads go out before anything else: they are the long pole
    s8_t bw[NB][U];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = u < n_k ? u : (n_k > 0 ? n_k - 1 : 0);
            if (n_k > 0) bw[b][u] = __builtin_nontemporal_load(wp[b] + (int64_t)k * 64);
        }

    if (NORM) {   // row scales once per workgroup: half-wave (w, h) owns row 2w + h, its 32 lanes split the partials
        const int row_i = 2 * w + h;
        if (row_i < MT * 32) {
            const int row = row_i < g.M ? row_i : g.M - 1;
            const float* p = g.rowsq + (int64_t)row * g.rowsq_n;
            float s = 0.f;
            for (int j = r; j < g.rowsq_n; j += 32) s += p[j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (r == 0) sh_inv[row_i] = rsqrtf(s / (float)g.K + g.eps);
        }
        __syncthreads();
    }
    int arow[MT];
    float inv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 32 + r;
        if (row >= g.M) row = g.M - 1;   // clamp: computed on valid memory, never stored
        arow[mt] = row;
        inv[mt] = NORM ? sh_inv[mt * 32 + r] : 1.f;
    }

    f16_t acc[NB][MT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][mt][i] = 0.f;

    for (int kt = 0; kt < n_k; kt += U) {
        if (kt > 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = kt + u < n_k ? kt + u : n_k - 1;      // tail: re-load the last tile, result discarded
                    bw[b][u] = __builtin_nontemporal_load(wp[b] + (int64_t)k * 64);
                }
        }
        s8_t af[MT][U];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = kt + u < n_k ? kt + u : n_k - 1;
                const int64_t off = (int64_t)arow[mt] * g.K + (int64_t)(kt_lo + k) * 16 + h * 8;
                if (NORM) {
                    const float* xp = reinterpret_cast<const float*>(g.A) + off;
                    const f4_t x0 = *reinterpret_cast<const f4_t*>(xp), x1 = *reinterpret_cast<const f4_t*>(xp + 4);
                    const float* wn = g.norm_w + (kt_lo + k) * 16 + h * 8;
                    const f4_t w0 = *reinterpret_cast<const f4_t*>(wn), w1 = *reinterpret_cast<const f4_t*>(wn + 4);
                    i4_t pk;
                    pk[0] = (int)pk2(w0[0] * (x0[0] * inv[mt]), w0[1] * (x0[1] * inv[mt]));
                    pk[1] = (int)pk2(w0[2] * (x0[2] * inv[mt]), w0[3] * (x0[3] * inv[mt]));
                    pk[2] = (int)pk2(w1[0] * (x1[0] * inv[mt]), w1[1] * (x1[1] * inv[mt]));
                    pk[3] = (int)pk2(w1[2] * (x1[2] * inv[mt]), w1[3] * (x1[3] * inv[mt]));
                    af[mt][u] = __builtin_bit_cast(s8_t, pk);
                } else {
                    af[mt][u] = *reinterpret_cast<const s8_t*>(reinterpret_cast<const bf16_t*>(g.A) + off);
                }
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt + u < n_k) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[b][mt] = mfma32(af[mt][u], bw[b][u], acc[b][mt]);
            }
        }
    }

    // ---- combine the 16 K-partials through LDS in a fixed order (waves 8-15 fold into 0-7, then 8 -> 1) and run the
    // fused epilogue, one accumulator tile at a time
    const int n = nt * 32 + (tid & 31);          // column of this thread in the reduce phase
    const int e_h = (tid >> 5) & 1;              // which half-wave -> +4 rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float val[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            __syncthreads();
            if (w >= 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[w - 8][i][lane] = acc[b][mt][i];
            }
            __syncthreads();
            if (w < 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][mt][i] += red[w][i][lane];
            }
            __syncthreads();
            if (w < 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[b][mt][i];
            }
            __syncthreads();
            {
                const int i = tid >> 6, l = tid & 63;     // 1024 threads = 16 x 64 elements of the 32x32 tile
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) s += red[ww][i][l];
                val[b] = s;
            }
        }
        {
            const int i = tid >> 6;
            const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * e_h;
            const bool ok = row < g.M && n < g.N;
            float v = val[0];
            if (EPI == COL_STORE) {
                if (ok) {
                    if (g.bias) v += g.bias[n];
                    g.out[(int64_t)row * g.ldc + n] = v;
                }
            } else if (EPI == COL_RESID) {
                float xn = 0.f;
                if (ok) {
                    if (g.bias) v += g.bias[n];
                    if (g.scale) v *= g.scale[n];
                    float* xp = g.out + (int64_t)row * g.ldc + n;
                    xn = *xp + v;
                    *xp = xn;
                }
                float sq = xn * xn;                   // half-wave = one row's 32 columns
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
                if ((tid & 31) == 0 && row < g.M) g.rowsq_out[(int64_t)row * g.rowsq_out_n + nt] = sq;
            } else {                                  // COL_SILU: gate = tile nt, up = tile nt + offset
                if (ok) {
                    const float u = val[NB - 1];
                    g.out_bf16[(int64_t)row * g.ldc + n] = f32_to_bf16(v / (1.f + __expf(-v)) * u);
                }
            }
        }
    }
}

template <int MT, bool NORM>
int dispatch_epi(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    if (!e0 && !e1) {   // plain launches are what a stream capture records
        switch (g.epi) {
            case COL_STORE: hipLaunchKernelGGL((k_gemm_col<MT, NORM, COL_STORE>), grid, dim3(1024), 0, ctx->stream, g); break;
            case COL_RESID: hipLaunchKernelGGL((k_gemm_col<MT, NORM, COL_RESID>), grid, dim3(1024), 0, ctx->stream, g); break;
            case COL_SILU: hipLaunchKernelGGL((k_gemm_col<MT, NORM, COL_SILU>), grid, dim3(1024), 0, ctx->stream, g); break;
            default: return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: bad epilogue %d", g.epi);
        }
        RT_HIP(ctx, hipGetLastError());
        return RT_OK;
    }
    switch (g.epi) {
        case COL_STORE: hipExtLaunchKernelGGL((k_gemm_col<MT, NORM, COL_STORE>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, g); break;
        case COL_RESID: hipExtLaunchKernelGGL((k_gemm_col<MT, NORM, COL_RESID>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, g); break;
        case COL_SILU: hipExtLaunchKernelGGL((k_gemm_col<MT, NORM, COL_SILU>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, g); break;
        default: return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: bad epilogue %d", g.epi);
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

}  // namespace

int launch_gemm_col(rt_ctx* ctx, const ColArgs& a, const PackedW& w, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.M < 1 || a.M > 32) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: M=%d outside 1..32 (callers split larger row blocks)", a.M);
    if (w.K != w.Kp || w.K != a.K) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: K mismatch (%d vs %d) or not a multiple of 16", w.K, a.K);
    ColArgs g = a;
    g.Wp = w.data;
    g.NT = w.Np / 32;
    g.KT = w.Kp / 16;
    int tiles = g.NT;
    if (g.epi == COL_SILU) {
        if (w.N % 64) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: gate/up width %d not a multiple of 64", w.N);
        tiles = g.NT / 2;
        g.up_tile_offset = g.NT / 2;
        g.N = w.N / 2;
    } else {
        g.N = w.N;
    }
    if (g.epi == COL_RESID && (!g.rowsq_out || g.rowsq_out_n < g.NT)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: rowsq_out too small");
    if (g.a_norm && (!g.rowsq || !g.norm_w || g.rowsq_n < 1)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: norm prologue without partials");
    dim3 grid(tiles);
    return g.a_norm ? dispatch_epi<1, true>(ctx, g, grid, ev_start, ev_stop) : dispatch_epi<1, false>(ctx, g, grid, ev_start, ev_stop);
}
