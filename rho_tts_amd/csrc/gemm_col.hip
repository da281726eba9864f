// Column-owner decode GEMM for gfx950 (M <= 64 rows): one workgroup owns a 32-column tile of the output for the
// WHOLE K, its 8 waves split K, every wave streams its own contiguous run of pre-tiled weights from HBM with
// non-temporal 1-KiB loads (each weight byte is read exactly once, by exactly one wave), partial accumulators are
// combined through LDS in a fixed order (deterministic, no float atomics, no partial slabs in HBM), and because a
// workgroup sees complete dot products the layer's elementwise work is fused into the GEMM:
//   RMSNorm         : the operand is bf16(w .* x) written by the producer; the row scale 1/rms comes from the producer's
//                     sum-of-squares partials and is applied to the accumulator (y = inv_row * ((w .* x) W^T))
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
__device__ __forceinline__ unsigned pk2(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }

constexpr int WAVES = 8;    // 512 threads, 2 waves per SIMD: 256 VGPRs per wave for deep load queues

// RMSNorm without touching the operand twice: y = rmsnorm(x) W^T = inv_row * ((w .* x) W^T).  The producer of x (the
// RESID epilogue, or k_rowsq for the first GEMM of a stack) already stored bf16(w_next .* x) in fragment-tiled order, so
// every GEMM reads a plain bf16 operand with one coalesced load per tile, and the row scale inv_row - from the
// producer's per-tile sums of squares - is applied to the accumulator in the epilogue (post_scale).
template <int EPI>
__global__ __launch_bounds__(512) void k_gemm_col(ColArgs g) {
    __shared__ float red[WAVES][16][64];   // 32 KiB: one accumulator tile per wave
    __shared__ float sh_inv[32];           // RMSNorm row scales
    constexpr int NB = (EPI == COL_SILU) ? 2 : 1;
    constexpr int C = (NB == 2) ? 4 : 8;   // k-tiles per super-chunk
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nt = blockIdx.x;
    const int kchunk = (g.KT + WAVES - 1) / WAVES;
    const int kt_lo = w * kchunk;
    int kt_hi = kt_lo + kchunk;
    if (kt_hi > g.KT) kt_hi = g.KT;
    const int n_k = kt_hi > kt_lo ? kt_hi - kt_lo : 0;
    const int n_sc = (n_k + C - 1) / C;

    const s8_t* wp[NB];
    wp[0] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)nt * g.KT + kt_lo) * 64 + lane;
    if (NB == 2) wp[1] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)(nt + g.up_tile_offset) * g.KT + kt_lo) * 64 + lane;
    const int arow = g.row_off + (r < g.M ? r : g.M - 1);          // clamp: computed on valid memory, never stored
    // fragment-tiled A: this lane's 8 elements of k-tile kt sit at tile_off(arow, 16 kt + 8 h); consecutive lanes are contiguous
    const s8_t* ap = reinterpret_cast<const s8_t*>(reinterpret_cast<const bf16_t*>(g.A) + tile_off(arow, kt_lo * 16 + h * 8, g.KT));

    struct Chunk {
        s8_t b[NB][C];
        s8_t a[C];
    };
    auto issue = [&](int sc, Chunk& ck) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            int k = sc * C + u;
            if (k >= n_k) k = n_k - 1;                // tail: re-load the last tile, its product is skipped
#pragma unroll
            for (int b = 0; b < NB; ++b) ck.b[b][u] = g.nt ? __builtin_nontemporal_load(wp[b] + (int64_t)k * 64) : wp[b][(int64_t)k * 64];
            ck.a[u] = ap[(int64_t)k * 64];            // 64 lanes x 16 B = the next contiguous KiB
        }
    };

    Chunk c0, c1;
    if (n_sc > 0) issue(0, c0);
    // residual values this thread will update in the epilogue (elements tid and tid + 512 of the 32x32 tile)
    const int n = nt * 32 + (tid & 31);
    const int e_h = (tid >> 5) & 1;
    float xres[2] = {0.f, 0.f};
    if (EPI == COL_RESID) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = (tid + q * 512) >> 6;
            const int row = (i & 3) + 8 * (i >> 2) + 4 * e_h;
            if (row < g.M && n < g.N) xres[q] = g.out[tile_off(g.row_off + row, n, (int)(g.ldc >> 4))];
        }
    }
    if (g.post_scale) {   // row scales: half-wave (w, h) owns rows 2w + h and 2w + h + 16, its 32 lanes split the partials
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row_i = 2 * w + h + 16 * rr;
            const int row = g.row_off + (row_i < g.M ? row_i : g.M - 1);
            const float* p = g.rowsq + (int64_t)row * g.rowsq_n;
            float s = 0.f;
            for (int j = r; j < g.rowsq_n; j += 32) s += p[j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (r == 0) sh_inv[row_i] = rsqrtf(s / (float)g.K + g.eps);
        }
    }
    if (n_sc > 1) issue(1, c1);

    f16_t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    auto consume = [&](int sc, Chunk& ck) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            if (sc * C + u < n_k) {
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = mfma32(ck.a[u], ck.b[b][u], acc[b]);
            }
        }
    };
    for (int sc = 0; sc < n_sc; sc += 2) {
        consume(sc, c0);
        if (sc + 2 < n_sc) issue(sc + 2, c0);
        if (sc + 1 < n_sc) {
            consume(sc + 1, c1);
            if (sc + 3 < n_sc) issue(sc + 3, c1);
        }
    }

    // ---- combine the 8 K-partials through LDS in a fixed order and run the fused epilogue
    float val[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[b][i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * 512;
            const int i = e >> 6, l = e & 63;
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) s += red[ww][i][l];
            val[b][q] = s;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = (tid + q * 512) >> 6;
        const int row = (i & 3) + 8 * (i >> 2) + 4 * e_h;
        const bool ok = row < g.M && n < g.N;
        const float inv = g.post_scale ? sh_inv[row & 31] : 1.f;     // (sh_inv was published before the reduce barriers)
        float v = val[0][q] * inv;
        if (EPI == COL_STORE) {
            if (ok) {
                if (g.bias) v += g.bias[n];
                g.out[(int64_t)(g.row_off + row) * g.ldc + n] = v;      // STORE outputs (qkv, logits, mtp rows) stay row-major
            }
        } else if (EPI == COL_RESID) {
            float xn = 0.f;
            if (ok) {
                if (g.bias) v += g.bias[n];
                if (g.scale) v *= g.scale[n];
                xn = xres[q] + v;
                const int64_t o = tile_off(g.row_off + row, n, (int)(g.ldc >> 4));
                g.out[o] = xn;
                if (g.next_bf16) g.next_bf16[o] = f32_to_bf16(g.next_norm_w[n] * xn);   // operand of the GEMM behind the next RMSNorm
            }
            float sq = xn * xn;                   // half-wave = one row's 32 columns
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            if ((tid & 31) == 0 && row < g.M) g.rowsq_out[(int64_t)(g.row_off + row) * g.rowsq_out_n + nt] = sq;
        } else {                                  // COL_SILU: gate = tile nt, up = tile nt + offset
            if (ok) {
                const float u = val[NB - 1][q] * inv;
                g.out_bf16[tile_off(g.row_off + row, n, (int)(g.ldc >> 4))] = f32_to_bf16(v / (1.f + __expf(-v)) * u);
            }
        }
    }
}

int dispatch_epi(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    if (!e0 && !e1) {   // plain launches are what a stream capture records
        switch (g.epi) {
            case COL_STORE: hipLaunchKernelGGL((k_gemm_col<COL_STORE>), grid, dim3(512), 0, ctx->stream, g); break;
            case COL_RESID: hipLaunchKernelGGL((k_gemm_col<COL_RESID>), grid, dim3(512), 0, ctx->stream, g); break;
            case COL_SILU: hipLaunchKernelGGL((k_gemm_col<COL_SILU>), grid, dim3(512), 0, ctx->stream, g); break;
            default: return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: bad epilogue %d", g.epi);
        }
    } else {            // device-side begin/end stamps for the roofline figure
        switch (g.epi) {
            case COL_STORE: hipExtLaunchKernelGGL((k_gemm_col<COL_STORE>), grid, dim3(512), 0, ctx->stream, e0, e1, 0, g); break;
            case COL_RESID: hipExtLaunchKernelGGL((k_gemm_col<COL_RESID>), grid, dim3(512), 0, ctx->stream, e0, e1, 0, g); break;
            case COL_SILU: hipExtLaunchKernelGGL((k_gemm_col<COL_SILU>), grid, dim3(512), 0, ctx->stream, e0, e1, 0, g); break;
            default: return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: bad epilogue %d", g.epi);
        }
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
    if (g.post_scale && (!g.rowsq || g.rowsq_n < 1)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: row scaling without partials");
    if (g.next_bf16 && !g.next_norm_w) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: next operand without its norm weight");
    return dispatch_epi(ctx, g, dim3(tiles), ev_start, ev_stop);
}
