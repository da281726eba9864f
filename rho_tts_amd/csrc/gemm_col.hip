// Column-owner decode GEMM for gfx950 (M <= 32 rows per launch): one workgroup owns a 16-column tile of the output for the
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
// HBM-bound: N*K*2 bytes per launch; MFMA v_mfma_f32_16x16x32_bf16 at a few % utilisation keeps it that way.
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

typedef __attribute__((ext_vector_type(4))) float f4acc_t;
__device__ __forceinline__ f4acc_t mfma16(s8_t a, s8_t b, f4acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
}

// 16-column tiles (v_mfma_f32_16x16x32_bf16): twice the workgroups of a 32-column tiling, and a CU cannot pull more
// than ~24 GB/s out of HBM on its own, so the narrow GEMMs of a decode step (N = hidden) only reach the chip's
// bandwidth when they are spread over that many CUs.
//
// RMSNorm without touching the operand twice: y = rmsnorm(x) W^T = inv_row * ((w .* x) W^T).  The producer of x (the
// RESID epilogue, or k_rowsq for the first GEMM of a stack) already stored bf16(w_next .* x) in fragment-tiled order, so
// every GEMM reads a plain bf16 operand with one coalesced load per tile, and the row scale inv_row - from the
// producer's per-tile sums of squares - is applied to the accumulator in the epilogue (post_scale).
// The gate/up instantiation is held to 128 VGPRs (shorter chunks) so that two workgroups share a CU: its N/32 workgroups are
// 1.5x the CUs on the 1.7B talker, and one-per-CU would run them as a full round plus a half-empty one.
// MT = 16-row sub-blocks per launch: 2 (M <= 32, every single-position decode pass) or 4 (M <= 64: the predictor's first
// pass carries two positions per sequence; one 64-row launch streams the weights once instead of twice).
// NPRE = super-chunks requested up front WITHOUT a branch around them (1 or 2, chosen by the launcher from K; indices are
// clamped so every request is a valid address).  Branch-free matters: the row-scale partials are requested just before the
// chunks, and only when the number of younger loads is a compile-time constant can the compiler wait for them with a
// counted `s_waitcnt vmcnt(N)` instead of draining the weight chunks too.  NPRE = 0 keeps the fully guarded form (any K).
// MT = 1 (M <= 16: one of several decode LANES, rt_debug_tune 40n) is held to 128 VGPRs so that two workgroups - of two
// different lanes' launches - share a CU: two dependent chains can then really run side by side (with one workgroup per CU a
// second stream's kernel only queues behind the first).
// X (gate/up only): the workgroup also owns HALF of a second gate/up pair - 8 gate and 8 up columns side by side in one more
// 16-column MFMA tile - so that 3 n pairs run as 2 n workgroups of 1.5 pairs each.  The 1.7B talker's 384 pairs are then ONE
// round of 256 workgroups instead of a full round plus a half-empty one (which costs as much as two: 14.8 us for 384
// workgroups, 15.8 for 512, 9.6 for 256 - tools/bench_gemm_col_sweep.py), and the extra columns reuse the A fragments already
// in registers.  Every column is still summed by the same wave split in the same order: the bits do not change.
constexpr int col_chunk(int epi, int mt, bool x) {                         // k-tiles (32 deep) per super-chunk
    return epi == COL_SILU ? (x ? (mt == 4 ? 2 : 4) : 2) : ((mt == 4 || mt == 1) ? 4 : 8);
}
// NORM = the launch applies an RMSNorm row scale (g.post_scale), as a compile-time property: its partials are then requested
// without any branch (see the row-scale block), and launches without a scale carry no requests for them at all.
template <int EPI, int MT, int NPRE, bool X = false, bool NORM = false>
__global__ __launch_bounds__(512, (((EPI == COL_SILU && MT == 2) || MT == 1) && !X) ? 4 : 2) void k_gemm_col(ColArgs g) {
    static_assert(!X || (EPI == COL_SILU && MT >= 2), "the extra half pair exists for gate/up only");
    __shared__ float red[WAVES][MT][4][64];  // 16 KiB per 32 rows: one 16x16 accumulator tile per sub-block and wave
    __shared__ float sh_inv[16 * MT];        // RMSNorm row scales
    constexpr int NB = (EPI == COL_SILU) ? (X ? 3 : 2) : 1;
    constexpr int C = col_chunk(EPI, MT, X);
    constexpr int PASSES = (MT + 1) / 2;     // epilogue / row-scale passes of 32 rows (MT = 1: half of one)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, qd = lane >> 4;
    const int sp_shift = g.split == 4 ? 2 : (g.split == 2 ? 1 : 0);
    const int nt = blockIdx.x >> sp_shift, sub = blockIdx.x & (g.split - 1);
    const int cw_shift = 4 - sp_shift;                        // log2 of the columns this workgroup owns
    const bool b_mine = (r >> cw_shift) == sub;               // lane holds a B column of this workgroup's sub-tile
    const int kchunk = (g.KT + WAVES - 1) / WAVES;
    const int kt_lo = w * kchunk;
    int kt_hi = kt_lo + kchunk;
    if (kt_hi > g.KT) kt_hi = g.KT;
    const int n_k = kt_hi > kt_lo ? kt_hi - kt_lo : 0;
    const int n_sc = (n_k + C - 1) / C;
    const int n_mt = (g.M + 15) >> 4;                         // sub-blocks that hold rows

    const s8_t* wp[NB];
    wp[0] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)nt * g.KT + kt_lo) * 64 + lane;
    if (NB >= 2) wp[1] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)(nt + g.up_tile_offset) * g.KT + kt_lo) * 64 + lane;
    // the extra tile: lanes r < 8 hold gate columns x_col0 + r of pair x_tile, lanes r >= 8 its up columns x_col0 + r - 8
    const int x_tile = g.x_tile0 + ((int)blockIdx.x >> 1), x_col0 = ((int)blockIdx.x & 1) * 8;
    if (NB == 3) wp[NB - 1] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)(x_tile + (r < 8 ? 0 : g.up_tile_offset)) * g.KT + kt_lo) * 64 + qd * 16 + x_col0 + (r & 7);
    // fragment-tiled A, 16-row sub-blocks: this lane's 8 elements of k-tile kt sit at tile_off(row, 32 kt + 8 qd)
    const s8_t* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + r;
        if (row >= g.M) row = g.M - 1;                  // clamp: computed on valid memory, never stored
        ap[mt] = reinterpret_cast<const s8_t*>(reinterpret_cast<const bf16_t*>(g.A) + tile_off(g.row_off + row, kt_lo * 32 + qd * 8, g.K));
    }

    struct Chunk {
        s8_t b[NB][C];
        s8_t a[MT][C];
    };
    auto issue = [&](int sc, Chunk& ck) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            int k = sc * C + u;
            if (k >= n_k) k = n_k - 1;                // tail: re-load the last tile (an L1 / L2 hit), its product is skipped
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const s8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                ck.b[b][u] = !b_mine ? zero : (g.nt ? __builtin_nontemporal_load(wp[b] + (int64_t)k * 64) : wp[b][(int64_t)k * 64]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ck.a[mt][u] = ap[mt][(int64_t)k * 64];   // 64 lanes x 16 B = the next contiguous KiB
        }
    };

    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[0] = wall_clock64();
    // The small operands of the prologue / epilogue are requested FIRST: vector-memory loads return in issue order, so anything
    // issued behind the 16 KiB weight / activation chunks would only arrive after them - the RMSNorm row scales used to cost
    // every NORM launch 1-1.5 us that way (in-kernel stamps, tools/bench_gemm_col.py), and the epilogue's per-column vectors a
    // dependent L2 round trip after the reduce.
    // the output elements this thread finishes in the epilogue: pass ps -> sub-block 2 ps + (tid >> 8), accumulator register i, lane l
    const int e_i = (tid >> 6) & 3, e_l = tid & 63;
    const int n = nt * 16 + (e_l & 15);
    const bool e_mine = ((e_l & 15) >> cw_shift) == sub;
    const bool n_ok = n < g.N && e_mine;
    float rq[PASSES][16];                      // rowsq partials of rows 32 ps + 4 w + qd, 16 per lane (<= 256 partials per row)
    constexpr int RQ_MAX = 256;
    if constexpr (NORM) {   // row scales: the 16-lane group (w, qd) owns rows 4w + qd (+ 32 per pass), its lanes split the partials.
        // NO runtime branch around these requests: behind `if (g.post_scale)` the compiler merged this block with the summation
        // below and placed the first add (0 + rq[0][0]) at the join - a `s_waitcnt` for the first partial IN FRONT of the weight
        // chunks' requests, so every NORM launch (qkv, gate/up, the heads: 222 per frame) began to stream its weights one L2
        // round trip late.  (Unconditional stand-in requests for the launches WITHOUT a scale were measured too: +0.2-0.4 us on
        // each of those, more than the NORM launches gain - hence the template parameter.)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row_i = 32 * ps + 4 * w + qd;
            const int row = g.row_off + (row_i < g.M ? row_i : g.M - 1);
            const float* p = g.rowsq + (int64_t)row * g.rowsq_n;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int j = r + 16 * it;
                rq[ps][it] = p[j < g.rowsq_n ? j : g.rowsq_n - 1];          // (unconditional request, see NPRE; masked when summed)
            }
        }
    }
    float xres[PASSES];
    float e_bias = 0.f, e_scale = 1.f, e_nw = 0.f;
    if (n_ok) {
        if (EPI != COL_SILU && g.bias) e_bias = g.bias[n];
        if (EPI == COL_RESID && g.scale) e_scale = g.scale[n];
        if (EPI == COL_RESID && g.next_bf16) e_nw = g.next_norm_w[n];
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int e_row = (2 * ps + (tid >> 8)) * 16 + (e_l >> 4) * 4 + e_i;
        xres[ps] = 0.f;
        if (EPI == COL_RESID && e_row < g.M && n_ok) xres[ps] = g.out[tile_off(g.row_off + e_row, n, (int)g.ldc)];
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the small requests ahead of the chunks ...
    Chunk c0, c1;
    if (NPRE >= 1 || n_sc > 0) issue(0, c0);
    if (NPRE >= 2 || (NPRE == 0 && n_sc > 1)) issue(1, c1);
    __builtin_amdgcn_sched_barrier(0);          // ... and their consumers behind them (the scheduler would hoist the row-scale sums)
    if constexpr (NORM) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row_i = 32 * ps + 4 * w + qd;
            float s = 0.f;
#pragma unroll
            for (int it = 0; it < 16; ++it) s += (r + 16 * it < g.rowsq_n) ? rq[ps][it] : 0.f;
            if (g.rowsq_n > RQ_MAX) {             // (more partials than the unrolled window: not a shape the model produces)
                const int row = g.row_off + (row_i < g.M ? row_i : g.M - 1);
                const float* p = g.rowsq + (int64_t)row * g.rowsq_n;
                for (int j = RQ_MAX + r; j < g.rowsq_n; j += 16) s += p[j];
            }
            s = group_sum_f32<16>(s);             // (DPP adds, no LDS round trips: common.h)
            if (r == 0 && row_i < 16 * MT) sh_inv[row_i] = rsqrtf(s / (float)g.K + g.eps);
        }
    }
    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[1] = wall_clock64();

    f4acc_t acc[NB][MT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[b][mt][i] = 0.f;

    auto consume = [&](int sc, Chunk& ck) {
#pragma unroll
        for (int u = 0; u < C; ++u) {
            if (sc * C + u < n_k) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (mt == 0 || mt < n_mt) acc[b][mt] = mfma16(ck.a[mt][u], ck.b[b][u], acc[b][mt]);
            }
        }
    };
    for (int sc = 0; sc < n_sc; sc += 2) {
        consume(sc, c0);
        if (g.stamps && sc == 0 && blockIdx.x == 0 && tid == 0) g.stamps[2] = wall_clock64();
        if (sc + 2 < n_sc) issue(sc + 2, c0);
        if (sc + 1 < n_sc) {
            consume(sc + 1, c1);
            if (sc + 3 < n_sc) issue(sc + 3, c1);
        }
    }

    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[3] = wall_clock64();
    // ---- combine the 8 K-partials through LDS in a fixed order and run the fused epilogue (one output per thread and pass)
    float val[NB][PASSES];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b > 0) __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[w][mt][i][lane] = acc[b][mt][i];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            float s = 0.f;
            const int smt = 2 * ps + (tid >> 8);               // (MT = 1: the upper half of the workgroup has no sub-block)
            if (smt < MT) {
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) s += red[ww][smt < MT ? smt : 0][e_i][e_l];
            }
            val[b][ps] = s;
        }
    }
    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[4] = wall_clock64();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int e_row = (2 * ps + (tid >> 8)) * 16 + (e_l >> 4) * 4 + e_i;
        const bool ok = e_row < g.M && n_ok;
        const float inv = (NORM && e_row < 16 * MT) ? sh_inv[e_row] : 1.f;       // (published before the reduce barriers)
        float v = val[0][ps] * inv;
        if (EPI == COL_STORE) {
            if (ok) {
                v += e_bias;
                g.out[(int64_t)(g.row_off + e_row) * g.ldc + n] = v;    // STORE outputs (qkv, logits, mtp rows) stay row-major
            }
        } else if (EPI == COL_RESID) {
            float xn = 0.f;
            if (ok) {
                v += e_bias;
                v *= e_scale;
                xn = xres[ps] + v;
                const int64_t o = tile_off(g.row_off + e_row, n, (int)g.ldc);
                g.out[o] = xn;
                if (g.next_bf16) g.next_bf16[o] = f32_to_bf16(e_nw * xn);   // operand of the GEMM behind the next RMSNorm
            }
            float sq = xn * xn;                   // a 16-lane group = one row's 16 columns
            // (DPP adds: the four dependent ds_bpermute round trips of a __shfl_xor butterfly sat in the tail of every RESID launch -
            //  two per layer - between the last store of x and the store of its sums of squares)
            sq = group_sum_f32<16>(sq);
            if ((e_l & 15) == 0 && e_row < g.M) g.rowsq_out[(int64_t)(g.row_off + e_row) * g.rowsq_out_n + blockIdx.x] = sq;
        } else {                                  // COL_SILU: gate = tile nt, up = tile nt + offset
            if (ok) {
                const float u = val[1][ps] * inv;
                g.out_bf16[tile_off(g.row_off + e_row, n, (int)g.ldc)] = f32_to_bf16(v / (1.f + __expf(-v)) * u);
            }
            if (X) {                              // the extra half pair: gate in columns 0..7 of its tile, up in 8..15 of the same row
                const float gx = val[NB - 1][ps] * inv;
                const float ux = __shfl_down(gx, 8, 16);
                const int nx = x_tile * 16 + x_col0 + (e_l & 7);
                if (e_row < g.M && (e_l & 15) < 8 && nx < g.N) g.out_bf16[tile_off(g.row_off + e_row, nx, (int)g.ldc)] = f32_to_bf16(gx / (1.f + __expf(-gx)) * ux);
            }
        }
    }
    if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[5] = wall_clock64();
}

template <int EPI, int MT, int NPRE, bool X, bool NORM>
void launch_one(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    if (!e0 && !e1) hipLaunchKernelGGL((k_gemm_col<EPI, MT, NPRE, X, NORM>), grid, dim3(512), 0, ctx->stream, g);   // plain launches are what a stream capture records
    else hipExtLaunchKernelGGL((k_gemm_col<EPI, MT, NPRE, X, NORM>), grid, dim3(512), 0, ctx->stream, e0, e1, 0, g);   // device-side begin/end stamps
}

template <int EPI, int MT, bool X, bool NORM>
void launch_npre2(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    // every wave needs at least one k-tile for the clamped (branch-free) requests to be valid addresses
    constexpr int C = col_chunk(EPI, MT, X);
    const int kchunk = (g.KT + WAVES - 1) / WAVES;
    const bool all_waves_busy = g.KT >= WAVES && (WAVES - 1) * kchunk < g.KT;
    const int n_sc = (kchunk + C - 1) / C;
    if (!all_waves_busy) launch_one<EPI, MT, 0, X, NORM>(ctx, g, grid, e0, e1);
    else if (n_sc >= 2) launch_one<EPI, MT, 2, X, NORM>(ctx, g, grid, e0, e1);
    else launch_one<EPI, MT, 1, X, NORM>(ctx, g, grid, e0, e1);
}

// (the residual epilogue never follows an RMSNorm in this model: one instantiation less per shape)
template <int EPI, int MT, bool X = false>
void launch_npre(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    if (g.post_scale) {
        if constexpr (EPI != COL_RESID) launch_npre2<EPI, MT, X, true>(ctx, g, grid, e0, e1);
    } else {
        launch_npre2<EPI, MT, X, false>(ctx, g, grid, e0, e1);
    }
}

int dispatch_epi(rt_ctx* ctx, const ColArgs& g, dim3 grid, hipEvent_t e0, hipEvent_t e1) {
    const bool wide = g.M > 32, narrow = g.M <= 16 && g_col_rows16;
    switch (g.epi) {
        case COL_STORE: wide ? launch_npre<COL_STORE, 4>(ctx, g, grid, e0, e1) : narrow ? launch_npre<COL_STORE, 1>(ctx, g, grid, e0, e1) : launch_npre<COL_STORE, 2>(ctx, g, grid, e0, e1); break;
        case COL_RESID: wide ? launch_npre<COL_RESID, 4>(ctx, g, grid, e0, e1) : narrow ? launch_npre<COL_RESID, 1>(ctx, g, grid, e0, e1) : launch_npre<COL_RESID, 2>(ctx, g, grid, e0, e1); break;
        case COL_SILU:
            if (g.x_tile0 > 0) wide ? launch_npre<COL_SILU, 4, true>(ctx, g, grid, e0, e1) : launch_npre<COL_SILU, 2, true>(ctx, g, grid, e0, e1);
            else wide ? launch_npre<COL_SILU, 4>(ctx, g, grid, e0, e1) : narrow ? launch_npre<COL_SILU, 1>(ctx, g, grid, e0, e1) : launch_npre<COL_SILU, 2>(ctx, g, grid, e0, e1);
            break;
        default: return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: bad epilogue %d", g.epi);
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

}  // namespace

// gate/up GEMM (N = 2 * inter): a workgroup owns a gate tile and its up tile, N/32 workgroups.  Halving the tiles to even out
// the 1.5 rounds of the 1.7B talker (384 workgroups on 256 CUs) was measured slower (21.5 vs 17.4 us: every workgroup
// re-reads the whole A operand), so that split is only taken when forced through rt_debug_tune(502/504); the 1.5 rounds
// are evened out the other way - fewer, larger workgroups (X, g_col_silu_x).
rt_knob g_col_silu_x{1};            // 1: gate/up GEMMs whose pairs are 1.5x the CUs run as one round of 1.5-pair workgroups (rt_debug_tune 2400 / 2401)
rt_knob g_col_rows16{0};            // 1: launches of <= 16 rows take the 128-VGPR MT = 1 instantiation (two workgroups per CU: decode lanes, rt_debug_tune 2301)
rt_knob g_col_split4{0};            // quarter tiles for N <= 1024 measured 1.4 ms/step slower than half tiles (rt_debug_tune 1601 to try)
int col_split_silu(int N, int n_cu) {
    (void)N; (void)n_cu;
    return (g_col_split == 2 || g_col_split == 4) ? g_col_split.load() : 1;
}

// sub-tile split for an N-wide decode GEMM: enough workgroups to put one on every CU
int col_split_for(int N, int n_cu) {
    if (g_col_split == 1 || g_col_split == 2 || g_col_split == 4) return g_col_split;
    const int tiles = (N + 15) / 16;
    if (g_col_split4 && tiles * 4 <= n_cu) return 4;      // N <= 1024 on 256 CUs: a quarter tile per workgroup puts one on every CU
    if (tiles * 2 <= n_cu) return 2;
    return 1;
}

int launch_gemm_col(rt_ctx* ctx, const ColArgs& a, const PackedW& w, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.M < 1 || a.M > 64) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: M=%d outside 1..64 (callers split larger row blocks)", a.M);
    if (w.K != a.K || w.K % 32) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: K mismatch (%d vs %d) or not a multiple of 32", w.K, a.K);
    if (!w.data16) return rt_fail(ctx, RT_ERR_STATE, "gemm_col: weight has no 16-column decode copy");
    ColArgs g = a;
    g.Wp = w.data16;
    g.NT = w.Np16 / 16;
    g.KT = w.K / 32;
    if (g.split != 1 && g.split != 2 && g.split != 4) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: split %d (1, 2 or 4)", g.split);
    int tiles = g.NT;
    if (g.epi == COL_SILU) {
        if (w.N % 32) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: gate/up width %d not a multiple of 32", w.N);
        tiles = g.NT / 2;
        g.up_tile_offset = g.NT / 2;
        g.N = w.N / 2;
        g.x_tile0 = 0;
        // more pairs than CUs but no more than 1.5x: 2/3 of them as workgroups, each with half of one of the others
        const int wg = tiles / 3 * 2;
        const bool narrow = g.M <= 16 && g_col_rows16;
        if (g_col_silu_x && g.split == 1 && !narrow && tiles % 3 == 0 && tiles > ctx->n_cu && wg <= ctx->n_cu && (w.N / 2) % 16 == 0) {
            g.x_tile0 = wg;
            tiles = wg;
        }
    } else {
        g.N = w.N;
    }
    if (g.epi == COL_RESID && (!g.rowsq_out || g.rowsq_out_n < g.NT * g.split)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: rowsq_out too small");
    if (g.post_scale && (!g.rowsq || g.rowsq_n < 1)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: row scaling without partials");
    if (g.post_scale && g.epi == COL_RESID) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "gemm_col: the residual epilogue has no row-scaled instantiation");
    if (g.next_bf16 && !g.next_norm_w) return rt_fail(ctx, RT_ERR_INVALID, "gemm_col: next operand without its norm weight");
    return dispatch_epi(ctx, g, dim3(tiles * g.split), ev_start, ev_stop);
}
