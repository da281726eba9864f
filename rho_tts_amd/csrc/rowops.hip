// Row-wise / elementwise kernels of the decode step and the codec decoder (gfx950).
// All are HBM/L2-streaming: vectorised where the layout allows, one wave-shuffle reduction per row.
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f4_t;

__device__ __forceinline__ float block_sum_f32(float v, float* sh) {
    v = wave_sum_f32(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// x += scale * (sum_s slab[s] + slab_bias); xn = rmsnorm(x) * w.   One workgroup (256) per row, 16-B accesses,
// the row kept in registers between the reduction and the normalisation (H <= 8192, H % 4 == 0).
__global__ __launch_bounds__(256) void k_add_rmsnorm(float* __restrict__ x, int H, const float* __restrict__ slabs, int n_slabs,
                                                     int64_t slab_stride, const float* __restrict__ slab_bias,
                                                     const float* __restrict__ scale, const float* __restrict__ w, float eps,
                                                     bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    __shared__ float sh[4];
    constexpr int MAXV = 8;
    const int64_t row = blockIdx.x;
    const int nv = H >> 2;
    f4_t* xr = reinterpret_cast<f4_t*>(x + row * H);
    f4_t keep[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i >= nv) break;
        f4_t v = xr[i];
        if (n_slabs > 0) {
            f4_t a = {0.f, 0.f, 0.f, 0.f};
            if (slab_bias) a = reinterpret_cast<const f4_t*>(slab_bias)[i];
            const f4_t* sp = reinterpret_cast<const f4_t*>(slabs + row * H) + i;
            const int64_t st4 = slab_stride >> 2;
#pragma unroll 8
            for (int s = 0; s < n_slabs; ++s) a += sp[s * st4];
            if (scale) a *= reinterpret_cast<const f4_t*>(scale)[i];
            v += a;
            xr[i] = v;
        }
        keep[j] = v;
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (!w) return;
    const float tot = block_sum_f32(ss, sh);
    const float inv = rsqrtf(tot / (float)H + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i >= nv) break;
        const f4_t wv = reinterpret_cast<const f4_t*>(w)[i];
        f4_t v = keep[j];
        v[0] = wv[0] * (v[0] * inv); v[1] = wv[1] * (v[1] * inv); v[2] = wv[2] * (v[2] * inv); v[3] = wv[3] * (v[3] * inv);
        if (out_f32) reinterpret_cast<f4_t*>(out_f32 + row * H)[i] = v;
        if (out_bf16) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            reinterpret_cast<uint2*>(out_bf16 + row * H)[i] = pk;
        }
    }
}

// The same for H = NV x 1024 with everything a thread needs requested up front (the generic form above leaves its loop by a
// per-lane `break`, and behind such a branch the compiler waits for each iteration's requests on their own; identical arithmetic).
template <int NV>
__global__ __launch_bounds__(256) void k_add_rmsnorm_v(float* __restrict__ x, const float* __restrict__ slabs, int n_slabs, int64_t slab_stride,
                                                       const float* __restrict__ slab_bias, const float* __restrict__ scale, const float* __restrict__ w,
                                                       float eps, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    __shared__ float sh[4];
    constexpr int H = NV * 1024;
    const int64_t row = blockIdx.x;
    f4_t* xr = reinterpret_cast<f4_t*>(x + row * H);
    f4_t keep[NV], a[NV], sc[NV], wv[NV];
    const f4_t zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
    // (absent vectors are read through a stand-in pointer and replaced afterwards: even a uniform `ptr ? load : const` makes the
    //  compiler wait for everything in flight where the two paths join)
    const f4_t* bp = reinterpret_cast<const f4_t*>(slab_bias ? slab_bias : x + row * H);
    const f4_t* sp_ = reinterpret_cast<const f4_t*>(scale ? scale : x + row * H);
    const f4_t* wp = reinterpret_cast<const f4_t*>(w ? w : x + row * H);
    const f4_t* s0 = reinterpret_cast<const f4_t*>(n_slabs > 0 ? slabs + row * H : x + row * H);
    f4_t t0[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = threadIdx.x + j * 256;
        keep[j] = xr[i];
        a[j] = bp[i];
        sc[j] = sp_[i];
        wv[j] = wp[i];
        t0[j] = s0[i];                       // the first slab (the only one behind the prompt-prefill GEMMs) rides along
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (!slab_bias) a[j] = zero;
        if (!scale) sc[j] = one;
        if (n_slabs > 0) a[j] += t0[j];
    }
    const int64_t st4 = slab_stride >> 2;
    for (int s = 1; s < n_slabs; ++s) {
        f4_t t[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) t[j] = (reinterpret_cast<const f4_t*>(slabs + row * H) + threadIdx.x + j * 256)[s * st4];
#pragma unroll
        for (int j = 0; j < NV; ++j) a[j] += t[j];
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        f4_t v = keep[j];
        if (n_slabs > 0) {
            f4_t add = a[j];
            if (scale) add *= sc[j];
            v += add;
            xr[threadIdx.x + j * 256] = v;
        }
        keep[j] = v;
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (!w) return;
    const float tot = block_sum_f32(ss, sh);
    const float inv = rsqrtf(tot / (float)H + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = threadIdx.x + j * 256;
        f4_t v = keep[j];
        v[0] = wv[j][0] * (v[0] * inv); v[1] = wv[j][1] * (v[1] * inv); v[2] = wv[j][2] * (v[2] * inv); v[3] = wv[j][3] * (v[3] * inv);
        if (out_f32) reinterpret_cast<f4_t*>(out_f32 + row * H)[i] = v;
        if (out_bf16) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            reinterpret_cast<uint2*>(out_bf16 + row * H)[i] = pk;
        }
    }
}

// rowsq[row][0] = sum x^2, rowsq[row][1..n) = 0: seeds the NORM prologue of the column-owner GEMM; optionally also
// re-tiles the row into the fragment layout (common.h tile_off) the decode GEMMs read.
__global__ __launch_bounds__(256) void k_rowsq(const float* __restrict__ x, int H, float* __restrict__ rowsq, int rowsq_n,
                                               float* __restrict__ x_tiled, bf16_t* __restrict__ a_tiled,
                                               const float* __restrict__ norm_w, const int32_t* __restrict__ src_rows,
                                               const int32_t* __restrict__ dst_rows, RowsqGather gt) {
    __shared__ float sh[4];
    // (optional row maps: block i reads row src_rows[i] of x and writes decode row dst_rows[i] - a queued item taking over a row)
    const int64_t row = dst_rows ? dst_rows[blockIdx.x] : blockIdx.x;
    const f4_t* xr = reinterpret_cast<const f4_t*>(x + (int64_t)(src_rows ? src_rows[blockIdx.x] : blockIdx.x) * H);
    // (optional gather: blocks from gt.first on read row idx[block - first] of an f32 table instead - the predictor's first pass,
    //  whose second half of rows are embeddings of the code just sampled: the same loop and sums as k_gather_f32 + this kernel)
    bool zero_row = false;                               // (a negative index is a row of zeros, as in k_gather_f32)
    if (gt.table && (int)blockIdx.x >= gt.first) {
        const int32_t* ix = gt.idx + (gt.frame_ptr ? (int64_t)(*gt.frame_ptr) * gt.idx_frame_stride : 0);
        const int id = ix[(int64_t)((int)blockIdx.x - gt.first) * gt.idx_stride];
        zero_row = id < 0;
        xr = reinterpret_cast<const f4_t*>(gt.table + (int64_t)(id < 0 ? 0 : id) * H);
    }
    float ss = 0.f;
    for (int i = threadIdx.x; i < (H >> 2); i += 256) {
        const f4_t v = zero_row ? f4_t{0.f, 0.f, 0.f, 0.f} : xr[i];
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (x_tiled) *reinterpret_cast<f4_t*>(x_tiled + tile_off((int)row, i * 4, H)) = v;
        if (a_tiled) {   // bf16(norm_w .* x): the first GEMM's operand (its row scale is applied after the product)
            const f4_t wv = reinterpret_cast<const f4_t*>(norm_w)[i];
            uint2 pk;
            pk.x = f32x2_to_bf16x2(wv[0] * v[0], wv[1] * v[1]);
            pk.y = f32x2_to_bf16x2(wv[2] * v[2], wv[3] * v[3]);
            *reinterpret_cast<uint2*>(a_tiled + tile_off((int)row, i * 4, H)) = pk;
        }
    }
    const float tot = block_sum_f32(ss, sh);
    for (int j = threadIdx.x; j < rowsq_n; j += 256) rowsq[row * rowsq_n + j] = j == 0 ? tot : 0.f;
}

// Decode-step input in one launch: row = add_vec + sum_j table_j[idx_j] (bf16 embedding rows, summed in source order) or
// one row of an f32 table, written straight in the column path's layouts - fragment-tiled f32 x, tiled bf16(norm_w .* x)
// and the RMSNorm sum of squares (rowsq[row][0], the other partial slots zeroed).  Replaces k_gather_* + k_rowsq: all
// embedding rows of a thread are loaded before the first add (one memory round trip instead of n_src dependent ones).
__global__ __launch_bounds__(256) void k_embed_rowsq(const GatherSrc* __restrict__ srcs, int n_src, const float* __restrict__ f32_table,
                                                     const int32_t* __restrict__ idx, int idx_stride,
                                                     const int32_t* __restrict__ frame_ptr, int64_t idx_frame_stride, int H,
                                                     const float* __restrict__ add_vec, float* __restrict__ rowsq, int rowsq_n,
                                                     float* __restrict__ x_tiled, bf16_t* __restrict__ a_tiled,
                                                     const float* __restrict__ norm_w, int32_t* frame_inc, unsigned* arrive) {
    constexpr int MAXS = 16;
    __shared__ const bf16_t* sh_tab[MAXS];
    __shared__ float sh[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    // (frame_inc: this launch also advances the frame counter it reads - see below; the counter is then read with an atomic load,
    //  which the compiler may neither duplicate nor re-issue behind the barrier)
    if (frame_ptr) idx += (int64_t)(frame_inc ? __atomic_load_n(frame_ptr, __ATOMIC_RELAXED) : *frame_ptr) * idx_frame_stride;
    if (tid < MAXS) {
        const bf16_t* p = nullptr;
        if (tid < n_src) {
            const int id = idx[(int64_t)row * idx_stride + tid];
            if (id >= 0) p = srcs[tid].table + (int64_t)id * srcs[tid].row_stride;
        }
        sh_tab[tid] = p;
    }
    const int fid = f32_table ? idx[(int64_t)row * idx_stride] : -1;
    __syncthreads();
    // frame += 1 by the LAST workgroup to get here (instead of a launch of its own, k_frame_inc).  The only readers of the counter
    // in this launch are the lanes that filled sh_tab: their index loads - which needed the counter - had to return before the
    // table pointers could be written, i.e. before the barrier above, so once every workgroup has arrived nobody reads it again.
    // (frame_inc is refused together with f32_table, whose index load is not tied to the barrier that way.)
    if (frame_inc && tid == 0 && atomicAdd(arrive, 1u) == gridDim.x - 1) {
        *arrive = 0u;
        atomicAdd(frame_inc, 1);
    }
    float ss = 0.f;
    for (int c = tid * 8; c < H; c += 2048) {
        uint4 raw[MAXS];
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const bf16_t* p = sh_tab[j];
            raw[j] = p ? *reinterpret_cast<const uint4*>(p + c) : uint4{0u, 0u, 0u, 0u};
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = add_vec ? add_vec[c + e] : 0.f;
        if (fid >= 0) {
            const f4_t a = *reinterpret_cast<const f4_t*>(f32_table + (int64_t)fid * H + c);
            const f4_t b = *reinterpret_cast<const f4_t*>(f32_table + (int64_t)fid * H + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            if (sh_tab[j]) {              // (a missing source adds nothing - not even +0.0 - exactly like k_gather_sum)
                const unsigned wds[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(wds[e] << 16);
                    v[2 * e + 1] += __uint_as_float(wds[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        const int64_t o = tile_off(row, c, H);
        *reinterpret_cast<f4_t*>(x_tiled + o) = f4_t{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f4_t*>(x_tiled + o + 4) = f4_t{v[4], v[5], v[6], v[7]};
        const f4_t w0 = *reinterpret_cast<const f4_t*>(norm_w + c), w1 = *reinterpret_cast<const f4_t*>(norm_w + c + 4);
        uint4 pk;
        pk.x = f32x2_to_bf16x2(w0[0] * v[0], w0[1] * v[1]);
        pk.y = f32x2_to_bf16x2(w0[2] * v[2], w0[3] * v[3]);
        pk.z = f32x2_to_bf16x2(w1[0] * v[4], w1[1] * v[5]);
        pk.w = f32x2_to_bf16x2(w1[2] * v[6], w1[3] * v[7]);
        *reinterpret_cast<uint4*>(a_tiled + o) = pk;
    }
    const float tot = block_sum_f32(ss, sh);
    for (int j = tid; j < rowsq_n; j += 256) rowsq[(int64_t)row * rowsq_n + j] = j == 0 ? tot : 0.f;
}

// rmsnorm of the column path's tiled residual stream back to plain rows: out[row][c] = w[c] * x[row][c] * rsqrt(sum(rowsq[row]) / H
// + eps).  Only an equal-width predictor needs it (its first input row is the talker's normalised hidden state itself).
__global__ __launch_bounds__(256) void k_norm_tiled_rows(const float* __restrict__ x_tiled, const float* __restrict__ rowsq, int rowsq_n,
                                                         const float* __restrict__ w, float eps, int H, float* __restrict__ out) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < rowsq_n; j += 256) s += rowsq[(int64_t)row * rowsq_n + j];
    const float inv = rsqrtf(block_sum_f32(s, sh) / (float)H + eps);
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        const f4_t v = *reinterpret_cast<const f4_t*>(x_tiled + tile_off(row, c, H));
        const f4_t wv = *reinterpret_cast<const f4_t*>(w + c);
        *reinterpret_cast<f4_t*>(out + (int64_t)row * H + c) = f4_t{wv[0] * (v[0] * inv), wv[1] * (v[1] * inv), wv[2] * (v[2] * inv), wv[3] * (v[3] * inv)};
    }
}

__global__ void k_silu_mul(const float* __restrict__ slabs, int n_slabs, int64_t slab_stride, int I, bf16_t* __restrict__ out,
                           int64_t total4, float* __restrict__ out_f32) {
    const int I4 = I >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / I4;
        const int i = (int)(idx - m * I4);
        f4_t g = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
        const f4_t* p = reinterpret_cast<const f4_t*>(slabs + m * 2 * I);
        const int64_t st4 = slab_stride >> 2;
#pragma unroll 8
        for (int s = 0; s < n_slabs; ++s) {
            g += p[s * st4 + i];
            u += p[s * st4 + I4 + i];
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = g[j] / (1.f + __expf(-g[j])) * u[j];
        if (out_f32) { reinterpret_cast<f4_t*>(out_f32)[idx] = f4_t{r[0], r[1], r[2], r[3]}; continue; }
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(r[0]) | ((unsigned)f32_to_bf16(r[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(r[2]) | ((unsigned)f32_to_bf16(r[3]) << 16);
        reinterpret_cast<uint2*>(out)[idx] = pk;
    }
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, int n_slabs, int64_t slab_stride, int N, const float* __restrict__ bias,
                               int act, float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        float v = bias ? bias[idx % N] : 0.f;
        for (int s = 0; s < n_slabs; ++s) v += slabs[s * slab_stride + idx];
        if (act == ACT_SILU) v = v / (1.f + __expf(-v));
        else if (act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        if (out_f32) out_f32[idx] = v;
        if (out_bf16) out_bf16[idx] = f32_to_bf16(v);
    }
}

// One wave per (row, head): sum split-K slabs, RMSNorm over head_dim, rotate-half RoPE; q -> f32 buffer,
// k/v -> bf16 cache row (slot, pos).  head index < heads: q; < heads+kv: k; else v.
__global__ __launch_bounds__(64) void k_qkv_post(const float* __restrict__ slabs, int n_slabs, int64_t slab_stride, int heads,
                                                 int kv_heads, int d, const float* __restrict__ qw, const float* __restrict__ kw,
                                                 float eps, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                 const int32_t* __restrict__ row_slot, const int32_t* __restrict__ row_pos,
                                                 int pos_add, float* __restrict__ q_out, bf16_t* __restrict__ kc,
                                                 bf16_t* __restrict__ vc, int max_pos, const int32_t* __restrict__ frame_ptr,
                                                 bf16_t* __restrict__ kc_lo, bf16_t* __restrict__ vc_lo) {
    const int row = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x;
    const int width = (heads + 2 * kv_heads) * d;
    const int half = d >> 1;
    const int pos = row_pos[row] + pos_add + (frame_ptr ? *frame_ptr : 0), slot = row_slot[row];
    // lane handles element pairs (i, i + half) for i = lane, lane+64, ...  (half <= 64 for d <= 128)
    float a = 0.f, b = 0.f;
    const bool act = lane < half;
    if (act) {
        const int64_t base = (int64_t)row * width + hd * d;
        for (int s = 0; s < n_slabs; ++s) {
            a += slabs[s * slab_stride + base + lane];
            b += slabs[s * slab_stride + base + lane + half];
        }
    }
    const bool is_q = hd < heads, is_k = !is_q && hd < heads + kv_heads;
    if (is_q || is_k) {
        const float* nw = is_q ? qw : kw;
        if (nw) {
            const float ss = wave_sum_rows_f32(act ? a * a + b * b : 0.f);   // (same association as the fused attention prologue)
            const float inv = rsqrtf(ss / (float)d + eps);
            if (act) { a = nw[lane] * (a * inv); b = nw[lane + half] * (b * inv); }
        }
        if (act && cosT) {                           // (cosT == nullptr: no rotary embedding - learned / sinusoidal positions, stt.hip)
            const float c = cosT[(int64_t)pos * half + lane], s = sinT[(int64_t)pos * half + lane];
            const float ra = a * c - b * s, rb = b * c + a * s;
            a = ra; b = rb;
        }
    }
    if (!act) return;
    if (is_q) {
        float* o = q_out + ((int64_t)row * heads + hd) * d;
        o[lane] = a;
        o[lane + half] = b;
    } else {
        const int kh = is_k ? hd - heads : hd - heads - kv_heads;
        const int64_t off = (((int64_t)slot * kv_heads + kh) * max_pos + pos) * d;
        bf16_t* o = (is_k ? kc : vc) + off;
        const bf16_t ha = f32_to_bf16(a), hb = f32_to_bf16(b);
        o[lane] = ha;
        o[lane + half] = hb;
        if (kc_lo) {                                  // low planes: what the bf16 rounding dropped
            bf16_t* l = (is_k ? kc_lo : vc_lo) + off;
            l[lane] = f32_to_bf16(a - bf16_to_f32(ha));
            l[lane + half] = f32_to_bf16(b - bf16_to_f32(hb));
        }
    }
}

__global__ __launch_bounds__(256) void k_gather_sum(const GatherSrc* __restrict__ srcs, int n_src, const int32_t* __restrict__ idx,
                                                    int H, const float* __restrict__ add_vec, const float* __restrict__ add_rows,
                                                    const int32_t* __restrict__ add_row_idx, float* __restrict__ out_f32,
                                                    bf16_t* __restrict__ out_bf16, int idx_stride,
                                                    const int32_t* __restrict__ frame_ptr, int64_t idx_frame_stride) {
    const int64_t row = blockIdx.x;
    if (frame_ptr) idx += (int64_t)(*frame_ptr) * idx_frame_stride;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = add_vec ? add_vec[i] : 0.f;
        if (add_rows) {
            const int ar = add_row_idx ? add_row_idx[row] : (int)row;
            if (ar >= 0) v += add_rows[(int64_t)ar * H + i];
        }
        for (int j = 0; j < n_src; ++j) {
            const int id = idx[row * idx_stride + j];
            if (id >= 0) v += bf16_to_f32(srcs[j].table[(int64_t)id * srcs[j].row_stride + i]);
        }
        if (out_f32) out_f32[row * H + i] = v;
        if (out_bf16) out_bf16[row * H + i] = f32_to_bf16(v);
    }
}

__global__ __launch_bounds__(256) void k_gather_f32(const float* __restrict__ table, int H, const int32_t* __restrict__ idx,
                                                    float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16, int idx_stride,
                                                    const int32_t* __restrict__ frame_ptr, int64_t idx_frame_stride) {
    const int64_t row = blockIdx.x;
    if (frame_ptr) idx += (int64_t)(*frame_ptr) * idx_frame_stride;
    const int id = idx[row * idx_stride];
    for (int i = threadIdx.x; i < H; i += 256) {
        const float v = id >= 0 ? table[(int64_t)id * H + i] : 0.f;
        if (out_f32) out_f32[row * H + i] = v;
        if (out_bf16) out_bf16[row * H + i] = f32_to_bf16(v);
    }
}


// ConvNeXt front half: depthwise causal conv k=7 over time, then LayerNorm over channels -> bf16.
// One workgroup per (b, t) row; channels-last so the 7 taps are 7 contiguous rows.
__global__ __launch_bounds__(256) void k_dwconv_ln(const float* __restrict__ x, int T, int C, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ lw,
                                                   const float* __restrict__ lb, float eps, float* __restrict__ out) {
    extern __shared__ float row[];
    __shared__ float sh[4];
    const int64_t r = blockIdx.x;
    const int t = (int)(r % T);
    float s1 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = b[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int tt = t - 6 + k;
            if (tt >= 0) v += w[k * C + c] * x[(r - 6 + k) * C + c];
        }
        row[c] = v;
        s1 += v;
    }
    const float mean = block_sum_f32(s1, sh) / (float)C;
    float s2 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float dlt = row[c] - mean; s2 += dlt * dlt; }
    const float var = block_sum_f32(s2, sh) / (float)C;
    const float inv = rsqrtf(var + eps);
    for (int c = threadIdx.x; c < C; c += 256) out[r * C + c] = (row[c] - mean) * inv * lw[c] + lb[c];
}

__global__ __launch_bounds__(256) void k_code_embed_mean(const bf16_t* __restrict__ table, int codebook, int Q, int H,
                                                         const int32_t* __restrict__ codes, float* __restrict__ out) {
    const int64_t r = blockIdx.x;
    const float inv = 1.f / (float)Q;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = 0.f;
        for (int q = 0; q < Q; ++q) {
            int c = codes[r * Q + q];
            c = c < 0 ? 0 : (c >= codebook ? codebook - 1 : c);
            v += bf16_to_f32(table[((int64_t)q * codebook + c) * H + i]);
        }
        out[r * H + i] = v * inv;
    }
}


// Last conv of the codec decoder (C channels -> 1, k = 7, causal) + clamp(-1, 1), input as the hi / lo bf16 planes the previous
// epilogue wrote.  As a one-column GEMM it wasted 127 of 128 tile columns (1.0 ms at 2.7 M samples); here a workgroup owns 128
// consecutive samples of one item: the (128 + 6) x C input window is summed (hi + lo, exact in f32) into LDS once with
// coalesced 16-B loads, then every thread does half a sample's 7 x C dot product out of LDS (row stride C + 1).
__global__ __launch_bounds__(256) void k_final_conv(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int T, int C,
                                                    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wav) {
    extern __shared__ float sm[];
    const int stride = C + 1;
    float* tile = sm;
    float* wt = sm + 134 * stride;
    const int b = blockIdx.y, t0 = blockIdx.x * 128, tid = threadIdx.x;
    for (int i = tid; i < 7 * C; i += 256) wt[i] = w[i];
    const int C8 = C >> 3;
    // the window, 16 B of each plane per request, seven requests of a thread in flight together: rows outside the item and the
    // surplus of the last round read a clamped address and are masked / rewrite the value of the clamped index (no branch around
    // a load: behind one the compiler waits for every request on its own - seven serial round trips per workgroup)
    constexpr int IT = 7;
    const int n_items = 134 * C8;
    for (int base = 0; base < n_items; base += 256 * IT) {
        uint4 h[IT], l[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            int idx = base + u * 256 + tid;
            idx = idx < n_items ? idx : n_items - 1;
            const int row = idx / C8, c = (idx - row * C8) * 8;
            int t = t0 - 6 + row;
            t = t < 0 ? 0 : (t < T ? t : T - 1);
            const int64_t o = ((int64_t)b * T + t) * C + c;
            h[u] = *reinterpret_cast<const uint4*>(hi + o);
            l[u] = *reinterpret_cast<const uint4*>(lo + o);
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            int idx = base + u * 256 + tid;
            idx = idx < n_items ? idx : n_items - 1;
            const int row = idx / C8, c = (idx - row * C8) * 8, t = t0 - 6 + row;
            const bool in = t >= 0 && t < T;
            const unsigned hw[4] = {h[u].x, h[u].y, h[u].z, h[u].w}, lw[4] = {l[u].x, l[u].y, l[u].z, l[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = __uint_as_float(hw[e] << 16) + __uint_as_float(lw[e] << 16);
                const float v1 = __uint_as_float(hw[e] & 0xffff0000u) + __uint_as_float(lw[e] & 0xffff0000u);
                tile[row * stride + c + 2 * e] = in ? v0 : 0.f;
                tile[row * stride + c + 2 * e + 1] = in ? v1 : 0.f;
            }
        }
    }
    __syncthreads();
    const int s = tid >> 1, half = tid & 1, ch = C >> 1;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k) {
        const float* x = tile + (s + k) * stride + half * ch;
        const float* ww = wt + k * C + half * ch;
        for (int c = 0; c < ch; ++c) acc += x[c] * ww[c];
    }
    acc += __shfl_xor(acc, 1, 64);
    const int t = t0 + s;
    if (half == 0 && t < T) wav[(int64_t)b * T + t] = fminf(1.f, fmaxf(-1.f, acc + bias[0]));
}

__global__ void k_f32_to_bf16(const float* __restrict__ x, int64_t n, bf16_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16(x[i]);
}

inline unsigned grid_for(int64_t total, int block = 256) {
    int64_t b = (total + block - 1) / block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

int launch_add_rmsnorm(rt_ctx* ctx, float* x, int M, int H, const float* slabs, int n_slabs, const float* slab_bias,
                       const float* scale, const float* w, float eps, bf16_t* out_bf16, float* out_f32) {
    if (M <= 0) return RT_OK;
    if (H % 4 || H > 8192) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rmsnorm: hidden size %d unsupported (multiple of 4, <= 8192)", H);
    const int64_t st = (int64_t)M * H;
    if (H == 1024) hipLaunchKernelGGL(k_add_rmsnorm_v<1>, dim3(M), dim3(256), 0, ctx->stream, x, slabs, n_slabs, st, slab_bias, scale, w, eps, out_bf16, out_f32);
    else if (H == 2048) hipLaunchKernelGGL(k_add_rmsnorm_v<2>, dim3(M), dim3(256), 0, ctx->stream, x, slabs, n_slabs, st, slab_bias, scale, w, eps, out_bf16, out_f32);
    else hipLaunchKernelGGL(k_add_rmsnorm, dim3(M), dim3(256), 0, ctx->stream, x, H, slabs, n_slabs, st, slab_bias, scale, w, eps, out_bf16, out_f32);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_rowsq(rt_ctx* ctx, const float* x, int M, int H, float* rowsq, int rowsq_n, float* x_tiled, bf16_t* a_tiled,
                 const float* norm_w, const int32_t* src_rows, const int32_t* dst_rows, const RowsqGather* gather) {
    if (M <= 0) return RT_OK;
    const RowsqGather gt = gather ? *gather : RowsqGather{};
    if (gt.table && (src_rows || dst_rows || gt.first < 0 || gt.first > M || !gt.idx))
        return rt_fail(ctx, RT_ERR_INVALID, "rowsq: gathered rows take no row maps (first %d of %d rows)", gt.first, M);
    hipLaunchKernelGGL(k_rowsq, dim3(M), dim3(256), 0, ctx->stream, x, H, rowsq, rowsq_n, x_tiled, a_tiled, norm_w, src_rows, dst_rows, gt);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_embed_rowsq(rt_ctx* ctx, const GatherSrc* d_srcs, int n_src, const float* f32_table, const int32_t* d_idx, int idx_stride,
                       const int32_t* frame_ptr, int64_t idx_frame_stride, int M, int H, const float* add_vec, float* rowsq, int rowsq_n,
                       float* x_tiled, bf16_t* a_tiled, const float* norm_w, int32_t* frame_inc, unsigned* arrive) {
    if (M <= 0) return RT_OK;
    if (n_src > 16 || H % 8 || !x_tiled || !a_tiled || !norm_w || !rowsq || rowsq_n < 1 || (n_src > 0 && !d_srcs))
        return rt_fail(ctx, RT_ERR_INVALID, "embed_rowsq: n_src %d (<= 16), H %d (multiple of 8) or a missing buffer", n_src, H);
    if (frame_inc && (f32_table || !arrive || frame_inc != frame_ptr))
        return rt_fail(ctx, RT_ERR_INVALID, "embed_rowsq: the frame counter is advanced by launches over bf16 sources only, with an arrival counter");
    hipLaunchKernelGGL(k_embed_rowsq, dim3(M), dim3(256), 0, ctx->stream, d_srcs, n_src, f32_table, d_idx, idx_stride, frame_ptr,
                       idx_frame_stride, H, add_vec, rowsq, rowsq_n, x_tiled, a_tiled, norm_w, frame_inc, arrive);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_norm_tiled_rows(rt_ctx* ctx, const float* x_tiled, const float* rowsq, int rowsq_n, const float* w, float eps, int M, int H,
                           float* out) {
    if (M <= 0) return RT_OK;
    if (H % 8) return rt_fail(ctx, RT_ERR_INVALID, "norm_tiled_rows: H %d not a multiple of 8", H);
    hipLaunchKernelGGL(k_norm_tiled_rows, dim3(M), dim3(256), 0, ctx->stream, x_tiled, rowsq, rowsq_n, w, eps, H, out);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_silu_mul(rt_ctx* ctx, const float* slabs, int n_slabs, int M, int I, bf16_t* out, float* out_f32) {
    const int64_t total = (int64_t)M * I / 4;
    if (total <= 0) return RT_OK;
    if (I % 4) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "silu_mul: intermediate size %d not a multiple of 4", I);
    hipLaunchKernelGGL(k_silu_mul, dim3(grid_for(total, 64)), dim3(64), 0, ctx->stream, slabs, n_slabs, (int64_t)M * 2 * I, I, out, total, out_f32);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_reduce_slabs(rt_ctx* ctx, const float* slabs, int n_slabs, int64_t M, int N, const float* bias, int act, float* out_f32,
                        bf16_t* out_bf16) {
    const int64_t total = M * N;
    if (total <= 0) return RT_OK;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(grid_for(total)), dim3(256), 0, ctx->stream, slabs, n_slabs, total, N, bias, act, out_f32,
                       out_bf16, total);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_qkv_post(rt_ctx* ctx, const float* slabs, int n_slabs, int M, int heads, int kv_heads, int head_dim, const float* q_norm_w,
                    const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, const int32_t* row_slot,
                    const int32_t* row_pos, int pos_add, float* q_out, const KvCache& kv, int layer, const int32_t* frame_ptr) {
    if (M <= 0) return RT_OK;
    if (head_dim > 128 || (head_dim & 1)) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "head_dim %d unsupported (even, <= 128)", head_dim);
    const int width = (heads + 2 * kv_heads) * head_dim;
    hipLaunchKernelGGL(k_qkv_post, dim3(M, heads + 2 * kv_heads), dim3(64), 0, ctx->stream, slabs, n_slabs, (int64_t)M * width, heads,
                       kv_heads, head_dim, q_norm_w, k_norm_w, eps, rope_cos, rope_sin, row_slot, row_pos, pos_add, q_out,
                       kv.k + layer * kv.layer_stride(), kv.v + layer * kv.layer_stride(), kv.max_pos, frame_ptr,
                       kv.k_lo ? kv.k_lo + layer * kv.layer_stride() : nullptr, kv.v_lo ? kv.v_lo + layer * kv.layer_stride() : nullptr);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_gather_sum(rt_ctx* ctx, const GatherSrc* d_srcs, int n_src, const int32_t* d_idx, int M, int H, const float* add_vec,
                      const float* add_rows, const int32_t* add_row_idx, float* out_f32, bf16_t* out_bf16, int idx_stride,
                      const int32_t* frame_ptr, int64_t idx_frame_stride) {
    if (M <= 0) return RT_OK;
    hipLaunchKernelGGL(k_gather_sum, dim3(M), dim3(256), 0, ctx->stream, d_srcs, n_src, d_idx, H, add_vec, add_rows, add_row_idx, out_f32,
                       out_bf16, idx_stride > 0 ? idx_stride : n_src, frame_ptr, idx_frame_stride);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_gather_f32(rt_ctx* ctx, const float* table, int H, const int32_t* d_idx, int M, float* out_f32, bf16_t* out_bf16,
                      int idx_stride, const int32_t* frame_ptr, int64_t idx_frame_stride) {
    if (M <= 0) return RT_OK;
    hipLaunchKernelGGL(k_gather_f32, dim3(M), dim3(256), 0, ctx->stream, table, H, d_idx, out_f32, out_bf16, idx_stride, frame_ptr,
                       idx_frame_stride);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}


int launch_dwconv_ln(rt_ctx* ctx, const float* x, int B, int T, int C, const float* w, const float* b, const float* ln_w,
                     const float* ln_b, float eps, float* out_f32) {
    const int64_t rows = (int64_t)B * T;
    if (rows <= 0) return RT_OK;
    hipLaunchKernelGGL(k_dwconv_ln, dim3((unsigned)rows), dim3(256), C * sizeof(float), ctx->stream, x, T, C, w, b, ln_w, ln_b, eps, out_f32);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_code_embed_mean(rt_ctx* ctx, const bf16_t* table, int codebook, int Q, int H, const int32_t* codes, int64_t rows, float* out_f32) {
    if (rows <= 0) return RT_OK;
    hipLaunchKernelGGL(k_code_embed_mean, dim3((unsigned)rows), dim3(256), 0, ctx->stream, table, codebook, Q, H, codes, out_f32);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}


bool launch_final_conv_ok(int C) { return C % 16 == 0 && (size_t)(134 * (C + 1) + 7 * C) * 4 <= 64 * 1024; }

int launch_final_conv(rt_ctx* ctx, const bf16_t* hi, const bf16_t* lo, int B, int T, int C, const float* w, const float* bias, float* wav) {
    if (B <= 0 || T <= 0) return RT_OK;
    if (!launch_final_conv_ok(C)) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "final conv: %d channels do not fit the LDS window", C);
    const size_t lds = (size_t)(134 * (C + 1) + 7 * C) * 4;
    hipLaunchKernelGGL(k_final_conv, dim3((T + 127) / 128, B), dim3(256), lds, ctx->stream, hi, lo, T, C, w, bias, wav);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int launch_f32_to_bf16(rt_ctx* ctx, const float* x, int64_t n, bf16_t* out) {
    if (n <= 0) return RT_OK;
    hipLaunchKernelGGL(k_f32_to_bf16, dim3(grid_for(n)), dim3(256), 0, ctx->stream, x, n, out);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
