// Token sampler for gfx950: one 256-thread workgroup per row of logits.
// Greedy = arg-max with lowest-index tie break.  Sampling = repetition penalty, suppression mask,
// temperature, exact top-k (k <= 64) by radix select on order-preserving keys, candidates sorted by
// (logit desc, index asc) with a one-wave bitonic network, then the fully ordered float32
// softmax / top-p / inverse-CDF walk that oracle/sampling.py defines, driven by the same
// counter-hash uniform keyed by (seed, item, frame, group).
#include "kernels.h"

namespace {

__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float rt_uniform(uint64_t seed, unsigned item, unsigned frame, unsigned group) {
    const unsigned a = mix32((unsigned)(seed & 0xffffffffu) ^ 0x85EBCA6Bu);
    const unsigned b = mix32(a + item * 0x9E3779B1u);
    const unsigned c = mix32(b + frame * 0x85EBCA77u);
    const unsigned d = mix32(c + group * 0xC2B2AE3Du + (unsigned)(seed >> 32));
    return __fmul_rn(__fadd_rn((float)(d >> 8), 0.5f), 1.0f / 16777216.0f);
}
__device__ __forceinline__ unsigned okey(float f) {  // larger float -> larger key; -inf smallest
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Cand { float v; int idx; };
__device__ __forceinline__ bool before(const Cand& a, const Cand& b) { return a.v > b.v || (a.v == b.v && a.idx < b.idx); }

__global__ __launch_bounds__(256) void k_sample(SampleArgs A) {
    extern __shared__ float vals[];          // [V]
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_u[8];
    __shared__ float sh_f[4];
    __shared__ int sh_i[4];
    __shared__ Cand cand[64];
    const int row = blockIdx.x, tid = threadIdx.x, V = A.V;
    const int64_t slab_stride = (int64_t)A.M * V;
    if (A.seed_ptr) A.seed = *A.seed_ptr;
    if (A.frame_ptr) {     // graph replay: frame index and frame-strided buffers resolved on the device
        const int f = *A.frame_ptr;
        A.frame = f;
        A.out += (int64_t)f * A.out_fs;
        if (A.eos_flag) A.eos_flag += (int64_t)f * A.eos_fs;
        if (A.forced) A.forced += (int64_t)f * A.forced_fs;
        if (A.logits_copy) A.logits_copy += (int64_t)f * A.copy_fs;
        A.allow_token = (A.eos_live && f >= A.min_frames) ? A.eos_token : -1;
    }
    uint8_t* seen = A.seen ? A.seen + (int64_t)row * V : nullptr;

    for (int i = tid; i < V; i += 256) {
        float l = 0.f;
        for (int s = 0; s < A.n_slabs; ++s) l += A.logits[s * slab_stride + (int64_t)row * V + i];
        if (A.logits_copy) A.logits_copy[(int64_t)row * V + i] = l;
        if (seen && A.rep_penalty != 1.0f && seen[i]) l = l > 0.f ? __fdiv_rn(l, A.rep_penalty) : __fmul_rn(l, A.rep_penalty);
        if (i >= A.suppress_from && i != A.allow_token) l = -INFINITY;
        if (A.do_sample) l = __fdiv_rn(l, A.temperature);
        vals[i] = l;
    }
    __syncthreads();

    int token = -1;
    const int forced = A.forced ? A.forced[row] : -1;
    if (forced >= 0) {
        token = forced;
    } else {
        // ---- arg-max (needed by greedy, and as the fallback when no candidate is finite)
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) { const float v = vals[i]; if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sh_f[tid >> 6] = bv; sh_i[tid >> 6] = bi; }
        __syncthreads();
        bv = sh_f[0]; bi = sh_i[0];
        for (int w = 1; w < 4; ++w) if (sh_f[w] > bv || (sh_f[w] == bv && sh_i[w] < bi)) { bv = sh_f[w]; bi = sh_i[w]; }
        if (bi == 0x7fffffff) bi = 0;
        token = bi;

        if (A.do_sample) {
            int k = A.top_k < V ? A.top_k : V;
            // ---- radix select: key of the k-th largest value
            unsigned prefix = 0, mask = 0;
            int remaining = k;
            if (tid == 0) sh_u[6] = 0;
            for (int pass = 3; pass >= 0; --pass) {
                hist[tid] = 0;
                __syncthreads();
                for (int i = tid; i < V; i += 256) {
                    const unsigned key = okey(vals[i]);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> (pass * 8)) & 255u], 1u);
                }
                __syncthreads();
                if (tid < 64) {   // wave 0: descending cumulative count over the 256 bins, 4 bins per lane
                    const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                    const unsigned tot = c0 + c1 + c2 + c3;
                    unsigned suf = tot;   // inclusive suffix sum over lanes >= tid
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned t = __shfl_down(suf, o, 64);
                        if (tid + o < 64) suf += t;
                    }
                    const unsigned above = suf - tot;
                    if (tid == 0) { sh_u[0] = 0; sh_u[1] = suf - c0; }          // fallback: bin 0
                    if (above < (unsigned)remaining && suf >= (unsigned)remaining) {
                        unsigned cum = above;
                        const unsigned c[4] = {c0, c1, c2, c3};
                        int pick = 0;
#pragma unroll
                        for (int b = 3; b >= 0; --b) {
                            if (cum + c[b] >= (unsigned)remaining) { pick = b; break; }
                            cum += c[b];
                        }
                        sh_u[4] = 4 * tid + pick; sh_u[5] = cum; sh_u[6] = 1;
                    }
                }
                __syncthreads();
                if (tid == 0) { if (sh_u[6]) { sh_u[0] = sh_u[4]; sh_u[1] = sh_u[5]; } sh_u[6] = 0; }
                __syncthreads();
                prefix |= sh_u[0] << (pass * 8);
                mask |= 0xffu << (pass * 8);
                remaining -= (int)sh_u[1];
                __syncthreads();
            }
            const unsigned thr = prefix;      // key of the k-th largest; `remaining` ties at thr are wanted
            if (tid == 0) { sh_u[2] = 0; sh_u[3] = 0; }
            if (tid < 64) { cand[tid].v = -INFINITY; cand[tid].idx = 0x7fffffff; }
            __syncthreads();
            int my_ties = 0;
            for (int i = tid; i < V; i += 256) {
                const unsigned key = okey(vals[i]);
                if (key > thr) { const unsigned s = atomicAdd(&sh_u[2], 1u); if (s < 64) { cand[s].v = vals[i]; cand[s].idx = i; } }
                else if (key == thr) ++my_ties;
            }
            if (my_ties) atomicAdd(&sh_u[3], (unsigned)my_ties);
            __syncthreads();
            const int n_gt = (int)sh_u[2], n_tie = (int)sh_u[3];
            if (n_tie == remaining) {
                for (int i = tid; i < V; i += 256)
                    if (okey(vals[i]) == thr) { const unsigned s = atomicAdd(&sh_u[2], 1u); if (s < 64) { cand[s].v = vals[i]; cand[s].idx = i; } }
            } else if (tid == 0) {           // rare: more equal values than wanted -> lowest indices win
                int s = n_gt, need = remaining;
                for (int i = 0; i < V && need > 0; ++i)
                    if (okey(vals[i]) == thr) { if (s < 64) { cand[s].v = vals[i]; cand[s].idx = i; } ++s; --need; }
            }
            __syncthreads();
            if (tid < 64) {
                // ---- bitonic sort of 64 candidates by (value desc, index asc), one per lane
                Cand c = cand[tid];
                if (tid >= k) { c.v = -INFINITY; c.idx = 0x7fffffff; }
                for (int size = 2; size <= 64; size <<= 1)
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        Cand o; o.v = __shfl_xor(c.v, stride, 64); o.idx = __shfl_xor(c.idx, stride, 64);
                        const bool up = ((tid & size) == 0);          // ascending "before" order in this block
                        const bool lower = ((tid & stride) == 0);
                        const bool take_o = (lower == up) ? before(o, c) : before(c, o);
                        if (take_o) c = o;
                    }
                cand[tid] = c;
            }
            __syncthreads();
            if (tid < 64) {
                // wave 0 holds the sorted candidates one per lane; every sum is accumulated in candidate order
                // (the oracle's order) by broadcasting lane j to all lanes - sequential semantics, no LDS round trips
                const Cand c = cand[tid];
                const unsigned long long fin = __ballot(tid < k && c.v > -INFINITY);
                const int n = __popcll(fin);                                  // finite candidates are a prefix after the sort
                if (n > 0) {
                    const float mx = __shfl(c.v, 0, 64);
                    const float p = tid < n ? expf(__fsub_rn(c.v, mx)) : 0.f;
                    float total = 0.f;
                    for (int j = 0; j < n; ++j) total = __fadd_rn(total, __shfl(p, j, 64));
                    int keep = n;
                    if (A.top_p < 1.0f) {
                        const float lim = __fmul_rn(A.top_p, total);
                        float cum = 0.f;
                        for (int j = 0; j < n; ++j) { cum = __fadd_rn(cum, __shfl(p, j, 64)); if (cum >= lim) { keep = j + 1; break; } }
                        total = cum;
                    }
                    const float u = rt_uniform(A.seed, (unsigned)A.item_ids[row], (unsigned)A.frame, (unsigned)A.group);
                    const float target = __fmul_rn(u, total);
                    float cum = 0.f;
                    int pick_lane = keep - 1;
                    for (int j = 0; j < keep; ++j) { cum = __fadd_rn(cum, __shfl(p, j, 64)); if (cum > target) { pick_lane = j; break; } }
                    const int pick = __shfl(c.idx, pick_lane, 64);
                    if (tid == 0) sh_i[0] = pick;
                } else if (tid == 0) {
                    sh_i[0] = token;
                }
            }
            __syncthreads();
            token = sh_i[0];
        }
    }
    if (tid == 0) {
        if (seen && token >= 0 && token < V) seen[token] = 1;
        int is_eos = 0;
        if (A.eos_token >= 0 && token == A.eos_token) { is_eos = 1; token = 0; }
        if (A.eos_flag) A.eos_flag[row] = is_eos;
        A.out[(int64_t)row * A.out_stride] = token;
    }
}

}  // namespace

int launch_sample(rt_ctx* ctx, const SampleArgs& a) {
    if (a.M <= 0) return RT_OK;
    if (a.do_sample && (a.top_k < 1 || a.top_k > 64)) return rt_fail(ctx, RT_ERR_INVALID, "sampling needs 1 <= top_k <= 64 (got %d)", a.top_k);
    if (a.do_sample && !(a.temperature > 0.f)) return rt_fail(ctx, RT_ERR_INVALID, "sampling needs temperature > 0");
    if (a.V > 16384) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "vocabulary %d too large for the sampler", a.V);
    hipLaunchKernelGGL(k_sample, dim3(a.M), dim3(256), a.V * sizeof(float), ctx->stream, a);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
