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
        const int fr = f - (A.frame_off ? A.frame_off[row] : 0);   // the row's own frame (rows are re-assigned to queued items)
        A.frame = fr;
        A.out += (int64_t)f * A.out_fs;
        if (A.eos_flag) A.eos_flag += (int64_t)f * A.eos_fs;
        if (A.forced) A.forced += (int64_t)f * A.forced_fs;
        if (A.logits_copy) A.logits_copy += (int64_t)f * A.copy_fs;
        A.allow_token = (A.eos_live && fr >= A.min_frames) ? A.eos_token : -1;
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
                    // running sums by the Kogge-Stone scan oracle/sampling.py defines (scan_f32), as in k_sample_w
                    float cum = p;
                    for (int d = 1; d < 64; d <<= 1) {
                        const float o = __shfl_up(cum, d, 64);
                        if (tid >= d) cum = __fadd_rn(cum, o);
                    }
                    float total = __shfl(cum, n - 1, 64);
                    int keep = n;
                    if (A.top_p < 1.0f) {
                        const float lim = __fmul_rn(A.top_p, total);
                        const unsigned long long reach = __ballot(tid < n && cum >= lim);
                        if (reach) { keep = __builtin_ctzll(reach) + 1; total = __shfl(cum, keep - 1, 64); }
                    }
                    const float u = rt_uniform(A.seed, (unsigned)A.item_ids[row], (unsigned)A.frame, (unsigned)A.group);
                    const float target = __fmul_rn(u, total);
                    const unsigned long long over = __ballot(tid < keep && cum > target);
                    const int pick_lane = over ? __builtin_ctzll(over) : keep - 1;
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

// ---------------------------------------------------------------------------------------------------------------
// Wave-native sampler (V <= 4096): every value lives in a register, the k-th largest is found by a bit-serial radix
// select made of 64-lane ballots (no LDS histograms, no atomics), each of the 4 waves keeps its own top-k, wave 0
// selects the final k out of those 4k with unique 64-bit (value, index) keys, sorts them with the one-wave bitonic
// network and walks the ordered CDF exactly as oracle/sampling.py defines it.
typedef __attribute__((ext_vector_type(4))) float f4s_t;
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ float lane_f32(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ int lane_i32(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

__device__ __forceinline__ float okey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// NW waves, NV values per thread: element j of thread (w, lane) has index (w * NV + j) * 64 + lane, so that inside a wave
// the index grows with (j, lane) and ties are resolved by position = by lowest index.
//   phase 1  every wave narrows ITS keys down to a superset of its own top-k: the bit-serial pivot search stops as soon as
//            at most 64 keys are >= the pivot (12-17 of the 32 bits on random logits) - no exact k-th key is needed here;
//   prune    a wave with >= k keys above its pivot proves that the global k-th key is >= that pivot: T = the largest such
//            pivot, and everything below T is dropped (typically ~100 candidates survive out of NW x 64);
//   phase 2  every surviving candidate counts the candidates ahead of it under the unique 64-bit key (value desc, index asc):
//            that rank is at once the exact top-k selection (rank < k, ties to the lowest indices) and the sort order;
//   walk     wave 0 holds candidate r on lane r and evaluates softmax / top-p / inverse CDF in oracle/sampling.py's order.
template <int NV, int NW>
__global__ __launch_bounds__(NW * 64) void k_sample_w(SampleArgs A) {
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) unsigned long long ckey[NW * 64];
    __shared__ unsigned long long sorted[64];
    __shared__ __attribute__((aligned(16))) unsigned hist[1024];
    __shared__ unsigned wpre[NW];
    __shared__ int wcnt[NW];
    __shared__ float sh_f[NW];
    __shared__ int sh_i[NW];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, V = A.V;
    const int64_t slab_stride = (int64_t)A.M * V;
    // all loads first (independent, in flight together): a load placed after the logits_copy store of the previous
    // element would have to wait for it (possible aliasing) and the loop degenerates into NV serial round trips
    // (and BRANCH-FREE: clamped indices and a stand-in pointer, the results masked afterwards - behind a per-lane `if (valid)`
    //  the compiler drains the load queue, `s_waitcnt vmcnt(0)`, after every single request: eight serial round trips, about
    //  4 us of a 10-us launch, is what the guarded form cost here).  They also come BEFORE the frame bookkeeping below: nothing
    //  they read depends on the frame index, so they need not wait for that scalar round trip.
    const float* __restrict__ lg = A.logits + (int64_t)row * V;
    const uint8_t* __restrict__ seen_r = A.seen ? A.seen + (int64_t)row * V : nullptr;
    const uint8_t* __restrict__ seen_or_any = seen_r ? seen_r : reinterpret_cast<const uint8_t*>(lg);
    // the row's item id (the random stream's key) rides along instead of costing a dependent round trip where it is used
    const unsigned item_id = (unsigned)*(A.item_ids ? reinterpret_cast<const int32_t*>(A.item_ids + row) : reinterpret_cast<const int32_t*>(lg));   // (low half)
    bool valid[NV];
    uint8_t sn[NV];
    float raw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (w * NV + j) * 64 + lane;
        valid[j] = i < V;
        const int ic = valid[j] ? i : V - 1;
        raw[j] = lg[ic];
        sn[j] = seen_or_any[ic];
    }
    if (A.seed_ptr) A.seed = *A.seed_ptr;
    if (A.frame_ptr) {     // graph replay: frame index and frame-strided buffers resolved on the device
        const int f = *A.frame_ptr;
        const int fr = f - (A.frame_off ? A.frame_off[row] : 0);   // the row's own frame (rows are re-assigned to queued items)
        A.frame = fr;
        A.out += (int64_t)f * A.out_fs;
        if (A.eos_flag) A.eos_flag += (int64_t)f * A.eos_fs;
        if (A.forced) A.forced += (int64_t)f * A.forced_fs;
        if (A.logits_copy) A.logits_copy += (int64_t)f * A.copy_fs;
        A.allow_token = (A.eos_live && fr >= A.min_frames) ? A.eos_token : -1;
    }
    uint8_t* seen = A.seen ? A.seen + (int64_t)row * V : nullptr;

    if (A.stamps && row == 0 && tid == 0) A.stamps[0] = wall_clock64();
    float v[NV];
    unsigned key[NV];
    // (the forced token's array moves with the frame: requested here, still well ahead of its use)
    const int forced_raw = *(A.forced ? A.forced + row : reinterpret_cast<const int32_t*>(lg));
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        v[j] = valid[j] ? 0.f + raw[j] : 0.f;
        sn[j] = (seen_r && valid[j]) ? sn[j] : (uint8_t)0;
    }
    for (int sb = 1; sb < A.n_slabs; ++sb) {
        float more[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (w * NV + j) * 64 + lane;
            more[j] = lg[sb * slab_stride + (valid[j] ? i : V - 1)];
        }
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (valid[j]) v[j] += more[j];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (w * NV + j) * 64 + lane;
        float l = -INFINITY;
        if (valid[j]) {
            l = v[j];
            if (A.logits_copy) A.logits_copy[(int64_t)row * V + i] = l;
            if (A.rep_penalty != 1.0f && sn[j]) l = l > 0.f ? __fdiv_rn(l, A.rep_penalty) : __fmul_rn(l, A.rep_penalty);
            if (i >= A.suppress_from && i != A.allow_token) l = -INFINITY;
            if (A.do_sample) l = __fdiv_rn(l, A.temperature);
        }
        v[j] = l;
        key[j] = valid[j] ? okey(l) : 0u;          // (okey(-inf) = 0x007FFFFF: every valid key is > 0)
    }

    if (A.stamps && row == 0 && tid == 0) A.stamps[1] = wall_clock64();
    int token = -1;
    const int forced = A.forced ? forced_raw : -1;
    bool need_argmax = !A.do_sample;
    if (forced >= 0) {
        token = forced;
        need_argmax = false;
    } else if (A.do_sample) {
        if (A.stamps && row == 0 && tid == 0) A.stamps[2] = wall_clock64();
        const int k = A.top_k < V ? A.top_k : V;
        // ---- fast path: ONE histogram over the distance from the row maximum.  d = max - v is monotone in v, and the top
        // 11 bits of the float d (exponent + 2 mantissa bits) are a log-scale bin, 19 % wide relative to the distance: the
        // bins up to the one where the running count reaches k hold a superset of the top-k that is typically k + 10..20.
        float mloc = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; ++j) mloc = (valid[j] && v[j] > mloc) ? v[j] : mloc;
        mloc = wave_max_rows_f32(mloc);              // (DPP + readlane: no LDS round trips, common.h)
        if (lane == 0) sh_f[w] = mloc;
        for (int i = tid; i < 1024; i += NT) hist[i] = 0u;
        if (tid == 0) { sh_i[2] = 0; sh_i[3] = 0; wcnt[0] = 0; }
        __syncthreads();
        float mx = sh_f[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) mx = sh_f[q] > mx ? sh_f[q] : mx;
        unsigned bin[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            bin[j] = __float_as_uint(__fsub_rn(mx, v[j])) >> 21;        // (max - (-inf) = +inf -> bin 1020; max = -inf -> NaN -> 1022)
            if (valid[j]) atomicAdd(&hist[bin[j]], 1u);
        }
        __syncthreads();
        const int kf = k < V ? k : V;                                   // every index < V is a candidate (suppressed ones as -inf)
        if (w == 0) {
            unsigned c16[16];
            unsigned sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[lane * 16 + q * 4]);
                c16[4 * q] = h4.x; c16[4 * q + 1] = h4.y; c16[4 * q + 2] = h4.z; c16[4 * q + 3] = h4.w;
                sum += h4.x + h4.y + h4.z + h4.w;
            }
            const unsigned incl = wave_scan_incl_u32(sum);
            unsigned run = incl - sum;
            if (run < (unsigned)kf && incl >= (unsigned)kf) {           // exactly one lane: the running count reaches k inside its 16 bins
                int bs = 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (run < (unsigned)kf) { run += c16[q]; bs = q; }
                }
                sh_i[2] = lane * 16 + bs; sh_i[3] = (int)run;
            }
        }
        __syncthreads();
        const unsigned bstar = (unsigned)sh_i[2];
        int n_c = sh_i[3];
        const bool fast = n_c <= NW * 64;
        if (fast) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const bool s = valid[j] && bin[j] <= bstar;
                const unsigned long long b_sel = __ballot(s);
                if (b_sel) {
                    int base = 0;
                    if (lane == (int)__builtin_ctzll(b_sel)) base = atomicAdd(&wcnt[0], __popcll(b_sel));
                    base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(b_sel));
                    if (s) ckey[base + lanes_below(b_sel)] = ((unsigned long long)key[j] << 32) | (0xFFFFFFFFu - (unsigned)((w * NV + j) * 64 + lane));
                }
            }
        } else {
        // ---- fallback (more than NW x 64 values share the bins up to the cut - e.g. a row of equal logits):
        // phase 1: narrow this wave's keys down to <= 64 that contain its own top-k
        int n_valid = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) n_valid += __popcll(__ballot(valid[j]));
        const int kw = k < n_valid ? k : n_valid;
        unsigned prefix = 1u;                       // key >= 1 <=> valid
        int cnt_p = n_valid;                        // number of keys >= prefix
        if (cnt_p > 64) {
            prefix = 0u;
            for (unsigned bm = 0x80000000u; bm != 0u; bm >>= 1) {
                const unsigned pivot = prefix | bm;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < NV; ++j) cnt += __popcll(__ballot(key[j] >= pivot));
                if (cnt >= kw) {
                    prefix = pivot; cnt_p = cnt;
                    if (cnt <= 64) break;
                }
            }
        }
        // (after all 32 bits the pivot is the wave's k-th key itself; more than 64 keys >= it means equal keys straddle the
        //  cut - e.g. a wave of suppressed tokens, all -inf: the lowest positions of the equal ones are kept)
        int need_eq = -1;
        if (cnt_p > 64) {
            int n_gt = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) n_gt += __popcll(__ballot(key[j] > prefix));
            need_eq = kw - n_gt;
        }
        if (lane == 0) wpre[w] = (kw == k && kw > 0) ? prefix : 0u;     // a wave with fewer than k keys proves nothing
        __syncthreads();
        unsigned T = 0u;
#pragma unroll
        for (int q = 0; q < NW; ++q) T = wpre[q] > T ? wpre[q] : T;
        bool sel[NV];
        int c_w = 0, tie_seen = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            bool s;
            if (need_eq < 0) {
                s = key[j] >= prefix;
            } else {
                const bool eq = key[j] == prefix;
                const unsigned long long b_eq = __ballot(eq);
                s = key[j] > prefix || (eq && tie_seen + lanes_below(b_eq) < need_eq);
                tie_seen += __popcll(b_eq);
            }
            sel[j] = s && kw > 0 && key[j] >= T;
            c_w += __popcll(__ballot(sel[j]));
        }
        if (lane == 0) wcnt[w] = c_w;
        __syncthreads();
        int off = 0;
        n_c = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) { const int cq = wcnt[q]; off += q < w ? cq : 0; n_c += cq; }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const unsigned long long b_sel = __ballot(sel[j]);
            if (sel[j]) ckey[off + lanes_below(b_sel)] = ((unsigned long long)key[j] << 32) | (0xFFFFFFFFu - (unsigned)((w * NV + j) * 64 + lane));
            off += __popcll(b_sel);
        }
        }
        if (A.stamps && row == 0 && tid == 0) A.stamps[3] = wall_clock64();
        __syncthreads();
        // ---- phase 2: rank = number of candidates ahead of mine (keys are unique: value, then lowest index) - at once the exact
        // top-k selection (rank < k) and the sort.  G consecutive lanes share one candidate and split the comparisons.
        {
            int slots = 64;
            while (slots < n_c) slots <<= 1;
            int G = NT / slots;
            G = G > 64 ? 64 : G;
            const int c = tid / G, part = tid - c * G;
            const unsigned long long my = c < n_c ? ckey[c] : 0ull;
            int rank = 0;
            int j = part;
            for (; j + 3 * G < n_c; j += 4 * G) {
                const unsigned long long o0 = ckey[j], o1 = ckey[j + G], o2 = ckey[j + 2 * G], o3 = ckey[j + 3 * G];
                rank += (o0 > my ? 1 : 0) + (o1 > my ? 1 : 0) + (o2 > my ? 1 : 0) + (o3 > my ? 1 : 0);
            }
            for (; j < n_c; j += G) rank += ckey[j] > my ? 1 : 0;
            if (G == 16) rank = group_sum_i32<16>(rank);
            else if (G == 8) rank = group_sum_i32<8>(rank);                  // (integer sums by DPP adds - every lane runs this; common.h)
            else if (G == 4) rank = group_sum_i32<4>(rank);
            else if (G == 2) rank = group_sum_i32<2>(rank);
            else for (int o = 1; o < G; o <<= 1) rank += __shfl_xor(rank, o, 64);
            if (part == 0 && c < n_c && rank < kf) sorted[rank] = my;
        }
        if (A.stamps && row == 0 && tid == 0) A.stamps[4] = wall_clock64();
        __syncthreads();
        if (w == 0) {
            const unsigned long long mk = lane < kf ? sorted[lane] : 0ull;
            Cand c;
            c.v = lane < kf ? okey_inv((unsigned)(mk >> 32)) : -INFINITY;
            c.idx = lane < kf ? (int)(0xFFFFFFFFu - (unsigned)mk) : 0x7fffffff;
            if (A.stamps && row == 0 && tid == 0) A.stamps[5] = wall_clock64();
            const unsigned long long fin_mask = __ballot(lane < kf && c.v > -INFINITY);
            const int n = __popcll(fin_mask);                                  // finite candidates are a prefix after the sort
            if (n > 0) {
                const float mx = lane_f32(c.v, 0);
                const float p = lane < n ? expf(__fsub_rn(c.v, mx)) : 0.f;
                // running sums in sorted order by a Kogge-Stone scan (6 shuffle + add steps, each add rounded to float32): cum_j
                // lands on lane j.  oracle/sampling.py defines the SAME evaluation order (scan_f32), so totals, the top-p cut and
                // the inverse-CDF pick are bit-identical on both sides; the 50-step scalar walk this replaces cost 1.8 us.
                float cum = p;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const float o = __shfl_up(cum, d, 64);
                    if (lane >= d) cum = __fadd_rn(cum, o);
                }
                const float mycum = cum;
                cum = lane_f32(mycum, n - 1);
                float tot = cum;
                int keep = n;
                if (A.top_p < 1.0f) {
                    const float lim = __fmul_rn(A.top_p, tot);
                    const unsigned long long reach = __ballot(lane < n && mycum >= lim);
                    if (reach) { keep = __builtin_ctzll(reach) + 1; tot = lane_f32(mycum, keep - 1); }
                }
                const float u = rt_uniform(A.seed, item_id, (unsigned)A.frame, (unsigned)A.group);
                const float target = __fmul_rn(u, tot);
                const unsigned long long over = __ballot(lane < keep && mycum > target);
                const int pick_lane = over ? __builtin_ctzll(over) : keep - 1;
                token = lane_i32(c.idx, pick_lane);
            }
            if (lane == 0) { sh_i[0] = n > 0 ? 0 : 1; sh_i[1] = token; }      // no finite candidate at all: fall back to the arg-max rule
        }
        __syncthreads();
        need_argmax = sh_i[0] != 0;
        token = sh_i[1];
        __syncthreads();
    }
    if (need_argmax) {
        // ---- arg-max, lowest index on ties (greedy result, and the fallback when no candidate is finite)
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = (w * NV + j) * 64 + lane;
            if (valid[j] && (v[j] > bv || (v[j] == bv && i < bi))) { bv = v[j]; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sh_f[w] = bv; sh_i[w] = bi; }
        __syncthreads();
        bv = sh_f[0]; bi = sh_i[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) if (sh_f[ww] > bv || (sh_f[ww] == bv && sh_i[ww] < bi)) { bv = sh_f[ww]; bi = sh_i[ww]; }
        if (bi == 0x7fffffff) bi = 0;
        token = bi;
        __syncthreads();
    }
    if (A.stamps && row == 0 && tid == 0) A.stamps[6] = wall_clock64();
    if (tid == 0) {
        if (seen && token >= 0 && token < V) seen[token] = 1;
        int is_eos = 0;
        if (A.eos_token >= 0 && token == A.eos_token) { is_eos = 1; token = 0; }
        if (A.eos_flag) A.eos_flag[row] = is_eos;
        A.out[(int64_t)row * A.out_stride] = token;
        sh_i[0] = token;
    }
    if (A.emb_table) {      // next pass's input straight from the drawn token (same arithmetic as k_embed_rowsq's f32-table mode:
        __syncthreads();    // 256 threads x 8 elements, wave sums, then waves 0..3 - threads beyond 256 only keep the barriers)
        const int tok = sh_i[0];
        const int H = A.emb_H;
        float ss = 0.f;
        if (tid < 256) {
            for (int c = tid * 8; c < H; c += 2048) {
                float e[8];
                const f4s_t a = *reinterpret_cast<const f4s_t*>(A.emb_table + (int64_t)tok * H + c);
                const f4s_t b = *reinterpret_cast<const f4s_t*>(A.emb_table + (int64_t)tok * H + c + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { e[q] = 0.f + a[q]; e[4 + q] = 0.f + b[q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) ss += e[q] * e[q];
                const int64_t o = tile_off(row, c, H);
                *reinterpret_cast<f4s_t*>(A.emb_x_tiled + o) = f4s_t{e[0], e[1], e[2], e[3]};
                *reinterpret_cast<f4s_t*>(A.emb_x_tiled + o + 4) = f4s_t{e[4], e[5], e[6], e[7]};
                const f4s_t w0 = *reinterpret_cast<const f4s_t*>(A.emb_norm_w + c), w1 = *reinterpret_cast<const f4s_t*>(A.emb_norm_w + c + 4);
                uint4 pk;
                pk.x = f32x2_to_bf16x2(w0[0] * e[0], w0[1] * e[1]);
                pk.y = f32x2_to_bf16x2(w0[2] * e[2], w0[3] * e[3]);
                pk.z = f32x2_to_bf16x2(w1[0] * e[4], w1[1] * e[5]);
                pk.w = f32x2_to_bf16x2(w1[2] * e[6], w1[3] * e[7]);
                *reinterpret_cast<uint4*>(A.emb_a_tiled + o) = pk;
            }
            ss = wave_sum_f32(ss);
        }
        __syncthreads();
        if (lane == 0 && w < 4) sh_f[w] = ss;
        __syncthreads();
        const float tot = ((sh_f[0] + sh_f[1]) + sh_f[2]) + sh_f[3];
        for (int j = tid; j < A.emb_rowsq_n; j += NT) A.emb_rowsq[(int64_t)row * A.emb_rowsq_n + j] = j == 0 ? tot : 0.f;
    }
}

}  // namespace

int launch_sample(rt_ctx* ctx, const SampleArgs& a) {
    if (a.M <= 0) return RT_OK;
    if (a.do_sample && (a.top_k < 1 || a.top_k > 64)) return rt_fail(ctx, RT_ERR_INVALID, "sampling needs 1 <= top_k <= 64 (got %d)", a.top_k);
    if (a.do_sample && !(a.temperature > 0.f)) return rt_fail(ctx, RT_ERR_INVALID, "sampling needs temperature > 0");
    if (a.V > 16384) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "vocabulary %d too large for the sampler", a.V);
    if (a.emb_table && (a.V > 4096 || a.emb_H % 8 || !a.emb_norm_w || !a.emb_rowsq || a.emb_rowsq_n < 1 || !a.emb_x_tiled || !a.emb_a_tiled))
        return rt_fail(ctx, RT_ERR_INVALID, "sample: fused embedding needs V <= 4096, H %% 8 == 0 and all its buffers");
    if (a.V <= 1024) hipLaunchKernelGGL((k_sample_w<4, 4>), dim3(a.M), dim3(256), 0, ctx->stream, a);
    else if (a.V <= 2048) hipLaunchKernelGGL((k_sample_w<4, 8>), dim3(a.M), dim3(512), 0, ctx->stream, a);
    else if (a.V <= 4096) hipLaunchKernelGGL((k_sample_w<4, 16>), dim3(a.M), dim3(1024), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_sample, dim3(a.M), dim3(256), a.V * sizeof(float), ctx->stream, a);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
