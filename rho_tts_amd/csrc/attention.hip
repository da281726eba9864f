// KV-cached attention for gfx950: one workgroup per (row, kv-head), K/V streamed straight to VGPRs
// (each K/V byte is used once: no LDS round trip), 16-B loads, D/8 lanes per cached position,
// wave-shuffle dot-product reduction, per-lane-group online softmax merged at the end.
// Serves the decode step (one row per sequence), the prompt prefill (row t sees rows <= t of its
// slot) and the codec pre-transformer (sliding window).  HBM-bound: 2 * ctx * head_dim * 2 bytes
// per (row, kv-head).
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

// FUSED (decode rows only, one row per sequence slot): `q` is the raw qkv projection [M][(heads+2kv)*D]; the workgroup
// first finishes its own head group - RMSNorm over the head, rotate-half RoPE, K/V rounded to bf16 and appended to
// the cache row (slot, pos) - then attends over the cache including the row it has just written.
// LO (codec pre-transformer, which has to stay float32-faithful for the waveform RMSE bar): the cache holds K/V as hi + lo
// bf16 planes (value = hi + lo, ~16 mantissa bits) and the output is written in float32.
template <int D, int REP, bool FUSED, int NW, bool LO = false>
__global__ __launch_bounds__(NW * 64) void k_attention(const float* __restrict__ q, int heads, int kv_heads,
                                                   const int32_t* __restrict__ row_slot, const int32_t* __restrict__ row_pos,
                                                   int pos_add, int window, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                   int max_pos, bf16_t* __restrict__ out, const float* __restrict__ qw,
                                                   const float* __restrict__ kw, float eps, const float* __restrict__ cosT,
                                                   const float* __restrict__ sinT, const int32_t* __restrict__ frame_ptr, int out_tiled, int prefix_slot,
                                                   int prefix_len, const bf16_t* __restrict__ kc_lo, const bf16_t* __restrict__ vc_lo,
                                                   float* __restrict__ out_f32, int slot_base, int pair_n) {
    constexpr int LPP = D / 8;        // lanes per cached position
    constexpr int PPW = 64 / LPP;     // positions per wave step
    // positions in flight per lane.  (8 - a talker row's whole ~500-position context in ONE batch requested before the prologue,
    // so that no second round trip is exposed - needs 64 registers for K / V alone: 100 spilled registers under the 128 a
    // 1024-thread workgroup may hold, 17.0 us against 10.7; measured in round 4.)
    constexpr int U = 4;
    __shared__ float sh[NW][REP][LPP][10];     // NW waves split the cached positions of one (row, kv head)
    __shared__ float sh_q[FUSED ? REP : 1][FUSED ? D : 1];
    __shared__ __attribute__((aligned(16))) bf16_t sh_kv[2][FUSED ? D : 8];   // the appended K / V row (as rounded for the cache)
    // pair_n > 0 (FUSED; the predictor's two-position first pass): rows >= pair_n sit ONE position behind row - pair_n of the same
    // slot, which this very launch appends.  Such a workgroup cannot read that position from the cache (another workgroup is
    // writing it), so it works out the partner's K / V itself - the same operations on the same operands, i.e. the bits the
    // cache will hold - keeps them here, and patches them over the cache read.  Replaces a k_qkv_post launch per layer.
    __shared__ __attribute__((aligned(16))) bf16_t sh_kv2[2][FUSED ? D : 8];

    // x = kv head (fastest): workgroups are dealt round-robin over the 8 XCDs, so with 8 kv heads every XCD's L2 holds
    // ONE head's shared-prefix K/V and serves it to all the rows of that head
    const int kh = blockIdx.x, row = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane % LPP, pg = lane / LPP;
    // FUSED: the wave's first head vector (q / k / v of the new position) and its norm weights depend on the row only - they are
    // requested BEFORE the row's position and slot are known, so their round trip overlaps the dependent scalar loads below
    // (frame -> position / slot) instead of following them; the RoPE factors are requested as soon as the position is known and
    // not behind the norm's reduction (three dependent memory round trips fewer per launch).
    constexpr int half = D / 2;
    const int width = (heads + 2 * kv_heads) * D;
    const bool act = lane < half;
    float pa = 0.f, pb = 0.f, pwa = 0.f, pwb = 0.f, pc = 0.f, psn = 0.f;
    // (branch-free on purpose: clamped lanes / vectors and a stand-in pointer instead of `if`s - behind a divergent branch the
    //  compiler drains every outstanding load, `s_waitcnt vmcnt(0)`, before it goes on, which would serialise the requests again)
    const int l0 = act ? lane : 0;
    if (FUSED) {
        const int v0 = w < REP + 2 ? w : REP + 1;
        const int col0 = v0 < REP ? (kh * REP + v0) * D : (v0 == REP ? (heads + kh) * D : (heads + kv_heads + kh) * D);
        const float* src = q + (int64_t)row * width + col0;
        pa = src[l0];
        pb = src[l0 + half];
        const float* nw = v0 < REP ? qw : kw;
        const float* nwp = nw ? nw : src;
        pwa = nwp[l0];
        pwb = nwp[l0 + half];
    }
    // (row_slot == nullptr: slot = slot_base + row; row_pos == nullptr: every row at pos_add - the predictor's passes, whose
    //  positions are known when the launch is recorded: no dependent scalar loads in front of the K / V requests at all)
    const int slot = row_slot ? row_slot[row] : slot_base + row;
    const int hi = (row_pos ? row_pos[row] : 0) + pos_add + (frame_ptr ? *frame_ptr : 0);
    if (FUSED) { pc = cosT[(int64_t)hi * half + l0]; psn = sinT[(int64_t)hi * half + l0]; }
    int lo = 0;
    if (window > 0 && hi - window + 1 > 0) lo = hi - window + 1;
    const float scale = rsqrtf((float)D);
    const bool partner = FUSED && pair_n > 0 && row >= pair_n && hi >= 1;     // (hi >= 1: a position-0 row has nothing in front of it)
    const int64_t base = ((int64_t)slot * kv_heads + kh) * max_pos;
    const bf16_t* kb = kc + base * D + sub * 8;
    const bf16_t* vb = vc + base * D + sub * 8;
    // rows below prefix_len come from the shared prefix slot (same bytes for every sequence -> cache hits)
    const int64_t pdelta = prefix_slot >= 0 ? (((int64_t)prefix_slot - slot) * kv_heads * max_pos) * D : 0;
    constexpr int STEP = NW * PPW;    // positions the workgroup covers per load slot
    u4_t kk[U], vv[U];
    u4_t kl[LO ? U : 1], vl[LO ? U : 1];
    auto load_batch = [&](int p0, int last) {      // positions p0 + u * STEP, clamped to [0, last]
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * STEP;
            int pc = p <= last ? p : last;
            if (pc < 0) pc = 0;
            const int64_t po = (int64_t)pc * D + (pc < prefix_len ? pdelta : 0);
            kk[u] = *reinterpret_cast<const u4_t*>(kb + po);
            vv[u] = *reinterpret_cast<const u4_t*>(vb + po);
            if (LO) {
                kl[LO ? u : 0] = *reinterpret_cast<const u4_t*>(kc_lo + base * D + sub * 8 + po);
                vl[LO ? u : 0] = *reinterpret_cast<const u4_t*>(vc_lo + base * D + sub * 8 + po);
            }
        }
    };
    const int p_first = lo + w * PPW + pg;
    // The cached positions of the first batch are requested BEFORE the q/k/v prologue, so their round trip overlaps the
    // prologue's own loads and reductions.  The row this launch appends (position hi) - and any position sharing a 128-B
    // line with it, which would sit stale in L1 - is left out and read after the barrier.
    constexpr int RPL = (D * 2 >= 128) ? 1 : 128 / (D * 2);
    const int pre_last = (hi / RPL) * RPL - 1;
    constexpr bool PRE = FUSED && REP <= 2;       // (4 query heads per kv head leave no registers for the early batch)
    if (PRE) load_batch(p_first, pre_last);

    if (FUSED) {
        // vectors of this head group: REP query heads, then K, then V; wave w takes vectors w, w+4, ...
        for (int vec = w; vec < REP + 2 + (partner ? 2 : 0); vec += NW) {
            const bool mate = vec >= REP + 2;        // K / V of the partner row (one position earlier), for this workgroup only
            const bool is_q = vec < REP, is_k = vec == REP || vec == REP + 2;
            const int col0 = is_q ? (kh * REP + vec) * D : (is_k ? (heads + kh) * D : (heads + kv_heads + kh) * D);
            const bool first = vec == w && !mate;    // (operands of the first vector are already on their way, see the top)
            const int vrow = mate ? row - pair_n : row, vpos = mate ? hi - 1 : hi;
            float a = pa, b = pb;
            if (!first && act) { a = q[(int64_t)vrow * width + col0 + lane]; b = q[(int64_t)vrow * width + col0 + lane + half]; }
            if (is_q || is_k) {
                const float* nw = is_q ? qw : kw;
                if (nw) {
                    float wa = pwa, wb = pwb;
                    if (!first && act) { wa = nw[lane]; wb = nw[lane + half]; }
                    const float ss = wave_sum_rows_f32(act ? a * a + b * b : 0.f);      // (wave-uniform: the vector loop is per wave)
                    const float inv = rsqrtf(ss / (float)D + eps);
                    if (act) { a = wa * (a * inv); b = wb * (b * inv); }
                }
                if (act) {
                    float c = pc, s = psn;
                    if (!first) { c = cosT[(int64_t)vpos * half + lane]; s = sinT[(int64_t)vpos * half + lane]; }
                    const float ra = a * c - b * s, rb = b * c + a * s;
                    a = ra; b = rb;
                }
            }
            if (act) {
                if (is_q) { sh_q[FUSED ? vec : 0][FUSED ? lane : 0] = a; sh_q[FUSED ? vec : 0][FUSED ? lane + half : 0] = b; }
                else if (mate) {                      // (the partner's own workgroup writes these to the cache)
                    sh_kv2[is_k ? 0 : 1][FUSED ? lane : 0] = f32_to_bf16(a);
                    sh_kv2[is_k ? 0 : 1][FUSED ? lane + half : 0] = f32_to_bf16(b);
                } else {
                    bf16_t* o = (is_k ? kc : vc) + (((int64_t)slot * kv_heads + kh) * max_pos + hi) * D;
                    const bf16_t ra = f32_to_bf16(a), rb = f32_to_bf16(b);
                    o[lane] = ra;
                    o[lane + half] = rb;
                    sh_kv[is_k ? 0 : 1][FUSED ? lane : 0] = ra;         // ... and kept in LDS: the workgroup attends to its own new
                    sh_kv[is_k ? 0 : 1][FUSED ? lane + half : 0] = rb;  // row without a round trip through global memory
                }
            }
        }
        __syncthreads();   // K/V row visible to the whole workgroup (same CU), q in LDS
    }

    float qr[REP][8];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (FUSED) qr[r][j] = sh_q[FUSED ? r : 0][FUSED ? sub * 8 + j : 0] * scale;
            else qr[r][j] = q[((int64_t)row * heads + kh * REP + r) * D + sub * 8 + j] * scale;
        }
    }
    float m[REP], l[REP], acc[REP][8];
#pragma unroll
    for (int r = 0; r < REP; ++r) {
        m[r] = -1e30f;
        l[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
    }
    for (int p0 = p_first; p0 <= hi; p0 += STEP * U) {
        if (PRE && p0 == p_first) {        // first batch: already in flight, patch the positions the prefetch had to skip
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * STEP;
                if (p == hi) {
                    kk[u] = *reinterpret_cast<const u4_t*>(&sh_kv[0][FUSED ? sub * 8 : 0]);
                    vv[u] = *reinterpret_cast<const u4_t*>(&sh_kv[1][FUSED ? sub * 8 : 0]);
                } else if (partner && p == hi - 1) {
                    kk[u] = *reinterpret_cast<const u4_t*>(&sh_kv2[0][FUSED ? sub * 8 : 0]);
                    vv[u] = *reinterpret_cast<const u4_t*>(&sh_kv2[1][FUSED ? sub * 8 : 0]);
                } else if (p > pre_last && p < hi) {
                    const int64_t po = (int64_t)p * D + (p < prefix_len ? pdelta : 0);
                    kk[u] = *reinterpret_cast<const u4_t*>(kb + po);
                    vv[u] = *reinterpret_cast<const u4_t*>(vb + po);
                }
            }
        } else {
            load_batch(p0, hi);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * NW * PPW;
            float kf[8], vf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = __uint_as_float(kk[u][j] << 16);
                kf[2 * j + 1] = __uint_as_float(kk[u][j] & 0xffff0000u);
                vf[2 * j] = __uint_as_float(vv[u][j] << 16);
                vf[2 * j + 1] = __uint_as_float(vv[u][j] & 0xffff0000u);
                if (LO) {
                    kf[2 * j] += __uint_as_float(kl[LO ? u : 0][j] << 16);
                    kf[2 * j + 1] += __uint_as_float(kl[LO ? u : 0][j] & 0xffff0000u);
                    vf[2 * j] += __uint_as_float(vl[LO ? u : 0][j] << 16);
                    vf[2 * j + 1] += __uint_as_float(vl[LO ? u : 0][j] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += qr[r][j] * kf[j];
                s = group_sum_f32<LPP>(s);          // (DPP: no LDS round trips - common.h)
                if (p <= hi) {   // uniform within the LPP-lane group
                    const float mn = fmaxf(m[r], s);
                    const float corr = __expf(m[r] - mn), pe = __expf(s - mn);
                    l[r] = l[r] * corr + pe;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][j] = acc[r][j] * corr + pe * vf[j];
                    m[r] = mn;
                }
            }
        }
    }
    // merge the PPW position groups of the wave (lanes sharing `sub`)
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int o = LPP; o < 64; o <<= 1) {
            const float m2 = __shfl_xor(m[r], o, 64), l2 = __shfl_xor(l[r], o, 64);
            const float mn = fmaxf(m[r], m2);
            const float c1 = __expf(m[r] - mn), c2 = __expf(m2 - mn);
            l[r] = l[r] * c1 + l2 * c2;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = acc[r][j] * c1 + __shfl_xor(acc[r][j], o, 64) * c2;
            m[r] = mn;
        }
        if (pg == 0) {
            sh[w][r][sub][0] = m[r];
            sh[w][r][sub][1] = l[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) sh[w][r][sub][2 + j] = acc[r][j];
        }
    }
    __syncthreads();
    // final merge over the NW waves: thread t -> (r, sub, j)
    for (int t = threadIdx.x; t < REP * LPP * 8; t += NW * 64) {
        const int j = t & 7, sb = (t >> 3) % LPP, r = t / (8 * LPP);
        float mn = -1e30f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) mn = fmaxf(mn, sh[ww][r][sb][0]);
        float lt = 0.f, at = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float c = __expf(sh[ww][r][sb][0] - mn);
            lt += sh[ww][r][sb][1] * c;
            at += sh[ww][r][sb][2 + j] * c;
        }
        const int kcol = (kh * REP + r) * D + sb * 8 + j;
        const int64_t oo = out_tiled ? tile_off(row, kcol, heads * D) : (int64_t)row * heads * D + kcol;
        if (LO) out_f32[(int64_t)row * heads * D + kcol] = lt > 0.f ? at / lt : 0.f;
        else out[oo] = f32_to_bf16(lt > 0.f ? at / lt : 0.f);
    }
}

struct FusedArgs { const float *qw, *kw, *cosT, *sinT; float eps; const int32_t* frame_ptr; int out_tiled; int prefix_slot, prefix_len;
                   const bf16_t *kc_lo = nullptr, *vc_lo = nullptr; float* out_f32 = nullptr; int slot_base = 0; int pair_n = 0; };

template <int D, bool FUSED, int NW, bool LO = false>
int dispatch_rep(rt_ctx* ctx, int rep, dim3 grid, const float* q, int heads, int kv_heads, const int32_t* rs, const int32_t* rp,
                 int pos_add, int window, bf16_t* kc, bf16_t* vc, int max_pos, bf16_t* out, const FusedArgs& f) {
    switch (rep) {
        case 1: hipLaunchKernelGGL((k_attention<D, 1, FUSED, NW, LO>), grid, dim3(NW * 64), 0, ctx->stream, q, heads, kv_heads, rs, rp, pos_add, window, kc, vc, max_pos, out, f.qw, f.kw, f.eps, f.cosT, f.sinT, f.frame_ptr, f.out_tiled, f.prefix_slot, f.prefix_len, f.kc_lo, f.vc_lo, f.out_f32, f.slot_base, f.pair_n); break;
        case 2: hipLaunchKernelGGL((k_attention<D, 2, FUSED, NW, LO>), grid, dim3(NW * 64), 0, ctx->stream, q, heads, kv_heads, rs, rp, pos_add, window, kc, vc, max_pos, out, f.qw, f.kw, f.eps, f.cosT, f.sinT, f.frame_ptr, f.out_tiled, f.prefix_slot, f.prefix_len, f.kc_lo, f.vc_lo, f.out_f32, f.slot_base, f.pair_n); break;
        case 4: hipLaunchKernelGGL((k_attention<D, 4, FUSED, NW, LO>), grid, dim3(NW * 64), 0, ctx->stream, q, heads, kv_heads, rs, rp, pos_add, window, kc, vc, max_pos, out, f.qw, f.kw, f.eps, f.cosT, f.sinT, f.frame_ptr, f.out_tiled, f.prefix_slot, f.prefix_len, f.kc_lo, f.vc_lo, f.out_f32, f.slot_base, f.pair_n); break;
        default: return rt_fail(ctx, RT_ERR_UNSUPPORTED, "attention: heads/kv_heads = %d unsupported (1, 2, 4)", rep);
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

template <bool FUSED>
int attention_any(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, int head_dim, const int32_t* row_slot, const int32_t* row_pos,
                  int pos_add, int window, const KvCache& kv, int layer, bf16_t* out, const FusedArgs& f0) {
    if (M <= 0) return RT_OK;
    FusedArgs f = f0;
    f.prefix_slot = kv.prefix_slot;
    f.prefix_len = kv.prefix_slot >= 0 ? kv.prefix_len : 0;
    if (heads % kv_heads) return rt_fail(ctx, RT_ERR_INVALID, "attention: heads %d not a multiple of kv_heads %d", heads, kv_heads);
    const int rep = heads / kv_heads;
    bf16_t* kc = kv.k + layer * kv.layer_stride();
    bf16_t* vc = kv.v + layer * kv.layer_stride();
    dim3 grid(kv_heads, M);
    if (kv.k_lo) {      // hi + lo planes, float32 output: the codec pre-transformer (many rows, short windows: 4 waves)
        if (FUSED || !f.out_f32 || !kv.v_lo) return rt_fail(ctx, RT_ERR_INVALID, "attention: hi/lo cache needs the unfused form and a float32 output");
        f.kc_lo = kv.k_lo + layer * kv.layer_stride();
        f.vc_lo = kv.v_lo + layer * kv.layer_stride();
        switch (head_dim) {
            case 32: return dispatch_rep<32, false, 4, true>(ctx, rep, grid, q, heads, kv_heads, row_slot, row_pos, pos_add, window, kc, vc, kv.max_pos, out, f);
            case 64: return dispatch_rep<64, false, 4, true>(ctx, rep, grid, q, heads, kv_heads, row_slot, row_pos, pos_add, window, kc, vc, kv.max_pos, out, f);
            case 128: return dispatch_rep<128, false, 4, true>(ctx, rep, grid, q, heads, kv_heads, row_slot, row_pos, pos_add, window, kc, vc, kv.max_pos, out, f);
            default: return rt_fail(ctx, RT_ERR_UNSUPPORTED, "attention: head_dim %d unsupported (32, 64, 128)", head_dim);
        }
    }
    // few rows with long contexts (the talker's decode step over a long KV row): 16 waves split the positions;
    // many rows or short contexts (prefill, predictor, sliding window): 4 waves are plenty
    // (decode steps only - the fused form, or frame-indexed rows: a PROMPT row must be attended with the same wave split,
    //  i.e. the same float32 merge order, whether it is prefilled alone or among 400 rows - a text's audio may not depend on
    //  what it was batched with)
    const bool wide = (FUSED || f0.frame_ptr) && M * kv_heads <= 512 && kv.max_pos > 64 && (window <= 0 || window > 256);
#define RT_ATT(DD)                                                                                                                   \
    return wide ? dispatch_rep<DD, FUSED, 16>(ctx, rep, grid, q, heads, kv_heads, row_slot, row_pos, pos_add, window, kc, vc, kv.max_pos, out, f) \
                : dispatch_rep<DD, FUSED, 4>(ctx, rep, grid, q, heads, kv_heads, row_slot, row_pos, pos_add, window, kc, vc, kv.max_pos, out, f)
    switch (head_dim) {
        case 32: RT_ATT(32);
        case 64: RT_ATT(64);
        case 128: RT_ATT(128);
        default: return rt_fail(ctx, RT_ERR_UNSUPPORTED, "attention: head_dim %d unsupported (32, 64, 128)", head_dim);
    }
#undef RT_ATT
}

}  // namespace

int launch_attention(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, int head_dim, const int32_t* row_slot,
                     const int32_t* row_pos, int pos_add, int window, const KvCache& kv, int layer, bf16_t* out, const int32_t* frame_ptr, int out_tiled,
                     float* out_f32) {
    // prompt rows behind a shared voice prefix: the matrix-core form (attention_mfma.hip).  The choice depends on the model and
    // the voice only, never on M: a text gets the same bits prefilled alone, in the first wave or at a hand-over
    if (!frame_ptr && !out_tiled && !out_f32 && out && attention_prefill_mfma_ok(heads, kv_heads, head_dim, window, kv))
        return launch_attention_prefill_mfma(ctx, q, M, heads, kv_heads, row_slot, row_pos, pos_add, kv, layer, out);
    FusedArgs f{};
    f.frame_ptr = frame_ptr;
    f.out_tiled = out_tiled;
    f.out_f32 = out_f32;
    return attention_any<false>(ctx, q, M, heads, kv_heads, head_dim, row_slot, row_pos, pos_add, window, kv, layer, out, f);
}

int launch_attention_fused(rt_ctx* ctx, const float* qkv, int M, int heads, int kv_heads, int head_dim, const float* q_norm_w,
                           const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, const int32_t* row_slot,
                           const int32_t* row_pos, int pos_add, int window, const KvCache& kv, int layer, bf16_t* out,
                           const int32_t* frame_ptr, int out_tiled, int slot_base, int pair_n) {
    if (!row_slot && (slot_base < 0 || slot_base + M > kv.slots))
        return rt_fail(ctx, RT_ERR_INVALID, "attention: rows %d..%d without a slot array do not fit the %d cache slots", slot_base, slot_base + M - 1, kv.slots);
    // pair_n: rows [pair_n, 2 pair_n) of the launch sit one position behind rows [0, pair_n) of the same slots (see the kernel)
    if (pair_n && (pair_n < 0 || M != 2 * pair_n || !row_slot || !row_pos || heads % kv_heads || heads / kv_heads > 2))
        return rt_fail(ctx, RT_ERR_INVALID, "attention: paired rows need M = 2 x %d rows with slot and position arrays and <= 2 query heads per kv head", pair_n);
    if (!pair_n && row_slot && row_pos && attention_mfma_ok(M, heads, kv_heads, head_dim, window, kv))
        return launch_attention_prefix_mfma(ctx, qkv, M, heads, kv_heads, q_norm_w, k_norm_w, eps, rope_cos, rope_sin, row_slot, row_pos, pos_add, kv, layer,
                                            out, frame_ptr, out_tiled);
    FusedArgs f{q_norm_w, k_norm_w, rope_cos, rope_sin, eps, frame_ptr, out_tiled, -1, 0, nullptr, nullptr, nullptr, slot_base, pair_n};
    return attention_any<true>(ctx, qkv, M, heads, kv_heads, head_dim, row_slot, row_pos, pos_add, window, kv, layer, out, f);
}
