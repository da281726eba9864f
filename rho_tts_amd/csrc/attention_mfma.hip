// Decode-step attention over a SHARED voice prefix with the matrix cores (gfx950).
//
// In the talker's decode step every sequence of the batch attends to the same ~460 prefix rows (the voice prompt, kept once in
// its own cache slot) plus its own few dozen rows.  k_attention (attention.hip) gives each (row, kv head) its own workgroup, so
// the prefix K/V of a head is re-read from L2 by 32 workgroups (60 MB per launch) and multiplied on the vector unit.  Here one
// 1024-thread workgroup owns a kv head and FOUR rows (4 rows x 2 query heads = 8 query vectors, padded to the 16 rows of
// v_mfma_f32_16x16x32_bf16): the prefix is read once per four rows and both products run as MFMAs -
//   S^T[key][q]  = K[keys x d] . Q^T[d x q]          (A = a K tile straight from the cache rows, B = Q^T from LDS, hi + lo planes)
//   O[q][d]     += P[q x keys] . V[keys x d]         (A = P: the S^T accumulator IS the A layout when the 32 keys of a block are
//                                                     dealt to the MFMA rows as key = 8 g + i (+4 for the second tile); B = V)
// Both prefix operands are read from FRAGMENT-TILED copies of the prefix K / V made once per voice (k_tile_prefix_kv): every
// operand fetch is one fully coalesced 1-KiB wave load (a plain transposed V cost 64 cache lines per load and made the first
// version of this kernel slower than the vector-unit one)
// - the fused prologue (q/k RMSNorm, RoPE, K/V append: one vector per wave), the rows' own suffix positions (vector unit, as
// k_attention) and the merge of all partial softmax states stay in the same launch.  16 waves: wave w takes prefix block w.
#include <algorithm>

#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((ext_vector_type(4))) float f4acc_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

__device__ __forceinline__ f4acc_t mfma16(s8_t a, s8_t b, f4acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), c, 0, 0, 0);
}

constexpr int D = 128, REP = 2;
constexpr int LPP = D / 8, PPW = 64 / LPP;                              // suffix part: lanes per position, positions per wave step
// Two instantiations:
//   decode  <ROWS 4, 16 waves, PREFILL false>: the fused decode step described above (4 rows x 2 heads = 8 of the 16 MFMA columns)
//   prefill <ROWS 8,  8 waves, PREFILL true>: PROMPT rows (the texts' suffixes behind a shared voice prefix; q already normed /
//           roped and K / V already appended by k_qkv_post): 8 rows x 2 heads fill the 16 MFMA columns, one wave per row takes
//           the row's own positions.  k_attention re-reads the 235 KB of prefix K / V per (row, kv head) from L2 - 416 suffix
//           rows x 8 heads x 28 layers per batch; here it is read once per 8 rows and multiplied on the matrix cores.
//           A row's result depends on nothing but the row: the prefix part is one MFMA column per query vector with the key
//           blocks dealt to the waves in a fixed order, the own part is one wave - so a text prefilled alone, among 400 rows or at
//           a hand-over gets the same bits (the property continuous batching rests on), whatever 8-row block it falls into.

struct MfmaAttnArgs {
    const float* qkv;                 // decode: raw projections [M][(heads + 2 kv) D]; prefill: q [M][heads][D], normed and roped
    int M, heads, kv_heads;
    const int32_t* row_slot; const int32_t* row_pos; int pos_add;
    bf16_t* kc; bf16_t* vc; int max_pos;
    const bf16_t* kt;                 // fragment-tiled prefix K / V of this layer: [kv_heads][vt_stride]
    const bf16_t* vt;
    int vt_stride;
    bf16_t* out; const float* qw; const float* kw; float eps; const float* cosT; const float* sinT;
    const int32_t* frame_ptr; int out_tiled; int prefix_slot, prefix_len;
    int block_prefix = 0;             // prefill of the prefix slot ITSELF (consecutive rows of one slot, causal): the keys in front of a
                                      // workgroup's first row play the part of the shared prefix, read from the tiles of the same slot
};

template <int ROWS, int NW, bool PREFILL>
__global__ __launch_bounds__(NW * 64) void k_attn_prefix_mfma(MfmaAttnArgs g) {
    constexpr int QR = ROWS * REP;                                         // live query vectors of the 16 MFMA columns
    constexpr int WPR = NW / ROWS;                                         // waves per row for the rows' own positions
    static_assert(QR <= 16 && NW % ROWS == 0 && (PREFILL || NW >= 4 * ROWS), "wave layout");
    constexpr int QP = D + 8;                                              // row pitch 272 B: the 16 rows of a b128 fragment read land on distinct banks
    __shared__ __attribute__((aligned(16))) bf16_t sq_hi[16][QP];         // Q (scaled), bf16 hi / lo planes, rows >= QR zero
    __shared__ __attribute__((aligned(16))) bf16_t sq_lo[16][QP];
    __shared__ float sq_f[QR][D];                                          // the same in float32 for the suffix part
    __shared__ __attribute__((aligned(16))) bf16_t s_kv[PREFILL ? 1 : ROWS][2][D];   // the appended K / V rows (as rounded for the cache); decode only
    __shared__ float p_m[NW][QR], p_l[NW][QR];                             // prefix partials per wave
    __shared__ float p_acc[NW][QR][D];
    __shared__ float s_part[NW][REP][LPP][10];                             // suffix partials per wave
    __shared__ int s_hi[ROWS];

    const int kh = blockIdx.x, r0 = blockIdx.y * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int heads = g.heads, kv_heads = g.kv_heads;
    const float scale = rsqrtf((float)D);
    const int frame = g.frame_ptr ? *g.frame_ptr : 0;

    // ---------------------------------------------------------------- A: fused prologue, one vector per wave
    for (int t = tid; t < 16 * QP; t += NW * 64) { (&sq_hi[0][0])[t] = 0; (&sq_lo[0][0])[t] = 0; }
    // decode: the operands of the wave's prologue vector (q / k / v of the new position, norm weights, RoPE factors) are requested
    // FIRST - vector-memory loads return in issue order, so behind the 16 KiB of prefix tiles below they would arrive last - and
    // without branches (clamped row, stand-in pointers): waves past 4 x ROWS load a vector they never use
    constexpr int half_d = D / 2;
    float pro_a = 0.f, pro_b = 0.f, pro_wa = 1.f, pro_wb = 1.f, pro_c = 1.f, pro_s = 0.f;
    int pro_hi = 0, pro_slot = 0;
    if constexpr (!PREFILL) {
        const int rr = (w >> 2) < ROWS ? (w >> 2) : ROWS - 1, vec = w & 3;
        const int row = r0 + rr < g.M ? r0 + rr : g.M - 1;
        const int width = (heads + 2 * kv_heads) * D;
        const int col0 = vec < REP ? (kh * REP + vec) * D : (vec == REP ? (heads + kh) * D : (heads + kv_heads + kh) * D);
        const float* src = g.qkv + (int64_t)row * width + col0;
        pro_a = src[lane]; pro_b = src[lane + half_d];
        const float* nw = vec < REP ? g.qw : g.kw;
        const float* nwp = nw ? nw : src;
        pro_wa = nwp[lane]; pro_wb = nwp[lane + half_d];
        pro_hi = g.row_pos[row] + g.pos_add + frame;
        pro_slot = g.row_slot[row];
        pro_c = g.cosT[(int64_t)pro_hi * half_d + lane]; pro_s = g.sinT[(int64_t)pro_hi * half_d + lane];
    }
    // the wave's first prefix block is requested BEFORE the prologue: it depends on nothing the prologue produces, so its L2
    // round trip runs under the prologue's own loads and reductions
    int Lp = g.prefix_len;
    if (PREFILL && g.block_prefix) {                         // (uniform per workgroup; rows r0 .. r0 + ROWS - 1 sit at consecutive positions)
        Lp = g.row_pos[r0 < g.M ? r0 : g.M - 1] + g.pos_add + frame;
        if (Lp < 0) Lp = 0;
    }
    const int gq = lane >> 4, n16 = lane & 15;               // MFMA lane split: k-octet / row-or-column
    const s8_t* ktp = reinterpret_cast<const s8_t*>(g.kt + (int64_t)kh * g.vt_stride) + lane;      // [block][tile 0..7][64 lanes] x 16 B
    const s8_t* vtp = reinterpret_cast<const s8_t*>(g.vt + (int64_t)kh * g.vt_stride) + lane;
    const int n_blk = (Lp + 31) >> 5;
    s8_t kf[2][4], vf[8];
    {
        const int blk = w < n_blk ? w : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dk = 0; dk < 4; ++dk) kf[t][dk] = ktp[((int64_t)blk * 8 + t * 4 + dk) * 64];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) vf[dt] = vtp[((int64_t)blk * 8 + dt) * 64];
    }
    __syncthreads();
    if constexpr (PREFILL) {
        // query vector qi = rr * REP + head of the block's rows: one wave per vector and turn; K / V are in the cache already
        for (int qi = w; qi < QR; qi += NW) {
            const int rr = qi / REP, vec = qi % REP, row = r0 + rr;
            constexpr int half = D / 2;
            if (row < g.M) {
                if (vec == 0 && lane == 0) s_hi[rr] = g.row_pos[row] + g.pos_add + frame;
                const float* qp = g.qkv + ((int64_t)row * heads + kh * REP + vec) * D;
                const float qa = qp[lane] * scale, qb = qp[lane + half] * scale;
                sq_f[qi][lane] = qa; sq_f[qi][lane + half] = qb;
                const bf16_t ha = f32_to_bf16(qa), hb = f32_to_bf16(qb);
                sq_hi[qi][lane] = ha; sq_hi[qi][lane + half] = hb;
                sq_lo[qi][lane] = f32_to_bf16(qa - bf16_to_f32(ha)); sq_lo[qi][lane + half] = f32_to_bf16(qb - bf16_to_f32(hb));
            } else if (vec == 0 && lane == 0) {
                s_hi[rr] = -1;
            }
        }
    } else {
        const int rr = w >> 2, vec = w & 3;                 // local row, vector: 0,1 = the two query heads, 2 = K, 3 = V
        const int row = r0 + rr;
        if (rr < ROWS && row < g.M) {
            constexpr int half = D / 2;
            const bool is_q = vec < REP, is_k = vec == REP;
            const int hi = pro_hi, slot = pro_slot;
            if (vec == 0 && lane == 0) s_hi[rr] = hi;
            float a = pro_a, b = pro_b;
            if (is_q || is_k) {
                const float* nw = is_q ? g.qw : g.kw;
                if (nw) {
                    const float ss = wave_sum_rows_f32(a * a + b * b);
                    const float inv = rsqrtf(ss / (float)D + g.eps);
                    a = pro_wa * (a * inv); b = pro_wb * (b * inv);
                }
                const float ra = a * pro_c - b * pro_s, rb = b * pro_c + a * pro_s;
                a = ra; b = rb;
            }
            if (is_q) {
                const int qi = rr * REP + vec;
                const float qa = a * scale, qb = b * scale;
                sq_f[qi][lane] = qa; sq_f[qi][lane + half] = qb;
                const bf16_t ha = f32_to_bf16(qa), hb = f32_to_bf16(qb);
                sq_hi[qi][lane] = ha; sq_hi[qi][lane + half] = hb;
                sq_lo[qi][lane] = f32_to_bf16(qa - bf16_to_f32(ha)); sq_lo[qi][lane + half] = f32_to_bf16(qb - bf16_to_f32(hb));
            } else {
                bf16_t* o = (is_k ? g.kc : g.vc) + (((int64_t)slot * kv_heads + kh) * g.max_pos + hi) * D;
                const bf16_t ra = f32_to_bf16(a), rb = f32_to_bf16(b);
                o[lane] = ra; o[lane + half] = rb;
                s_kv[rr][is_k ? 0 : 1][lane] = ra; s_kv[rr][is_k ? 0 : 1][lane + half] = rb;
            }
        } else if (rr < ROWS && vec == 0 && lane == 0) {
            s_hi[rr] = -1;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- B: shared prefix, one 32-key block per wave and turn
    float m_run = -1e30f, l_run = 0.f;                       // per query vector n16 (replicated over the four lane groups)
    const int hi_q = n16 < QR ? s_hi[n16 / REP] : 0x7fffffff; // causal bound of this lane's query vector (rows normally sit past the prefix)
    f4acc_t oacc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) oacc[dt] = f4acc_t{0.f, 0.f, 0.f, 0.f};
    for (int blk = w; blk < n_blk; blk += NW) {
        const int kbase = blk * 32;
        // A fragments of the two K tiles (MFMA row rho = n16 is key kbase + 8 (rho >> 2) + (rho & 3), + 4 for tile 1) and B
        // fragments of V (lane (gq, n16): keys kbase + 8 gq .. + 8 of column 16 dt + n16): 16 coalesced 1-KiB loads; the first
        // block's are already in flight
        if (blk != w) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dk = 0; dk < 4; ++dk) kf[t][dk] = ktp[((int64_t)blk * 8 + t * 4 + dk) * 64];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) vf[dt] = vtp[((int64_t)blk * 8 + dt) * 64];
        }
        f4acc_t st[2] = {f4acc_t{0.f, 0.f, 0.f, 0.f}, f4acc_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            const s8_t qh = *reinterpret_cast<const s8_t*>(&sq_hi[n16][dk * 32 + gq * 8]);
            const s8_t ql = *reinterpret_cast<const s8_t*>(&sq_lo[n16][dk * 32 + gq * 8]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                st[t] = mfma16(kf[t][dk], qh, st[t]);
                st[t] = mfma16(kf[t][dk], ql, st[t]);
            }
        }
        // lane (gq, n16): scores of query vector n16 against keys kbase + 8 gq + i (tile 0) and + 4 + i (tile 1)
        float sc[8];
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = kbase + 8 * gq + i + 4 * t;
                sc[t * 4 + i] = (key < Lp && key <= hi_q) ? st[t][i] : -1e30f;
                mx = fmaxf(mx, sc[t * 4 + i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float corr = m_new > -1e29f ? __expf(m_run - m_new) : 1.f;       // (nothing visible yet: keep the empty state)
        float ps = 0.f;
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { pe[e] = sc[e] > -1e29f ? __expf(sc[e] - m_new) : 0.f; ps += pe[e]; }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * corr + ps;
        m_run = m_new;
        // P as the A operand of the second product: element e of lane (gq, n16) is key 8 gq + e of the block - already in place
        s8_t pa;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const unsigned pk = f32x2_to_bf16x2(pe[e], pe[e + 1]);
            pa[e] = (short)(pk & 0xffffu); pa[e + 1] = (short)(pk >> 16);
        }
        // rescale the running output (rows 4 gq + i of the accumulator <- corr of query vector 4 gq + i) and accumulate
        float cr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cr[i] = __shfl(corr, 4 * gq + i, 64);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) oacc[dt][i] *= cr[i];
            oacc[dt] = mfma16(pa, vf[dt], oacc[dt]);
        }
    }
    if (lane < 16 && n16 < QR) { p_m[w][n16] = m_run; p_l[w][n16] = l_run; }
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qi = 4 * gq + i;
            if (qi < QR) p_acc[w][qi][dt * 16 + n16] = oacc[dt][i];
        }

    // ---------------------------------------------------------------- C: the rows' own positions [Lp, hi] on the vector unit
    {
        const int rr = w / WPR, sw = w % WPR;                // WPR waves per local row
        const int hi = s_hi[rr];
        const int sub = lane % LPP, pg = lane / LPP;
        float m[REP], l[REP], acc[REP][8];
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            m[r] = -1e30f; l[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
        }
        if (hi >= 0) {
            const int slot = g.row_slot[r0 + rr];
            const int64_t base = ((int64_t)slot * kv_heads + kh) * g.max_pos;
            const bf16_t* kb = g.kc + base * D + sub * 8;
            const bf16_t* vb = g.vc + base * D + sub * 8;
            float qr[REP][8];
#pragma unroll
            for (int r = 0; r < REP; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) qr[r][j] = sq_f[rr * REP + r][sub * 8 + j];
            constexpr int U = 4;                             // positions in flight per lane: WPR waves x 4 positions x 4 per batch
            for (int p0 = (Lp <= hi + 1 ? Lp : hi + 1) + sw * PPW + pg; p0 <= hi; p0 += WPR * PPW * U) {
                u4_t kk[U], vv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * WPR * PPW;
                    if (PREFILL) {                           // every own row sits in the cache (k_qkv_post ran first); slots past the row re-read it
                        const int pc = p <= hi ? p : hi;
                        kk[u] = *reinterpret_cast<const u4_t*>(kb + (int64_t)pc * D);
                        vv[u] = *reinterpret_cast<const u4_t*>(vb + (int64_t)pc * D);
                    } else if (p >= hi) {                    // the row this launch appends (and clamped slots past it): from LDS, not through global memory
                        kk[u] = *reinterpret_cast<const u4_t*>(&s_kv[PREFILL ? 0 : rr][0][sub * 8]);
                        vv[u] = *reinterpret_cast<const u4_t*>(&s_kv[PREFILL ? 0 : rr][1][sub * 8]);
                    } else {
                        kk[u] = *reinterpret_cast<const u4_t*>(kb + (int64_t)p * D);
                        vv[u] = *reinterpret_cast<const u4_t*>(vb + (int64_t)p * D);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = p0 + u * WPR * PPW;
                    float kf2[8], vf2[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        kf2[2 * j] = __uint_as_float(kk[u][j] << 16); kf2[2 * j + 1] = __uint_as_float(kk[u][j] & 0xffff0000u);
                        vf2[2 * j] = __uint_as_float(vv[u][j] << 16); vf2[2 * j + 1] = __uint_as_float(vv[u][j] & 0xffff0000u);
                    }
#pragma unroll
                    for (int r = 0; r < REP; ++r) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) s += qr[r][j] * kf2[j];
                        s = group_sum_f32<LPP>(s);  // (DPP: no LDS round trips - common.h)
                        if (p <= hi) {                       // uniform within the 16-lane group
                            const float mn = fmaxf(m[r], s);
                            const float c1 = __expf(m[r] - mn), pe = __expf(s - mn);
                            l[r] = l[r] * c1 + pe;
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[r][j] = acc[r][j] * c1 + pe * vf2[j];
                            m[r] = mn;
                        }
                    }
                }
            }
        }
        // merge the wave's PPW position groups (lanes sharing `sub`); divergent trip counts above are over: all lanes take part
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
                s_part[w][r][sub][0] = m[r];
                s_part[w][r][sub][1] = l[r];
#pragma unroll
                for (int j = 0; j < 8; ++j) s_part[w][r][sub][2 + j] = acc[r][j];
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- D: merge prefix and suffix partials, normalise, store
    for (int t = tid; t < QR * D; t += NW * 64) {
        const int qi = t / D, d = t % D;
        const int rr = qi / REP, r = qi % REP, row = r0 + rr;
        if (row >= g.M) continue;
        const int sb = d >> 3, j = d & 7;
        float mn = -1e30f;
        const int nb_w = n_blk < NW ? n_blk : NW;             // waves that saw a prefix block
        for (int ww = 0; ww < nb_w; ++ww) mn = fmaxf(mn, p_m[ww][qi]);
#pragma unroll
        for (int sw = 0; sw < WPR; ++sw) mn = fmaxf(mn, s_part[rr * WPR + sw][r][sb][0]);
        float lt = 0.f, at = 0.f;
        for (int ww = 0; ww < nb_w; ++ww) {
            const float c = __expf(p_m[ww][qi] - mn);
            lt += p_l[ww][qi] * c;
            at += p_acc[ww][qi][d] * c;
        }
#pragma unroll
        for (int sw = 0; sw < WPR; ++sw) {
            const float c = __expf(s_part[rr * WPR + sw][r][sb][0] - mn);
            lt += s_part[rr * WPR + sw][r][sb][1] * c;
            at += s_part[rr * WPR + sw][r][sb][2 + j] * c;
        }
        const int kcol = (kh * REP + r) * D + d;
        const int64_t oo = g.out_tiled ? tile_off(row, kcol, heads * D) : (int64_t)row * heads * D + kcol;
        g.out[oo] = f32_to_bf16(lt > 0.f ? at / lt : 0.f);
    }
}

// K / V rows [pos][D] of one (layer, kv head) of the prefix slot -> the fragment order the kernel above loads, zero padded to
// whole 32-key blocks: per block 8 tiles of 64 lanes x 8 elements.
//   K tile (t, dk), lane (gq, rho): K[32 blk + 8 (rho >> 2) + (rho & 3) + 4 t][32 dk + 8 gq + e]
//   V tile dt,      lane (gq, n)  : V[32 blk + 8 gq + e][16 dt + n]
__global__ void k_tile_prefix_kv(const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc, int kv_heads, int max_pos, int slot, int prefix_len,
                                 bf16_t* __restrict__ kt, bf16_t* __restrict__ vt, int stride) {
    const int kh = blockIdx.x;
    const bf16_t* ks = kc + ((int64_t)slot * kv_heads + kh) * max_pos * D;
    const bf16_t* vs = vc + ((int64_t)slot * kv_heads + kh) * max_pos * D;
    bf16_t* kd = kt + (int64_t)kh * stride;
    bf16_t* vd = vt + (int64_t)kh * stride;
    const int n_blk = (prefix_len + 31) / 32;
    const int last = prefix_len - 1;
    // one thread per lane fragment (8 elements, one 16-B store each for K and V): the K fragment is 16 contiguous bytes of a cache
    // row, the V fragment one column of 8 consecutive rows; rows past the prefix read a clamped row and are zeroed (no branches
    // around the loads: all nine requests of a thread are in flight together)
    for (int g = threadIdx.x + blockIdx.y * blockDim.x; g < n_blk * 512; g += blockDim.x * gridDim.y) {
        const int ln = g & 63, tile = (g >> 6) & 7, blk = g >> 9;
        const int gq = ln >> 4, n = ln & 15;
        const int kkey = 32 * blk + 8 * (n >> 2) + (n & 3) + 4 * (tile >> 2);
        u4_t kf = *reinterpret_cast<const u4_t*>(ks + (int64_t)(kkey < prefix_len ? kkey : last) * D + 32 * (tile & 3) + 8 * gq);
        unsigned short vf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int vkey = 32 * blk + 8 * gq + e;
            vf[e] = vs[(int64_t)(vkey < prefix_len ? vkey : last) * D + 16 * tile + n];
        }
        if (kkey >= prefix_len) kf = u4_t{0u, 0u, 0u, 0u};
        u4_t vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int vkey = 32 * blk + 8 * gq + 2 * e;
            const unsigned lo16 = vkey < prefix_len ? vf[2 * e] : 0u, hi16 = vkey + 1 < prefix_len ? vf[2 * e + 1] : 0u;
            vv[e] = lo16 | (hi16 << 16);
        }
        *reinterpret_cast<u4_t*>(kd + (int64_t)g * 8) = kf;
        *reinterpret_cast<u4_t*>(vd + (int64_t)g * 8) = vv;
    }
}

// one thread per 8-element fragment: 512 fragments per 32-key block and kv head
inline unsigned tile_grid_y(int prefix_len) { return (unsigned)std::max(1, ((prefix_len + 31) / 32 * 512 + 255) / 256); }

}  // namespace

bool attention_mfma_ok(int M, int heads, int kv_heads, int head_dim, int window, const KvCache& kv) {
    return g_attn_mfma && head_dim == D && kv.head_dim == D && heads == REP * kv_heads && window <= 0 && kv.prefix_slot >= 0 && kv.vt_prefix && kv.prefix_len >= 64 &&
           kv.kt_prefix && (kv.prefix_len + 31) / 32 * 4096 <= kv.vt_stride && M >= 1;
}

int launch_attention_prefix_mfma(rt_ctx* ctx, const float* qkv, int M, int heads, int kv_heads, const float* q_norm_w, const float* k_norm_w, float eps,
                                 const float* rope_cos, const float* rope_sin, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                                 const KvCache& kv, int layer, bf16_t* out, const int32_t* frame_ptr, int out_tiled) {
    MfmaAttnArgs g;
    g.qkv = qkv; g.M = M; g.heads = heads; g.kv_heads = kv_heads; g.row_slot = row_slot; g.row_pos = row_pos; g.pos_add = pos_add;
    g.kc = kv.k + layer * kv.layer_stride(); g.vc = kv.v + layer * kv.layer_stride(); g.max_pos = kv.max_pos;
    g.kt = kv.kt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt = kv.vt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt_stride = kv.vt_stride;
    g.out = out; g.qw = q_norm_w; g.kw = k_norm_w; g.eps = eps; g.cosT = rope_cos; g.sinT = rope_sin; g.frame_ptr = frame_ptr; g.out_tiled = out_tiled;
    g.prefix_slot = kv.prefix_slot; g.prefix_len = kv.prefix_len;
    // g_attn_mfma 1: four rows per workgroup (the prefix tiles read once per four rows, 64 workgroups at batch 32: round 2's form);
    // 2: ONE row per workgroup - the grid of the vector-unit kernel (a workgroup per (row, kv head), every CU busy), each of the 16
    // waves multiplies one 32-key prefix block on the matrix cores and all 16 share the row's own positions.  The prefix tiles are
    // re-read per row from L2 like the cache rows of the vector kernel, which the counters show is not what bounds it
    // (profiles/r04_pmc_attention.csv: 33 MB of L2 -> L1 reads in a 12.9-us launch); what the matrix cores remove is the vector unit's
    // 2.8 us of dot products and rescales over the 460 prefix keys.
    if (g_attn_mfma == 2) hipLaunchKernelGGL((k_attn_prefix_mfma<1, 16, false>), dim3(kv_heads, M), dim3(16 * 64), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_attn_prefix_mfma<4, 16, false>), dim3(kv_heads, (M + 3) / 4), dim3(16 * 64), 0, ctx->stream, g);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// prompt rows behind a shared prefix (see the kernel's header): the same preconditions as the decode form but q comes prepared
rt_knob g_prefill_attn_mfma{1};
bool attention_prefill_mfma_ok(int heads, int kv_heads, int head_dim, int window, const KvCache& kv) {
    return g_prefill_attn_mfma && head_dim == D && kv.head_dim == D && heads == REP * kv_heads && window <= 0 && !kv.k_lo && kv.prefix_slot >= 0 &&
           kv.vt_prefix && kv.kt_prefix && kv.prefix_len >= 64 && kv.tiles_len == kv.prefix_len && (kv.prefix_len + 31) / 32 * 4096 <= kv.vt_stride;
}
int launch_attention_prefill_mfma(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, const int32_t* row_slot, const int32_t* row_pos, int pos_add,
                                  const KvCache& kv, int layer, bf16_t* out) {
    MfmaAttnArgs g;
    g.qkv = q; g.M = M; g.heads = heads; g.kv_heads = kv_heads; g.row_slot = row_slot; g.row_pos = row_pos; g.pos_add = pos_add;
    g.kc = kv.k + layer * kv.layer_stride(); g.vc = kv.v + layer * kv.layer_stride(); g.max_pos = kv.max_pos;
    g.kt = kv.kt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt = kv.vt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt_stride = kv.vt_stride;
    g.out = out; g.qw = nullptr; g.kw = nullptr; g.eps = 0.f; g.cosT = nullptr; g.sinT = nullptr; g.frame_ptr = nullptr; g.out_tiled = 0;
    g.prefix_slot = kv.prefix_slot; g.prefix_len = kv.prefix_len;
    hipLaunchKernelGGL((k_attn_prefix_mfma<8, 8, true>), dim3(kv_heads, (M + 7) / 8), dim3(8 * 64), 0, ctx->stream, g);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// The voice prefix's OWN prefill (rt_model_set_voice: n consecutive rows of the prefix slot, causal, no shared prefix yet): layer
// by layer, right after k_qkv_post has written the layer's K / V rows, their fragment-tiled copies are made - the copies every later
// prompt prefill reads anyway - and the attention of the layer runs on them: for a workgroup's 8 rows the keys in front of its
// first row go through the matrix cores, the <= 8 keys from there on through the vector part.
bool attention_block_prefix_ok(int M, int heads, int kv_heads, int head_dim, int window, const KvCache& kv) {
    return g_prefill_attn_mfma && head_dim == D && kv.head_dim == D && heads == REP * kv_heads && window <= 0 && !kv.k_lo && kv.vt_prefix && kv.kt_prefix &&
           kv.prefix_slot_alloc >= 0 && M >= 64 && (M + 31) / 32 * 4096 <= kv.vt_stride;
}
int launch_attention_block_prefix(rt_ctx* ctx, const float* q, int M, int heads, int kv_heads, const int32_t* row_slot, const int32_t* row_pos,
                                  KvCache& kv, int layer, bf16_t* out) {
    hipLaunchKernelGGL(k_tile_prefix_kv, dim3(kv.kv_heads, tile_grid_y(M)), dim3(256), 0, ctx->stream, kv.k + layer * kv.layer_stride(), kv.v + layer * kv.layer_stride(),
                       kv.kv_heads, kv.max_pos, kv.prefix_slot_alloc, M, kv.kt_prefix + (int64_t)layer * kv.kv_heads * kv.vt_stride,
                       kv.vt_prefix + (int64_t)layer * kv.kv_heads * kv.vt_stride, kv.vt_stride);
    RT_HIP(ctx, hipGetLastError());
    MfmaAttnArgs g;
    g.qkv = q; g.M = M; g.heads = heads; g.kv_heads = kv_heads; g.row_slot = row_slot; g.row_pos = row_pos; g.pos_add = 0;
    g.kc = kv.k + layer * kv.layer_stride(); g.vc = kv.v + layer * kv.layer_stride(); g.max_pos = kv.max_pos;
    g.kt = kv.kt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt = kv.vt_prefix + (int64_t)layer * kv_heads * kv.vt_stride; g.vt_stride = kv.vt_stride;
    g.out = out; g.qw = nullptr; g.kw = nullptr; g.eps = 0.f; g.cosT = nullptr; g.sinT = nullptr; g.frame_ptr = nullptr; g.out_tiled = 0;
    g.prefix_slot = -1; g.prefix_len = 0; g.block_prefix = 1;
    hipLaunchKernelGGL((k_attn_prefix_mfma<8, 8, true>), dim3(kv_heads, (M + 7) / 8), dim3(8 * 64), 0, ctx->stream, g);
    RT_HIP(ctx, hipGetLastError());
    if (layer + 1 == kv.layers) kv.tiles_len = M;            // every layer's tiles now hold the prefix
    return RT_OK;
}

int launch_transpose_prefix_v(rt_ctx* ctx, KvCache& kv, int prefix_len) {
    if (!kv.vt_prefix || !kv.kt_prefix || kv.head_dim != D) return RT_OK;       // (the matrix-core path exists for head_dim 128 only)
    if ((prefix_len + 31) / 32 * 4096 > kv.vt_stride) return rt_fail(ctx, RT_ERR_LENGTH, "prefix of %d rows exceeds the tiled prefix buffers", prefix_len);
    for (int layer = 0; layer < kv.layers; ++layer)
        hipLaunchKernelGGL(k_tile_prefix_kv, dim3(kv.kv_heads, tile_grid_y(prefix_len)), dim3(256), 0, ctx->stream, kv.k + layer * kv.layer_stride(), kv.v + layer * kv.layer_stride(),
                           kv.kv_heads, kv.max_pos, kv.prefix_slot_alloc, prefix_len, kv.kt_prefix + (int64_t)layer * kv.kv_heads * kv.vt_stride,
                           kv.vt_prefix + (int64_t)layer * kv.kv_heads * kv.vt_stride, kv.vt_stride);
    RT_HIP(ctx, hipGetLastError());
    kv.tiles_len = prefix_len;
    return RT_OK;
}
