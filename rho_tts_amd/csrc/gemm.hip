// bf16 MFMA GEMMs for gfx950 (v_mfma_f32_32x32x16_bf16, 64-lane waves).
//
//   k_pack_weight   W[N][K] row-major -> B-fragment tile order (one coalesced 1-KiB load per operand)
//   k_gemm_skinny   decode form, M <= 64 rows: every weight byte streamed from HBM exactly once
//                   (non-temporal), activations from L2, split-K partial slabs (deterministic sum
//                   in the consumer instead of float atomics).  HBM-bound: N*K*2 bytes per launch.
//   k_gemm_tiled    128x128x32 LDS-staged implicit GEMM for prefill and the codec decoder's
//                   conv-as-GEMM contractions (causal dilated conv1d, transposed conv1d), fused
//                   bias / activation / layer-scale / residual / SnakeBeta epilogue.  MFMA-bound.
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

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return f32x2_to_bf16x2(lo, hi);
}

// ------------------------------------------------------------------------------------------ pack
__global__ void k_pack_weight(const bf16_t* __restrict__ src, int N, int K, int Np, int Kp, bf16_t* __restrict__ dst) {
    const int KT = Kp / 16;
    const int64_t n_pieces = (int64_t)(Np / 32) * KT * 64;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pieces; p += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(p & 63);
        const int64_t tile = p >> 6;
        const int kt = (int)(tile % KT);
        const int nt = (int)(tile / KT);
        const int n = nt * 32 + (lane & 31);
        const int k0 = kt * 16 + (lane >> 5) * 8;
        bf16_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (n < N && k0 + j < K) ? src[(int64_t)n * K + k0 + j] : (bf16_t)0;
        *reinterpret_cast<s8_t*>(dst + p * 8) = *reinterpret_cast<s8_t*>(v);
    }
}

__global__ void k_pack_weight16(const bf16_t* __restrict__ src, int N, int K, int Np, bf16_t* __restrict__ dst) {
    const int KT = K / 32;
    const int64_t n_pieces = (int64_t)(Np / 16) * KT * 64;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pieces; p += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(p & 63);
        const int64_t tile = p >> 6;
        const int kt = (int)(tile % KT);
        const int nt = (int)(tile / KT);
        const int n = nt * 16 + (lane & 15);
        const int k0 = kt * 32 + (lane >> 4) * 8;
        bf16_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = n < N ? src[(int64_t)n * K + k0 + j] : (bf16_t)0;
        *reinterpret_cast<s8_t*>(dst + p * 8) = *reinterpret_cast<s8_t*>(v);
    }
}

// ---------------------------------------------------------------------------------------- skinny
// grid.x = Np/32 n-tiles (one wave each, 4 waves per workgroup), grid.y = split_k.
template <int MT>
__global__ __launch_bounds__(256) void k_gemm_skinny(const bf16_t* __restrict__ A, int M, int K, const bf16_t* __restrict__ Wp,
                                                     int NT, int KT, int kt_per_split, float* __restrict__ out, int64_t ldc,
                                                     int N, int64_t slab_stride) {
    const int lane = threadIdx.x & 63;
    const int nt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (nt >= NT) return;
    const int r = lane & 31, h = lane >> 5;
    const int kt0 = blockIdx.y * kt_per_split;
    int kt1 = kt0 + kt_per_split;
    if (kt1 > KT) kt1 = KT;
    const s8_t* wp = reinterpret_cast<const s8_t*>(Wp) + ((int64_t)nt * KT + kt0) * 64 + lane;
    const bf16_t* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 32 + r;
        if (row >= M) row = M - 1;  // clamp: rows >= M are computed on valid memory and never stored
        ap[mt] = A + (int64_t)row * K + h * 8;
    }
    f16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    const int n_k = kt1 - kt0;
    int kt = 0;
    constexpr int U = 8;
    for (; kt + U <= n_k; kt += U) {
        s8_t b[U];
        s8_t a[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = __builtin_nontemporal_load(wp + (int64_t)(kt + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = *reinterpret_cast<const s8_t*>(ap[mt] + (int64_t)(kt0 + kt + u) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(a[u][mt], b[u], acc[mt]);
    }
    for (; kt < n_k; ++kt) {
        const s8_t b = __builtin_nontemporal_load(wp + (int64_t)kt * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const s8_t a = *reinterpret_cast<const s8_t*>(ap[mt] + (int64_t)(kt0 + kt) * 16);
            acc[mt] = mfma32(a, b, acc[mt]);
        }
    }
    float* o = out + (int64_t)blockIdx.y * slab_stride;
    const int n = nt * 32 + r;
    if (n < N) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M) o[(int64_t)row * ldc + n] = acc[mt][i];
            }
    }
}

// Variant 2 of the decode GEMM: the workgroup's four waves share one K-slice, so the activation slab
// A[M][k0:k1] is staged ONCE per workgroup into LDS with fully coalesced 16-B loads (instead of four waves each
// issuing row-strided fragment loads), and the weight stream is software-pipelined: the 1-KiB B loads of step i+1
// are in flight while step i's MFMAs run.  LDS rows are padded by 16 B so the 32-row x 16-B fragment reads of a
// ds_read_b128 spread over all banks.
template <int MT, int U>
__global__ __launch_bounds__(256) void k_gemm_skinny2(const bf16_t* __restrict__ A, int M, int K, const bf16_t* __restrict__ Wp,
                                                      int NT, int KT, int kt_per_split, float* __restrict__ out, int64_t ldc,
                                                      int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_a[];   // [MT*32][ks*2 + 16] bytes
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nt = blockIdx.x * 4 + w;
    const int r = lane & 31, h = lane >> 5;
    const int kt0 = blockIdx.y * kt_per_split;
    int kt1 = kt0 + kt_per_split;
    if (kt1 > KT) kt1 = KT;
    const int n_k = kt1 - kt0;
    const int ks = n_k * 16;                    // k elements of this slice
    const int row_bytes = ks * 2 + 16;
    // ---- stage A: rows x ks bf16, 16-B pieces, coalesced along k
    {
        const int pieces_per_row = ks / 8;
        const int total = MT * 32 * pieces_per_row;
        for (int p = threadIdx.x; p < total; p += 256) {
            const int row = p / pieces_per_row, pc = p - row * pieces_per_row;
            const int src_row = row < M ? row : M - 1;
            const s8_t v = *reinterpret_cast<const s8_t*>(A + (int64_t)src_row * K + kt0 * 16 + pc * 8);
            *reinterpret_cast<s8_t*>(lds_a + row * row_bytes + pc * 16) = v;
        }
    }
    __syncthreads();
    if (nt >= NT) return;
    const s8_t* wp = reinterpret_cast<const s8_t*>(Wp) + ((int64_t)nt * KT + kt0) * 64 + lane;
    f16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    const unsigned char* abase = lds_a + r * row_bytes + h * 16;

    s8_t b_cur[U], b_nxt[U];
    int kt = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) b_cur[u] = (u < n_k) ? __builtin_nontemporal_load(wp + (int64_t)u * 64) : s8_t{0, 0, 0, 0, 0, 0, 0, 0};
    for (; kt < n_k; kt += U) {
        const bool more = kt + U < n_k;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                b_nxt[u] = (kt + U + u < n_k) ? __builtin_nontemporal_load(wp + (int64_t)(kt + U + u) * 64) : s8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt + u < n_k) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const s8_t a = *reinterpret_cast<const s8_t*>(abase + (mt * 32) * row_bytes + (kt + u) * 32);
                    acc[mt] = mfma32(a, b_cur[u], acc[mt]);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) b_cur[u] = b_nxt[u];
        }
    }
    float* o = out + (int64_t)blockIdx.y * M * ldc;
    const int n = nt * 32 + r;
    if (n < N) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M) o[(int64_t)row * ldc + n] = acc[mt][i];
            }
    }
}

// ----------------------------------------------------------------------------------------- tiled
constexpr int BM = 128, BN = 128, BK = 32;

struct TiledArgs {
    GemmA a;
    const bf16_t* Wp;
    int N, K, NT, KT;
    GemmEpi e;
    int kt_per_split;  // in 16-wide k tiles, multiple of 2
    int xcd_order = 0;   // 1: XCD-aware tile order (see k_gemm_tiled)
    // k_conv_win<..., FUSE = true>: the 1x1 conv behind this conv's activation runs in the same launch (see the kernel)
    const bf16_t* Wp2 = nullptr;
    int NT2 = 0, KT2 = 0, N2 = 0;
    GemmEpi e2;
};

__device__ __forceinline__ int lds_a_off(int row, int slot) {  // bytes; 64-B rows, 16-B slots XOR-swizzled by row/4
    return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4);
}

// Fused epilogue of one wave's MT x NTT accumulator tiles (shared by k_gemm_tiled and k_conv_win).
// Every load and store is a BUFFER operation: one resource descriptor per operand in scalar registers - base = the workgroup's
// tile corner, size = up to the last valid row - and a 32-bit per-lane byte offset.  Rows past M are simply out of range (loads
// return 0, stores are dropped by the hardware), so there is no row test, no branch and no 64-bit address per element: the
// per-element `if (row < M)` form compiled to an exec-mask branch per element, serialised the 16 residual loads of a tile behind a
// `s_waitcnt vmcnt(0)` each, and its address registers spilled the 128 x 128 instantiations into scratch.
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(T* p, int64_t tile_base, int64_t elems) {
    int64_t bytes = p ? elems * (int64_t)sizeof(T) : 0;          // (an absent operand: an empty buffer nobody touches)
    if (bytes < 0) bytes = 0;
    if (bytes > 0x40000000) bytes = 0x40000000;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p + tile_base), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld_f32(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void buf_st_f32(__amdgpu_buffer_rsrc_t r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0); }
__device__ __forceinline__ void buf_st_b16(__amdgpu_buffer_rsrc_t r, unsigned off, bf16_t v) { __builtin_amdgcn_raw_buffer_store_b16((short)v, r, off, 0, 0); }

template <int MT, int NTT>
__device__ __forceinline__ void tile_epilogue_e(const GemmEpi& e, int64_t M, int N, f16_t (&acc)[MT][NTT], int64_t m0, int n0, int row_blk0, int wn, int r, int h);
template <int MT, int NTT>
__device__ __forceinline__ void tile_epilogue(const TiledArgs& g, f16_t (&acc)[MT][NTT], int64_t m0, int n0, int wm, int wn, int r, int h) {
    tile_epilogue_e<MT, NTT>(g.e, g.a.M, g.N, acc, m0, n0, wm * MT, wn, r, h);
}
// row_blk0: index (in 32-row blocks from m0) of the wave's first accumulator tile
template <int MT, int NTT>
__device__ __forceinline__ void tile_epilogue_e(const GemmEpi& e, int64_t M, int N, f16_t (&acc)[MT][NTT], int64_t m0, int n0, int row_blk0, int wn, int r, int h) {
    // ---- epilogue.  Every option of GemmEpi is uniform over the launch, so each one is tested ONCE per 32x32 accumulator
    // tile with the 16-element loops inside (tested per element, the option branches and 64-bit index arithmetic made the
    // epilogue ~220 instructions per output - several times the cost of the K loop for the codec decoder's short-K convs).
    const int64_t ldc = e.ldc;
    const int64_t tb = m0 * ldc + n0;                                  // element offset of the workgroup's tile corner (uniform)
    const int64_t span = (M - m0) * ldc - n0;                          // elements from the corner to the end of the last valid row
    const unsigned ldc32 = (unsigned)ldc;
    const __amdgpu_buffer_rsrc_t r_res = tile_rsrc(e.residual, tb, span);
    const __amdgpu_buffer_rsrc_t r_f32 = tile_rsrc(e.out_f32, tb + (e.split_k > 1 ? (int64_t)blockIdx.z * M * ldc : 0), span);
    const __amdgpu_buffer_rsrc_t r_bf = tile_rsrc(e.out_bf16, tb, span);
    const __amdgpu_buffer_rsrc_t r_hi = tile_rsrc(e.out_hi, tb, span), r_lo = tile_rsrc(e.out_lo, tb, span);
    const __amdgpu_buffer_rsrc_t r2_hi = tile_rsrc(e.out2_hi, tb, span), r2_lo = tile_rsrc(e.out2_lo, tb, span);
    const __amdgpu_buffer_rsrc_t r2_bf = tile_rsrc(e.out2_bf16, tb, span), r2_f32 = tile_rsrc(e.out2_f32, tb, span);
    // the per-column constants of every option, for all of the wave's column tiles, requested together up front through stand-in
    // pointers (an absent vector reads the bias / a valid address and is ignored): fetched where they are used, each sat behind
    // its own uniform branch - up to four dependent round trips per 32 x 32 tile, drained one by one
    float c_bias[NTT], c_sa[NTT], c_sib[NTT], c_scale[NTT], c_s2a[NTT], c_s2ib[NTT];
    {
        const bool snake = e.act == ACT_SNAKE, two = (e.out2_hi || e.out2_bf16 || e.out2_f32) && e.act2 != ACT_ELU;
        const float* any = e.bias ? e.bias : (e.scale ? e.scale : (snake ? e.snake_a : (two ? e.snake2_a : nullptr)));
        if (any) {
            const float *p_b = e.bias ? e.bias : any, *p_sc = e.scale ? e.scale : any;
            const float *p_sa = snake ? e.snake_a : any, *p_sib = snake ? e.snake_ib : any;
            const float *p_2a = two ? e.snake2_a : any, *p_2ib = two ? e.snake2_ib : any;
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                int n = n0 + (wn * NTT + nt) * 32 + r;
                n = n < N ? n : N - 1;
                c_bias[nt] = p_b[n]; c_scale[nt] = p_sc[n]; c_sa[nt] = p_sa[n]; c_sib[nt] = p_sib[n]; c_s2a[nt] = p_2a[n]; c_s2ib[nt] = p_2ib[n];
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
            if (!e.bias) c_bias[nt] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // element i of the accumulator sits (i & 3) + 8 * (i >> 2) rows below element 0
#define RT_ROW(i) ((i & 3) + 8 * (i >> 2))
        // Byte offsets of the sub-tile's 16 rows from the tile corner, computed ONCE per 32-row sub-tile for 4-byte and 2-byte
        // elements; a column tile further right and every store kind (f32, bf16, the hi / lo planes, the second output) then differ by
        // a compile-time constant that the buffer instruction carries in its immediate offset.  Computed per store - as
        // (lo0 + RT_ROW(i) * ldc) * bytes for every column tile and every kind - the addressing was 27 vector instructions per 16
        // stores, ~7 of the ~25 per output element in kernels whose epilogue, not their K loop, sets the pace (round 4).
        unsigned ro4[16], ro2[16];
        {
            const unsigned base = (unsigned)(((row_blk0 + mt) * 32 + 4 * h) * ldc32 + wn * NTT * 32 + r);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned el = base + RT_ROW(i) * ldc32;
                ro4[i] = el * 4u;
                ro2[i] = el * 2u;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
            const int n = n0 + (wn * NTT + nt) * 32 + r;
            if (n >= N) continue;
#define RT_OFF(i, bytes) ((bytes) == 4 ? ro4[i] + (unsigned)(nt * 128) : ro2[i] + (unsigned)(nt * 64))
            if (e.split_k > 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) buf_st_f32(r_f32, RT_OFF(i, 4), acc[mt][nt][i]);
                continue;
            }
            // residual rows first, all 16 in flight: the residual usually aliases out_f32 (in-place update), so loads issued
            // between the stores would be serialised behind them
            float res[16];
            if (e.residual) {
#pragma unroll
                for (int i = 0; i < 16; ++i) res[i] = buf_ld_f32(r_res, RT_OFF(i, 4));
            }
            const float bias = c_bias[nt];
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[mt][nt][i] + bias;
            if (e.act == ACT_SILU) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = v[i] / (1.f + __expf(-v[i]));
            } else if (e.act == ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = 0.5f * v[i] * (1.f + erff(v[i] * 0.70710678118654752f));
            } else if (e.act == ACT_SNAKE) {
                const float sa = c_sa[nt], sib = c_sib[nt];
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float sn = __sinf(v[i] * sa); v[i] = v[i] + sib * sn * sn; }
            } else if (e.act == ACT_CLAMP1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fminf(1.f, fmaxf(-1.f, v[i]));
            } else if (e.act == ACT_ELU) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = v[i] > 0.f ? v[i] : expm1f(v[i]);
            }
            if (e.scale) {
                const float scale = c_scale[nt];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] *= scale;
            }
            if (e.residual) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += res[i];
            }
            if (e.out_f32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) buf_st_f32(r_f32, RT_OFF(i, 4), v[i]);
            }
            if (e.out_bf16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) buf_st_b16(r_bf, RT_OFF(i, 2), f32_to_bf16(v[i]));
            }
            if (e.out_hi) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bf16_t hi = f32_to_bf16(v[i]);
                    const bf16_t lo = f32_to_bf16(v[i] - bf16_to_f32(hi));
                    buf_st_b16(r_hi, RT_OFF(i, 2), hi);
                    buf_st_b16(r_lo, RT_OFF(i, 2), lo);
                }
            }
            if (e.out2_hi || e.out2_bf16 || e.out2_f32) {
                if (e.act2 == ACT_ELU) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = v[i] > 0.f ? v[i] : expm1f(v[i]);
                } else {
                    const float s2a = c_s2a[nt], s2ib = c_s2ib[nt];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float sn = __sinf(v[i] * s2a); v[i] = v[i] + s2ib * sn * sn; }
                }
                if (e.out2_hi) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bf16_t hi = f32_to_bf16(v[i]);
                        const bf16_t lo = f32_to_bf16(v[i] - bf16_to_f32(hi));
                        buf_st_b16(r2_hi, RT_OFF(i, 2), hi);
                        buf_st_b16(r2_lo, RT_OFF(i, 2), lo);
                    }
                }
                if (e.out2_bf16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) buf_st_b16(r2_bf, RT_OFF(i, 2), f32_to_bf16(v[i]));
                }
                if (e.out2_f32) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) buf_st_f32(r2_f32, RT_OFF(i, 4), v[i]);
                }
            }
#undef RT_OFF
        }
#undef RT_ROW
    }
}

// SPLIT: the f32 A operand is fed as two bf16 planes, hi = bf16(x) and lo = bf16(x - hi), and every B fragment
// is multiplied with both (2x MFMA work): products stay exact, activation precision goes from 2^-9 to ~2^-17.
// Used by the codec decoder, whose waveform has to agree with the f32-activation oracle to RMSE < 1e-3.
// Waves are arranged WGM x WGN, each computing MT x NTT 32x32 tiles: the workgroup tile is (WGM*MT*32) x (WGN*NTT*32).
// 2,2,2,2 = 128 x 128 (default); 4,1,1,3 = 128 x 96 for the codec decoder's 96- and 192-channel stages, where a 128-wide
// tile would spend a quarter of its MFMAs on padding columns.
template <bool A_F32, bool SPLIT, int WGM, int WGN, int MT, int NTT>
__global__ __launch_bounds__(256, 3) void k_gemm_tiled(TiledArgs g) {
    static_assert(WGM * WGN == 4 && WGM * MT * 32 == BM, "4 waves, 128 rows");
    constexpr int BNT = WGN * NTT * 32;
    constexpr int PLANES = SPLIT ? 2 : 1;
    constexpr int BUF = BM * 64 * PLANES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WGN, wn = w % WGN;
    const int r = lane & 31, h = lane >> 5;
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs in launch order, each XCD with its own 4-MiB L2.  With
    // g.xcd_order the launch index is re-read so that ONE XCD runs all the column tiles of a row tile back to back: the A
    // rows (7 taps x n column tiles of re-reads in the codec decoder) are then fetched into that L2 once instead of once per
    // column tile (the 768-channel stage missed L2 on half of 68 M requests per launch with the plain x-fastest order).
    int64_t rt = blockIdx.x;                       // x: row tiles (can exceed 65535), y: column tiles
    int ct = blockIdx.y;
    if (g.xcd_order) {
        const int64_t L = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;      // gridDim.x is a multiple of 8 here
        const int xcd = (int)(L & 7);
        const int64_t j = L >> 3;
        ct = (int)(j % gridDim.y);
        rt = (j / gridDim.y) * 8 + xcd;
        if (rt * BM >= g.a.M) return;              // padding tiles of the rounded-up row count
    }
    const int64_t m0 = rt * BM;
    const int n0 = ct * BNT;
    const int kt0 = blockIdx.z * g.kt_per_split;
    int kt1 = kt0 + g.kt_per_split;
    if (kt1 > g.KT) kt1 = g.KT;
    const int n_it = (kt1 - kt0 + 1) / 2;

    // ---- per-thread A staging state: pieces p = tid, tid+256 -> (row = p>>2, slot = p&3)
    int64_t row_base[2];   // element offset of (b, t=0... ) i.e. b*rows_in*Cin
    int row_t[2];          // t + tap_offset
    bool row_ok[2];
    int p_tap[2], p_ci[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = tid + i * 256;
        const int row = p >> 2, slot = p & 3;
        const int64_t m = m0 + row;
        row_ok[i] = m < g.a.M;
        int64_t b = 0, t = m;
        if (g.a.rows_out > 0) { b = m / g.a.rows_out; t = m - b * g.a.rows_out; }
        const int rows_in = g.a.rows_out > 0 ? g.a.rows_in : 0;
        row_base[i] = b * (int64_t)rows_in * g.a.Cin;
        row_t[i] = (int)t + g.a.tap_offset;
        const int kk = kt0 * 16 + slot * 8;
        p_tap[i] = kk / g.a.Cin;
        p_ci[i] = kk - p_tap[i] * g.a.Cin;
    }
    const int t_limit = g.a.rows_out > 0 ? g.a.rows_in : 0x7fffffff;

    auto load_piece = [&](int i, s8_t& dst, s8_t& dst_lo) {
        const int ti = row_t[i] + p_tap[i] * g.a.tap_stride;
        const bool ok = row_ok[i] && p_tap[i] < g.a.taps && ti >= 0 && (g.a.rows_out == 0 || ti < t_limit);
        if (ok) {
            const int64_t off = row_base[i] + (int64_t)ti * g.a.Cin + p_ci[i];
            if (A_F32) {
                const float* s = reinterpret_cast<const float*>(g.a.ptr) + off;
                const f4_t v0 = *reinterpret_cast<const f4_t*>(s), v1 = *reinterpret_cast<const f4_t*>(s + 4);
                i4_t pk;
                pk[0] = (int)pack_bf16x2(v0[0], v0[1]);
                pk[1] = (int)pack_bf16x2(v0[2], v0[3]);
                pk[2] = (int)pack_bf16x2(v1[0], v1[1]);
                pk[3] = (int)pack_bf16x2(v1[2], v1[3]);
                dst = __builtin_bit_cast(s8_t, pk);
                if (SPLIT) {
                    float rs[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        rs[j] = v0[j] - __uint_as_float(((unsigned)(unsigned short)dst[j]) << 16);
                        rs[4 + j] = v1[j] - __uint_as_float(((unsigned)(unsigned short)dst[4 + j]) << 16);
                    }
                    i4_t pl;
                    pl[0] = (int)pack_bf16x2(rs[0], rs[1]);
                    pl[1] = (int)pack_bf16x2(rs[2], rs[3]);
                    pl[2] = (int)pack_bf16x2(rs[4], rs[5]);
                    pl[3] = (int)pack_bf16x2(rs[6], rs[7]);
                    dst_lo = __builtin_bit_cast(s8_t, pl);
                }
            } else {
                dst = *reinterpret_cast<const s8_t*>(reinterpret_cast<const bf16_t*>(g.a.ptr) + off);
                if (SPLIT) dst_lo = *reinterpret_cast<const s8_t*>(reinterpret_cast<const bf16_t*>(g.a.ptr_lo) + off);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { dst[j] = 0; if (SPLIT) dst_lo[j] = 0; }
        }
        // advance this piece by BK along K for the next iteration
        p_ci[i] += BK;
        while (p_ci[i] >= g.a.Cin) { p_ci[i] -= g.a.Cin; ++p_tap[i]; }
    };
    auto store_piece = [&](int i, int buf, const s8_t& v, const s8_t& v_lo) {
        const int p = tid + i * 256;
        *reinterpret_cast<s8_t*>(lds + buf * BUF + lds_a_off(p >> 2, p & 3)) = v;
        if (SPLIT) *reinterpret_cast<s8_t*>(lds + buf * BUF + BM * 64 + lds_a_off(p >> 2, p & 3)) = v_lo;
    };
    // ---- B fragments straight from the packed weights (global -> VGPR), one 16-B load per (nt, kt)
    const int nt_base = (n0 >> 5) + wn * NTT;
    auto load_b = [&](int it, s8_t (&b)[NTT][2]) {
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int t_n = nt_base + nt, t_k = kt0 + it * 2 + kk;
                if (t_n < g.NT && t_k < kt1)
                    b[nt][kk] = *(reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)t_n * g.KT + t_k) * 64 + lane);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[nt][kk][j] = 0;
                }
            }
    };

    f16_t acc[MT][NTT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    s8_t ra[2], rl[2], rb[NTT][2], rb_next[NTT][2];
    load_piece(0, ra[0], rl[0]);
    load_piece(1, ra[1], rl[1]);
    load_b(0, rb);
    store_piece(0, 0, ra[0], rl[0]);
    store_piece(1, 0, ra[1], rl[1]);
    __syncthreads();

    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < n_it;
        if (more) {
            load_piece(0, ra[0], rl[0]);
            load_piece(1, ra[1], rl[1]);
            load_b(it + 1, rb_next);
        }
        s8_t fa[MT][2], fl[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fa[mt][kk] = *reinterpret_cast<const s8_t*>(lds + buf * BUF + lds_a_off((wm * MT + mt) * 32 + r, kk * 2 + h));
                if (SPLIT) fl[mt][kk] = *reinterpret_cast<const s8_t*>(lds + buf * BUF + BM * 64 + lds_a_off((wm * MT + mt) * 32 + r, kk * 2 + h));
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt) {
                    acc[mt][nt] = mfma32(fa[mt][kk], rb[nt][kk], acc[mt][nt]);
                    if (SPLIT) acc[mt][nt] = mfma32(fl[mt][kk], rb[nt][kk], acc[mt][nt]);
                }
        if (more) {
            store_piece(0, buf ^ 1, ra[0], rl[0]);
            store_piece(1, buf ^ 1, ra[1], rl[1]);
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) rb[nt][kk] = rb_next[nt][kk];
        }
        __syncthreads();
    }

    tile_epilogue<MT, NTT>(g, acc, m0, n0, wm, wn, r, h);
}

// ----------------------------------------------------------------------------------------- mid-M (prompt prefill)
// out[M][N] (f32) = A[M][K] (bf16, row-major) W^T for a few hundred rows (the voice-prefix and suffix prefills: 400-500 rows).
// The 128 x 128 kernel above only fills the chip for such M by splitting K eight ways, and then moves more float32 partial
// slabs (S x M x N x 4 B written, then re-read by the consumer) than weights.  Here a workgroup owns a 64 x 64 tile for the
// WHOLE K - M = 460, N = 2048 already gives 256 workgroups - and writes final sums.  Both operands go through LDS (the packed
// weight fragments verbatim, so no wave re-reads another's through the vector L1), 64 deep per stage, two stages of loads in
// flight in registers behind the one being multiplied.  Per-CU bound: (64 + 64) x K x 2 B through a 64 B/clk L1.
constexpr int MID_BK = 64;
struct MidArgs {
    const bf16_t* A;
    const bf16_t* Wp;
    float* out;
    int M, N, K, NT, KT;
    int64_t ldc;
    int seg_it;        // > 0: every seg_it iterations the accumulator is added to a running total and cleared (see below)
};
// WM x WN 32 x 32 MFMA tiles per wave, 2 x 2 waves: the workgroup tile is (64 WM) x (64 WN).  Cache-to-CU traffic is
// M N K 2 B x (1 / BM + 1 / BN): 64 x 64 tiles ran every prefill GEMM of the 1.7B talker at 12-14 TB/s of L2 reads (the chip's
// practical limit for re-read lines) and no faster than the split-K kernel; 128 x 128 halves that traffic.
// seg_it: a prompt row must get the SAME float32 sums whether it is prefilled among 460 rows here or among 13 rows by the
// skinny kernel (continuous batching hands rows over a few at a time; a text's audio may not depend on its batch).  The skinny
// kernel accumulates K in `split` contiguous segments and its consumers add the segment sums in order; with seg_it this
// kernel adds in exactly that association: ascending k inside a segment, then total = (...((0 + s0) + s1) + ...).
template <int WM, int WN, int PF>
__global__ __launch_bounds__(256, 2) void k_gemm_mid(MidArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_BYTES = BM * MID_BK * 2;                     // BM rows x 128 B, 16-B slots XOR-swizzled by row
    constexpr int B_BYTES = (BN / 32) * (MID_BK / 16) * 1024;    // packed 1-KiB weight fragments, verbatim
    constexpr int NA = BM / 32, NB = BN / 32;                    // 16-B pieces per thread and stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (A_BYTES + B_BYTES)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 31, h = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs in launch order, each XCD with its own L2: XCD x takes the column tiles
    // x, x + 8, ... and runs ALL row tiles of one column tile back to back, so a weight tile is fetched into one L2 once and
    // re-read there by the other row tiles (the plain x-fastest order puts the row tiles of a column tile on different XCDs).
    const int n_rt = (g.M + BM - 1) / BM, n_ct = (g.N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ct = (jj / n_rt) * 8 + xcd, rt = jj % n_rt;
    if (ct >= n_ct) return;
    const int64_t m0 = (int64_t)rt * BM;
    const int n0 = ct * BN;
    const int n_it = g.K / MID_BK;
    // A pieces: thread -> (row = tid >> 3 (+32 i), 16-B slot tid & 7); B pieces: thread -> fragment (tid >> 6) (+4 i), lane
    const int a_row = tid >> 3, a_slot = tid & 7;
    // Every request below is UNCONDITIONAL on a clamped address (a row past M re-reads row M - 1, a column tile past N the last
    // tile, an iteration past the end the last one): rows / columns that do not exist only feed accumulator elements that are
    // never stored, and with no branch around a load the compiler can wait for an older stage with a counted `s_waitcnt vmcnt(n)`
    // instead of draining the loads it has just issued (a full round trip per iteration otherwise).
    const bf16_t* a_src[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int64_t m = m0 + a_row + 32 * i;
        if (m >= g.M) m = g.M - 1;
        a_src[i] = g.A + m * g.K + a_slot * 8;
    }
    const int nt0 = n0 >> 5;
    const s8_t* b_src[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = (tid >> 6) + 4 * i;                        // fragment f -> (column tile f >> 2, k tile f & 3)
        int t_n = nt0 + (f >> 2);
        if (t_n >= g.NT) t_n = g.NT - 1;
        b_src[i] = reinterpret_cast<const s8_t*>(g.Wp) + ((int64_t)t_n * g.KT + (f & 3)) * 64 + lane;
    }
    struct Regs { s8_t a[NA], b[NB]; };
    auto issue = [&](int it, Regs& rg) {
        if (it >= n_it) it = n_it - 1;
#pragma unroll
        for (int i = 0; i < NA; ++i) rg.a[i] = *reinterpret_cast<const s8_t*>(a_src[i] + (int64_t)it * MID_BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rg.b[i] = b_src[i][(int64_t)it * (MID_BK / 16) * 64];
    };
    auto stash = [&](int buf, const Regs& rg) {
        unsigned char* base = lds + buf * (A_BYTES + B_BYTES);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = a_row + 32 * i;
            *reinterpret_cast<s8_t*>(base + row * 128 + ((a_slot ^ (row & 7)) << 4)) = rg.a[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<s8_t*>(base + A_BYTES + ((tid >> 6) + 4 * i) * 1024 + lane * 16) = rg.b[i];
    };
    f16_t acc[WM][WN], tot[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[a][b][i] = 0.f; tot[a][b][i] = 0.f; }
    int seg_left = g.seg_it;
    auto flush = [&]() {                       // end of a K segment (uniform branch): total += segment sum
        if (g.seg_it > 0 && --seg_left == 0) {
            seg_left = g.seg_it;
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { tot[a][b][i] += acc[a][b][i]; acc[a][b][i] = 0.f; }
        }
    };
    auto multiply = [&](int buf) {
        const unsigned char* base = lds + buf * (A_BYTES + B_BYTES);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s8_t fa[WM], fb[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int row = (wm * WM + a) * 32 + r;
                fa[a] = *reinterpret_cast<const s8_t*>(base + row * 128 + (((kt * 2 + h) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) fb[b] = *reinterpret_cast<const s8_t*>(base + A_BYTES + ((wn * WN + b) * 4 + kt) * 1024 + lane * 16);
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = mfma32(fa[a], fb[b], acc[a][b]);
        }
    };
    // stage s: requested at iteration s - PF into register set s % PF, stashed at the end of iteration s - 1, multiplied at
    // iteration s (PF even: the LDS buffer of stage it + j is j & 1 whenever it is a multiple of PF)
    static_assert(PF % 2 == 0, "PF even");
    Regs rg[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) issue(j, rg[j]);
    stash(0, rg[0]);
    __syncthreads();
    int it = 0;
    for (; it + PF <= n_it; it += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            issue(it + j + PF, rg[j]);                   // (rg[j] held stage it + j, stashed an iteration ago)
            multiply(j & 1);
            flush();
            stash((j + 1) & 1, rg[(j + 1) % PF]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) {
        if (it + j < n_it) {
            multiply(j & 1);
            flush();
            stash((j + 1) & 1, rg[j + 1]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const int n = n0 + (wn * WN + b) * 32 + r;
            if (n >= g.N) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = m0 + (wm * WM + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < g.M) g.out[m * g.ldc + n] = g.seg_it > 0 ? tot[a][b][i] : acc[a][b][i];
            }
        }
}

// Causal dilated conv as an implicit GEMM with the input window held in LDS (codec decoder k=7 convs, operand = hi / lo bf16
// planes).  k_gemm_tiled walks K = tap x channel in 32-wide steps and re-loads a shifted copy of the same input rows for every
// tap: 7 global loads, 7 barriers and 7 exposed round trips per 32 channels.  Here a workgroup loads the rows its 128 outputs
// can see - 128 + (taps-1)*dilation of them - ONCE per 32-channel chunk, and the taps are fragment reads at shifted LDS rows:
// one barrier per `taps` MFMA steps, the next chunk's window in flight during all of them.
// Requires rows_in == rows_out, tap_offset = -(taps-1)*tap_stride, Cin % 32 == 0, (taps-1)*tap_stride <= 63, split_k = 1
// (the LDS window has 64 rows of halo room; its last row is then never a window row and always holds zeros: the ZERO ROW below).
// The workgroup tile is (WGM*MT*32) rows: 128 by default; 256 (MT doubled) for the 96- and 192-channel stages, whose four
// waves all need the SAME weight fragments (the waves split the rows, not the 96 columns) - with 3 workgroups per CU the
// per-CU vector L1 (64 B/clk) then moves as many weight bytes per tap as the SIMDs spend cycles on its MFMAs; twice the
// rows per weight fragment halves that.
//
// FUSE (the 96-channel residual units of the codec decoder, where both convs are bound by HBM traffic, not by the matrix
// cores): the unit's 1x1 conv runs in the SAME launch.  A wave holds all 96 output channels of its rows (WGN = 1), so after
// the K loop it applies bias + SnakeBeta, writes the hi / lo bf16 planes of the result - exactly what the unfused form stores
// to HBM for the next launch - to a private LDS region in A-operand order, multiplies by the 96 x 96 weight of the 1x1 conv
// (fragments from L2) and runs that conv's epilogue (bias, residual, the next unit's SnakeBeta planes).  Saved per element:
// the 4-byte write and the 4-byte re-read of the intermediate planes (24 -> 16 bytes of HBM traffic per element and unit) and
// one launch.
// TAPS > 0: the tap count is a compile-time constant (7 for every k = 7 conv of the codec decoder) and the tap loop is unrolled -
// the weight-fragment double buffer then alternates by renaming instead of by 16-24 register copies per tap.  Round 4's counters
// (SQ_INSTS_VALU against SQ_INSTS_MFMA, profiles/r04_pmc_vocoder_sq.txt) showed 4.5-6 vector instructions per MFMA in these
// kernels - copies, the zero select of the causal padding, LDS addressing - competing with the MFMAs for the SIMD's issue slots:
// the matrix pipe sat at 0.3-0.5 while the waves waited to ISSUE (SQ_WAIT_INST_ANY 0.4-0.47 of their cycles), not for memory.
template <int WGM, int WGN, int MT, int NTT, bool FUSE = false, int TAPS = 0>
__global__ __launch_bounds__(256, (WGM * MT * 32 > 128) ? 2 : 3) void k_conv_win(TiledArgs g) {
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(!FUSE || (WGN == 1 && NTT == 3), "the fused 1x1 conv needs every output channel of a row in one wave");
    constexpr int BMT = WGM * MT * 32;
    constexpr int WIN_ROWS = BMT + 64;
    constexpr int NP = (WIN_ROWS * 4 + 255) / 256;        // 16-B window pieces per thread and plane
    constexpr int BNT = WGN * NTT * 32;
    constexpr int PLANE = WIN_ROWS * 64;                  // bytes per plane and stage
    constexpr int F_ROWB = NTT * 32 * 2 + 16;             // fused form: bytes per row of a wave's staged tile (96 bf16 + padding)
    constexpr int F_WAVE = 2 * 32 * F_ROWB;               // ... hi and lo planes of 32 rows
    constexpr int LDS_BYTES = (FUSE && 4 * F_WAVE > 2 * 2 * PLANE) ? 4 * F_WAVE : 2 * 2 * PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WGN, wn = w % WGN;
    const int r = lane & 31, h = lane >> 5;
    int64_t rt = blockIdx.x;
    int ct = blockIdx.y;
    if (g.xcd_order) {                                    // same XCD-aware tile order as k_gemm_tiled
        const int64_t L = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = (int)(L & 7);
        const int64_t j = L >> 3;
        ct = (int)(j % gridDim.y);
        rt = (j / gridDim.y) * 8 + xcd;
        if (rt * BMT >= g.a.M) return;
    }
    const int64_t m0 = rt * BMT;
    const int n0 = ct * BNT;
    const int taps = TAPS > 0 ? TAPS : g.a.taps, stride = g.a.tap_stride, Cin = g.a.Cin;
    const int halo = (taps - 1) * stride, WR = BMT + halo, n_cc = Cin >> 5;
    const bf16_t* __restrict__ hi = reinterpret_cast<const bf16_t*>(g.a.ptr);
    const bf16_t* __restrict__ lo = reinterpret_cast<const bf16_t*>(g.a.ptr_lo);

    // ---- window staging: piece p = tid + 256 i -> (window row p >> 2, 16-B slot p & 3); global row = m0 - halo + row.
    // Buffer loads: one resource descriptor per plane in SGPRs (base = the first window row that exists, size = what is left of
    // the buffer), one 32-bit byte offset per piece in a VGPR, rows before the buffer's start get an offset beyond the size - the
    // hardware returns zeros for every out-of-range piece.  The 64-bit per-piece addresses of plain global loads (hoisted out of
    // the chunk loop by the compiler: 12 registers) had pushed this kernel over its 168-VGPR budget (3 waves per SIMD) into
    // scratch, and the zero fill needed a branch per piece.
    const int64_t first_row = m0 - halo > 0 ? m0 - halo : 0;          // first window row that exists
    const int skip = (int)(first_row - (m0 - halo));                  // window rows in front of the buffer
    int64_t rows_left = g.a.M - first_row;
    if (rows_left > WR - skip) rows_left = WR - skip;
    if (rows_left < 0) rows_left = 0;
    const unsigned win_bytes = (unsigned)(rows_left * Cin * 2);
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(hi + first_row * Cin), 0, win_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(lo + first_row * Cin), 0, win_bytes, 0x00020000);
    unsigned p_off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = tid + i * 256, row = p >> 2, slot = p & 3;
        p_off[i] = (row >= skip && row < WR) ? (unsigned)(((row - skip) * Cin + slot * 8) * 2) : 0x80000000u;
    }
    auto load_win = [&](int cc, s8_t (&a)[NP], s8_t (&l)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            a[i] = __builtin_bit_cast(s8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, p_off[i] + cc * 64, 0, 0));
            l[i] = __builtin_bit_cast(s8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, p_off[i] + cc * 64, 0, 0));
        }
    };
    auto store_win = [&](int buf, const s8_t (&a)[NP], const s8_t (&l)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = tid + i * 256;
            if ((p >> 2) < WIN_ROWS) {
                *reinterpret_cast<s8_t*>(lds + buf * 2 * PLANE + lds_a_off(p >> 2, p & 3)) = a[i];
                *reinterpret_cast<s8_t*>(lds + buf * 2 * PLANE + PLANE + lds_a_off(p >> 2, p & 3)) = l[i];
            }
        }
    };
    // ---- B fragments straight from the packed weights: k tile of (tap, chunk, kk) = (tap * Cin + 32 cc) / 16 + kk.
    // Buffer loads as well: the lane's byte offset inside its column tile is loop-invariant (one VGPR per column tile), the k tile
    // is a scalar offset, a column tile beyond the matrix reads zeros
    const int nt_base = (n0 >> 5) + wn * NTT;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)g.Wp, 0, (unsigned)((int64_t)g.NT * g.KT * 1024), 0x00020000);
    unsigned b_off[NTT];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) b_off[nt] = nt_base + nt < g.NT ? (unsigned)(((nt_base + nt) * g.KT * 64 + lane) * 16) : 0x80000000u;
    auto load_b = [&](int tap, int cc, s8_t (&b)[NTT][2]) {
        const int t_k0 = (tap * Cin + cc * 32) >> 4;
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[nt][kk] = __builtin_bit_cast(s8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[nt], (t_k0 + kk) * 1024, 0));
    };
    // time index of this lane's fragment rows: a tap reaching before the start of its item reads zeros
    int t_row[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m0 + (wm * MT + mt) * 32 + r;
        t_row[mt] = (int)(m % g.a.rows_out);
    }

    f16_t acc[MT][NTT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    s8_t ra[NP], rl[NP], rb[NTT][2], rb_next[NTT][2];
    load_win(0, ra, rl);
    load_b(0, 0, rb);
    store_win(0, ra, rl);
    __syncthreads();
    for (int cc = 0; cc < n_cc; ++cc) {
        const int buf = cc & 1;
        const bool more = cc + 1 < n_cc;
        if (more) load_win(cc + 1, ra, rl);
#pragma unroll
        for (int tap = 0; tap < (TAPS > 0 ? TAPS : taps); ++tap) {
            const bool last = tap + 1 == taps;
            if (!last) load_b(tap + 1, cc, rb_next);
            else if (more) load_b(0, cc + 1, rb_next);
            const int reach = (taps - 1 - tap) * stride;          // rows this tap looks back
            s8_t fa[MT][2], fl[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // a tap reaching before the start of its item reads zeros: such a lane reads the window's ZERO ROW instead of its
                // own row - one select on the row index per sub-tile and tap, where zeroing the loaded fragments took 16 (2
                // v_cndmask per MFMA: 193 selects against 96 MFMAs in the unrolled block, competing with them for issue slots)
                const int wrow = t_row[mt] >= reach ? (wm * MT + mt) * 32 + r + tap * stride : WIN_ROWS - 1;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    fa[mt][kk] = *reinterpret_cast<const s8_t*>(lds + buf * 2 * PLANE + lds_a_off(wrow, kk * 2 + h));
                    fl[mt][kk] = *reinterpret_cast<const s8_t*>(lds + buf * 2 * PLANE + PLANE + lds_a_off(wrow, kk * 2 + h));
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTT; ++nt) {
                        acc[mt][nt] = mfma32(fa[mt][kk], rb[nt][kk], acc[mt][nt]);
                        acc[mt][nt] = mfma32(fl[mt][kk], rb[nt][kk], acc[mt][nt]);
                    }
            if (!last || more) {
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) rb[nt][kk] = rb_next[nt][kk];
            }
        }
        if (more) store_win(buf ^ 1, ra, rl);
        __syncthreads();
    }
    if constexpr (!FUSE) {
        tile_epilogue<MT, NTT>(g, acc, m0, n0, wm, wn, r, h);
    } else {
        // (every wave is past the last barrier of the K loop: the window buffers are free, each wave takes its own region)
        unsigned char* my = lds + w * F_WAVE;
        const GemmEpi& e = g.e;
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)g.Wp2, 0, (unsigned)((int64_t)g.NT2 * g.KT2 * 1024), 0x00020000);
        // the per-channel constants of this conv's epilogue and the first weight fragments of the 1x1 conv: requested once, together
        // (inside the row-tile loop they were a dependent round trip per column tile and row tile, and every MFMA of the 1x1 conv
        // waited for a fragment requested just in front of it)
        float f_bias[NTT], f_sa[NTT], f_sib[NTT];
        {
            const float* bp = e.bias ? e.bias : e.snake_a;           // (stand-in: no branch around a request)
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                const int n = nt * 32 + r;
                f_bias[nt] = bp[n];
                f_sa[nt] = e.snake_a[n];
                f_sib[nt] = e.snake_ib[n];
            }
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
                if (!e.bias) f_bias[nt] = 0.f;
        }
        auto load_b2 = [&](int k2, s8_t (&b)[NTT]) {
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
                b[nt] = __builtin_bit_cast(s8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, (unsigned)(((nt * g.KT2 + k2) * 64 + lane) * 16), 0, 0));
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            s8_t b2[2][NTT];
            load_b2(0, b2[0]);
            // 1. this conv's epilogue (bias, SnakeBeta) into LDS as hi / lo planes, [row][channel]
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                const int n = nt * 32 + r;
                const float bias = f_bias[nt], sa = f_sa[nt], sib = f_sib[nt];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[mt][nt][i] + bias;
                    const float sn = __sinf(v * sa);
                    v = v + sib * sn * sn;
                    const bf16_t hi = f32_to_bf16(v);
                    const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    *reinterpret_cast<bf16_t*>(my + row * F_ROWB + n * 2) = hi;
                    *reinterpret_cast<bf16_t*>(my + 32 * F_ROWB + row * F_ROWB + n * 2) = lo;
                }
            }
            // (the same wave reads what it wrote: a wave's LDS operations complete in order)
            // 2. the 1x1 conv: [32 rows][96] x W2^T, hi + lo planes against each weight fragment
            f16_t acc2[1][NTT];
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[0][nt][i] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < NTT * 2; ++k2) {
                if (k2 + 1 < NTT * 2) load_b2(k2 + 1, b2[(k2 + 1) & 1]);          // the next k-step's fragments fly during this step's MFMAs
                const s8_t a_hi = *reinterpret_cast<const s8_t*>(my + r * F_ROWB + (k2 * 16 + h * 8) * 2);
                const s8_t a_lo = *reinterpret_cast<const s8_t*>(my + 32 * F_ROWB + r * F_ROWB + (k2 * 16 + h * 8) * 2);
#pragma unroll
                for (int nt = 0; nt < NTT; ++nt) {
                    acc2[0][nt] = mfma32(a_hi, b2[k2 & 1][nt], acc2[0][nt]);
                    acc2[0][nt] = mfma32(a_lo, b2[k2 & 1][nt], acc2[0][nt]);
                }
            }
            // 3. the 1x1 conv's epilogue on this wave's 32 rows
            tile_epilogue_e<1, NTT>(g.e2, g.a.M, g.N2, acc2, m0, 0, wm * MT + mt, 0, r, h);
        }
    }
}

}  // namespace

size_t packed_bytes(int N, int K) {
    const size_t Np = (size_t)(N + 31) / 32 * 32, Kp = (size_t)(K + 15) / 16 * 16;
    return Np * Kp * 2;
}

size_t packed16_bytes(int N, int K) { return (size_t)((N + 15) / 16 * 16) * K * 2; }

int launch_pack_weight16(rt_ctx* ctx, const bf16_t* d_src, int N, int K, bf16_t* d_dst, PackedW* out) {
    if (K % 32) return rt_fail(ctx, RT_ERR_INVALID, "pack16: K=%d must be a multiple of 32", K);
    const int Np = (N + 15) / 16 * 16;
    const int64_t pieces = (int64_t)(Np / 16) * (K / 32) * 64;
    int64_t blocks = (pieces + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_weight16, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_src, N, K, Np, d_dst);
    RT_HIP(ctx, hipGetLastError());
    out->data16 = d_dst;
    out->Np16 = Np;
    return RT_OK;
}

int launch_pack_weight(rt_ctx* ctx, const bf16_t* d_src, int N, int K, bf16_t* d_dst, PackedW* out) {
    const int Np = (N + 31) / 32 * 32, Kp = (K + 15) / 16 * 16;
    const int64_t pieces = (int64_t)(Np / 32) * (Kp / 16) * 64;
    int64_t blocks = (pieces + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_src, N, K, Np, Kp, d_dst);
    RT_HIP(ctx, hipGetLastError());
    out->data = d_dst;
    out->N = N; out->K = K; out->Np = Np; out->Kp = Kp;
    return RT_OK;
}

rt_knob g_pred_nt{0};              // predictor weights: 0 = cacheable loads (Infinity-Cache resident across its 15 passes), 1 = non-temporal
rt_knob g_use_graph{1};            // 1: the decode frame is replayed from captured hipGraphs
rt_knob g_col_rows64{1};           // 1: one 64-row decode GEMM launch for the predictor's two-position pass, 0: two 32-row launches
rt_knob g_pair_attn{1};            // 1: two-position decode passes append both positions inside the fused attention launch, 0: k_qkv_post + k_attention (rt_debug_tune 2800/2801)
rt_knob g_frame_inc_fold{0};       // 1: frame += 1 by the last workgroup of the frame's talker-input launch, 0: k_frame_inc (rt_debug_tune 2700/2701; the fold measured 1-1.8 ms per step SLOWER)
rt_knob g_fuse_sample_embed{1};    // 1: sampler + next-input embedding in one launch (predictor groups), 0: separate k_embed_rowsq
rt_knob g_prefill_fill{3};          // workgroups per CU a prefill GEMM's split-K aims for
rt_knob g_xcd_order{1};             // 1: tiled GEMMs run a row tile's column tiles back to back on one XCD
rt_knob g_final_conv{1};           // 1: the codec decoder's last conv runs in its own LDS-window kernel
rt_knob g_col_max_rows{64};         // batches up to this many rows decode on the column-owner path (tune 20nn).  Above 32 the talker's GEMMs take 64 rows per
                                // launch and the predictor's two-position first pass runs as two 64-row launches: 715 audio-s/s at batch 64 against
                                // 510 at batch 32 (1.7B, bench.py --batch 64) - every weight byte serves twice the rows for ~1.4x the launch time
rt_knob g_conv_tall{1};            // 1: 256-row tiles for the k>1 convs of the 96- / 192-channel stages
rt_knob g_conv_unroll{1};          // 1: k = 7 convs run the tap-unrolled instantiation of k_conv_win (rt_debug_tune 2600 / 2601)
rt_knob g_conv_win{1};             // 1: k>1 convs on operand planes keep their input window in LDS (k_conv_win)
rt_knob g_tile96{1};               // 1: 128x96 workgroup tiles for N = 96 / 192 (codec decoder), 0: always 128x128
rt_knob g_col_split{0};            // 0: automatic (col_split_for), else forced 1 / 2 / 4
rt_knob g_attn_mfma{0};             // 1: the talker's decode attention runs its shared-prefix part on the matrix cores (attention_mfma.hip);
                                // measured 16.6 us per launch against 13.3 us for the vector-unit kernel at batch 32 / 460 prefix rows, so off
rt_knob g_handover_every{4};       // queued items (n_items > rows): frames between two looks at the flags + row hand-overs.  Measured on the
                                // 1.7B model, 512 / 64 ragged texts on 32 rows: 2 -> 487 / 445, 3 -> 489 / 445, 4 -> 491 / 447, 6 -> 485 / 445,
                                // 8 -> 477 / 428, 12 -> 475 / 434 audio-s/s (a hand-over costs ~1.4 ms, a waiting row 0.13 ms per frame)
rt_knob g_eos_check_every{8};      // frames between two host looks at the device-side end-of-sequence flags (1 = every frame)
rt_knob g_sync_parts{0};           // 1: rt_generate waits for the stream after every frame part (bounds the dispatches in flight; profiling aid)
rt_knob g_decode_lanes{1};         // decode lanes: groups of items decoding concurrently on their own streams (rt_generate)
rt_knob g_decode_col{1};           // 1: decode stacks use the column-owner GEMM + fused attention (5 launches per layer)
rt_knob g_skinny_variant{0};       // 0: k_gemm_skinny, 1: k_gemm_skinny2<.,4>, 2: k_gemm_skinny2<.,8>
rt_knob g_skinny_waves_per_cu{4};  // split-K is chosen so that about this many waves per CU stream weights

int skinny_pick_split(int M, int N, int K, int n_cu) {
    const int tiles = (N + 31) / 32, KT = (K + 15) / 16;
    const int target = n_cu * g_skinny_waves_per_cu;
    int s = 1;
    // (a K segment is a whole number of the prompt-prefill kernel's 64-deep steps whenever K allows it: k_gemm_mid adds K in
    // these very segments, so that a prompt row gets the same float32 sums from either kernel - 0.6B down-projection, K = 3072:
    // 16 segments of 12 k-tiles, not 32 of 6)
    while (s < 32 && tiles * (s * 2) <= target && KT % (s * 2) == 0 && KT / (s * 2) >= 4 && (KT % 4 != 0 || (KT / (s * 2)) % 4 == 0)) s *= 2;
    return s;
}

int launch_gemm_skinny(rt_ctx* ctx, const bf16_t* d_a, int M, const PackedW& w, float* d_out, int64_t ldc, int split_k,
                       hipEvent_t ev_start, hipEvent_t ev_stop, int64_t slab_stride) {
    if (slab_stride <= 0) slab_stride = (int64_t)M * ldc;
    if (M < 1 || M > 64) return rt_fail(ctx, RT_ERR_INVALID, "gemm_skinny: M=%d outside 1..64", M);
    if (w.K != w.Kp) return rt_fail(ctx, RT_ERR_INVALID, "gemm_skinny: K=%d must be a multiple of 16", w.K);
    const int NT = w.Np / 32, KT = w.Kp / 16;
    if (split_k < 1 || KT % split_k) return rt_fail(ctx, RT_ERR_INVALID, "gemm_skinny: split_k=%d does not divide %d k-tiles", split_k, KT);
    dim3 grid((NT + 3) / 4, split_k);
    if (g_skinny_variant > 0 && (ev_start || ev_stop) && slab_stride == (int64_t)M * ldc) {
        const int mt = M <= 32 ? 1 : 2;
        const size_t lds = (size_t)mt * 32 * ((KT / split_k) * 32 + 16);
        if (lds <= 64 * 1024) {
            if (g_skinny_variant == 1) {
                if (mt == 1) hipExtLaunchKernelGGL((k_gemm_skinny2<1, 4>), grid, dim3(256), lds, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N);
                else hipExtLaunchKernelGGL((k_gemm_skinny2<2, 4>), grid, dim3(256), lds, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N);
            } else {
                if (mt == 1) hipExtLaunchKernelGGL((k_gemm_skinny2<1, 8>), grid, dim3(256), lds, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N);
                else hipExtLaunchKernelGGL((k_gemm_skinny2<2, 8>), grid, dim3(256), lds, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N);
            }
            RT_HIP(ctx, hipGetLastError());
            return RT_OK;
        }
    }
    // hipExtLaunchKernelGGL stamps the events at the kernel's own begin/end on the device (no launch gaps inside)
    if (!ev_start && !ev_stop) {
        if (M <= 32) hipLaunchKernelGGL(k_gemm_skinny<1>, grid, dim3(256), 0, ctx->stream, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N, slab_stride);
        else hipLaunchKernelGGL(k_gemm_skinny<2>, grid, dim3(256), 0, ctx->stream, d_a, M, w.K, w.data, NT, KT, KT / split_k, d_out, ldc, w.N, slab_stride);
    } else if (M <= 32)
        hipExtLaunchKernelGGL(k_gemm_skinny<1>, grid, dim3(256), 0, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT,
                              KT / split_k, d_out, ldc, w.N, slab_stride);
    else
        hipExtLaunchKernelGGL(k_gemm_skinny<2>, grid, dim3(256), 0, ctx->stream, ev_start, ev_stop, 0, d_a, M, w.K, w.data, NT, KT,
                              KT / split_k, d_out, ldc, w.N, slab_stride);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

rt_knob g_prefill_mid{1};           // 1: prompt prefills of 65..1024 rows run their GEMMs on k_gemm_mid (no split-K slabs); 2 / 3 force its 64 / 128 tiles
bool gemm_mid_shape_ok(const PackedW& w) { return g_prefill_mid && w.K % MID_BK == 0 && w.K >= 128 && w.Kp == w.K; }
bool gemm_mid_ok(int M, const PackedW& w) { return gemm_mid_shape_ok(w) && M > 64 && M <= 1024; }
int launch_gemm_mid(rt_ctx* ctx, const bf16_t* A, int M, const PackedW& w, float* out, int64_t ldc) {
    if (!gemm_mid_ok(M, w)) return rt_fail(ctx, RT_ERR_INVALID, "gemm_mid: M=%d K=%d outside its range", M, w.K);
    MidArgs g{A, w.data, out, M, w.N, w.K, w.Np / 32, w.Kp / 16, ldc, 0};
    // same association of the K sum as the skinny kernel's slabs (its split depends on N and K only): see the kernel
    const int S = skinny_pick_split(M, w.N, w.K, ctx->n_cu);
    if (S > 1) {
        // (skinny_pick_split only returns such splits for K % 64 == 0; anything else would silently break the batch invariance)
        if ((g.KT / S) % (MID_BK / 16)) return rt_fail(ctx, RT_ERR_STATE, "gemm_mid: K segments of %d k-tiles are not whole 64-deep steps (K=%d, split %d)", g.KT / S, w.K, S);
        g.seg_it = (g.KT / S) / (MID_BK / 16);
    }
    // 128 x 64 tiles (3/4 of the cache-to-CU traffic of 64 x 64) only where they still give >= 3 workgroups per CU - the
    // gate/up projection; with fewer, a bigger tile cannot overlap its LDS phase with its MFMA phase (128 x 128 tiles at M = 460:
    // 25-42 us on the 64-128 workgroups of the narrow projections against 21-23 us for 64 x 64, 42.7 us on gate/up - but with the
    // segment totals they need 2 x 64 accumulator registers on top of the staging registers and spill)
    const bool big = g_prefill_mid == 1 ? (int64_t)((M + 127) / 128) * ((w.N + 63) / 64) >= 3 * ctx->n_cu : g_prefill_mid == 3;      // (2 / 3: forced)
    const int bm = big ? 128 : 64;
    const int n_rt = (M + bm - 1) / bm, n_ct = (w.N + 63) / 64;
    dim3 grid(8 * ((n_ct + 7) / 8) * n_rt);               // (XCD-aware order, see the kernel)
    if (big) hipLaunchKernelGGL((k_gemm_mid<2, 1, 4>), grid, dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL((k_gemm_mid<1, 1, 6>), grid, dim3(256), 0, ctx->stream, g);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

rt_knob g_fuse_conv{1};            // 1: a 96-channel k>1 conv and the 1x1 conv behind its activation run as one launch (launch_conv_pair)

// a k > 1 conv with SnakeBeta whose hi / lo output planes feed only the 1x1 conv (w2, e2): both in one launch when the first
// conv's workgroup tile holds every output channel (96 channels, the LDS-window kernel) - else two launches through the planes
bool conv_pair_fusable(const GemmA& a, const PackedW& w, const GemmEpi& e, const PackedW& w2) {
    return g_fuse_conv && g_conv_win && g_tile96 && w.N == 96 && w2.N == 96 && w2.K == 96 && w2.Kp == 96 && e.act == ACT_SNAKE && a.split && !a.is_f32 && a.ptr_lo && a.taps >= 2 &&
           a.Cin % 32 == 0 && a.rows_out > 0 && a.rows_in == a.rows_out && a.tap_offset == -(a.taps - 1) * a.tap_stride && (a.taps - 1) * a.tap_stride <= 63 &&
           a.M % a.rows_out == 0;
}

int launch_gemm(rt_ctx* ctx, const GemmA& a, const PackedW& w, const GemmEpi& e, const PackedW* w2, const GemmEpi* e2) {
    if (a.M <= 0) return RT_OK;
    if (a.Cin % 8 || (int64_t)a.Cin * a.taps != w.K)
        return rt_fail(ctx, RT_ERR_INVALID, "gemm: A has taps=%d Cin=%d but W has K=%d", a.taps, a.Cin, w.K);
    if (e.split_k < 1 || (e.split_k > 1 && !e.out_f32)) return rt_fail(ctx, RT_ERR_INVALID, "gemm: bad split_k");
    TiledArgs g;
    g.a = a;
    g.Wp = w.data;
    g.N = w.N; g.K = w.K; g.NT = w.Np / 32; g.KT = w.Kp / 16;
    g.e = e;
    int per = (g.KT + e.split_k - 1) / e.split_k;
    per = (per + 1) / 2 * 2;
    g.kt_per_split = per;
    // 96-wide workgroup tiles when N is a multiple of 96 but not of 128 (the decoder's 96- and 192-channel stages)
    const bool narrow = g_tile96 && e.split_k == 1 && w.N % 96 == 0 && w.N % 128 != 0;
    // codec decoder k>1 convs on operand planes: input window in LDS (k_conv_win)
    const bool conv_win = g_conv_win && a.split && !a.is_f32 && a.ptr_lo && a.taps >= 2 && a.Cin % 32 == 0 && a.rows_out > 0 && a.rows_in == a.rows_out &&
        a.tap_offset == -(a.taps - 1) * a.tap_stride && (a.taps - 1) * a.tap_stride <= 63 && e.split_k == 1 && a.M % a.rows_out == 0;
    const bool tall = conv_win && narrow && (g_conv_tall == 2 || (g_conv_tall == 1 && a.M >= 256 * 1024));   // 256-row tiles (see k_conv_win; 2 = forced, tests)
    const int bm = tall ? 256 : BM;
    const int64_t my = (a.M + bm - 1) / bm;
    if (my > 0x7fffffff) return rt_fail(ctx, RT_ERR_LENGTH, "gemm: length %lld rows too large", (long long)a.M);
    const int bn = narrow ? 96 : BN;
    const int ny = (w.N + bn - 1) / bn;
    g.xcd_order = (g_xcd_order && e.split_k == 1 && ny > 1 && my >= 64) ? 1 : 0;
    dim3 grid((unsigned)(g.xcd_order ? (my + 7) / 8 * 8 : my), ny, e.split_k);
    if (a.split && !a.is_f32 && !a.ptr_lo) return rt_fail(ctx, RT_ERR_INVALID, "gemm: split precision needs an f32 A operand or a low plane");
    if ((e.out_hi && !e.out_lo) || (e.out2_hi && (!e.out2_lo || (e.act2 != ACT_ELU && (!e.snake2_a || !e.snake2_ib)))) ||
        ((e.out_hi || e.out2_hi) && e.split_k > 1))
        return rt_fail(ctx, RT_ERR_INVALID, "gemm: incomplete hi/lo plane output");
    if (w2) {           // fused conv pair: only the form conv_pair_fusable() describes
        if (!e2 || !conv_win || !narrow || ny != 1 || w.N != 96 || w2->N != 96 || w2->K != 96 || e.act != ACT_SNAKE || !e.snake_a || !e.snake_ib ||
            e2->split_k != 1 || e2->ldc < 96)
            return rt_fail(ctx, RT_ERR_INVALID, "gemm: this conv pair cannot be fused (96 channels, k > 1 conv with SnakeBeta, then a 96 x 96 1x1 conv)");
        g.Wp2 = w2->data; g.NT2 = w2->Np / 32; g.KT2 = w2->Kp / 16; g.N2 = w2->N; g.e2 = *e2;
        const bool t7f = a.taps == 7 && g_conv_unroll;
        if (tall && t7f) hipLaunchKernelGGL((k_conv_win<4, 1, 2, 3, true, 7>), grid, dim3(256), 0, ctx->stream, g);
        else if (tall) hipLaunchKernelGGL((k_conv_win<4, 1, 2, 3, true>), grid, dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL((k_conv_win<4, 1, 1, 3, true>), grid, dim3(256), 0, ctx->stream, g);
        RT_HIP(ctx, hipGetLastError());
        return RT_OK;
    }
    if (conv_win) {
        const bool t7 = a.taps == 7 && g_conv_unroll;
        if (tall && t7) hipLaunchKernelGGL((k_conv_win<4, 1, 2, 3, false, 7>), grid, dim3(256), 0, ctx->stream, g);
        else if (tall) hipLaunchKernelGGL((k_conv_win<4, 1, 2, 3>), grid, dim3(256), 0, ctx->stream, g);
        else if (narrow) hipLaunchKernelGGL((k_conv_win<4, 1, 1, 3>), grid, dim3(256), 0, ctx->stream, g);
        else if (t7) hipLaunchKernelGGL((k_conv_win<2, 2, 2, 2, false, 7>), grid, dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL((k_conv_win<2, 2, 2, 2>), grid, dim3(256), 0, ctx->stream, g);
        RT_HIP(ctx, hipGetLastError());
        return RT_OK;
    }
#define RT_TILED(AF, SP)                                                                                          \
    do {                                                                                                          \
        if (narrow) hipLaunchKernelGGL((k_gemm_tiled<AF, SP, 4, 1, 1, 3>), grid, dim3(256), 0, ctx->stream, g);   \
        else hipLaunchKernelGGL((k_gemm_tiled<AF, SP, 2, 2, 2, 2>), grid, dim3(256), 0, ctx->stream, g);          \
    } while (0)
    if (a.split && !a.is_f32) RT_TILED(false, true);
    else if (a.split) RT_TILED(true, true);
    else if (a.is_f32) RT_TILED(true, false);
    else RT_TILED(false, false);
#undef RT_TILED
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}
