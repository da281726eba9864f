// Internal helpers shared by the HIP translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/rho_tts_amd_debug.h"

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string last_error;
    int n_cu = 0;
    char arch[64] = {0};
    // small reusable device/host staging areas
    void* d_scratch = nullptr;
    size_t d_scratch_bytes = 0;
    void* h_pinned = nullptr;
    size_t h_pinned_bytes = 0;
};

// Process-wide launch-plan switches (rt_debug_tune) against calls in flight: every entry point that does work holds g_tune_mu
// SHARED for its whole duration (CtxLock, together with its context's mutex); rt_debug_tune takes it EXCLUSIVELY - so it waits
// until no call is executing on any context and no call can see a half-changed plan - and refuses while a resumable generation
// (rt_generate_begin .. rt_generate_end) is in flight on any model (g_runs_in_flight).  The switches themselves are atomics
// (kernels.h): read relaxed on the launch paths, written only under the exclusive lock.
typedef std::atomic<int> rt_knob;
extern std::shared_mutex g_tune_mu;
extern std::atomic<int> g_runs_in_flight;
struct CtxLock {
    std::shared_lock<std::shared_mutex> tune;
    std::lock_guard<std::mutex> ctx;
    explicit CtxLock(rt_ctx* c) : tune(g_tune_mu), ctx(c->mu) {}
};

inline int rt_fail(rt_ctx* ctx, int status, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return status;
}

inline int rt_hip_status(hipError_t e) {
    return e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP;
}

// "out of memory" must appear verbatim in the text for OOM: the pipeline's retry
// policy matches on that substring (base_tts.py:789).
#define RT_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            return rt_fail((ctx), rt_hip_status(_e), "%s%s failed: %s (%s:%d)",                    \
                           _e == hipErrorOutOfMemory ? "out of memory: " : "", #expr,              \
                           hipGetErrorString(_e), __FILE__, __LINE__);                             \
        }                                                                                          \
    } while (0)

int rt_ctx_scratch(rt_ctx* ctx, size_t bytes, void** out);  // grows ctx->d_scratch
int rt_ctx_pinned(rt_ctx* ctx, size_t bytes, void** out);   // grows ctx->h_pinned

// ---------------------------------------------------------------- device utils
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over aligned groups of G = 2, 4, 8 or 16 lanes (every lane of the group gets the total), by DPP modifiers on the adds instead of
// __shfl_xor: hipcc lowers __shfl_xor to ds_bpermute_b32 - an LDS-crossbar round trip with an `s_waitcnt lgkmcnt(0)` behind each of
// the log2(G) DEPENDENT steps (~100 cycles apiece) - where a DPP operand costs nothing beyond the add.  Pairings: xor 1 and xor 2 are
// quad permutations; for xor 4 / xor 8 the row is mirrored within halves / as a whole (lane i <-> 7 - i / 15 - i) - a different lane
// than i ^ 4 / i ^ 8, but one that holds the SAME value at that step (after the earlier steps every lane of a quad / half holds the
// quad's / half's total), so the sums are bit-identical to the xor butterfly in ascending order.  All lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_lane_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int G>
__device__ __forceinline__ float group_sum_f32(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "aligned lane groups within a DPP row");
    if (G >= 2) v += dpp_lane_f32<0xB1>(v);     // quad_perm [1, 0, 3, 2]
    if (G >= 4) v += dpp_lane_f32<0x4E>(v);     // quad_perm [2, 3, 0, 1]
    if (G >= 8) v += dpp_lane_f32<0x141>(v);    // row_half_mirror
    if (G >= 16) v += dpp_lane_f32<0x140>(v);   // row_mirror
    return v;
}
// Whole-wave sum on every lane without LDS round trips: DPP adds inside the four 16-lane rows, then the four row totals through
// v_readlane (scalar registers).  Six dependent ds_bpermute steps of wave_sum_f32 are ~0.3 us on a launch's critical path; this is
// four DPP adds and four readlanes.  ALL 64 lanes must be active (wave-uniform call sites only); the association differs from
// wave_sum_f32's (rows first), so the two are not interchangeable where another kernel must reproduce the bits.
__device__ __forceinline__ float wave_sum_rows_f32(float v) {
    const int x = __float_as_int(group_sum_f32<16>(v));
    return (__int_as_float(__builtin_amdgcn_readlane(x, 0)) + __int_as_float(__builtin_amdgcn_readlane(x, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(x, 32)) + __int_as_float(__builtin_amdgcn_readlane(x, 48)));
}
// The same for order-independent integer / max reductions of the sampler (results equal the shuffle forms bit for bit).
template <int CTRL, bool ZERO_FILL = false>
__device__ __forceinline__ int dpp_lane_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, ZERO_FILL); }
template <int G>
__device__ __forceinline__ int group_sum_i32(int v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "aligned lane groups within a DPP row");
    if (G >= 2) v += dpp_lane_i32<0xB1>(v);
    if (G >= 4) v += dpp_lane_i32<0x4E>(v);
    if (G >= 8) v += dpp_lane_i32<0x141>(v);
    if (G >= 16) v += dpp_lane_i32<0x140>(v);
    return v;
}
// Largest value of the wave on every lane (`a > b ? a : b` at every step, as the shuffle form it replaces).  All lanes active.
__device__ __forceinline__ float wave_max_rows_f32(float v) {
    float o;
    o = dpp_lane_f32<0xB1>(v);  v = o > v ? o : v;
    o = dpp_lane_f32<0x4E>(v);  v = o > v ? o : v;
    o = dpp_lane_f32<0x141>(v); v = o > v ? o : v;
    o = dpp_lane_f32<0x140>(v); v = o > v ? o : v;
    const int x = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    const float a = r1 > r0 ? r1 : r0, b = r3 > r2 ? r3 : r2;
    return b > a ? b : a;
}
// Inclusive prefix sum over the 64 lanes: Hillis-Steele inside the 16-lane rows by row_shr (zero fill), then the totals of the
// rows below through v_readlane.  All lanes active.
__device__ __forceinline__ unsigned wave_scan_incl_u32(unsigned u) {
    int v = (int)u;
    v += dpp_lane_i32<0x111, true>(v);          // row_shr:1
    v += dpp_lane_i32<0x112, true>(v);          // row_shr:2
    v += dpp_lane_i32<0x114, true>(v);          // row_shr:4
    v += dpp_lane_i32<0x118, true>(v);          // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return (unsigned)(v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0));
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Fragment-tiled activation layout shared by the decode kernels: a [rows][K] matrix is stored as
// [row block of 16][k tile of 32][lane = (k octet << 4) | row][8 elements], i.e. exactly the A-operand order of
// v_mfma_f32_16x16x32_bf16, so a wave fetches an operand tile with ONE fully coalesced 1-KiB load instead of 16
// row-strided pieces.  K % 32 == 0.
__host__ __device__ __forceinline__ int64_t tile_off(int m, int k, int K) {
    return ((((int64_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + ((((k >> 3) & 3) << 4) | (m & 15))) << 3) + (k & 7);
}

// bf16 <-> f32 (round-to-nearest-even, NaN preserved by the hardware cast on gfx950)
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even by the hardware converter (v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN) - the integer
// emulation costs ~10 VALU instructions per element, which showed up as microseconds in the conversion-heavy kernels
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) __bf16 rt_bf2_t;
__device__ __forceinline__ unsigned f32x2_to_bf16x2(float lo, float hi) {
    const rt_bf2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// Grid-wide barrier for a persistent kernel whose workgroups are ALL co-resident (grid <= CUs x occupancy, checked on
// the host).  bar[0] is a monotonic arrival counter, bar[1] an abort flag; both zeroed by the host before the launch.
// Every workgroup calls it with the same epoch 1, 2, 3 ...  The wait is bounded: after `max_spins` polls (or when
// another workgroup has given up) the barrier raises the abort flag and returns false, and the caller must return - a
// launch that is not fully resident therefore ends with an error instead of hanging the GPU.
// MODE 0: bulk cache maintenance (release write-back by thread 0, acquire invalidate by every wave) - any plain store
//         before the barrier is visible to any plain load after it, at the price of L2-wide flushes per workgroup;
// MODE 2: no cache maintenance - only data moved with agent-scope accesses (st_agent / ld_agent below: write-through
//         stores, L2-bypassing loads) is exchanged, which costs nothing at the barrier itself.
template <int MODE = 0>
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned n_wg, unsigned epoch, unsigned max_spins = 1u << 22) {
    __shared__ int gb_ok;
    if (MODE == 2) __builtin_amdgcn_s_waitcnt(0);           // this wave's write-through stores have been acknowledged
    __syncthreads();                                        // every wave's stores are issued and complete at workgroup scope
    if (threadIdx.x == 0) {
        if (MODE == 2) __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // release: write back for the other XCDs
        const unsigned target = epoch * n_wg;
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > max_spins || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        gb_ok = good;
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave: drop stale cache lines before reading other workgroups' data
    return gb_ok != 0;
}
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
