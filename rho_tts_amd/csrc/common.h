// Internal helpers shared by the HIP translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rho_tts_amd.h"

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
