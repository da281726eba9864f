// Context lifecycle of the C ABI (include/rho_tts_amd.h, "context" group).
#include <cstdlib>
#include <hip/hip_ext.h>

#include "common.h"

std::shared_mutex g_tune_mu;               // launch-plan switches against calls in flight (common.h, CtxLock)
std::atomic<int> g_runs_in_flight{0};      // resumable generations between rt_generate_begin and rt_generate_end, all models

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_status_string(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID: return "invalid argument";
        case RT_ERR_OOM: return "out of memory";
        case RT_ERR_LENGTH: return "length limit exceeded";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_CANCELLED: return "cancelled";
        case RT_ERR_STATE: return "invalid state";
        case RT_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

int rt_create(int device_ordinal, rt_ctx** out_ctx) {
    if (!out_ctx) return RT_ERR_INVALID;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_ERR_HIP;
    if (device_ordinal < 0 || device_ordinal >= n) return RT_ERR_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return RT_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return RT_ERR_HIP;
    // gfx950 code objects only: refuse anything else loudly instead of failing at first launch.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RT_ERR_UNSUPPORTED;
    rt_ctx* ctx = new rt_ctx();
    ctx->device = device_ordinal;
    ctx->n_cu = prop.multiProcessorCount;
    snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
    // RHO_TTS_AMD_CU_MASK="first:count" (measurement aid, tools/overlap_probe.py): the context's stream may only use that range of CUs
    const char* cm = getenv("RHO_TTS_AMD_CU_MASK");
    int first = 0, count = 0;
    hipError_t se;
    if (cm && sscanf(cm, "%d:%d", &first, &count) == 2 && first >= 0 && count > 0 && first + count <= ctx->n_cu) {
        uint32_t mask[32] = {0};
        for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
        se = hipExtStreamCreateWithCUMask(&ctx->own_stream, (uint32_t)((ctx->n_cu + 31) / 32), mask);
    } else {
        se = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out_ctx = ctx;
    return RT_OK;
}

int rt_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int rt_set_stream(rt_ctx* ctx, void* hip_stream) {
    if (!ctx) return RT_ERR_INVALID;
    CtxLock g(ctx);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RT_OK;
}

int rt_synchronize(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_device_info(rt_ctx* ctx, char* arch, size_t arch_cap, int* n_cu, int64_t* hbm_free, int64_t* hbm_total) {
    if (!ctx) return RT_ERR_INVALID;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (arch && arch_cap) snprintf(arch, arch_cap, "%s", ctx->arch);
    if (n_cu) *n_cu = ctx->n_cu;
    size_t f = 0, t = 0;
    RT_HIP(ctx, hipMemGetInfo(&f, &t));
    if (hbm_free) *hbm_free = (int64_t)f;
    if (hbm_total) *hbm_total = (int64_t)t;
    return RT_OK;
}

}  // extern "C"

int rt_ctx_scratch(rt_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->d_scratch_bytes) {
        if (ctx->d_scratch) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            RT_HIP(ctx, hipFree(ctx->d_scratch));
            ctx->d_scratch = nullptr;
            ctx->d_scratch_bytes = 0;
        }
        size_t want = bytes + bytes / 2 + 4096;
        RT_HIP(ctx, hipMalloc(&ctx->d_scratch, want));
        ctx->d_scratch_bytes = want;
    }
    *out = ctx->d_scratch;
    return RT_OK;
}

int rt_ctx_pinned(rt_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->h_pinned_bytes) {
        if (ctx->h_pinned) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            RT_HIP(ctx, hipHostFree(ctx->h_pinned));
            ctx->h_pinned = nullptr;
            ctx->h_pinned_bytes = 0;
        }
        size_t want = bytes + bytes / 2 + 4096;
        RT_HIP(ctx, hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
        ctx->h_pinned_bytes = want;
    }
    *out = ctx->h_pinned;
    return RT_OK;
}
