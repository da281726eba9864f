/*
 * rho_tts_amd.h — C ABI of the MI355X-native generation path for rho-tts.
 *
 * Everything a host binds is declared here: extern "C", plain pointers and
 * sizes, no torch / C++ types.  The reference (rhofield/rho-tts v1.1.4) is pure
 * Python and has no FFI of its own; each entry point therefore cites the
 * reference *Python* interface it stands behind (paths relative to
 * /root/reference/src/rho_tts/).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns an rt_status (0 = ok); rt_last_error(ctx) holds text
 *   - "d_" parameters are device (HBM) pointers, "h_" parameters host pointers
 *   - all device work is issued on the context's stream (rt_set_stream lets a
 *     host share its own, e.g. torch's current stream)
 *   - the library is callable from any host thread; calls on one context are
 *     serialised by an internal mutex (base_tts.py has no re-entrancy guard and
 *     the UI shares one instance across threads, ui/state.py:54,85-87)
 *
 * Status -> Python exception mapping used by the provider
 * (base_tts.py:786-797 decides retry vs. propagate on exactly these classes):
 *   RT_ERR_INVALID   -> ValueError                  (configuration error, never retried)
 *   RT_ERR_OOM       -> RuntimeError("out of memory ...")   (empty_cache + retry)
 *   RT_ERR_LENGTH    -> RuntimeError("length ...")          (retry)
 *   RT_ERR_CANCELLED -> CancelledException          (cancellation.py:14)
 *   anything else    -> RuntimeError
 */
#ifndef RHO_TTS_AMD_H
#define RHO_TTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API __attribute__((visibility("default")))

#define RT_ABI_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = 1,
    RT_ERR_OOM = 2,
    RT_ERR_LENGTH = 3,
    RT_ERR_HIP = 4,
    RT_ERR_CANCELLED = 5,
    RT_ERR_STATE = 6,
    RT_ERR_UNSUPPORTED = 7
} rt_status;

typedef struct rt_ctx rt_ctx;     /* one per (process, GPU) */
typedef struct rt_model rt_model; /* weights + KV cache + workspaces of one Qwen3-TTS-shaped model */

/* ------------------------------------------------------------------ context */

RT_API int rt_abi_version(void);
RT_API const char* rt_status_string(int status);
/* device_ordinal: HIP device index (LOCAL_RANK in a one-process-per-GPU job). */
RT_API int rt_create(int device_ordinal, rt_ctx** out_ctx);
RT_API int rt_destroy(rt_ctx* ctx);
RT_API const char* rt_last_error(rt_ctx* ctx);
/* hip_stream: a hipStream_t (NULL restores the context's own stream). */
RT_API int rt_set_stream(rt_ctx* ctx, void* hip_stream);
RT_API int rt_synchronize(rt_ctx* ctx);
/* Name of the GPU architecture the context runs on, e.g. "gfx950". */
RT_API int rt_device_info(rt_ctx* ctx, char* arch, size_t arch_cap, int* n_cu, int64_t* hbm_free, int64_t* hbm_total);

/* --------------------------------------------------------- post-processing
 * One fused launch, one workgroup per text item, standing behind the numeric
 * leaves of the pipeline:
 *   _trim_silence          base_tts.py:348-392      RT_POST_TRIM_START / _END
 *   _remove_dc_offset      base_tts.py:394-399      RT_POST_DC
 *   _apply_fades           base_tts.py:401-433      RT_POST_FADE_IN / _OUT
 *   _smooth_segment_join   base_tts.py:435-536      RT_POST_JOIN (implies per-position trim flags)
 *   QwenTTS._post_process_audio + _apply_windowed_normalization
 *                          providers/qwen.py:268-378  RT_POST_LOUDNESS
 *   _validate_sound_decay  base_tts.py:297-323      RT_POST_DECAY (statistics only)
 * The full per-item tail of _run_pipeline (base_tts.py:912-926) is
 * RT_POST_PIPELINE.
 */
#define RT_POST_TRIM_START 0x01u
#define RT_POST_TRIM_END   0x02u
#define RT_POST_DC         0x04u
#define RT_POST_FADE_IN    0x08u
#define RT_POST_FADE_OUT   0x10u
#define RT_POST_JOIN       0x20u
#define RT_POST_LOUDNESS   0x40u
#define RT_POST_DECAY      0x80u
#define RT_POST_PIPELINE   (RT_POST_TRIM_START | RT_POST_TRIM_END | RT_POST_DC | RT_POST_FADE_IN | \
                            RT_POST_FADE_OUT | RT_POST_JOIN | RT_POST_LOUDNESS | RT_POST_DECAY)

typedef struct rt_post_params {
    int32_t sample_rate;        /* BaseTTS.sample_rate                                   */
    float   silence_threshold;  /* 10^(silence_threshold_db/20), base_tts.py:367         */
    int32_t window;             /* int(sr*0.01)                base_tts.py:366           */
    int32_t fade;               /* int(sr*fade_duration_sec)   base_tts.py:420           */
    int32_t crossfade;          /* int(sr*crossfade_duration_sec) base_tts.py:455        */
    int32_t pause;              /* int(sr*inter_sentence_pause_sec) base_tts.py:519; 0 = none */
    int32_t loud_window;        /* int(sr*2.0)                 qwen.py:293               */
    int32_t trim_enabled;       /* BaseTTS.trim_silence                                  */
    double  target_rms_db;      /* -23.0   qwen.py:278                                   */
    double  max_gain_db;        /* 18.0    qwen.py:280                                   */
    double  max_amplitude;      /* 0.95    qwen.py:309                                   */
    double  decay_threshold;    /* BaseTTS.sound_decay_threshold                         */
    uint32_t stages;            /* RT_POST_* mask                                        */
    uint32_t reserved;
} rt_post_params;

typedef struct rt_post_stats {
    int64_t out_len;            /* samples written for the item                           */
    int64_t first_trim_start;   /* trim bounds of the item's first segment (leaf calls)   */
    int64_t first_trim_end;
    double  decay_ratio;        /* last-third RMS / first-third RMS (1.0 on the guards)   */
    double  rms_out;            /* RMS of the written samples                             */
    int32_t decay_ok;
    int32_t all_silent;         /* every segment was below the silence threshold          */
    int32_t fallback_concat;    /* reference's "direct concatenation" path was taken      */
    int32_t windowed_applied;   /* the 2-s windowed gain envelope was applied             */
} rt_post_stats;

/* Upper bound of the output length of an item (sum of segment lengths + pauses). */
RT_API int64_t rt_post_capacity(const rt_post_params* p, int32_t n_segments, const int64_t* h_seg_len);

/* Device-resident form: segment samples already in HBM (the vocoder's output).
 *   h_item_first_seg [n_items+1]  segments of item i are [first[i], first[i+1])
 *   h_seg_ptr        [n_seg]      device pointers to float32 samples
 *   h_seg_len        [n_seg]
 *   h_seg_trim       [n_seg] or NULL: per-segment RT_POST_TRIM_* override (ignored with RT_POST_JOIN
 *                    for items of >1 segment, where the position decides, base_tts.py:469-474)
 *   h_out_ptr        [n_items]    device pointers, capacity >= rt_post_capacity()
 *   h_stats          [n_items]    results (host)
 * Returns after the results are on the host. */
RT_API int rt_post_process(rt_ctx* ctx, const rt_post_params* p, int32_t n_items,
                           const int32_t* h_item_first_seg, const float* const* h_seg_ptr,
                           const int64_t* h_seg_len, const uint8_t* h_seg_trim,
                           float* const* h_out_ptr, const int64_t* h_out_cap, rt_post_stats* h_stats);

/* Host-buffer form (what the reference hands over: CPU float32 tensors, qwen.py:265).
 * Same arguments with host sample pointers; stages through HBM (PCIe-inclusive). */
RT_API int rt_post_process_host(rt_ctx* ctx, const rt_post_params* p, int32_t n_items,
                                const int32_t* h_item_first_seg, const float* const* h_seg_ptr,
                                const int64_t* h_seg_len, const uint8_t* h_seg_trim,
                                float* const* h_out_ptr, const int64_t* h_out_cap, rt_post_stats* h_stats);

/* float32 [-1,1] -> int16 PCM with the reference's truncating conversion (base_tts.py:664-666). */
RT_API int rt_pcm16(rt_ctx* ctx, const float* d_in, int64_t n, int16_t* d_out);

#ifdef __cplusplus
}
#endif
#endif /* RHO_TTS_AMD_H */
