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

#define RT_ABI_VERSION 6

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

/* Sub-segment streaming (an EXTENSION - BaseTTS.stream yields whole segments, base_tts.py:1132-1190): the
 * per-segment leaves of stream() (base_tts.py:1170-1176: _post_process_audio, _trim_silence, _remove_dc_offset, _apply_fades)
 * for ONE chunk of a segment that is still being decoded.  The two segment-wide quantities are taken from the segment's first
 * chunk and carried: h_dc_gain[0] = DC offset, h_dc_gain[1] = linear gain to the target RMS (in/out).
 *   RT_STREAM_MEASURE     first chunk: measure gain (target RMS / RMS of the raw chunk's AUDIBLE span - first to last 10-ms frame above
 *                         the silence threshold - held to +-30 dB; over the WHOLE chunk, unclamped, when the chunk is also the last: a
 *                         segment handed over whole is levelled as the reference levels it) and dc (after the trim) and return
 *                         them; else apply as given.
 *                         A chunk with no audible frame measures nothing and returns gain 0: pass RT_STREAM_MEASURE (with the
 *                         first-chunk trim and fade flags) again with the next chunk and drop this one's output
 *   RT_STREAM_TRIM_START  drop leading silence (first chunk)     RT_STREAM_TRIM_END  drop trailing silence (last chunk)
 *   RT_STREAM_FADE_IN / _OUT  raised-cosine ramps of p->fade samples at the chunk's start / end (skipped below 2 x fade samples)
 * y = 0.95 tanh(x gain / 0.95) - dc; the 2-s windowed decay correction needs the whole segment and is not applied.
 * d_out: capacity n samples; *h_out_len = samples written. */
#define RT_STREAM_MEASURE    0x01u
#define RT_STREAM_TRIM_START 0x02u
#define RT_STREAM_TRIM_END   0x04u
#define RT_STREAM_FADE_IN    0x08u
#define RT_STREAM_FADE_OUT   0x10u
RT_API int rt_stream_chunk(rt_ctx* ctx, const rt_post_params* p, const float* d_in, int64_t n, uint32_t flags, double* h_dc_gain,
                           float* d_out, int64_t* h_out_len);

/* ------------------------------------------------------------------ model
 * Stands behind the third-party model object the reference drives
 * (providers/qwen.py:160-165 from_pretrained, :247-251 generate_custom_voice,
 * :253-258 generate_voice_clone): talker decode + residual-code predictor +
 * codec decoder, shape-parametrised (the real dimensions are UNVERIFIED offline).
 */
typedef struct rt_stack_dims {
    int32_t hidden, layers, heads, kv_heads, head_dim, inter;
    float rope_theta, rms_eps;
} rt_stack_dims;

/* Conditioning front-end (reference audio -> codes + speaker embedding): a causal conv encoder (filters x 2 per stage,
 * strides ratios[]), a transformer at twice the frame rate, a stride-2 conv and a split residual vector quantiser.
 * filters = 0: the model has no encoder (rt_voice_encode / rt_model_set_voice_pcm return RT_ERR_UNSUPPORTED). */
typedef struct rt_encoder_config {
    int32_t filters, n_ratios, ratios[8], kernel, res_kernel, last_kernel;
    rt_stack_dims tf;             /* hidden = encoder width                                  */
    int32_t window, vq_dim, spk_hidden;
    int32_t max_ref_frames;       /* longest reference clip, in codec frames                  */
} rt_encoder_config;

typedef struct rt_model_config {
    rt_stack_dims talker, predictor, codec_tf;
    int32_t codec_vocab, predictor_vocab, text_vocab, text_hidden, n_groups;
    int32_t codebook_size, num_quantizers, codec_sliding_window;
    int32_t n_upsampling;         /* ConvNeXt upsample stages            */
    int32_t upsampling_ratios[4];
    int32_t n_upsample_rates;     /* decoder blocks                      */
    int32_t upsample_rates[8];
    int32_t decoder_dim;
    int32_t codec_eos_id;
    int32_t max_batch;            /* sequences decoded together          */
    int32_t max_positions;        /* KV rows per sequence (prompt + frames) */
    int32_t max_codec_frames;     /* frames per sequence one rt_code2wav call may decode */
    int32_t reserved[4];
    rt_encoder_config enc;
} rt_model_config;

#define RT_DTYPE_BF16 0
#define RT_DTYPE_F32  1

typedef struct rt_sampling {
    int32_t do_sample;            /* 0: greedy (lowest index on ties)    */
    float   temperature;
    int32_t top_k;                /* 1..64 when sampling                 */
    float   top_p;
    float   repetition_penalty;   /* talker group 0 only                 */
} rt_sampling;

RT_API int rt_model_create(rt_ctx* ctx, const rt_model_config* cfg, rt_model** out_model);
RT_API int rt_model_destroy(rt_model* m);
/* Number of named tensors the model expects, and the name / shape of the i-th (INTEGRATION.md lists them). */
RT_API int rt_model_tensor_count(rt_model* m);
RT_API int rt_model_tensor_info(rt_model* m, int32_t index, char* name, size_t name_cap, int64_t* shape2, int32_t* kind);
/* Upload one tensor (matrices [N][K] row-major bf16; vectors bf16 or f32).  on_device != 0: data is an HBM pointer. */
RT_API int rt_model_set_tensor(rt_model* m, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols,
                               int32_t on_device);
/* After the last tensor: verifies completeness, uploads RoPE tables (h_* are [max_positions][head_dim/2] float32,
 * one pair per stack: talker, predictor, codec_tf) and derives the projected predictor embedding tables. */
RT_API int rt_model_finalize(rt_model* m, const float* h_rope_cos[3], const float* h_rope_sin[3]);
RT_API int64_t rt_model_weight_bytes(rt_model* m);

/* Voice conditioning = the shared prompt prefix, computed ONCE per voice (the reference re-derives it on every
 * call by passing ref_audio=path each time, qwen.py:253-258).  Row recipe of the prefix, one entry per row:
 *   h_text_ids[r]   text token of the row, or -1 for tts_pad
 *   h_codec_ids[r*n_groups + g]  codec-side ids summed into the row (g = 0: talker table / control ids;
 *                   g >= 1: residual codebook g), -1 = none
 *   h_speaker_row   row that additionally receives h_speaker_embed[hidden] (float32), or -1
 */
RT_API int rt_model_set_voice(rt_model* m, int32_t n_rows, const int32_t* h_text_ids, const int32_t* h_codec_ids,
                              int32_t h_speaker_row, const float* h_speaker_embed);
/* The step BEFORE the path: the reference hands the model a reference-audio PATH on every call
 * (ref_audio=..., providers/qwen.py:253-258) and the model encodes it each time; here the clip is encoded once, on the GPU.
 *   h_pcm       mono float32 in [-1, 1] at the model's sample rate (host memory); n_samples is cut down to whole codec frames
 *   h_codes     out [n_frames][num_quantizers] (capacity max_frames rows), h_n_frames out
 *   h_speaker_embed  out [talker hidden] float32 (statistics-pooled speaker head), may be NULL */
RT_API int rt_voice_encode(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t* h_codes, int32_t max_frames, int32_t* h_n_frames,
                           float* h_speaker_embed);
/* rt_voice_encode + rt_model_set_voice in one call: the recipe arrays describe the rows BEFORE the reference frames (role /
 * control / speaker / bos / reference-text rows, as for rt_model_set_voice); one row per encoded frame is appended, with text id
 * frame_text_id (tts_pad) and the frame's codes; the speaker row receives the encoder's speaker embedding.  Needs
 * num_quantizers == n_groups.  h_codes / h_n_frames (optional) return what was encoded. */
RT_API int rt_model_set_voice_pcm(rt_model* m, const float* h_pcm, int64_t n_samples, int32_t n_head_rows, const int32_t* h_text_ids,
                                  const int32_t* h_codec_ids, int32_t h_speaker_row, int32_t frame_text_id, int32_t max_ref_frames,
                                  int32_t* h_codes, int32_t* h_n_frames);
RT_API int32_t rt_voice_prefix_len(rt_model* m);
/* Prefix KV as one HBM blob [2][layers][kv_heads][prefix_len][head_dim] bf16, for an RCCL broadcast. */
RT_API int64_t rt_voice_blob_bytes(rt_model* m);
RT_API int rt_voice_export(rt_model* m, void* d_blob, int64_t bytes);
RT_API int rt_voice_import(rt_model* m, int32_t prefix_len, const void* d_blob, int64_t bytes);

/* n_items may exceed the model's max_batch: the first max_batch items start on the decode rows, the others wait in a queue
 * (in array order - put the longest first) and take over a row as soon as the host has seen its item finish: the new item's
 * prompt suffix is prefilled into the row's KV slot between two frames (continuous batching).  An item's codes depend only
 * on (h_item_ids[i], seed), not on the row, the moment it ran or what it was batched with (bit for bit: tested at the 1.7B and
 * 0.6B shapes, first waves of fewer than 64, of 416 and of more than 1024 prompt rows - the prompt prefill runs in chunks of
 * <= 1024 rows on one kernel whose K association equals the <= 64-row kernel's; tests/test_model_shapes_gpu.py).  Teacher forcing and the logit traces need
 * n_items <= max_batch. */
typedef struct rt_generate_args {
    int32_t n_items;
    const int32_t* h_text_ids;      /* concatenated target-text token ids                          */
    const int32_t* h_text_offsets;  /* [n_items + 1]                                               */
    const int32_t* h_max_frames;    /* [n_items]                                                   */
    const int64_t* h_item_ids;      /* [n_items] RNG stream of each item (its global index)        */
    uint64_t seed;                  /* BaseTTS.seed (base_tts.py:45,142-149)                       */
    rt_sampling talker, predictor;
    int32_t ignore_eos;             /* synthetic weights never emit EOS: length = max_frames       */
    int32_t min_frames;
    int32_t tts_eos_id, tts_pad_id, codec_pad_id, codec_bos_id;
    const int32_t* h_forced_codes;  /* optional teacher forcing: item i frames at offset forced_offsets[i], [frames][n_groups] */
    const int32_t* h_forced_offsets;/* [n_items + 1], in frames                                    */
    const volatile int32_t* h_cancel_flag; /* polled between frames (CancellationToken, cancellation.py:19-65) */
    int32_t* h_codes;               /* out: item i at (sum_{j<i} max_frames[j]) * n_groups, [frames][n_groups] */
    int32_t* h_n_frames;            /* out: [n_items]                                              */
    float* d_trace_talker;          /* optional: [max_frames_max][n_items][codec_vocab] talker logits */
    float* d_trace_predictor;       /* optional: [max_frames_max][n_groups-1][n_items][predictor_vocab] */
    int32_t max_rows;               /* 0: decode on min(n_items, max_batch) rows; else on at most this many (a short queue is
                                       served faster by 32 busy rows than by 64 half-empty ones: a 64-row frame costs ~1.4x a
                                       32-row one)                                                                           */
} rt_generate_args;

RT_API int rt_generate(rt_model* m, const rt_generate_args* args);
/* The same generation in pieces (sub-segment streaming, SURVEY.md 8f-4: BaseTTS.stream, base_tts.py:1132-1190, hands audio over
 * per segment; here the first codec frames of a segment can be vocoded while the rest is still being decoded).
 *   rt_generate_begin   validates, prefills the prompts, builds the decode state; args->h_codes / h_n_frames may be NULL; every
 *                       input array is copied (only h_cancel_flag must stay valid until rt_generate_end).  A run that was
 *                       begun and never ended is dropped.
 *   rt_generate_step    runs up to n_frames more frames and returns once the host holds their codes; the next talker step is
 *                       already in flight on the stream.  *h_frames_run = frames run so far, *h_all_done = every item ended.
 *   rt_generate_peek    codes of `item` produced so far: frames [first_frame, first_frame + *h_n_frames) into h_codes
 *                       [max_frames][n_groups]; *h_finished = the item has ended.
 *   rt_generate_end     writes the results as rt_generate does (either pointer may be NULL) and releases the run; ending a run
 *                       that has not finished drops it (an error if results were asked for).
 * Between two steps other calls on the model are allowed (rt_code2wav of the frames already decoded); a voice change is not.
 * rt_generate == begin + one step of every frame + end: the codes are the same however the frames are cut into steps. */
RT_API int rt_generate_begin(rt_model* m, const rt_generate_args* args);
RT_API int rt_generate_step(rt_model* m, int32_t n_frames, int32_t* h_frames_run, int32_t* h_all_done);
RT_API int rt_generate_peek(rt_model* m, int32_t item, int32_t first_frame, int32_t max_frames, int32_t* h_codes, int32_t* h_n_frames,
                            int32_t* h_finished);
RT_API int rt_generate_end(rt_model* m, int32_t* h_codes, int32_t* h_n_frames);
/* Figures of the last rt_generate: decode frames launched, rows, frames kept over all items (row occupancy =
 * frames_kept / (frames_run * rows)) and the number of row hand-overs to queued items.  Any pointer may be NULL. */
RT_API int rt_generate_stats(rt_model* m, int64_t* frames_run, int64_t* rows, int64_t* frames_kept, int64_t* hand_overs);

/* Codec decoder: h_codes [n_items][t_max][num_quantizers] (right-padded), h_n_frames [n_items];
 * d_wav [n_items][wav_stride] float32 in HBM, h_wav_len out.  rt_wav_length(frames) = samples produced. */
RT_API int64_t rt_wav_length(rt_model* m, int32_t n_frames);
RT_API int rt_code2wav(rt_model* m, int32_t n_items, int32_t t_max, const int32_t* h_codes, const int32_t* h_n_frames,
                       float* d_wav, int64_t wav_stride, int64_t* h_wav_len);

/* ------------------------------------------------------------------ speech-to-text for validation
 * Stands behind validation/stt/stt_validator.py:42-148 (_get_whisper_model / transcribe_audio: faster-whisper "tiny" on the CPU,
 * or transformers' Whisper - :85-107) and the temporary-WAV round trip of base_tts.py:821-830: a generated segment is
 * transcribed where it already lives, in HBM.  Whisper-shaped encoder-decoder, shape-parametrised:
 *   PCM (any rate) -> windowed-sinc resampler to `sample_rate` -> log-mel (n_fft-point DFT, hop, n_mels slaney filters, log10,
 *   (max - 8) floor, (x + 4) / 4; padded to chunk_seconds) -> conv k3 + conv k3 stride 2 (GELU) + positions -> enc_layers pre-LN
 *   layers -> dec_layers decoder layers (causal self-attention, cross-attention) -> tied LM head -> greedy ids after the forced
 *   prefix.  Tensor names and layouts: rt_stt_tensor_info; INTEGRATION.md lists them. */
typedef struct rt_stt rt_stt;
typedef struct rt_stt_config {
    int32_t d_model, heads, ffn, enc_layers, dec_layers;
    int32_t n_mels, n_ctx, n_text_ctx, vocab;          /* n_ctx: encoder positions (1500); n_text_ctx: decoder positions (448) */
    int32_t n_fft, hop, sample_rate, chunk_seconds;    /* 400, 160, 16000, 30: chunk_seconds * sample_rate / hop == 2 * n_ctx       */
    int32_t eos_id;
    int32_t n_prefix, prefix[8];                       /* forced decoder prefix (start-of-transcript, language, task, no-timestamps) */
    int32_t suppress_from;                             /* ids >= suppress_from are never produced, eos_id excepted (0: none)         */
    int32_t n_begin_suppress, begin_suppress[4];       /* ids not allowed as the FIRST generated token (space, end-of-sequence)      */
    int32_t max_new_tokens;
    int32_t reserved[4];
} rt_stt_config;
RT_API int rt_stt_create(rt_ctx* ctx, const rt_stt_config* cfg, rt_stt** out);
RT_API int rt_stt_destroy(rt_stt* s);
RT_API int rt_stt_tensor_count(rt_stt* s);
RT_API int rt_stt_tensor_info(rt_stt* s, int32_t index, char* name, size_t name_cap, int64_t* shape2, int32_t* kind);
RT_API int rt_stt_set_tensor(rt_stt* s, const char* name, const void* data, int32_t dtype, int64_t rows, int64_t cols, int32_t on_device);
RT_API int rt_stt_finalize(rt_stt* s);
/* Ids that are never produced, besides [suppress_from, vocab): the `suppress_tokens` list of a checkpoint's generation config.
 * Before rt_stt_finalize. */
RT_API int rt_stt_set_suppress(rt_stt* s, const int32_t* h_ids, int32_t n);
/* d_pcm: mono float32 in HBM at `sample_rate_in` (the TTS output as it stands).  h_tokens receives the ids generated after the
 * prefix, end-of-sequence excluded.  Audio longer than one chunk (chunk_seconds, 30 s) is NOT truncated: it is transcribed in
 * consecutive chunk_seconds windows (features, encoder, greedy decode behind the forced prefix per window, at most
 * cfg.max_new_tokens ids each) and the ids are concatenated, at most max_tokens in all - size h_tokens for
 * ceil(seconds / chunk_seconds) * cfg.max_new_tokens.  d_first_logits (optional, HBM, [vocab]): the logits behind the forced
 * prefix of the FIRST window, for tests. */
RT_API int rt_stt_transcribe(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate_in, int32_t* h_tokens, int32_t max_tokens,
                             int32_t* h_n_tokens, float* d_first_logits);
/* Stages on their own (tests): the log-mel features [2 n_ctx][n_mels] and the encoder states [n_ctx][d_model], float32 in HBM. */
RT_API int rt_stt_log_mel(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate_in, float* d_mel);
RT_API int rt_stt_encode(rt_stt* s, const float* d_pcm, int64_t n_samples, int32_t sample_rate_in, float* d_states);

/* ------------------------------------------------------------------ drift-classifier features
 * Stands behind the front half of validation/classifier/trainer.py:23-96 (extract_features, _estimate_formants: librosa on a
 * temporary WAV): the per-sample work of the 30 hand-crafted dimensions, on the waveform where it lives.
 *   d_pcm: mono float32 in HBM at sample_rate_in; resampled to 16 kHz on the device (the speech-to-text front-end's resampler).
 *   h_mfcc_stats26: mean (13) then population standard deviation (13) over the frames of the 13 MFCCs (n_fft 2048, hop 512,
 *     zero-padded centre, periodic Hann, 128 slaney mel filters, power_to_db with top_db 80, orthonormal DCT-II).
 *   h_cmnd [n_frames][max_period - min_period + 1] float64 (optional, room for cmnd_cap_frames frames): the cumulative-mean-
 *     normalised difference function of probabilistic YIN per pitch frame (frame 2048, window 1024, hop 512, zero-padded
 *     centre) for the lags min_period .. max_period (rt_features_geometry); the trough statistics and the Viterbi pass that
 *     turn it into F0 are host arithmetic (rho_tts_amd/features.py).
 *   h_lpc [lpc_order + 1] float64: Burg LPC of the pre-emphasised (0.97), symmetric-Hann-windowed 25-ms frame about the middle
 *     sample; the formants are the angles of its roots.
 * Frames: 1 + n16 / 512 for both (n16 = samples at 16 kHz).  Parity with librosa is UNPINNED (not installable offline). */
typedef struct rt_features rt_features;
RT_API int rt_features_create(rt_ctx* ctx, rt_features** out);
RT_API int rt_features_destroy(rt_features* f);
/* min / max period of probabilistic YIN for a search range [fmin, fmax] Hz at the rate the analysis ASSUMES (the reference
 * leaves librosa.pyin's default 22050 in place for 16-kHz audio, trainer.py:52): floor(sr / fmax), min(ceil(sr / fmin), 1023). */
RT_API int rt_features_geometry(int32_t pitch_sr, double fmin, double fmax, int32_t* min_period, int32_t* max_period);
RT_API int rt_features_extract(rt_features* f, const float* d_pcm, int64_t n_samples, int32_t sample_rate_in, int32_t min_period,
                               int32_t max_period, int32_t lpc_order, double* h_mfcc_stats26, int32_t* h_n_mfcc_frames, double* h_cmnd,
                               int32_t cmnd_cap_frames, int32_t* h_n_pitch_frames, double* h_lpc);

/* Measurement and test entry points (rt_profile_*, rt_debug_*, rt_bench_*) are declared in rho_tts_amd_debug.h: a host binding of
 * the generation path needs none of them. */

#ifdef __cplusplus
}
#endif
#endif /* RHO_TTS_AMD_H */
