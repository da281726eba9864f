// Model group, part 5 of 5: the codec decoder (rt_code2wav): codes -> code-embedding mean -> pre-transformer -> ConvNeXt upsampling
// -> SnakeBeta / transposed-conv / dilated residual stack -> waveform.  Every conv is an implicit GEMM of gemm.hip in split
// precision.  Host-side orchestration only.
#include "model_internal.h"

using namespace rtm;

extern "C" {

// --------------------------------------------------------------------------------------- code2wav
int64_t rt_wav_length(rt_model* m, int32_t n_frames) {
    if (!m || n_frames < 0) return -1;
    int64_t L = n_frames;
    for (int i = 0; i < m->cfg.n_upsampling; ++i) L *= m->cfg.upsampling_ratios[i];
    for (int i = 0; i < m->cfg.n_upsample_rates; ++i) L = (L - 1) * m->cfg.upsample_rates[i];
    return L > 0 ? L : 0;
}

int rt_code2wav(rt_model* m, int32_t n_items, int32_t t_max, const int32_t* h_codes, const int32_t* h_n_frames, float* d_wav,
                int64_t wav_stride, int64_t* h_wav_len) {
    if (!m || !h_codes || !h_n_frames || !d_wav || !h_wav_len) return rt_fail(m ? m->ctx : nullptr, RT_ERR_INVALID, "rt_code2wav: null argument");
    rt_ctx* ctx = m->ctx;
    CtxLock g(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!m->finalized) return rt_fail(ctx, RT_ERR_STATE, "rt_code2wav: model not finalized");
    const rt_model_config& c = m->cfg;
    const int B = n_items, T = t_max, Q = c.num_quantizers, Hc = c.codec_tf.hidden;
    if (B < 1 || B > c.max_batch || T < 1) return rt_fail(ctx, RT_ERR_INVALID, "rt_code2wav: n_items %d / t_max %d out of range", B, T);
    if (T > c.max_codec_frames) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length %d frames exceeds max_codec_frames %d", T, c.max_codec_frames);
    const int64_t L_out = rt_wav_length(m, T);
    if (L_out < 1) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length %d frames too short for the decoder", T);
    if (wav_stride < L_out) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: wav_stride %lld < length %lld", (long long)wav_stride, (long long)L_out);
    for (int b = 0; b < B; ++b)
        if (h_n_frames[b] < 0 || h_n_frames[b] > T) return rt_fail(ctx, RT_ERR_INVALID, "rt_code2wav: n_frames[%d] out of range", b);
    pool_release_all(m);
    const int64_t rows0 = (int64_t)B * T;
    int32_t *d_codes, *d_slot, *d_pos;
    RT_TRY(pool_arr(m, (size_t)rows0 * Q, &d_codes));
    RT_TRY(pool_arr(m, rows0, &d_slot));
    RT_TRY(pool_arr(m, rows0, &d_pos));
    RT_HIP(ctx, hipMemcpyAsync(d_codes, h_codes, (size_t)rows0 * Q * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        std::vector<int32_t> sl(rows0), ps(rows0);
        for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) { sl[(size_t)b * T + t] = b; ps[(size_t)b * T + t] = t; }
        RT_HIP(ctx, hipMemcpyAsync(d_slot, sl.data(), rows0 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(d_pos, ps.data(), rows0 * 4, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    }
    float* h = nullptr;
    RT_TRY(pool_arr(m, (size_t)rows0 * Hc, &h));
    RT_TRY(launch_code_embed_mean(ctx, TBL(m, "codec.code_embedding"), c.codebook_size, Q, Hc, d_codes, rows0, h));
    float* hn = nullptr;
    RT_TRY(pool_arr(m, (size_t)rows0 * Hc, &hn));
    {
        StackWs w;
        RT_TRY(alloc_stack_ws(m, c.codec_tf, (int)rows0, &w, true));
        RT_TRY(stack_forward(m, m->ctf, w, h, (int)rows0, d_slot, d_pos, 0, nullptr, hn));
    }
    // ---- ConvNeXt upsampling stages (transposed conv k = stride = r is a plain GEMM in channels-last).
    // Every conv-as-GEMM below runs with split (hi + lo) activations: see k_gemm_tiled.
    float* cur = hn;
    int64_t Tc = T;
    for (int i = 0; i < c.n_upsampling; ++i) {
        const std::string u = "codec.up" + std::to_string(i);
        const int r = c.upsampling_ratios[i];
        float* up = nullptr;
        RT_TRY(pool_arr(m, (size_t)B * Tc * r * Hc, &up));
        GemmA a; a.ptr = cur; a.is_f32 = 1; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = Hc;
        GemmEpi e; e.bias = VEC(m, u + ".tconv_b"); e.out_f32 = up; e.ldc = (int64_t)r * Hc;
        RT_TRY(launch_gemm(ctx, a, PW(m, u + ".tconv"), e));
        Tc *= r;
        const int64_t rows = (int64_t)B * Tc;
        float *ln = nullptr, *mid = nullptr;
        RT_TRY(pool_arr(m, (size_t)rows * Hc, &ln));
        RT_TRY(pool_arr(m, (size_t)rows * 4 * Hc, &mid));
        RT_TRY(launch_dwconv_ln(ctx, up, B, (int)Tc, Hc, VEC(m, u + ".dw_w"), VEC(m, u + ".dw_b"), VEC(m, u + ".ln_w"), VEC(m, u + ".ln_b"), 1e-6f, ln));
        GemmA a1; a1.ptr = ln; a1.is_f32 = 1; a1.split = 1; a1.M = rows; a1.Cin = Hc;
        GemmEpi e1; e1.bias = VEC(m, u + ".pw1_b"); e1.act = ACT_GELU; e1.out_f32 = mid; e1.ldc = 4 * (int64_t)Hc;
        RT_TRY(launch_gemm(ctx, a1, PW(m, u + ".pw1"), e1));
        GemmA a2; a2.ptr = mid; a2.is_f32 = 1; a2.split = 1; a2.M = rows; a2.Cin = 4 * Hc;
        GemmEpi e2; e2.bias = VEC(m, u + ".pw2_b"); e2.scale = VEC(m, u + ".gamma"); e2.residual = up; e2.out_f32 = up; e2.ldc = Hc;
        RT_TRY(launch_gemm(ctx, a2, PW(m, u + ".pw2"), e2));
        cur = up;
    }
    // ---- decoder: conv k7 -> [SnakeBeta, transposed conv, 3 residual units] x n -> SnakeBeta -> conv k7 -> clamp
    // The snake-activated operands (s_in, s1, s2) are kept as hi + lo bf16 planes written by the producing epilogue: the same
    // 4 bytes per element as f32, but the split is done once per element instead of once per consuming workgroup and tap.
    struct Planes { bf16_t* hi = nullptr; bf16_t* lo = nullptr; };
    auto planes = [&](size_t n, Planes* p) -> int {
        RT_TRY(pool_arr(m, n, &p->hi));
        RT_TRY(pool_arr(m, n, &p->lo));
        return RT_OK;
    };
    Planes s_in;  // snake-activated input of the next transposed conv
    {
        RT_TRY(planes((size_t)B * Tc * m->dec_ch[0], &s_in));
        GemmA a; a.ptr = cur; a.is_f32 = 1; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = Hc; a.taps = 7; a.tap_stride = 1; a.tap_offset = -6;
        a.rows_out = (int)Tc; a.rows_in = (int)Tc;
        GemmEpi e; e.bias = VEC(m, "codec.dec0_b"); e.out2_hi = s_in.hi; e.out2_lo = s_in.lo;
        e.snake2_a = VEC(m, "codec.b0.sa"); e.snake2_ib = VEC(m, "codec.b0.sib");
        e.ldc = m->dec_ch[0];
        RT_TRY(launch_gemm(ctx, a, PW(m, "codec.dec0"), e));
    }
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        const std::string bn = "codec.b" + std::to_string(i);
        const int cin = m->dec_ch[i], cout = m->dec_ch[i + 1], r = c.upsample_rates[i];
        const int64_t To = (Tc - 1) * r;
        if (To < 1) return rt_fail(ctx, RT_ERR_LENGTH, "rt_code2wav: length collapsed in decoder block %d", i);
        const int64_t rows = (int64_t)B * To;
        float* xr = nullptr;
        Planes s1, s2;
        RT_TRY(pool_arr(m, (size_t)rows * cout, &xr));
        RT_TRY(planes((size_t)rows * cout, &s1));
        RT_TRY(planes((size_t)rows * cout, &s2));
        {
            // transposed conv k = 2r, stride r, r samples trimmed on both sides: out[m*r + j] = x[m+1] W[j] + x[m] W[j + r]
            GemmA a; a.ptr = s_in.hi; a.ptr_lo = s_in.lo; a.split = 1; a.M = (int64_t)B * (Tc - 1); a.Cin = cin; a.taps = 2; a.tap_stride = 1; a.tap_offset = 0;
            a.rows_out = (int)(Tc - 1); a.rows_in = (int)Tc;
            GemmEpi e; e.bias = VEC(m, bn + ".tconv_b"); e.out_f32 = xr; e.out2_hi = s1.hi; e.out2_lo = s1.lo;
            e.snake2_a = m->xvec[bn + ".u0.a1"]; e.snake2_ib = m->xvec[bn + ".u0.ib1"];
            e.ldc = (int64_t)r * cout;
            RT_TRY(launch_gemm(ctx, a, PW(m, bn + ".tconv"), e));
        }
        for (int j = 0; j < 3; ++j) {
            const std::string u = bn + ".u" + std::to_string(j);
            const int dil = j == 0 ? 1 : (j == 1 ? 3 : 9);
            GemmA a; a.ptr = s1.hi; a.ptr_lo = s1.lo; a.split = 1; a.M = rows; a.Cin = cout; a.taps = 7; a.tap_stride = dil; a.tap_offset = -6 * dil;
            a.rows_out = (int)To; a.rows_in = (int)To;
            GemmEpi e; e.bias = VEC(m, u + ".c1_b"); e.act = ACT_SNAKE; e.snake_a = VEC(m, u + ".a2"); e.snake_ib = VEC(m, u + ".ib2");
            e.out_hi = s2.hi; e.out_lo = s2.lo; e.ldc = cout;
            GemmA a2; a2.ptr = s2.hi; a2.ptr_lo = s2.lo; a2.split = 1; a2.M = rows; a2.Cin = cout;
            GemmEpi e2; e2.bias = VEC(m, u + ".c2_b"); e2.residual = xr; e2.out_f32 = xr; e2.out2_hi = s1.hi; e2.out2_lo = s1.lo; e2.ldc = cout;
            if (j < 2) { e2.snake2_a = VEC(m, bn + ".u" + std::to_string(j + 1) + ".a1"); e2.snake2_ib = VEC(m, bn + ".u" + std::to_string(j + 1) + ".ib1"); }
            else if (i + 1 < c.n_upsample_rates) { e2.snake2_a = VEC(m, "codec.b" + std::to_string(i + 1) + ".sa"); e2.snake2_ib = VEC(m, "codec.b" + std::to_string(i + 1) + ".sib"); }
            else { e2.snake2_a = VEC(m, "codec.fin_a"); e2.snake2_ib = VEC(m, "codec.fin_ib"); }
            // NOTE: the fused form writes s1 (the NEXT unit's operand planes) while other workgroups still read s1 as THIS unit's
            // input window, so it needs a second pair of planes to write to: s1 and s2 swap roles from unit to unit
            if (conv_pair_fusable(a, PW(m, u + ".c1"), e, PW(m, u + ".c2"))) {
                e2.out2_hi = s2.hi; e2.out2_lo = s2.lo;
                RT_TRY(launch_gemm(ctx, a, PW(m, u + ".c1"), e, &PW(m, u + ".c2"), &e2));
                std::swap(s1, s2);
            } else {
                RT_TRY(launch_gemm(ctx, a, PW(m, u + ".c1"), e));
                RT_TRY(launch_gemm(ctx, a2, PW(m, u + ".c2"), e2));
            }
        }
        s_in = s1;
        Tc = To;
    }
    float* wav_tmp = nullptr;
    RT_TRY(pool_arr(m, (size_t)B * Tc, &wav_tmp));
    {
        // last conv: channels -> 1, k = 7, causal, then clamp(-1, 1): the 7 x C taps of one output sample are contiguous
        // in the channels-last buffer, so it is the same implicit GEMM with a single output column
        const int cl = m->dec_ch.back();
        GemmA a; a.ptr = s_in.hi; a.ptr_lo = s_in.lo; a.split = 1; a.M = (int64_t)B * Tc; a.Cin = cl; a.taps = 7; a.tap_stride = 1; a.tap_offset = -6;
        a.rows_out = (int)Tc; a.rows_in = (int)Tc;
        if (launch_final_conv_ok(cl) && g_final_conv) {
            RT_TRY(launch_final_conv(ctx, s_in.hi, s_in.lo, B, (int)Tc, cl, VEC(m, "codec.fin_wv"), VEC(m, "codec.fin_b"), wav_tmp));
        } else {
            GemmEpi e; e.bias = VEC(m, "codec.fin_b"); e.act = ACT_CLAMP1; e.out_f32 = wav_tmp; e.ldc = 1;
            RT_TRY(launch_gemm(ctx, a, PW(m, "codec.fin_w"), e));
        }
    }
    RT_HIP(ctx, hipMemcpy2DAsync(d_wav, (size_t)wav_stride * 4, wav_tmp, (size_t)Tc * 4, (size_t)Tc * 4, B, hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < B; ++b) h_wav_len[b] = rt_wav_length(m, h_n_frames[b]);
    pool_release_all(m);
    return RT_OK;
}
}  // extern "C"
