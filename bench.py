#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per wall-second of the batched generation hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
Qwen3-TTS-1.7B-shaped model, bf16 weights, batch 32 per GPU, 30-s reference-audio clone,
10-word sentences (fixed 44 frames each: synthetic weights never emit EOS — SURVEY.md 8d).
One step = one pass of the hot path over one batch, everything `_run_pipeline` does after the
host-side text handling: conditioning (reference clip through the audio encoder, voice prefix
through the talker) -> batched autoregressive decode
-> codec decoder -> fused post-processing -> waveforms on the host.  Inputs (weights, reference
conditioning arrays, token ids) are resident before the timed region.  Data is synthetic and
weights are seeded random of the named architecture (no checkpoints offline).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling — every rank
decodes its own 32 texts; rank 0 computes the voice prefix and broadcasts its KV blob, and the
finished waveforms are gathered to rank 0, both inside the timed step.

`--corpus N` switches to BASELINE.json configs[3] (C4): ONE corpus of N ragged texts (seed 789,
U{6..24} words) for the whole job — strong scaling.  The corpus is dealt over the ranks by
estimated frames (dist.plan_corpus); every rank decodes its share on its `--batch` rows with
continuous batching (ONE rt_generate call: texts queued longest first, a finished row is handed to
the next queued text), vocodes in batches sorted by the lengths produced, post-processes, and the
waveforms are gathered to rank 0 and put back in corpus order (dist.unshard).  The line also
reports the measured row occupancy (`extra.padding_efficiency` = frames kept / (frames launched x
rows)).  `--static-batches`: length-bucketed static batches instead (dist.bucket_batches).
`--length-error e`: every text ends at estimate x (1 + U(-e, e)) frames while the schedule is
planned on the estimates - what a real checkpoint's end-of-sequence does to a planner.

`value` counts DELIVERED audio: the samples of the post-processed waveforms the caller receives
(after silence trim), not the vocoder's raw output.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_PROFILE = "r04_pmc_fetch_bench_step.json"      # counter pass over this bench's own step (tools/pmc_step_split.py), keyed on the library's hash

WORDS = ("time year people way day man thing woman life child world school state family student group country problem hand "
         "part place case week company system program question work government number night point home water room mother area "
         "money story fact month lot right study book eye job word business issue side kind head house service friend father "
         "power hour game line end member law car city community name president team minute idea kid body information back "
         "parent face others level office door health person art war history party result change morning reason research girl "
         "guy moment air teacher force education").split()


def sentences(n: int, n_words, seed: int):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = n_words if isinstance(n_words, int) else int(rng.integers(n_words[0], n_words[1] + 1))
        out.append(" ".join(WORDS[int(j)] for j in rng.integers(0, len(WORDS), k)).capitalize() + ".")
    return out


_T0 = time.perf_counter()


def log(msg: str) -> None:
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def launch_plan(gpus: int, env, n_devices: int, backend: str = "nccl"):
    """What `python bench.py --gpus N` has to do, decided from the flags and the environment alone (no GPU call):

      ("run", world)   this process is one rank of `world` (world == gpus; 1 = the plain single-process run)
      ("spawn", gpus)  no launcher started us: start `torch.distributed.run --nproc-per-node gpus` as a CHILD process
      ("fail", why)    the request cannot be honoured - exit non-zero rather than report a number for fewer GPUs

    The multi-GPU number is only ever printed by a job whose RCCL group has `--gpus` ranks (`--gpus` used to be
    parsed and ignored, so `python bench.py --gpus 8` printed the one-GPU figure)."""
    if gpus < 1:
        return "fail", f"--gpus {gpus}: need at least one GPU"
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        try:
            world = int(ws)
        except ValueError:
            return "fail", f"WORLD_SIZE={ws!r} is not a number"
        if world != gpus:
            return "fail", (f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks: pass --gpus {world} "
                            f"(or start `python bench.py --gpus {gpus}` without a launcher and let it start its own ranks)")
        local_world = int(env.get("LOCAL_WORLD_SIZE", world))
        if backend == "nccl" and n_devices < local_world:
            return "fail", f"{local_world} ranks on this node but only {n_devices} GPU(s) visible (one rank per GPU over RCCL)"
        return "run", world
    if gpus == 1:
        return "run", 1
    if backend == "nccl" and n_devices < gpus:
        return "fail", f"--gpus {gpus} but only {n_devices} GPU(s) visible on this node (one rank per GPU over RCCL)"
    return "spawn", gpus


def spawn_ranks(gpus: int, argv) -> int:
    """Start the N ranks as a child `torch.distributed.run` (never os.exec*: this process may already hold a HIP runtime) on
    127.0.0.1 and a free port; rank 0's JSON line goes to our stdout unchanged.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    log(f"--gpus {gpus} without a launcher: starting {gpus} ranks as a child process: {' '.join(cmd[1:8])} bench.py ...")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="1.7b", help="config preset: 1.7b | 0.6b | small | tiny")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--words", type=int, default=10)
    ap.add_argument("--ref-seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--phase-times", action="store_true", help="wait for the GPU after every phase of a step and report the wall ms per phase in extra.phase_ms "
                    "(a diagnostic: the waits cost a little throughput)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: probe 8, 16, 32, ... host cores in ascending order while more threads still pay, take the fastest)")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--corpus", type=int, default=0, help="C4: one corpus of this many ragged texts for the whole job (strong scaling)")
    ap.add_argument("--length-error", type=float, default=0.0, help="corpus mode: every text ends at estimate x (1 + U(-e, e)) frames - the schedule is "
                    "planned on the estimates, as with a real checkpoint whose end-of-sequence the host cannot know (synthetic weights never emit it)")
    ap.add_argument("--static-batches", action="store_true", help="corpus mode: length-bucketed static batches instead of continuous batching")
    ap.add_argument("--eos-live", action="store_true", help="decode with end-of-sequence live (as a real checkpoint does) instead of fixed lengths")
    ap.add_argument("--tune", default="", help="comma-separated rt_debug_tune codes (100/101 legacy/column decode, 200/201 eager/graph)")
    args = ap.parse_args()

    # ---- how many ranks?  decided before anything touches the GPU (torch.cuda.device_count() does not initialise it here)
    what, detail = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), args.backend)
    if what == "fail":
        print(f"bench.py: {detail}", file=sys.stderr, flush=True)
        sys.exit(2)
    if what == "spawn":
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, (world, args.gpus)
    dist = None
    # (RHO_TTS_AMD_FORCE_DIST=1: take the collective path with ONE rank too - how the RCCL call sites are exercised on a one-GPU box)
    if world > 1 or os.environ.get("RHO_TTS_AMD_FORCE_DIST", "") not in ("", "0"):
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        if args.backend == "nccl":
            dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist_mod.init_process_group(backend=args.backend, rank=rank, world_size=world)
            local_rank = local_rank % max(1, n_dev)           # rehearsal: several ranks may share a GPU
        dist = dist_mod
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    comm_dev = dev if (dist is None or args.backend == "nccl") else torch.device("cpu")
    # proof of the job's width for the JSON line: what the process group itself reports, not what the flags asked for
    ranks_info = {"world": world, "backend": None, "devices": [f"cuda:{local_rank}"], "rccl_ranks_seen": None, "launched_by": "direct"}
    if dist is not None:
        one = torch.ones(1, dtype=torch.int64, device=comm_dev)
        dist.all_reduce(one)                                   # every rank adds 1: the number of ranks the collective really spans
        mine = torch.tensor([local_rank], dtype=torch.int64, device=comm_dev)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ranks_info = {"world": dist.get_world_size(), "backend": str(dist.get_backend()), "devices": [f"cuda:{int(x)}" for x in seen],
                      "rccl_ranks_seen": int(one), "launched_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "env"}
        if int(one) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the process group spans {int(one)} rank(s)", file=sys.stderr, flush=True)
            sys.exit(2)

    from rho_tts_amd import _native, config
    from rho_tts_amd.engine import Engine
    from rho_tts_amd.voice import synthetic_reference_clip
    from rho_tts_amd.dist import broadcast_voice, gather_waveforms, padding_efficiency, plan_corpus, unshard, waveforms_to_host

    cfg = config.PRESETS[args.model]()
    log(f"rank {rank}/{world}: building {cfg.name} engine (synthetic weights) ...")
    eng = Engine(cfg=cfg, model_path=cfg.name, device_ordinal=local_rank, max_batch=args.batch, synthetic=True)
    if args.eos_live:
        eng.ignore_eos = False
    if args.greedy:
        eng.params.do_sample = False
    for code in [c for c in args.tune.split(",") if c]:
        eng.ctx.lib.rt_debug_tune(int(code), 0)
    B = args.batch
    ref_words = 75 if args.ref_seconds >= 10 else max(3, int(args.ref_seconds * 2.5))
    clip = synthetic_reference_clip(args.ref_seconds, cfg.sample_rate, 789)
    ref_text = " ".join(WORDS[i % len(WORDS)] for i in range(ref_words))
    post = _native.make_post_params(sample_rate=cfg.sample_rate, stages=_native.POST_PIPELINE)
    corpus = None
    if args.corpus > 0:
        # C4: the same corpus on every rank (same seed), dealt by estimated frames; item ids = corpus indices, so a text's
        # audio does not depend on the number of ranks or on the batch it lands in
        corpus = sentences(args.corpus, (6, 24), seed=789)
        c_frames = [eng.frames_for(t, 0) for t in corpus]
        shards, plans = plan_corpus(c_frames, world, B)
        texts = [corpus[i] for i in shards[rank]]
        item_ids = list(shards[rank])
        my_plan = [c_frames[i] for i in shards[rank]]
        if args.length_error > 0:
            import random
            rnd = random.Random(4242)
            actual = [max(2, int(round(f * (1.0 + args.length_error * (2.0 * rnd.random() - 1.0))))) for f in c_frames]
        else:
            actual = c_frames
        my_frames = [actual[i] for i in shards[rank]]
        pad_eff = padding_efficiency(c_frames, [b for pl in plans for b in pl])
    else:
        texts = sentences(B, args.words, seed=789 + rank)
        item_ids = list(range(rank * B, (rank + 1) * B))
        my_frames = my_plan = None

    sched = {}                                                # kept / launched row-frames of the decode schedule (this rank, all steps)

    phases = {}                                                # --phase-times: wall ms per phase of the timed steps on this rank (synchronising)

    def mark(name, t_prev):
        if not args.phase_times:
            return t_prev
        torch.cuda.synchronize()
        eng.ctx.synchronize()
        now = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + (now - t_prev) * 1e3
        return now

    def step():
        tp = time.perf_counter()
        if dist is None or rank == 0:
            # conditioning, once per step as the reference does per call (ref_audio=path, qwen.py:253-258): the 30-s clip through
            # the audio encoder (codes + speaker embedding) and the voice prefix through the talker (its K/V)
            eng.set_voice_from_audio(clip, ref_text)
        tp = mark("conditioning", tp)
        if dist is not None:
            broadcast_voice(eng, dist, src=0, comm_device=comm_dev)
        tp = mark("voice broadcast", tp)
        raw = eng.synthesize(texts, seed=789, item_ids=item_ids, max_frames=my_frames, plan_frames=my_plan, stats=sched,
                             continuous=False if args.static_batches else None) if texts else []
        tp = mark("decode + codec decoder", tp)
        outs, stats = eng.post_process([[w] for w in raw], post) if raw else ([], [])
        tp = mark("post-processing", tp)
        audio_s = sum(o.numel() for o in outs) / cfg.sample_rate   # delivered (post-processed) samples
        if dist is not None:
            host = gather_waveforms(outs, dist, dst=0, device=comm_dev, copy=False)   # (views of pinned staging, read before the step after next)
            if corpus is not None and rank == 0:
                host = unshard(host, shards, len(corpus))
        else:
            host = waveforms_to_host(outs, copy=False)
        tp = mark("waveforms to rank 0 / host", tp)
        return audio_s, host

    log(f"engine ready: {eng.model.weight_bytes() / 1e9:.2f} GB of weights; warmup x{args.warmup}")
    for _ in range(args.warmup):
        step()
        log("warmup step done")

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.ctx.synchronize()

    sync()
    sched.clear()
    phases.clear()
    t0 = time.perf_counter()
    audio_local = 0.0
    for _ in range(args.steps):
        a, _host = step()
        audio_local += a
    sync()
    if corpus is not None and rank == 0:
        assert len(_host) == len(corpus) and all(w is not None and w.numel() > 0 for w in _host), "corpus: a waveform is missing"
    dt = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    phase_ms = {k: round(v / max(1, args.steps), 3) for k, v in phases.items()} if args.phase_times else None
    t = torch.tensor([dt, audio_local], dtype=torch.float64, device=comm_dev)
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, audio_total = float(tmax[0]), float(tsum[1])
    else:
        audio_total = audio_local

    # ---- roofline of the dominant kernel (weight-streaming skinny GEMM), HIP events on the library's stream
    roof = None
    extra = {}
    extra_families = None
    if not args.no_roofline:
        eng.model.profile(True)
        step()
        n_l, ms, by = eng.model.profile_read()
        prof_classes = [eng.model.profile_read_class(c) for c in range(3)]
        eng.model.profile(False)
        if n_l > 0 and ms > 0:
            achieved = by / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_gemm_col", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": None, "launches": int(n_l),
                    "avg_launch_us": round(ms * 1e3 / n_l, 3), "avg_bytes_per_launch": round(by / n_l, 1)}
            # HBM-side bytes per launch: counters cannot be read from inside this process, so `traffic` is the FETCH_SIZE of the
            # SAME step (same model, batch, prompt and launch sequence) collected by `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py
            # --steps 1 --warmup 0 --tune 200,1301` and reduced by tools/pmc_step_split.py (x 1024 x 2: KiB, gfx950 correction)
            # The counters belong to ONE build of the library: the profile records the SHA-256 of the native sources it was taken with
            # (rho_tts_amd._build.source_hash), and any other build (any kernel change) reports traffic = null rather than stale bytes.
            default_workload = args.model == "1.7b" and B == 32 and args.words == 10 and args.ref_seconds == 30.0 and corpus is None
            try:
                from rho_tts_amd._build import source_hash
                lib_sha = source_hash()
                with open(os.path.join(ROOT, "profiles", PMC_PROFILE)) as f:
                    pmc = json.load(f)
                roof["build_sha256"] = lib_sha[:16]
                if default_workload and pmc["k_gemm_col_dispatches"] == int(n_l) and pmc.get("build_sha256", "")[:16] == lib_sha[:16]:
                    roof["traffic"] = float(pmc["all"]["fetched_bytes_per_dispatch"])
                    roof["traffic_source"] = ("collected by the builder, not in this run: FETCH_SIZE x 1024 x 2 per k_gemm_col dispatch, rocprofv3 counter pass over bench.py itself on this build "
                                              "(profiles/" + PMC_PROFILE + ": fetched / algorithmic = %.3f; talker layers %.3f, "
                                              "predictor layers %.3f - the predictor is Infinity-Cache resident, FETCH_SIZE counts its L2 misses)"
                                              % (pmc["all"]["fetched_over_algorithmic"], pmc["classes"]["talker layers"]["fetched_over_algorithmic"],
                                                 pmc["classes"]["predictor layers"]["fetched_over_algorithmic"]))
                elif default_workload:
                    roof["traffic_source"] = ("null: profiles/" + PMC_PROFILE + " was collected with another build of the native sources "
                                              "(sha256 %s..., %d k_gemm_col dispatches) - re-run tools/collect_profiles.sh" % (pmc.get("build_sha256", "?")[:16], pmc["k_gemm_col_dispatches"]))
                # HBM GB/s and MFMA-busy per kernel family against the chip's peaks (north_star): from the committed rocprofv3 passes of
                # this same build (kernel trace for time, FETCH_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes for bytes and matrix-core use)
                if pmc.get("build_sha256", "")[:16] == lib_sha[:16] and "families" in pmc:
                    extra_families = pmc["families"]
                else:
                    extra_families = None
            except (OSError, KeyError, ValueError):
                extra_families = None
        # decode-step view: algorithmic bytes per frame for the local batch
        t_, p_ = cfg.talker, cfg.predictor
        w_talker = 2 * (t_.weight_params() + t_.hidden * cfg.codec_vocab)
        w_pred = 2 * (p_.weight_params() + (cfg.n_groups - 1) * p_.hidden * cfg.predictor_vocab
                      + (t_.hidden * p_.hidden if cfg.has_mtp_proj else 0))
        frames = eng.frames_for(texts[0], 0) if corpus is None else int(round(sum(c_frames) / len(c_frames)))
        ctx_len = eng.model.prefix_len() + args.words + 3 + frames // 2
        kv = B * ctx_len * t_.layers * 2 * t_.kv_heads * t_.head_dim * 2
        extra = {"bytes_per_frame": int(w_talker + w_pred + kv), "frames_per_item": frames, "prefix_rows": eng.model.prefix_len()}
        if roof is not None:
            # (1) the dominant kernel's launches by what they stream (rt_profile_read_class): bytes that cross HBM once per frame
            # against their own peak, and the predictor's passes 2..15, whose 157 MB come back from the Infinity Cache
            names = ("talker layers + codec head + mtp", "predictor first pass + 15 heads", "predictor passes 2..15")
            cls = prof_classes
            (n0, ms0, by0), (n1, ms1, by1), (n2, ms2, by2) = cls
            def rate(by_, ms_):
                return round(by_ / (ms_ * 1e-3) / 1e9, 1) if ms_ > 0 else None
            roof["hbm_unique"] = {"what": names[0] + "; " + names[1] + ": every distinct weight byte of the frame, once (SURVEY.md 8d)",
                                  "launches": int(n0 + n1), "bytes": int(by0 + by1), "ms": round(ms0 + ms1, 3), "GB/s": rate(by0 + by1, ms0 + ms1),
                                  "peak": 8000.0, "frac": round((rate(by0 + by1, ms0 + ms1) or 0.0) / 8000.0, 4)}
            roof["infinity_cache_restream"] = {"what": names[2] + ": the same 5 layers again, served on-die (256 MiB Infinity Cache)",
                                               "launches": int(n2), "bytes": int(by2), "ms": round(ms2, 3), "GB/s": rate(by2, ms2),
                                               "peak": 8600.0, "peak_source": "MI355X_MICROARCH.md, gather table: 8.6 TB/s chip-wide from an Infinity-Cache-resident table",
                                               "frac": round((rate(by2, ms2) or 0.0) / 8600.0, 4)}
            # (2) the decode frame on SURVEY.md 8d's definition: each distinct weight byte ONCE per frame + the K/V read and written,
            # over the measured wall time of a frame (the whole dependent chain: GEMMs, attention, samplers, embeddings)
            ids_ = [eng.tokenizer.encode(t) for t in texts[:B]]
            fr_ = [frames] * len(ids_)
            best = None
            for _ in range(2):
                eng.model.generate_begin(ids_, fr_, eng.params.talker(), eng.params.predictor(), seed=789, item_ids=item_ids[:len(ids_)],
                                         ignore_eos=True)
                sync()
                t1 = time.perf_counter()
                eng.model.generate_step(frames)
                sync()
                d_ms = (time.perf_counter() - t1) * 1e3
                eng.model.generate_end()
                best = d_ms if best is None else min(best, d_ms)
            ms_frame = best / frames
            roof["decode_step"] = {"definition": "SURVEY.md 8d: W_talker + W_predictor (each distinct byte once) + K/V read + written per frame of the local batch, "
                                                 "over the wall time of one decode frame (rt_generate_step over all frames, graphs as in the timed region)",
                                   "bytes_per_frame": extra["bytes_per_frame"], "decode_ms_per_frame": round(ms_frame, 4),
                                   "achieved": round(extra["bytes_per_frame"] / (ms_frame * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(extra["bytes_per_frame"] / (ms_frame * 1e-3) / 1e9 / 8000.0, 4),
                                   "audio_s_per_s_at_roofline": round(B * 0.08 / (extra["bytes_per_frame"] / 8.0e12), 1)}
        if roof is not None and extra_families:
            extra["kernel_families"] = extra_families
    if phase_ms is not None:
        extra["phase_ms"] = phase_ms
    if corpus is not None:
        extra.update({"corpus_texts": len(corpus), "corpus_frames": int(sum(actual)), "length_error": args.length_error,
                      # rows kept busy by the decode schedule of rank 0, measured (kept frames / (frames launched x rows));
                      # `planned` is what static length-bucketed batches of the same shards would give
                      "padding_efficiency": round(sched.get("frames", 0) / max(1, sched.get("padded_frames", 1)), 4),
                      "padding_efficiency_static_plan": round(pad_eff, 4),
                      "schedule": "static length-bucketed batches" if (args.static_batches or B > 64) else "continuous batching (finished rows handed to queued texts)",
                      "row_hand_overs_per_step": sched.get("hand_overs", 0) // max(1, args.steps),
                      "batches_per_rank_static_plan": [len(pl) for pl in plans]})

    # time to first audio of the streaming entry (BaseTTS.stream yields per segment, base_tts.py:1132-1190): one text alone,
    # call -> decoded, vocoded, post-processed, 16-bit PCM of the first segment on the host
    if rank == 0 and not args.no_roofline:
        one = sentences(1, args.words, seed=4242)
        tt = []
        for _ in range(3):
            sync()
            t1 = time.perf_counter()
            raw1 = eng.synthesize(one, seed=789, item_ids=[0])
            o1, _s = eng.post_process([[raw1[0]]], post)
            _pcm = eng.ctx.pcm16(o1[0]).cpu()
            tt.append((time.perf_counter() - t1) * 1e3)
        extra["ttfa_ms_one_text"] = round(min(tt), 2)
        extra["ttfa_audio_s"] = round(o1[0].numel() / cfg.sample_rate, 2)
        # ... and of the SUB-SEGMENT streaming entry (provider.stream with stream_chunk_frames = 12, an extension): call -> the first
        # 12 codec frames decoded (rt_generate_begin / _step), vocoded, levelled / trimmed / faded (rt_stream_chunk), 16-bit PCM on the host
        tc, first_s = [], 0.0
        for _ in range(3):
            sync()
            t1 = time.perf_counter()
            state = [0.0, 1.0]
            gen = eng.stream_wav(one[0], seed=789, item_id=0, first_chunk=12, chunk=36)
            raw, last = next(gen)
            piece = eng.ctx.stream_chunk(post, raw, state, True, last)
            _pcm = eng.ctx.pcm16(piece).cpu()
            tc.append((time.perf_counter() - t1) * 1e3)
            first_s = piece.numel() / cfg.sample_rate
            gen.close()
        extra["ttfa_ms_first_chunk"] = round(min(tc), 2)
        extra["ttfa_first_chunk_audio_s"] = round(first_s, 2)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args, eng)

    if rank == 0:
        line = {
            # BASELINE.json's metric at the default flags; any other model / batch is named as what it is
            "metric": f"audio-sec/wall-sec (RTF) Qwen3-TTS-{args.model.upper() if args.model[0].isdigit() else args.model} batch={B}",
            "value": round(audio_total / dt, 2),
            "unit": "audio-s/s",
            "n_gpus": world,
            "ranks": ranks_info,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak" if corpus is None else "strong",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": (f"{cfg.name} bf16, batch {B}/GPU, {args.ref_seconds:g}-s reference clone, "
                                    + (f"{args.words}-word sentences ({eng.frames_for(texts[0], 0)} frames each)" if corpus is None else
                                       f"ONE corpus of {len(corpus)} texts of 6-24 words ({min(c_frames)}-{max(c_frames)} frames) sharded by length over the ranks, "
                                       + ("length-bucketed static batches" if (args.static_batches or B > 64) else "continuous batching on the rank's decode rows"))
                                    + f", sampling={'greedy' if args.greedy else 'top-k 50 T 0.9'}, seeded synthetic weights; "
                                    "value counts delivered (post-processed) audio"),
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def cpu_baseline(cfg, args, eng):
    """The CPU oracle (oracle/model.py, a port - the reference's own CPU path needs the absent qwen-tts package) timed on this
    box's host cores on a BOUNDED sample of the same workload: same model shape, batch and prompt (30-s
    reference clone, 460-row prefix, prefilled once and shared), 8 decoded frames per item + their codec decode + post-processing;
    the per-frame costs are then extended linearly to the workload's frame count.  Reported beside the GPU number, never as the
    target."""
    from oracle import encoder as OE
    from oracle import postprocess as OP
    from oracle.model import OracleModel, Voice
    from oracle.sampling import SamplingParams
    from rho_tts_amd.voice import synthetic_reference_clip
    from rho_tts_amd.weights import synthetic_state

    host_cores = os.cpu_count() or 1
    log(f"cpu baseline: copying weights to the host ({host_cores} host cores) ...")
    state = {k: v.cpu() for k, v in synthetic_state(cfg, 789, device=eng.device).items()}
    om = OracleModel(cfg, state)
    del state
    B = args.batch
    # Thread count: SURVEY.md 8d asks for os.cpu_count(), but eager PyTorch does not scale to every core of a 256-core host on
    # 32-row GEMMs - so it is MEASURED here, not assumed: 2 decode frames of the workload's batch behind a 4-frame voice prefix
    # at each candidate, and the sample below runs with the fastest (the probe's seconds per frame are reported in `thread_probe`).
    probe = {}
    cands = [c for c in (8, 16, 32, 64, 128, host_cores) if c <= host_cores]
    cands = sorted(set(cands))
    if args.cpu_threads > 0:
        cands = [min(args.cpu_threads, host_cores)]
    if len(cands) > 1:
        H_ = cfg.talker.hidden
        pv = Voice("english", None, torch.zeros(H_), [3, 4, 5], torch.zeros(4, cfg.n_groups, dtype=torch.int64))
        p_ids = [eng.tokenizer.encode(t) for t in sentences(B, args.words, seed=789)]
        for c in cands:
            # ascending, and stop once a count is clearly slower than the best so far: eager PyTorch gets SLOWER with more threads on
            # these 32-row GEMMs (first measurement on a 256-core host: 0.38 / 0.63 / 1.55 / 4.2 s per frame at 16 / 32 / 64 / 128
            # threads; at 256 the probe alone ran for minutes), so the large counts are only tried while they still pay
            if probe and min(probe.values()) * 1.5 < probe[max(probe)]:
                break
            torch.set_num_threads(c)
            tmp = {}
            with torch.no_grad():
                tp0 = time.perf_counter()
                om.generate(pv, p_ids, [2] * B, SamplingParams(True, 0.9, 50, 1.0, 1.05), seed=789, share_prefix=True, timing=tmp)
                probe[c] = round((time.perf_counter() - tmp["prefill_done"]) / 2.0, 3)
            log(f"cpu baseline: thread probe {c} threads: {probe[c]:.2f} s per decode frame (prefill {tmp['prefill_done'] - tp0:.1f} s)")
    cores = min(probe, key=probe.get) if probe else cands[0]
    torch.set_num_threads(cores)
    frames = 8
    full_frames = eng.frames_for(sentences(1, args.words, seed=789)[0], 0)
    texts = sentences(B, args.words, seed=789)
    ref_words = 75 if args.ref_seconds >= 10 else max(3, int(args.ref_seconds * 2.5))
    clip = synthetic_reference_clip(args.ref_seconds, cfg.sample_rate, 789)
    ids = [eng.tokenizer.encode(t) for t in texts]
    log("cpu baseline: oracle conditioning (audio encoder) ...")
    tm = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        n_clip = (clip.shape[0] // cfg.codec.total_upsample) * cfg.codec.total_upsample
        ref_codes, spk = OE.encode(om.W, cfg, clip[:n_clip])
        v = Voice("english", None, spk, eng.tokenizer.encode(" ".join(WORDS[i % len(WORDS)] for i in range(ref_words))), ref_codes)
        t_enc = time.perf_counter()
        log(f"cpu baseline: conditioning {t_enc - t0:.1f} s; prefill + decode running ...")
        codes = om.generate(v, ids, [frames] * B, SamplingParams(True, 0.9, 50, 1.0, 1.05), seed=789, share_prefix=True, timing=tm)
        t_dec = time.perf_counter()
        q = cfg.codec.num_quantizers
        log(f"cpu baseline: prefill {tm['prefill_done'] - t_enc:.1f} s, {frames} frames in {t_dec - tm['prefill_done']:.1f} s, codec decoder ...")
        wav = om.code2wav(torch.stack([c[:, :q].T for c in codes]))
        t_voc = time.perf_counter()
        p = OP.PostParams(sample_rate=cfg.sample_rate)
        for b in range(B):
            OP.finish_item([wav[b]], p)
    t1 = time.perf_counter()
    cond_s, prefill, decode, vocode, post = t_enc - t0, tm["prefill_done"] - t_enc, t_dec - tm["prefill_done"], t_voc - t_dec, t1 - t_voc
    scale = full_frames / frames
    t_full = cond_s + prefill + (decode + vocode + post) * scale
    audio_full = B * om.wav_length(full_frames) / cfg.sample_rate
    return {"value": round(audio_full / t_full, 4), "unit": "audio-s/s", "cores": cores, "host_cores": host_cores, "kind": "port",
            "thread_probe_s_per_frame": {str(k): v for k, v in probe.items()},
            "sample": f"oracle/model.py (PyTorch eager f32 on bf16-valued weights), {cfg.name}, batch {B}, {args.ref_seconds:g}-s reference clone "
                      f"({om.prefix_embeddings(v).shape[0]}-row prefix prefilled once), {frames} of {full_frames} frames/item decoded + codec decode + "
                      f"post-processing = {t1 - t0:.1f} s of CPU work (conditioning {cond_s:.1f}, prefill {prefill:.1f}, decode {decode:.1f}, codec {vocode:.1f}, post {post:.1f}); "
                      f"value = the per-frame costs extended linearly to {full_frames} frames ({t_full:.0f} s)"}


if __name__ == "__main__":
    main()
