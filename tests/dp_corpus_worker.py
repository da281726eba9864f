"""Worker of tests/test_provider_gpu.py::test_c4_corpus_through_the_provider (also usable by hand under torch.distributed.run):
BASELINE.json configs[3] - the 512-text ragged corpus on the 1.7B preset - through ``MI355XQwenTTS.generate`` in its
data-parallel mode, process group on RCCL.  Prints one JSON line with what the test asserts on.

    RHO_TTS_AMD_FORCE_DIST=1 python -m torch.distributed.run --nproc-per-node 1 tests/dp_corpus_worker.py --texts 512 --jitter 0.3
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RHO_TTS_AMD_SYNTHETIC", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--texts", type=int, default=512)
    ap.add_argument("--jitter", type=float, nargs="+", default=[0.0, 0.3])
    ap.add_argument("--model", default="Qwen/Qwen3-TTS-12Hz-1.7B-Base")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--check", type=int, default=8, help="items regenerated alone and compared bit for bit")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()

    import numpy as np
    import torch

    from rho_tts_amd.dist import init_from_env
    rank, local_rank, world = init_from_env(args.backend)               # before anything touches the GPU
    import bench
    from rho_tts_amd.provider import MI355XQwenTTS
    from rho_tts_amd.voice import synthetic_reference_clip

    texts = bench.sentences(args.texts, (6, 24), seed=789)
    ref_text = " ".join(bench.WORDS[i % len(bench.WORDS)] for i in range(75))
    with tempfile.TemporaryDirectory() as d:
        clip = os.path.join(d, "ref.npy")
        np.save(clip, synthetic_reference_clip(30.0, 24000, 789))
        tts = MI355XQwenTTS(device=f"cuda:{local_rank}", reference_audio=clip, reference_text=ref_text, model_path=args.model, batch_size=args.batch)
        # (a text whose random-weight audio "decays" would be regenerated under a wall-clock seed - base_tts.py:741-748 - and could
        # then not be compared with itself alone: one attempt per text here)
        tts.max_decay_retries = 1
        out = {"rank": rank, "world": world, "texts": len(texts), "runs": []}
        for jitter in args.jitter:
            eng = tts._load_engine()
            eng.length_jitter = jitter
            tts.last_schedule.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = tts.generate(texts)
            dt = time.perf_counter() - t0
            run = {"jitter": jitter, "seconds": round(dt, 3)}
            if rank == 0:
                sched = dict(tts.last_schedule)
                run["complete"] = bool(res is not None and len(res) == len(texts) and all(r is not None and r.audio.numel() > 0 for r in res))
                run["audio_s"] = round(sum(r.duration_sec for r in res if r is not None), 2) if res else 0.0
                run["occupancy"] = round(sched.get("frames", 0) / max(1, sched.get("padded_frames", 1)), 4)
                run["hand_overs"], run["generate_calls"] = int(sched.get("hand_overs", 0)), int(sched.get("batches", 0))
                # corpus order: item i's audio cannot be longer than what its own frame budget vocodes to, and (silence trim aside)
                # not much shorter
                frames = [eng.frames_actual(t, 0) for t in texts]
                lens = [int(r.audio.numel()) for r in res]
                cap = [eng.model.wav_length(f) for f in frames]
                run["order_ok"] = bool(all(0.5 * c <= n <= c for n, c in zip(lens, cap)))
                run["distinct_lengths"] = len(set(frames))
                # a sample of items regenerated ALONE (same RNG stream = their position in the call) must come out bit for bit
                step = max(1, len(texts) // max(1, args.check))
                idx = list(range(0, len(texts), step))[: args.check]
                same = 0
                for i in idx:
                    raw = eng.synthesize([texts[i]], seed=int(tts.seed), item_ids=[i])
                    alone, _, _ = tts._finish_items([[raw[0]]])[0]
                    same += int(torch.equal(alone.cpu().reshape(-1), res[i].audio.reshape(-1)))
                run["alone_checked"], run["alone_equal"] = len(idx), same
            out["runs"].append(run)
        if rank == 0:
            print(json.dumps(out), flush=True)
        tts.close()
    import torch.distributed as td
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
