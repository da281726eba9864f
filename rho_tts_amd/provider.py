"""``MI355XQwenTTS`` — the drop-in generation provider.

Registers through ``TTSFactory.register_provider()`` (factory.py:110-123) and implements the two
abstract members of ``BaseTTS`` (``_generate_audio`` base_tts.py:587-599, ``sample_rate`` :1192-1196)
with the constructor surface of the reference's ``QwenTTS`` (providers/qwen.py:48-66).

What differs from the reference, on purpose:
  * ``batch_size`` is live.  The reference stores it and never reads it (qwen.py:59,83) and its
    ``_run_pipeline`` is strictly sequential (base_tts.py:726,753,770).  Here ``_run_pipeline`` is
    overridden to flatten texts -> segments, run them ``batch_size`` at a time through
    ``_generate_audio(list)``, and finish every item (join -> loudness -> decay check) in ONE fused
    HIP launch — with the reference's per-item semantics (listed at ``BatchedPipeline``).
  * the voice conditioning is computed once per voice, not on every call (qwen.py:253-258).
  * ``sample_rate`` is a constant; the reference generates a throw-away sample to learn it (qwen.py:408-413).
  * every numeric leaf runs on the GPU through the C ABI; there is no CPU path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import logging
import threading
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _native
from .api import (BaseTTS, CancellationToken, CancelledException, ProviderInfo, TTSFactory, VoiceInfo)

logger = logging.getLogger("rho_tts_amd")

PROVIDER_NAME = "qwen_mi355x"
BUILTIN_VOICES = ["Chelsie", "Aidan", "Vivian", "Ryan", "Aria", "Ethan", "Luna", "Harper", "James"]


class BatchedPipeline:
    """Batched restatement of ``BaseTTS._run_pipeline`` (base_tts.py:708-956).  Kept from the reference:
      - phonetic mapping first (:721); segmentation per text with the memory-aware limit (:730-731)
      - ``CancelledException`` is raised, not returned, at text / segment / iteration granularity (:727,754,771)
      - ``ValueError`` propagates; ``RuntimeError`` mentioning "out of memory" or "length" and any other
        exception fail only the segment they hit and are retried up to ``max_iterations`` (:786-797);
        other ``RuntimeError`` propagate (:794)
      - ``max_iterations == 1`` means no validation calls at all (:800)
      - a text whose audio decays is regenerated whole, up to ``max_decay_retries`` times, with a seed taken
        from the wall clock (:741-748, 926-941)
      - per-item ``None`` on failure (:943-946); metadata: ``decay_ratio`` always, ``drift_prob`` = max and
        ``text_similarity`` = min only when validation ran (:948-953)
      - ``progress_callback("Generating segment i/n...")`` once per segment and attempt (:760-761)
    """

    def _run_pipeline(self, texts, cancellation_token, progress_callback=None):
        plans = self._plan_texts(texts, cancellation_token)
        self._begin_retry_seeds()
        return self._run_plans(plans, list(range(len(plans))), cancellation_token, progress_callback)

    def _plan_texts(self, texts, token, max_chars: Optional[int] = None) -> List[List[str]]:
        """Phonetic mapping, then segmentation per text with the memory-aware limit (base_tts.py:721,727-731).
        ``max_chars``: a limit agreed beforehand (data-parallel runs: every rank must cut the texts alike) instead of this
        process's own free-memory reading."""
        mapped = [self._apply_phonetic_mapping(t) for t in texts]
        plans: List[List[str]] = []
        for idx, text in enumerate(mapped):
            if token.is_cancelled():
                raise CancelledException(f"Cancelled during text item {idx}")
            plans.append(self._split_text_into_segments(text, self._compute_max_chars() if max_chars is None else max_chars))
        return plans

    # ---- retry seeds.  The reference reseeds from the wall clock at every decay retry and validation retry
    # (``self.seed = int(time.time() * 1000) % 100000``, base_tts.py:743,778) and leaves ``self.seed`` mutated.  Here the clock is
    # read ONCE per call and the seed of retry stage (decay attempt a, validation iteration k) is derived from that reading: still
    # a fresh clock-dependent seed per retry, still left in ``self.seed`` - but a function of (clock reading, a, k) alone, so the
    # ranks of a data-parallel call (which share rank 0's reading) give an item the audio one GPU would, retries included, and
    # can agree afterwards on the seed the call leaves behind.
    @staticmethod
    def _retry_clock() -> int:
        return int(time.time() * 1000) % 100000

    def _begin_retry_seeds(self, base: Optional[int] = None, entry_seed: Optional[int] = None) -> None:
        self._retry_base = int(self._retry_clock() if base is None else base)
        self._entry_seed = int(self.seed if entry_seed is None else entry_seed)
        self.seed = self._entry_seed
        self._last_stage = (0, 0)

    def _stage_seed(self, attempt: int, iteration: int) -> int:
        if (attempt, iteration) == (0, 0):
            return self._entry_seed
        return int((self._retry_base + 7919 * attempt + 104729 * iteration) % 100000)

    def _enter_stage(self, attempt: int, iteration: int) -> None:
        if not hasattr(self, "_retry_base"):
            self._begin_retry_seeds()
        self.seed = self._stage_seed(attempt, iteration)
        self._last_stage = max(getattr(self, "_last_stage", (0, 0)), (attempt, iteration))

    def _run_plans(self, plans, owned, token, progress_callback=None):
        """The pipeline for the text items ``owned`` (indices into ``plans``; all of them on one GPU, this rank's share in a
        data-parallel run).  Returns one entry per owned item, in that order.  The RNG stream of a segment is its position in
        the work list of ALL items, owned or not: a text's audio does not depend on how many ranks shared the list."""
        seg_base, tot = [], 0
        for pl in plans:
            seg_base.append(tot)
            tot += len(pl)
        n = len(plans)
        final: List[Optional[torch.Tensor]] = [None] * n
        seg_audio: List[List[torch.Tensor]] = [[] for _ in range(n)]
        decay: List[Tuple[float, bool]] = [(0.0, True)] * n
        scores: List[Tuple[List[float], List[float]]] = [([], []) for _ in range(n)]
        pending = list(owned)
        for attempt in range(self.max_decay_retries):
            if not pending:
                break
            if attempt > 0:
                self._enter_stage(attempt, 0)
                logger.warning(f"  sound decay detected in {len(pending)} item(s), regenerating (attempt {attempt + 1}/{self.max_decay_retries})")
            work = [(i, s, seg) for i in pending for s, seg in enumerate(plans[i])]
            for i in pending:
                seg_audio[i], scores[i] = [], ([], [])
            got = self._generate_work(work, plans, token, progress_callback, scores, [seg_base[i] + s for i, s, _ in work], attempt)
            for (i, s, _), a in zip(work, got):
                if a is not None:
                    seg_audio[i].append(a)
            ready = [i for i in pending if seg_audio[i]]
            finished = self._finish_items([seg_audio[i] for i in ready]) if ready else []
            still = []
            for i, res in zip(ready, finished):
                if res is None:
                    continue
                final[i], ratio, ok = res
                decay[i] = (ratio, ok)
                logger.info(f"  Item {i + 1} sound decay ratio: {ratio:.3f} (threshold: {self.sound_decay_threshold})")
                if not ok:
                    still.append(i)
            pending = still
        out: List[Optional[Tuple[torch.Tensor, int, dict]]] = []
        for i in owned:
            if final[i] is None or not seg_audio[i]:
                logger.error(f"Item {i + 1} failed: no audio generated")
                out.append(None)
                continue
            meta: Dict[str, float] = {}
            if scores[i][0]:
                meta["drift_prob"] = max(scores[i][0])
            if scores[i][1]:
                meta["text_similarity"] = min(scores[i][1])
            meta["decay_ratio"] = decay[i][0]
            out.append((final[i], len(seg_audio[i]), meta))
        return out

    # one entry of `work` per (item, segment index, text); returns the accepted audio (or None) per entry
    def _generate_work(self, work, plans, token, progress_callback, scores, stream_ids=None, attempt: int = 0):
        if stream_ids is None:
            stream_ids = list(range(len(work)))
        accepted: List[Optional[torch.Tensor]] = [None] * len(work)
        # per-segment validation state, exactly the reference's locals (base_tts.py:765-768): best audio by drift, the
        # MINIMUM drift seen, the LAST text similarity computed (whenever the voice check passed), the last audio
        state = [{"best": None, "best_drift": float("inf"), "text_sim": None, "last": None} for _ in work]
        todo = list(range(len(work)))
        bs = max(1, int(getattr(self, "batch_size", 1)))
        for iteration in range(self.max_iterations):
            if not todo:
                break
            if iteration > 0:
                self._enter_stage(attempt, iteration)
            retry: List[int] = []
            for chunk in self._cut_batches(todo, work, bs):
                for w in chunk:
                    i, s, _ = work[w]
                    if token.is_cancelled():
                        raise CancelledException(f"Cancelled during segment {s + 1} of item {i + 1}")
                    if progress_callback and iteration == 0:
                        progress_callback(f"Generating segment {s + 1}/{len(plans[i])}...")
                self._set_seeds()
                # RNG stream of a segment = its index in the work list of the whole call: independent of how the list is cut into
                # batches and of how many ranks share it
                audios = self._generate_chunk([work[w][2] for w in chunk], [stream_ids[w] for w in chunk], token)
                for w, a in zip(chunk, audios):
                    if a is None:
                        retry.append(w)
                        continue
                    state[w]["last"] = a
                    if self.max_iterations == 1:
                        state[w]["best"] = a
                        accepted[w] = a
                        if getattr(self, "auto_sort_good_dir", None) or getattr(self, "auto_sort_bad_dir", None):
                            self._auto_sort_only(a)          # drift detection for auto-sort even without validation retries (:801-818)
                        continue
                    if self._validate_segment(a, work[w][2], state[w]):
                        accepted[w] = a                      # valid: this attempt's audio is kept (:859)
                    else:
                        retry.append(w)
            todo = retry
        for w in todo:                                       # iterations exhausted: best by drift, else the last audio (:887-898)
            st = state[w]
            accepted[w] = st["best"] if st["best"] is not None else st["last"]
        if self.max_iterations > 1:
            for w, a in enumerate(accepted):                 # scores of every segment that produced audio (:901-906)
                if a is None:
                    continue
                i = work[w][0]
                if state[w]["best_drift"] != float("inf"):
                    scores[i][0].append(state[w]["best_drift"])
                if state[w]["text_sim"] is not None:
                    scores[i][1].append(state[w]["text_sim"])
        return accepted

    def _estimate_cost(self, segment: str) -> Optional[float]:
        """Expected decode length of a segment (any monotone unit), or None when the provider cannot tell.
        A batch decodes until its LONGEST member ends, so ragged batches idle their short rows."""
        return None

    def _cut_batches(self, todo: List[int], work, bs: int) -> List[List[int]]:
        """Cut the work list into batches of ``bs``.  More work than one batch is bucketed by estimated length (longest first,
        so every batch holds segments of similar length); fewer keeps arrival order.  The RNG stream of a segment is its index
        in the work list, so the audio of every segment is the same for any cut; inside a batch the indices stay ascending."""
        if len(todo) > bs:
            costs = {w: self._estimate_cost(work[w][2]) for w in todo}
            if all(c is not None for c in costs.values()):
                order = sorted(todo, key=lambda w: (-costs[w], w))
                return [sorted(order[b0:b0 + bs]) for b0 in range(0, len(order), bs)]
        return [todo[b0:b0 + bs] for b0 in range(0, len(todo), bs)]

    def _generate_chunk(self, segs: List[str], item_idx: List[int], token) -> List[Optional[torch.Tensor]]:
        """One batched ``_generate_audio`` call; on failure fall back to one call per segment so that the
        reference's per-segment error policy decides who fails."""
        try:
            out = self._generate_audio(list(segs), item_ids=item_idx, cancellation_token=token)
            return list(out)
        except (ValueError, CancelledException):
            raise
        except Exception as first:  # noqa: BLE001
            if len(segs) == 1:
                return [self._classify(first)]
        res: List[Optional[torch.Tensor]] = []
        for seg, it in zip(segs, item_idx):
            try:
                res.append(self._generate_audio(seg, item_ids=[it], cancellation_token=token))
            except (ValueError, CancelledException):
                raise
            except Exception as e:  # noqa: BLE001
                res.append(self._classify(e))
        return res

    @staticmethod
    def _classify(e: Exception):
        if isinstance(e, RuntimeError):
            msg = str(e).lower()
            if "out of memory" in msg or "length" in msg:
                logger.error(f"    segment OOM/length: {e}")
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
                return None
            raise e
        logger.warning(f"    generation error ({e})")
        return None

    # Optional tensor-level validators: callables on the waveform the provider still holds (GPU or CPU,
    # 1-D float32) - when both are set a segment is validated without the reference's temp-WAV round trip
    # (base_tts.py:821-830); either one alone replaces its file-based counterpart.
    drift_scorer = None        # Callable[[torch.Tensor, int], float]            -> accent-drift probability
    transcriber = None         # Callable[[torch.Tensor, int], Optional[str]]    -> transcription (None = failed)

    def _validate_segment(self, audio: torch.Tensor, text: str, st: dict) -> bool:
        """One validation attempt (base_tts.py:821-886), updating the segment's state in the reference's order: drift ->
        auto-sort -> best-by-drift -> text match only if the voice passed.  A validator that raises counts as a failed
        attempt but keeps what it had already recorded.  Returns True when the attempt is accepted."""
        scorer, transcriber = getattr(self, "drift_scorer", None), getattr(self, "transcriber", None)
        file_based = hasattr(self, "_validate_accent_drift")          # the reference's file-based validators (absent in the host mirror)
        if scorer is None and transcriber is None and not file_based:
            st["best"] = audio
            return True
        need_file = file_based and (scorer is None or transcriber is None or bool(getattr(self, "auto_sort_good_dir", None) or getattr(self, "auto_sort_bad_dir", None)))
        try:
            with (self._validation_input(audio) if need_file else contextlib.nullcontext(None)) as path:
                if scorer is not None:
                    drift = float(scorer(audio, self.sample_rate))
                    voice_ok = drift < self.accent_drift_threshold
                elif file_based:
                    drift, voice_ok = self._validate_accent_drift(path)
                else:
                    drift, voice_ok = 0.0, True                        # no classifier: passes, as base_tts.py:218-220 does
                if path is not None and hasattr(self, "_auto_sort_audio"):
                    self._auto_sort_audio(path, drift)
                if drift < st["best_drift"]:
                    st["best_drift"], st["best"] = drift, audio.clone()
                text_ok = True
                if voice_ok:
                    if transcriber is not None:
                        from .validation import validate_text_match
                        text_ok, sim, _ = validate_text_match(transcriber(audio, self.sample_rate), text, self.text_similarity_threshold)
                        st["text_sim"] = sim
                    elif file_based:
                        text_ok, sim, _ = self._validate_text_match(path, text)
                        st["text_sim"] = sim
                if voice_ok and text_ok:
                    st["best"] = audio
                    return True
                return False
        except Exception as e:  # noqa: BLE001
            logger.warning(f"    validation error ({e})")
            return False

    def _auto_sort_only(self, audio: torch.Tensor) -> None:
        """``max_iterations == 1`` with an auto-sort directory set (base_tts.py:801-818): score the accepted audio and copy it to
        the good / bad folder; nothing is retried and no score reaches the metadata.  As in the reference, only the copy itself
        is non-fatal (``_auto_sort_audio`` swallows ``OSError``)."""
        scorer = getattr(self, "drift_scorer", None)
        with self._validation_input(audio) as path:
            drift = float(scorer(audio, self.sample_rate)) if scorer is not None else self._validate_accent_drift(path)[0]
            self._auto_sort_audio(path, drift)

    def _validation_input(self, audio: torch.Tensor):
        """What the validators are handed: the reference's file-based validators get a temporary 16-bit WAV
        (base_tts.py:821-830), written from a device-side int16 conversion when the audio lives on the GPU."""
        import os
        import tempfile

        @contextlib.contextmanager
        def cm():
            fd, path = tempfile.mkstemp(suffix=".wav", prefix="rho_tts_validate_")
            os.close(fd)
            try:
                mono = audio.detach()
                self._save_wav(path, mono.unsqueeze(0) if mono.dim() == 1 else mono, self.sample_rate)
                yield path
            finally:
                try:
                    os.remove(path)
                except OSError:
                    pass
        return cm()


class DataParallelPipeline(BatchedPipeline):
    """``_run_pipeline`` across the GPUs of a node (BASELINE.json configs[3]): one process per GPU under
    ``torch.distributed`` (backend "nccl" = RCCL over xGMI), every rank calling ``generate()`` with the SAME texts.
    The reference's loop over texts is strictly sequential (base_tts.py:726-954) and keeps no state between items, so whole text
    items are the unit: they are dealt over the ranks by estimated decode length (dist.shard_items), every rank runs its share
    through the batched pipeline above (continuous batching on its decode rows, validation and decay retries included), and the
    finished waveforms + per-item records are gathered to rank 0 and put back in the caller's order.  Two collectives carry
    data - the broadcast of the voice conditioning rank 0 computed (once per voice) and the gather of the waveforms - plus
    one-word status exchanges so that a failure on one rank ends the call on all of them.
    Rank 0 returns the complete result list; the other ranks return their own items and ``None`` elsewhere (``generate()`` on
    them returns nothing and saves nothing: MI355XQwenTTS.generate).  A text's audio is a function of (text, voice, seed,
    position in the call), not of the number of ranks.
    ``data_parallel``: "auto" (on when a process group with more than one rank is initialised), True, False."""

    data_parallel = "auto"

    def _dp(self):
        mode = getattr(self, "data_parallel", "auto")
        if mode is False or mode in ("off", "0"):
            return None
        import os

        import torch.distributed as td
        if not (td.is_available() and td.is_initialized()):
            if mode is True:
                raise ValueError("data_parallel=True needs an initialised torch.distributed process group "
                                 "(launch one process per GPU with torch.distributed.run; rho_tts_amd.dist.init_from_env)")
            return None
        world = td.get_world_size()
        forced = mode is True or os.environ.get("RHO_TTS_AMD_FORCE_DIST", "") not in ("", "0")    # (one rank: exercises the RCCL call sites)
        if world < 2 and not forced:
            return None
        return td, td.get_rank(), world

    def _dp_device(self, td) -> torch.device:
        return torch.device(self.device) if str(td.get_backend()) == "nccl" else torch.device("cpu")

    @staticmethod
    def _dp_agree(td, dev, err: Optional[BaseException], what: str) -> None:
        from . import dist as D
        code = 0 if err is None else (2 if isinstance(err, CancelledException) else (3 if isinstance(err, ValueError) else 1))
        worst = D.agree(td, dev, code)
        if err is not None:
            raise err
        if worst == 2:
            raise CancelledException(f"cancelled on another rank during {what}")
        if worst == 3:
            raise ValueError(f"configuration error on another rank during {what}")
        if worst:
            raise RuntimeError(f"another rank failed during {what}")

    def _dp_share_voice(self, td, rank: int, dev) -> None:
        """Rank 0 computes the conditioning (audio encoder + prefix prefill), everyone else imports its KV blob."""
        raise NotImplementedError

    def _dp_costs(self, plans) -> List[float]:
        out = []
        for pl in plans:
            c = 0.0
            for seg in pl:
                e = self._estimate_cost(seg)
                c += float(e) if e is not None else float(max(1, len(seg.split())))
            out.append(c)
        return out

    def _run_pipeline(self, texts, cancellation_token, progress_callback=None):
        dp = self._dp()
        if dp is None:
            return super()._run_pipeline(texts, cancellation_token, progress_callback)
        from . import dist as D
        td, rank, world = dp
        dev = self._dp_device(td)
        plans, err, mc = [], None, 0
        # ---- what every rank must hold alike BEFORE anything is planned: the segment limit (it reads this process's free memory,
        # base_tts.py:158-185, and is refined by the loaded checkpoint, qwen.py:131-139 - so the engine is loaded first and the
        # ranks take the smallest limit any of them computed), the seed the call starts from and the clock reading its retry
        # seeds derive from (rank 0's)
        try:
            self._load_engine()
            mc = int(self._compute_max_chars())
        except BaseException as e:  # noqa: BLE001
            err = e
        self._dp_agree(td, dev, err, "loading")
        mc, _ = D.min_max(td, dev, mc)
        seed0, base0 = D.broadcast_ints(td, dev, [int(self.seed), self._retry_clock()], src=0)
        self._begin_retry_seeds(base0, seed0)
        try:
            plans = self._plan_texts(texts, cancellation_token, max_chars=mc)
        except BaseException as e:  # noqa: BLE001  (a token cancelled on ONE rank: re-raised by _dp_agree, on every rank)
            err = e
        self._dp_agree(td, dev, err, "planning")
        # the plans decide who owns what and where every waveform goes: a rank that cut its texts differently would drop or double
        # items without any error - so the plans' digest must agree before the deal
        import hashlib
        digest = int.from_bytes(hashlib.sha256(repr(plans).encode("utf-8")).digest()[:7], "big")
        lo, hi = D.min_max(td, dev, digest)
        if lo != hi:
            raise ValueError("data-parallel ranks planned different segmentations of the same texts (different texts, phonetic "
                             "mappings or segment limits per rank): every rank must call generate() with the same arguments")
        self._dp_share_voice(td, rank, dev)
        shards = D.shard_items(self._dp_costs(plans), world)
        local = []
        try:
            local = self._run_plans(plans, shards[rank], cancellation_token, progress_callback)
        except BaseException as e:  # noqa: BLE001  (re-raised by _dp_agree, on every rank)
            err = e
        # the seed the call leaves in self.seed is the last retry stage ANY rank reached - what one process running every item
        # would hold (its loops are attempt-major, iteration-minor) - so the next call starts alike on every rank
        a_k = getattr(self, "_last_stage", (0, 0))
        _, top = D.min_max(td, dev, a_k[0] * 100000 + a_k[1])
        self.seed = self._stage_seed(top // 100000, top % 100000)
        self._dp_agree(td, dev, err, "generation")
        nan = float("nan")
        wavs = [None if r is None else r[0].reshape(-1) for r in local]
        rows = torch.tensor([[0.0, nan, nan, nan, 0.0] if r is None else
                             [float(r[1]), float(r[2]["decay_ratio"]), float(r[2].get("drift_prob", nan)), float(r[2].get("text_similarity", nan)),
                              1.0 if r[0].dim() == 2 else 0.0] for r in local], dtype=torch.float64).reshape(len(local), 5)
        per_rank = D.gather_waveforms(wavs, td, dst=0, device=dev)
        recs = D.gather_rows(rows, td, dst=0, device=dev)
        if rank != 0:
            out: List[Optional[Tuple[torch.Tensor, int, dict]]] = [None] * len(plans)
            for i, r in zip(shards[rank], local):
                out[i] = r
            return out
        items = []
        for r in range(world):
            got = []
            for w, rec in zip(per_rank[r], recs[r].tolist()):
                if w is None:
                    got.append(None)
                    continue
                meta = {}
                if rec[2] == rec[2]:
                    meta["drift_prob"] = rec[2]
                if rec[3] == rec[3]:
                    meta["text_similarity"] = rec[3]
                meta["decay_ratio"] = rec[1]
                got.append((w.unsqueeze(0) if rec[4] else w, int(rec[0]), meta))
            items.append(got)
        return D.unshard(items, shards, len(plans))


class HipAudioLeaves:
    """The numeric leaves of the pipeline as calls into the fused HIP kernel (one stage mask each).
    Reference: base_tts.py:297-536, providers/qwen.py:268-378.  Inputs may be CPU or GPU tensors; the result lives
    where the input lived."""

    def _post_params(self, stages: int) -> _native.PostParams:
        return _native.make_post_params(
            sample_rate=self.sample_rate, silence_threshold_db=self.silence_threshold_db,
            fade_duration_sec=self.fade_duration_sec, crossfade_duration_sec=self.crossfade_duration_sec,
            inter_sentence_pause_sec=self.inter_sentence_pause_sec, trim_silence=self.trim_silence,
            sound_decay_threshold=getattr(self, "sound_decay_threshold", 0.3), stages=stages)

    def _post(self, items, stages, seg_trim=None):
        return self._native_ctx().post_process(self._post_params(stages), items, seg_trim)

    def _trim_silence(self, audio: torch.Tensor, from_start: bool = True, from_end: bool = True) -> torch.Tensor:
        if not self.trim_silence or audio.numel() == 0:
            return audio
        st = (_native.POST_TRIM_START if from_start else 0) | (_native.POST_TRIM_END if from_end else 0)
        if st == 0:
            return audio.reshape(-1) if audio.dim() == 2 else audio
        (out,), (s,) = self._post([[audio]], st)
        return out.unsqueeze(0) if s.all_silent else out       # the reference returns a (1, window) view for silence (:379-380)

    def _remove_dc_offset(self, audio: torch.Tensor) -> torch.Tensor:
        if audio.numel() == 0:
            return audio
        (out,), _ = self._post([[audio]], _native.POST_DC)
        return out.view(audio.shape)

    def _apply_fades(self, audio: torch.Tensor, fade_in: bool = True, fade_out: bool = True) -> torch.Tensor:
        if audio.numel() == 0:
            return audio
        st = (_native.POST_FADE_IN if fade_in else 0) | (_native.POST_FADE_OUT if fade_out else 0)
        (out,), _ = self._post([[audio]], st)
        return out.view(audio.shape)

    def _smooth_segment_join(self, audio_segments):
        if len(audio_segments) == 0:
            return None
        st = _native.POST_PIPELINE & ~(_native.POST_LOUDNESS | _native.POST_DECAY)
        (out,), (s,) = self._post([list(audio_segments)], st)
        return out.unsqueeze(0) if s.all_silent else out

    def _post_process_audio(self, audio: torch.Tensor) -> torch.Tensor:
        (out,), _ = self._post([[audio]], _native.POST_LOUDNESS)
        return out.view(audio.shape)

    def _validate_sound_decay(self, audio: torch.Tensor) -> tuple:
        if audio.numel() == 0:
            return 1.0, True
        _, (s,) = self._post([[audio]], _native.POST_DECAY)
        return s.decay_ratio, bool(s.decay_ok)

    def _save_wav(self, path: str, audio: torch.Tensor, sample_rate: int) -> None:
        """Output stage (base_tts.py:654-671).  A waveform that still lives in HBM is converted to the reference's 16-bit PCM
        (clip, x 32767, truncate - the stdlib fallback's definition) ON the device and crosses PCIe as int16: half the bytes of
        the float32 copy the reference makes first (:807,824,1046).  CPU tensors take the host API's own writer."""
        if not audio.is_cuda:
            return super()._save_wav(path, audio, sample_rate)
        import wave
        pcm = self._native_ctx().pcm16(audio).cpu().numpy()
        with wave.open(path, "wb") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(sample_rate)
            wf.writeframes(pcm.tobytes())

    def _finish_items(self, items: Sequence[Sequence[torch.Tensor]]):
        """join -> loudness -> decay for every item, one launch (base_tts.py:912-926)."""
        outs, stats = self._post([list(it) for it in items], _native.POST_PIPELINE)
        return [(o.unsqueeze(0) if s.all_silent else o, s.decay_ratio, bool(s.decay_ok)) for o, s in zip(outs, stats)]


class MI355XQwenTTS(DataParallelPipeline, HipAudioLeaves, BaseTTS):
    """Qwen3-TTS generation on one MI355X.  Same keyword arguments as the reference's ``QwenTTS``."""

    MAX_MODEL_CHARS = 4000
    BYTES_PER_CHAR_ESTIMATE = 500_000

    def __init__(self, device: str = "cuda", seed: int = 789, deterministic: bool = False,
                 reference_audio: Optional[str] = None, reference_text: Optional[str] = None, speaker: Optional[str] = None,
                 language: str = "English", model_path: str = "Qwen/Qwen3-TTS-12Hz-1.7B-Base",
                 max_chars_per_segment: Optional[int] = None, batch_size: int = 32, max_iterations: int = 10,
                 accent_drift_threshold: float = 0.17, text_similarity_threshold: float = 0.85,
                 sound_decay_threshold: float = 0.3, drift_model_path: Optional[str] = None,
                 phonetic_mapping: Optional[Dict[str, str]] = None):
        # Defaults are QwenTTS.__init__'s (providers/qwen.py:48-66) with ONE deviation: batch_size 32 instead of 5.  The reference
        # stores batch_size and never reads it (qwen.py:59,83), so no behaviour of its can depend on the value; here it is the
        # number of decode rows (BASELINE.json's headline batch).  max_iterations keeps the reference's 10: a caller who swaps
        # providers keeps the validation retries (they run whenever validators are installed or the tensor-level hooks are set).
        super().__init__(device, seed, deterministic, phonetic_mapping=phonetic_mapping)
        if reference_audio is not None and reference_text is None:
            raise ValueError("reference_text (transcript of reference audio) is required when reference_audio is set")
        if not str(device).startswith("cuda"):
            raise ValueError("MI355XQwenTTS runs on an AMD GPU only (device='cuda' or 'cuda:N'); there is no CPU path")
        self.reference_audio_path = reference_audio
        self.reference_text = reference_text
        self.speaker = speaker
        self.language = language
        self.voice_cloning = reference_audio is not None
        self.model_path = model_path
        self.drift_model_path = drift_model_path
        self._max_chars_explicit = max_chars_per_segment is not None
        self.max_chars_per_segment = max_chars_per_segment if max_chars_per_segment is not None else 1000
        self.batch_size = batch_size
        self.force_sentence_split = False
        self.max_iterations = max_iterations
        self.accent_drift_threshold = accent_drift_threshold
        self.text_similarity_threshold = text_similarity_threshold
        self.sound_decay_threshold = sound_decay_threshold
        self._engine = None
        self._ctx = None
        self._lock = threading.RLock()
        self._voice_key = None
        # decode-schedule figures of this instance's engine calls, accumulated until cleared: kept / launched row-frames
        # ("frames" / "padded_frames": their quotient is the row occupancy), rt_generate calls, row hand-overs
        self.last_schedule: Dict[str, int] = {}

    # ---------------------------------------------------------------- native handles
    def _device_ordinal(self) -> int:
        d = str(self.device)
        return int(d.split(":")[1]) if ":" in d else 0

    def _native_ctx(self):
        eng = self._engine
        if eng is not None:
            return eng.ctx
        if self._ctx is None:
            self._ctx = _native.Context(self._device_ordinal())
        return self._ctx

    def _load_engine(self):
        with self._lock:
            if self._engine is None:
                from .engine import Engine
                if self._ctx is not None:
                    self._ctx.close()
                    self._ctx = None
                self._engine = Engine(self.model_path, self._device_ordinal(), max_batch=max(1, min(64, int(self.batch_size))))
                # the reference refines the limit from the checkpoint's max_position_embeddings (qwen.py:131-139); a checkpoint
                # without one keeps MAX_MODEL_CHARS.  (The KV allocation - cfg.max_positions - is a separate number.)
                hf_pos = int(getattr(self._engine.cfg, "hf_max_position_embeddings", 0) or 0)
                self._max_model_chars = min(self.MAX_MODEL_CHARS, hf_pos) if hf_pos > 0 else self.MAX_MODEL_CHARS
            return self._engine

    def _voice_config_key(self):
        """The mode checks of QwenTTS._generate_audio (qwen.py:231-242), and what identifies the configured voice."""
        is_custom = "CustomVoice" in self.model_path
        if is_custom and not self.speaker:
            raise ValueError("CustomVoice model requires a named speaker. Select a built-in voice (e.g. Vivian, Ryan) "
                             "or provide reference audio with a Base model for voice cloning.")
        if not is_custom and not self.voice_cloning:
            raise ValueError("Qwen Base model requires reference audio for voice cloning. "
                             "Use a CustomVoice model with a named speaker, or provide reference audio.")
        return (self.speaker, self.language) if is_custom else (self.reference_audio_path, self.reference_text, self.language)

    def _dp_share_voice(self, td, rank: int, dev) -> None:
        from . import dist as D
        eng, err, key = None, None, None
        try:
            eng = self._load_engine()
            key = self._voice_config_key()
            if key != self._voice_key and rank == 0:
                with self._lock:
                    self._ensure_voice(eng)
        except BaseException as e:  # noqa: BLE001  (re-raised below, on every rank)
            err = e
        self._dp_agree(td, dev, err, "voice conditioning")
        # (`_dp_voice_sent` only changes here, on every rank at once - every rank runs the same calls with the same configuration -
        # so the ranks agree on whether the broadcast is due)
        if getattr(self, "_dp_voice_sent", None) != key:
            D.broadcast_voice(eng, td, src=0, comm_device=dev)
            self._voice_key = key
            self._dp_voice_sent = key

    def _ensure_voice(self, eng) -> None:
        is_custom = "CustomVoice" in self.model_path
        key = self._voice_config_key()
        if key == self._voice_key:
            return
        if is_custom:
            eng.set_builtin_voice(self.speaker, self.language)
        else:
            eng.set_voice_from_audio(self.reference_audio_path, self.reference_text, self.language)
        self._voice_key = key

    def _estimate_cost(self, segment: str) -> Optional[float]:
        eng = self._engine
        if eng is not None:
            return float(eng.frames_for(segment, len(eng.tokenizer.encode(segment))))
        return float(max(1, len(segment.split())))

    def _compute_speaker_similarity(self, wav_tensor: torch.Tensor) -> float:
        """Cosine similarity between generated audio and the reference voice (base_tts.py:325-346, which embeds both with
        resemblyzer on the CPU and is never called by the reference's own pipeline).  Here both embeddings come from the model's
        OWN speaker encoder on the GPU - the statistics-pooling head of the audio encoder that also conditions the talker
        (rt_voice_encode) - so the score says how close the output is in the space the model itself clones from.  Not comparable
        in value with resemblyzer's; same range and direction (1 = same voice)."""
        eng = self._load_engine()
        with self._lock:
            self._ensure_voice(eng)
            ref = None if eng.voice is None else eng.voice.speaker_embed
            if ref is None:
                raise ValueError("speaker similarity needs a cloned voice (reference audio with a Base model)")
            pcm = wav_tensor.detach().to(torch.float32).reshape(-1).cpu().numpy()
            cap = eng.model.rt_cfg.enc.max_ref_frames * eng.cfg.codec.total_upsample     # (samples the encoder's frame budget covers)
            _, emb = eng.model.encode_voice(pcm[:cap])
        ref = ref.detach().to(torch.float32).cpu().reshape(-1)
        return float(torch.dot(ref, emb) / (ref.norm() * emb.norm()).clamp_min(1e-12))

    def _cut_batches(self, todo: List[int], work, bs: int) -> List[List[int]]:
        """The engine decodes any number of segments on its ``batch_size`` rows with continuous batching (finished rows are handed
        to the next queued segment, Engine.synthesize): the whole work list goes down in one call instead of being cut here."""
        if len(todo) > bs and bs <= 64:
            return [list(todo)]
        return super()._cut_batches(todo, work, bs)

    # ---------------------------------------------------------------- provider contract
    def _generate_audio(self, text: Union[str, List[str]], **kwargs) -> Union[torch.Tensor, List[torch.Tensor]]:
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        eng = self._load_engine()
        token = kwargs.get("cancellation_token")
        flag = C.c_int32(0)
        stop = threading.Event()
        watcher = None
        if token is not None:                                   # forward CancellationToken.cancel() into the native frame loop
            def watch():
                while not stop.wait(0.02):
                    if token.is_cancelled():
                        flag.value = 1
                        return
            watcher = threading.Thread(target=watch, daemon=True)
            watcher.start()
        try:
            with self._lock:
                self._ensure_voice(eng)
                wavs = eng.synthesize(texts, seed=int(self.seed), item_ids=kwargs.get("item_ids"), cancel_flag=flag, stats=self.last_schedule)
        except _native.CancelledError as e:
            raise CancelledException(str(e))
        finally:
            stop.set()
            if watcher is not None:
                watcher.join(timeout=1.0)
        return wavs[0] if single else wavs

    def generate(self, texts, output_path: Optional[str] = None, cancellation_token: Optional[CancellationToken] = None,
                 format: str = "wav", speed: float = 1.0, pitch_semitones: float = 0.0, progress_callback: Optional[Callable[[str], None]] = None):
        """``BaseTTS.generate`` (base_tts.py:960-1101), inherited unchanged - except on the WORKER ranks of a data-parallel run
        (DataParallelPipeline): those do their share of the texts, deliver it to rank 0 and return nothing (``None`` for one
        text, a list of ``None`` for a list) without writing any file; rank 0 returns and saves everything."""
        dp = self._dp()
        if dp is None or dp[1] == 0:
            return super().generate(texts, output_path, cancellation_token, format, speed, pitch_semitones, progress_callback)
        if format not in ("wav", "mp3", "flac", "ogg"):          # the check rank 0 makes before it enters the pipeline (base_tts.py:996-999)
            from .api import FormatConversionError
            raise FormatConversionError(f"Unsupported format '{format}'. Supported: flac, mp3, ogg, wav")
        single = isinstance(texts, str)
        try:
            self._run_pipeline([texts] if single else list(texts), cancellation_token or CancellationToken(), progress_callback)
        except CancelledException as e:
            logger.warning(f"Generation cancelled: {e}")
        except ValueError:
            raise
        except Exception as e:  # noqa: BLE001  (as the inherited generate does: log, return nothing)
            logger.error(f"Error in TTS generation: {e}")
        return None if single else [None] * len(texts)

    def stream(self, text: str, cancellation_token: Optional[CancellationToken] = None, speed: float = 1.0,
               pitch_semitones: float = 0.0):
        """``BaseTTS.stream`` (base_tts.py:1132-1190): one result per segment, no crossfade, failed segments skipped - with the
        same audio per segment, produced in two calls instead of one per segment: the FIRST segment alone (time to first
        audio = one short decode), then ALL the others in one batched call while the caller is busy with the first.
        Every segment keeps RNG stream 0, which is what a one-segment ``_generate_audio`` call uses, so the results equal the
        sequential loop's."""
        from .api import GenerationResult
        token = cancellation_token or CancellationToken()
        segments = self._split_text_into_segments(self._apply_phonetic_mapping(text), self._compute_max_chars())
        if int(getattr(self, "stream_chunk_frames", 0) or 0) > 0:
            yield from self._stream_chunks(segments, token, speed, pitch_semitones)
            return

        def finish(raw: torch.Tensor):
            audio = self._post_process_audio(raw)
            audio = self._apply_fades(self._remove_dc_offset(self._trim_silence(audio, True, True)), True, True)
            if speed != 1.0 or pitch_semitones != 0.0:
                audio = self._apply_speed_pitch(audio, speed, pitch_semitones)
            n_samples = audio.shape[-1] if audio.dim() == 2 else audio.numel()
            return GenerationResult(audio=audio, sample_rate=self.sample_rate, duration_sec=n_samples / self.sample_rate,
                                    segments_count=1, format="wav")

        for first, group in ((0, segments[:1]), (1, segments[1:])):
            if not group or token.is_cancelled():
                continue
            self._set_seeds()
            try:
                raws: List[Optional[torch.Tensor]] = list(self._generate_audio(list(group), item_ids=[0] * len(group), cancellation_token=token))
            except CancelledException:
                return
            except Exception as first_error:  # noqa: BLE001  (the batch failed: let every segment fail or pass on its own)
                raws = []
                for seg in group:
                    try:
                        raws.append(self._generate_audio(seg, cancellation_token=token))
                    except CancelledException:
                        return
                    except Exception as e:  # noqa: BLE001
                        logger.warning(f"Segment {first + len(raws) + 1} failed: {e} (batched call: {first_error})")
                        raws.append(None)
            for k, raw in enumerate(raws):
                if token.is_cancelled():
                    return
                if raw is None:
                    continue
                try:
                    yield finish(raw)
                except Exception as e:  # noqa: BLE001
                    logger.warning(f"Segment {first + k + 1} failed: {e}")

    # Opt-in SUB-SEGMENT streaming (an extension: the reference hands audio over per segment): with ``stream_chunk_frames`` = n > 0,
    # ``stream()`` yields the first n codec frames of a segment (n x 80 ms of audio) as soon as they are decoded and vocoded, then
    # the rest in pieces of ``stream_next_chunk_frames`` while the decode goes on - one GenerationResult per piece, to be played
    # back to back.  Loudness gain and DC offset are taken from a segment's first piece and kept (rt_stream_chunk); trims and
    # fades happen at the segment's two ends only.
    stream_chunk_frames = 0
    stream_next_chunk_frames = 36

    def _stream_chunks(self, segments, token, speed, pitch_semitones):
        from .api import GenerationResult
        eng = self._load_engine()
        params = self._post_params(0)
        for seg_idx, seg in enumerate(segments):
            if token.is_cancelled():
                return
            self._set_seeds()
            flag = C.c_int32(0)
            state = [0.0, 1.0]
            try:
                with self._lock:
                    self._ensure_voice(eng)
                    started = False                                  # audible audio of this segment has been handed out
                    for raw, last in eng.stream_wav(seg, seed=int(self.seed), item_id=0, first_chunk=int(self.stream_chunk_frames),
                                                    chunk=int(self.stream_next_chunk_frames), cancel_flag=flag):
                        if token.is_cancelled():
                            flag.value = 1
                            return
                        audio = self._native_ctx().stream_chunk(params, raw, state, first=not started, last=last) if raw.numel() else raw
                        if not started and raw.numel():
                            if state[1] > 0.0:
                                started = True
                            else:                                    # a chunk of lead-in silence: nothing measured, nothing to play -
                                continue                             # the next chunk is the segment's first (gain, trim, fade-in)
                        if audio.numel() == 0:
                            continue
                        if speed != 1.0 or pitch_semitones != 0.0:
                            audio = self._apply_speed_pitch(audio, speed, pitch_semitones)
                        n_samples = audio.shape[-1] if audio.dim() == 2 else audio.numel()
                        yield GenerationResult(audio=audio, sample_rate=self.sample_rate, duration_sec=n_samples / self.sample_rate,
                                               segments_count=1, format="wav")
            except _native.CancelledError:
                return
            except Exception as e:  # noqa: BLE001  (as the reference: a failed segment is skipped, base_tts.py:1166-1168)
                logger.warning(f"Segment {seg_idx + 1} failed: {e}")

    @property
    def sample_rate(self) -> int:
        return 24000 if self._engine is None else self._engine.cfg.sample_rate

    def close(self) -> None:
        with self._lock:
            if self._engine is not None:
                self._engine.close()
                self._engine = None
            if self._ctx is not None:
                self._ctx.close()
                self._ctx = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    @classmethod
    def provider_info(cls) -> ProviderInfo:
        return ProviderInfo(name=PROVIDER_NAME, supports_voice_cloning=True,
                            supported_languages=["English", "Chinese", "Japanese", "Korean"],
                            builtin_voices=[VoiceInfo(id=v, name=v, language="English") for v in BUILTIN_VOICES])


def register(name: str = PROVIDER_NAME) -> str:
    """``TTSFactory.register_provider(name, MI355XQwenTTS)`` on whichever host API is active (see api.py)."""
    TTSFactory.register_provider(name, MI355XQwenTTS)
    return name
