"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards by independent units — every text (indeed every segment) is generated independently
(base_tts.py:726-954 keeps no cross-item state) — so there is no data-path collective in the decode.
Exactly two collectives exist:
  * broadcast of the voice conditioning (the prefix KV blob, tens of MB, once per voice)
  * gather of the finished 24 kHz waveforms to rank 0 (96 KB per audio-second)
The reference has no distributed code at all; nothing here translates an NCCL call pattern.
All functions also run on the gloo backend with CPU tensors, which is how the CPU tests cover them.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def shard_items(costs: Sequence[float], world: int) -> List[List[int]]:
    """Length-balanced deal: sort by estimated cost (descending) and give each item to the least-loaded rank.
    Returns the item indices of every rank, each list in ascending original order."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], len(out[k]), k))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def bucket_batches(costs: Sequence[float], batch: int) -> List[List[int]]:
    """Cut items into batches of ``batch`` indices.  A batch decodes until its longest member ends, so when there is more
    than one batch the items are bucketed by cost (longest first, ties by index) - each batch then holds items of similar
    length - and the indices inside a batch stay ascending.  One batch or less: arrival order."""
    n = len(costs)
    if n <= batch:
        return [list(range(n))] if n else []
    order = sorted(range(n), key=lambda i: (-costs[i], i))
    return [sorted(order[b0:b0 + batch]) for b0 in range(0, n, batch)]


def padding_efficiency(costs: Sequence[float], batches: Sequence[Sequence[int]]) -> float:
    """Useful row-frames / decoded row-frames = sum(frames) / sum(batch max x batch size)."""
    padded = sum(max(costs[i] for i in b) * len(b) for b in batches if len(b))
    return float(sum(costs[i] for b in batches for i in b)) / padded if padded else 1.0


def plan_corpus(costs: Sequence[float], world: int, batch: int) -> Tuple[List[List[int]], List[List[List[int]]]]:
    """The C4 plan: deal the corpus over ``world`` ranks by cost (shard_items), then bucket each rank's share
    into batches.  Returns (shards, batches): shards[r] = rank r's item indices (ascending), batches[r] = its batches as
    lists of GLOBAL item indices."""
    shards = shard_items(costs, world)
    plans = []
    for sh in shards:
        local = bucket_batches([costs[i] for i in sh], batch)
        plans.append([[sh[k] for k in b] for b in local])
    return shards, plans


def broadcast_voice(engine, dist, src: int = 0, comm_device=None) -> None:
    """Rank ``src`` has computed the voice prefix (Engine.set_voice); every other rank imports its KV blob.
    ``comm_device``: where the collective runs (the GPU for RCCL; "cpu" to rehearse on gloo)."""
    rank = dist.get_rank()
    dev = comm_device or engine.device
    meta = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        blob = engine.model.export_voice().to(dev)
        meta[0], meta[1] = engine.model.prefix_len(), blob.numel()
    dist.broadcast(meta, src=src)
    if rank != src:
        blob = torch.empty(int(meta[1]), dtype=torch.bfloat16, device=dev)
    dist.broadcast(blob, src=src)
    if rank != src:
        engine.model.import_voice(int(meta[0]), blob.to(engine.device))
        if getattr(engine, "voice", None) is None:          # the conditioning now lives in the imported prefix KV
            from .voice import VoiceConditioning
            engine.voice = VoiceConditioning("(imported)", None, None, [], None)


_PINNED: dict = {}


def _to_host(t: torch.Tensor, copy: bool = True) -> torch.Tensor:
    """Device -> host through cached PINNED staging buffers (a pageable ``.cpu()`` runs at a third of the PCIe rate).
    ``copy=True``: the result is a fresh pageable tensor (one more pass over the bytes on the host).  ``copy=False``: the result
    is a view of the staging buffer itself - two buffers alternate per (dtype, device), so it stays valid until the call AFTER the
    next one; for callers that consume a step's waveforms before the step after next (bench.py: 86 MB per step on rank 0 of an
    8-GPU job, where the extra host pass costs more than the PCIe copy)."""
    if not t.is_cuda or t.numel() == 0:
        return t.cpu()
    key = (t.dtype, t.device.index)
    ring = _PINNED.setdefault(key, {"bufs": [None, None], "next": 0})
    i = ring["next"]
    ring["next"] = 1 - i
    buf = ring["bufs"][i]
    if buf is None or buf.numel() < t.numel():
        buf = torch.empty(max(t.numel(), 1 << 20), dtype=t.dtype, pin_memory=True)
        ring["bufs"][i] = buf
    view = buf[: t.numel()]
    view.copy_(t.reshape(-1), non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return (view.clone() if copy else view).reshape(t.shape)


def waveforms_to_host(wavs: Sequence[torch.Tensor], copy: bool = True) -> List[torch.Tensor]:
    """One packed device buffer, one pinned copy, then views per waveform (instead of one pageable copy per item).
    ``copy=False``: views of the pinned staging buffer, valid until the call after next (_to_host)."""
    live = [w for w in wavs if w is not None and w.numel()]
    if not live or not live[0].is_cuda:
        return [None if w is None else w.cpu() for w in wavs]
    host = _to_host(torch.cat([w.reshape(-1) for w in live]), copy)
    out, o = [], 0
    for w in wavs:
        if w is None:
            out.append(None)
        elif w.numel() == 0:
            out.append(torch.zeros(0))
        else:
            out.append(host[o: o + w.numel()])
            o += w.numel()
    return out


def gather_waveforms(wavs: Sequence[torch.Tensor], dist, dst: int = 0, device=None, copy: bool = True) -> Optional[List[List[torch.Tensor]]]:
    """Variable-length gather: returns on ``dst`` a list (per rank) of lists of CPU float32 waveforms, else None.
    One length exchange plus one padded payload gather; ``None`` items travel as length -1.  ``copy=False``: the waveforms are
    views of pinned staging memory, valid until the call after next (_to_host)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or (wavs[0].device if len(wavs) and wavs[0] is not None else torch.device("cpu"))
    lens = torch.tensor([(-1 if w is None else int(w.numel())) for w in wavs], dtype=torch.int64, device=device)
    n_local = torch.tensor([lens.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local)
    max_items = max(1, int(max(int(c) for c in counts)))       # (a collective on empty tensors is not something every backend accepts)
    lens_pad = torch.full((max_items,), -1, dtype=torch.int64, device=device)
    lens_pad[: lens.numel()] = lens
    all_lens = [torch.empty_like(lens_pad) for _ in range(world)]
    dist.all_gather(all_lens, lens_pad)
    totals = [int(l.clamp(min=0).sum()) for l in all_lens]
    cap = max(1, max(totals))
    # (one concatenation + one pad instead of a copy per waveform: a rank's 32 waveforms are 32 small launches otherwise)
    live = [w.reshape(-1).to(device=device, dtype=torch.float32) for w in wavs if w is not None and w.numel()]
    mine = torch.cat(live) if live else torch.zeros(0, dtype=torch.float32, device=device)
    payload = mine if mine.numel() == cap else torch.cat([mine, torch.zeros(cap - mine.numel(), dtype=torch.float32, device=device)])
    if rank == dst:
        bufs = [torch.empty(cap, dtype=torch.float32, device=device) for _ in range(world)]
        dist.gather(payload, gather_list=bufs, dst=dst)
        out: List[List[torch.Tensor]] = []
        # every rank's bytes cross PCIe in ONE pinned copy
        host_all = _to_host(torch.cat([bufs[r][: totals[r]] for r in range(world)]), copy)
        base = 0
        for r in range(world):
            items, o = [], base
            for n in all_lens[r].tolist()[: int(counts[r])]:
                if n < 0:
                    items.append(None)
                else:
                    items.append(host_all[o: o + n])
                    o += n
            base += totals[r]
            out.append(items)
        return out
    dist.gather(payload, gather_list=None, dst=dst)
    return None


def gather_rows(rows: torch.Tensor, dist, dst: int = 0, device=None) -> Optional[List[torch.Tensor]]:
    """Per-item records beside the waveforms (segment counts, decay ratios, validation scores): ``rows`` is [n_local, k]
    float64; returns on ``dst`` one [n_r, k] CPU tensor per rank, else None.  One count exchange + one padded gather."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or rows.device
    k = int(rows.shape[1])
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local)
    cap = max(1, max(int(c) for c in counts))
    pad = torch.zeros(cap, k, dtype=torch.float64, device=device)
    pad[: rows.shape[0]] = rows.to(device=device, dtype=torch.float64)
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=bufs, dst=dst)
        return [bufs[r][: int(counts[r])].cpu() for r in range(world)]
    dist.gather(pad, gather_list=None, dst=dst)
    return None


def agree(dist, device, code: int) -> int:
    """Largest status code over the ranks (one tiny all-reduce): every rank learns whether any rank failed BEFORE the next
    collective, so that a failure on one rank ends the call on all of them instead of leaving the others waiting."""
    t = torch.tensor([int(code)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def min_max(dist, device, value: int) -> Tuple[int, int]:
    """(smallest, largest) of one integer per rank - a limit every rank must apply alike (take the min), a digest that must be
    the same everywhere (min == max), the furthest stage any rank reached (max).  One two-word all-reduce."""
    t = torch.tensor([-int(value), int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return -int(t[0].item()), int(t[1].item())


def broadcast_ints(dist, device, values: Sequence[int], src: int = 0) -> List[int]:
    """Rank ``src``'s integers on every rank (the seed a call starts from, the clock reading its retry seeds derive from)."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    return [int(v) for v in t.tolist()]


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """One process per GPU under ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment): initialise the process group - "nccl" IS RCCL on ROCm - BEFORE anything touches the GPU, and return
    (rank, local_rank, world).  Build the provider with ``device=f"cuda:{local_rank}"`` afterwards."""
    import os

    import torch.distributed as td
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not td.is_initialized():
        if backend == "nccl":
            td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def unshard(per_rank: Sequence[Sequence], shards: Sequence[Sequence[int]], n_total: int) -> List:
    """Inverse of ``shard_items``: place rank r's k-th result at original index shards[r][k]."""
    out = [None] * n_total
    for r, idxs in enumerate(shards):
        for k, i in enumerate(idxs):
            out[i] = per_rank[r][k]
    return out
