"""N > 1 path on the CPU: world_size-2 gloo run of the two collectives the provider uses
(variable-length waveform gather, conditioning broadcast) and of the work sharding."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from rho_tts_amd import dist as D


def test_shard_items_balances_and_round_trips():
    costs = [24, 6, 7, 19, 11, 12, 6, 23, 9, 15, 8]
    sh = D.shard_items(costs, 4)
    assert sorted(i for s in sh for i in s) == list(range(len(costs)))
    loads = [sum(costs[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= max(costs)
    assert all(s == sorted(s) for s in sh)
    per_rank = [[f"r{r}k{k}" for k in range(len(s))] for r, s in enumerate(sh)]
    back = D.unshard(per_rank, sh, len(costs))
    for r, s in enumerate(sh):
        for k, i in enumerate(s):
            assert back[i] == f"r{r}k{k}"
    assert D.shard_items([], 3) == [[], [], []]
    assert D.shard_items([5.0], 2) == [[0], []]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeModel:
    def __init__(self, rank):
        self.rank, self.n, self.blob = rank, 0, None

    def export_voice(self):
        return torch.arange(10, dtype=torch.float32).to(torch.bfloat16)

    def prefix_len(self):
        return 5

    def import_voice(self, n, blob):
        self.n, self.blob = n, blob.clone()


class _FakeEngine:
    def __init__(self, rank):
        self.device = torch.device("cpu")
        self.model = _FakeModel(rank)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wavs = [torch.full((100 + 10 * rank + k,), float(rank * 10 + k)) for k in range(2 + rank)]
        if rank == 1:
            wavs[1] = None                                                 # a failed item travels as a hole
        got = D.gather_waveforms(wavs, dist, dst=0, device=torch.device("cpu"))
        eng = _FakeEngine(rank)
        D.broadcast_voice(eng, dist, src=0)
        ok = True
        if rank == 0:
            ok = len(got) == world and [len(g) for g in got] == [2, 3]
            ok = ok and got[1][1] is None and got[1][2].numel() == 112 and float(got[1][2][0]) == 12.0
            ok = ok and got[0][1].numel() == 101 and float(got[0][1][5]) == 1.0
        else:
            ok = got is None and eng.model.n == 5 and eng.model.blob.tolist() == list(range(10))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gather_and_broadcast_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_bucket_batches_and_padding_efficiency():
    import numpy as np
    rng = np.random.default_rng(789)
    words = rng.integers(6, 25, 512)
    frames = [int(round(12.5 * 0.35 * w)) for w in words]                  # C4: 26..105 frames
    # one batch or less: arrival order, untouched
    assert D.bucket_batches(frames[:20], 32) == [list(range(20))]
    assert D.bucket_batches([], 32) == []
    b = D.bucket_batches(frames, 32)
    assert sorted(i for x in b for i in x) == list(range(512)) and all(x == sorted(x) for x in b) and all(len(x) == 32 for x in b)
    # longest first, batches do not interleave in length
    mins, maxs = [min(frames[i] for i in x) for x in b], [max(frames[i] for i in x) for x in b]
    assert all(mins[k] >= maxs[k + 1] for k in range(len(b) - 1))
    arrival = [list(range(i, i + 32)) for i in range(0, 512, 32)]
    e_sorted, e_arrival = D.padding_efficiency(frames, b), D.padding_efficiency(frames, arrival)
    assert e_sorted > 0.93 and e_arrival < 0.70, (e_sorted, e_arrival)     # 512 texts on one rank: 16 buckets
    # the 8-rank plan: every item exactly once, shards balanced, per-rank batches are buckets of the shard
    shards, plans = D.plan_corpus(frames, 8, 32)
    assert sorted(i for s in shards for i in s) == list(range(512))
    assert all(sorted(i for x in pl for i in x) == sh for sh, pl in zip(shards, plans))
    loads = [sum(frames[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(frames)
    # 64 texts per rank in two batches of 32 can only reach ~0.76 (a uniform 26..105 spread cut in two); smaller batches buy it back
    e8 = D.padding_efficiency(frames, [x for pl in plans for x in pl])
    e8_small = D.padding_efficiency(frames, [x for pl in D.plan_corpus(frames, 8, 8)[1] for x in pl])
    assert 0.70 < e8 < 0.85 and e8_small > 0.90, (e8, e8_small)


def _corpus_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = [30 + (7 * i) % 50 for i in range(21)]                     # the same corpus plan on every rank
        shards, plans = D.plan_corpus(frames, world, 4)
        mine = {}
        for batch in plans[rank]:                                           # "decode" each bucket: value = corpus index, length = frames
            for i in batch:
                mine[i] = None if i == 13 else torch.full((frames[i],), float(i))     # item 13 fails: travels as a hole
        outs = [mine[i] for i in shards[rank]]                              # shard order
        got = D.gather_waveforms(outs, dist, dst=0, device=torch.device("cpu"))
        ok = True
        if rank == 0:
            back = D.unshard(got, shards, len(frames))
            for i, w in enumerate(back):
                if i == 13:
                    ok = ok and w is None
                else:
                    ok = ok and w is not None and w.numel() == frames[i] and float(w[0]) == float(i)
        else:
            ok = got is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_corpus_plan_gather_unshard_world_size_2_gloo():
    """C4 on two ranks: shard by length, bucket, gather, unshard - corpus order restored, a failed item stays a hole."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_corpus_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
