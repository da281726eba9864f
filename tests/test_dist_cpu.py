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
