"""N > 1 path on CPU: world_size-2 gloo.  The hot path has no data-path collective (front-ends are
independent); what is exercised is the sharding rule, max-over-ranks timing and the all-gather of
detected-peak lists that bench.py / a multi-GPU scan perform over RCCL on the real node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allgather_peaks_torch(dist, torch, multigpu, freqs_hz, device, cap=None):
    """the product's record format (multigpu.pack_peaks / unpack_peaks) over torch.distributed -- "gloo" here, as "nccl"
    (== RCCL) would carry it on a node; the product itself exchanges through the C ABI (rcf_allgather_peaks) or the host
    rendezvous and has no torch in it"""
    cap = multigpu.PEAK_CAP if cap is None else cap
    mine = torch.from_numpy(multigpu.pack_peaks(freqs_hz, cap)).to(device)
    gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    return multigpu.unpack_peaks([g.cpu().numpy() for g in gathered])


def _max_over_ranks(dist, torch, seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from rcf import multigpu, scan
    from oracle import peaks as P
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = multigpu.sources_for_rank(8, world, rank)
        # each rank owns a spectrum slice: a synthetic float32 spectrum with rank-specific carriers
        rng = np.random.default_rng(40 + rank)
        n, fs = 16384, 2.4e6
        fc = 855e6 + rank * fs
        x = rng.normal(-40.0, 5.5, n)
        for c in (3000 + 500 * rank, 9000 + 300 * rank):
            x += 300 * np.exp(-0.5 * ((np.arange(n) - c) / 30.0) ** 2)
        x = x.astype(np.float32)
        lines, freqs = scan.peak_detect(x, fs, fc)              # product host picker (librcf, no GPU needed)
        l_ref, f_ref = P.peak_detect_scipy(x, fs, fc)
        assert list(lines) == list(l_ref) and freqs == f_ref and len(freqs) == 2
        everyone = _allgather_peaks_torch(dist, torch, multigpu, freqs, "cpu")
        tmax = _max_over_ranks(dist, torch, 1.0 + rank, "cpu")
        q.put((rank, mine, freqs, everyone, tmax))
    finally:
        dist.destroy_process_group()


def test_world2_peak_allgather_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, f0, all0, t0), (r1, s1, f1, all1, t1) = res
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]
    assert all0 == all1 == sorted(f0 + f1) and len(all0) == 4
    assert t0 == t1 == 2.0


def test_routing_and_packing_rules():
    from rcf import multigpu
    centers = [851e6 + 25e6 * g for g in range(8)]
    rates = [25e6] * 8
    assert multigpu.route_frequency(851e6 + 1e6, centers, rates) == 0
    assert multigpu.route_frequency(851e6 + 13e6, centers, rates) == 1       # nearest centre wins
    assert multigpu.route_frequency(700e6, centers, rates) is None
    rec = multigpu.pack_peaks([5, 3, 9], cap=8)
    assert rec.tolist() == [3, 5, 3, 9, -1, -1, -1, -1, -1]
    assert multigpu.unpack_peaks([rec, multigpu.pack_peaks([], cap=8)]) == [3, 5, 9]


def _host_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rcf import multigpu
    g = multigpu.HostGroup(rank, world, "127.0.0.1", port, timeout=60)
    try:
        uid = g.broadcast(bytes(range(128)) if rank == 0 else None)      # what init_comm does with the RCCL id
        g.barrier()
        tmax = g.max(1.0 + rank)
        everyone = multigpu.allgather_peaks_host(g, [851000000 + 12500 * rank, 852000000 - rank], cap=8)
        parts = g.all_gather(b"r%d" % rank * (rank + 1))
        q.put((rank, uid, tmax, everyone, parts))
    finally:
        g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_host_group_rendezvous_without_torch(world):
    """The torch-free transport bench.py uses for N > 1: TCP star on 127.0.0.1 -- RCCL id broadcast, barrier,
    max over ranks and the host fallback of the peak all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_host_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = sorted([851000000 + 12500 * r for r in range(world)] + [852000000 - r for r in range(world)])
    for rank, uid, tmax, everyone, parts in res:
        assert uid == bytes(range(128))
        assert tmax == float(world)
        assert everyone == want
        assert parts == [b"r%d" % r * (r + 1) for r in range(world)]
