"""Multi-GPU glue: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm,
"gloo" in CPU tests).  The hot path shards by independent SDR front-ends / spectrum slices -- the
reference already runs one channelizer process per SDR (systemd radiocapture-channelizer@<i>,
/root/reference/rc_frontend/receiver.py:67-70) -- so there is NO data-path collective.  The only
exchange is the all-gather of detected-peak lists after a scan (BASELINE.json configs[4])."""
from __future__ import annotations

import numpy as np

PEAK_CAP = 1024


def sources_for_rank(n_sources: int, world: int, rank: int):
    """Front-end g -> rank g % world (one per GPU when n_sources == world)."""
    return [s for s in range(n_sources) if s % world == rank]


def route_frequency(freq, centers, samp_rates):
    """Same rule as receiver.connect_channel_xlat (receiver.py:288-291): the source whose centre is
    nearest among those with abs(f - center) < samp_rate / 2; None when out of band."""
    best, dist_ = None, None
    for g, (c, r) in enumerate(zip(centers, samp_rates)):
        d = abs(freq - c)
        if d < r / 2 and (dist_ is None or d < dist_):
            best, dist_ = g, d
    return best


def pack_peaks(freqs_hz, cap=PEAK_CAP):
    """Fixed-capacity record: [count, f0, f1, ...] int64, -1 padded (latency-bound 8 KiB payload)."""
    rec = np.full(cap + 1, -1, dtype=np.int64)
    n = min(len(freqs_hz), cap)
    rec[0] = n
    rec[1:1 + n] = np.asarray(freqs_hz[:n], dtype=np.int64)
    return rec


def unpack_peaks(records):
    out = []
    for rec in records:
        n = int(rec[0])
        out.extend(int(v) for v in rec[1:1 + n])
    return sorted(out)


def allgather_peaks(dist, torch, freqs_hz, device, cap=PEAK_CAP):
    """Every rank ends with the global sorted list of detected peak frequencies."""
    mine = torch.from_numpy(pack_peaks(freqs_hz, cap)).to(device)
    world = dist.get_world_size()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return unpack_peaks([g.cpu().numpy() for g in gathered])


def max_over_ranks(dist, torch, seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
