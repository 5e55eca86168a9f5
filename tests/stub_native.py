"""A stand-in for rcf.native with the surface bench.py's multi-rank protocol touches (RCF_BENCH_NATIVE=stub_native):
lets the CPU suite run the rank launcher, the host rendezvous, max-over-ranks timing and the peak gather of
`python bench.py --gpus 2` end to end without a GPU.  It computes nothing: a commit is a short sleep."""
import os
import time

import numpy as np

WIN_HAMMING, WIN_BLACKMAN, WIN_KAISER, WIN_BLACKMAN_HARRIS = 0, 2, 4, 5
T_FIR, T_PFB, T_FIR_DERIVED, T_DISC, T_SCAN_FFT, T_SCAN_MOVSUM, T_HISTORY, T_FIR_MFMA, T_AUDIO, T_TAPS = range(10)


def device_count():
    return int(os.environ.get("STUB_DEVICES", "2"))


def design_low_pass_2(gain, fs, fc, tw, att, window=WIN_HAMMING):
    return np.ones(33, dtype=np.float32) / 33


def comm_unique_id():
    if os.environ.get("STUB_RCCL") == "1":              # a pretend ncclUniqueId: 128 bytes, like the real one
        return bytes(range(128))
    raise RuntimeError("stub: no RCCL")


def peak_frequency(index, fs, n, center):
    return int(index * fs / n - fs / 2 + center)


class Frontend:
    def __init__(self, samp_rate, center_freq=0.0, device=0, block_capacity=0, hist_capacity=0, out_capacity=0):
        self.device, self.samples_in, self._launches, self._chans = device, 0, 0, 0
        if os.environ.get("STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("stub: this rank was told to fail")

    def pfb_open(self, nb, D, taps):
        self.nb = nb

    def pfb_chan_open(self, bin_, cr, delta):
        self._chans += 1
        return self._chans

    def ingest_write(self, x, at):
        pass

    def commit(self, n):
        self.samples_in += n
        self._launches += 1
        time.sleep(0.0005 * (1 + int(os.environ.get("RANK", "0"))))      # rank 1 is the slower one

    def sync(self):
        pass

    def timing_enable(self, on=True, classes=None):
        pass

    def timing_stride(self, n):
        pass

    def timing_read(self, cls, reset=False):
        n, self._launches = self._launches, 0
        return 0.1 * max(n, 1), max(n, 1)

    def chan_produced(self, c):
        return 1

    def chan_read_fm(self, c, gain, max_samples=0):
        return np.zeros(8, dtype=np.float32)

    def scan_start(self, n, f, l):
        pass

    def scan_frames_done(self):
        return 1 << 30

    def scan_find_peaks(self, cap=1024):
        return np.array([100 + self.device], dtype=np.int64), 0.0, None

    # STUB_RCCL=1: a pretend communicator (a second host rendezvous) with the semantics of rcf_comm_init /
    # rcf_allgather_peaks / rcf_allreduce_max, so that the RCCL branch of bench.py's protocol -- id broadcast, every
    # rank joins, the proof before anything is timed, barrier and gather through the communicator -- runs on CPU
    _comm = None

    def comm_init(self, rank, n, uid=None):
        if os.environ.get("STUB_RCCL") != "1" or os.environ.get("STUB_RCCL_INIT_FAILS") == "1":
            raise RuntimeError("stub: no RCCL")
        assert uid == bytes(range(128))
        if os.environ.get("STUB_RCCL_HANGS") == "1":
            time.sleep(120)                               # a join that never completes
        from rcf import multigpu
        self._comm = multigpu.HostGroup(rank, n, "127.0.0.1", int(os.environ["MASTER_PORT"]) + 202)

    def comm_destroy(self):
        if self._comm is not None:
            self._comm.close()
            self._comm = None

    def comm_size(self):
        return self._comm.world if self._comm is not None else 1

    def allgather_peaks(self, mine, cap=1024):
        mine = np.ascontiguousarray(mine, dtype=np.int64)[:cap]
        return [np.frombuffer(b, dtype=np.int64).copy() for b in self._comm.all_gather(mine.tobytes())]

    def allreduce_max(self, value):
        return self._comm.max(float(value)) if self._comm is not None else float(value)

    def close(self):
        pass


# ---- what bench.py's per-GPU real-time point touches (N > 1): groups of front-ends and their native pumps
FMT_CF32, FMT_U8, FMT_S8, FMT_S16 = 0, 1, 2, 3


def channel_params(fs, cr, rule=0):
    return int(fs / cr) // 2, 2909


class PinnedArray:
    def __init__(self, n, dtype):
        self.array = np.zeros(int(n), dtype=dtype)

    def free(self):
        self.array = None


class Group:
    def __init__(self, frontends):
        self.frontends = list(frontends)

    def __len__(self):
        return len(self.frontends)

    def close(self):
        pass


class Pump:
    """finishes at once: every block of every member judged, nothing late (STUB_RT_MISS=<rank>: that rank's pumps miss)"""

    def __init__(self, group, rings, block_samples, samp_rate, subscriptions, n_blocks=0, warm_blocks=0, **kw):
        assert len(rings) == len(group) and all(len(r) >= 2 * block_samples for r in rings)
        self.n, self.judged = len(group), len(group) * (n_blocks - warm_blocks)
        self.late = 3 if os.environ.get("STUB_RT_MISS") == os.environ.get("RANK", "0") else 0

    def stats(self):
        return dict(blocks_done=self.judged, blocks_judged=self.judged, late=self.late, overruns=0, group_blocks=max(1, self.judged // 4),
                    max_batch=4, samples_out=0, latency_ms_p50=0.5, latency_ms_p99=1.0 + self.late, latency_ms_max=2.0,
                    host_plan_ms=1.0, host_wait_ms=1.0, elapsed_s=0.1, max_plan_ms=0.1, max_wait_ms=0.1,
                    max_sleep_overshoot_ms=0.0, slow_plans=0, slow_waits=0, slow_sleeps=0, rt_priority_granted=0, running=0, error=0)

    def stop(self):
        pass
