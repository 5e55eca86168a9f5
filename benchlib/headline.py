"""The timed configuration: communicator set-up, the W warm-up + K timed steps between two barriers, the roofline of
the filterbank launch, the per-rank record and (cfg5 / N > 1) the scan of the rank's slice + the peak-list all-gather."""
import json
import os
import struct
import sys
import threading
import time

import numpy as np

from .common import FS, NB, N_ACTIVE, HBM_PEAK_GBS, ROOT, proto_taps, cgroup_cpu_stat
from .sustained import sustained_leg
from .traffic import measure_traffic_live

SCAN_N, SCAN_F, SCAN_L = 1 << 20, 1000, 100


class Ranks:
    """rank / world / device of this process and, for N > 1, the host rendezvous + the RCCL communicator on `fe`"""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if "RCF_BENCH_DEVICE" in os.environ:             # two ranks on ONE GPU: exercises the N > 1 code on a 1-GPU box
            self.local_rank = int(os.environ["RCF_BENCH_DEVICE"])
        self.n_gpus = self.world if self.world > 1 else 1
        self.group = None
        self.use_rccl = os.environ.get("RCF_BENCH_TRANSPORT", "rccl") == "rccl"
        self.rccl_ranks, self.rccl_proof = 0, None

    def join(self, fe, native, multigpu):
        """N > 1: host rendezvous next to MASTER_PORT, ncclCommInitRank on this rank's GPU, and the proof -- one
        ncclAllGather of the rank numbers and one ncclAllReduce(max) -- BEFORE anything is timed.  If any rank cannot
        join (no librccl, two ranks told to share one GPU, ...) every rank falls back to the host rendezvous for the
        barrier and the gather: the data path has no collective, so the measurement does not depend on it."""
        if self.world <= 1:
            return
        rank, world = self.rank, self.world
        self.group = group = multigpu.HostGroup(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                                int(os.environ.get("MASTER_PORT", "29500")) + 101)
        # a communicator that never comes up (a rank missing, a fabric problem) must not hang the run for an hour: if
        # the join + proof are not through in RCF_BENCH_RCCL_TIMEOUT seconds (default 300) this rank says so and exits
        done = threading.Event()

        def _watchdog(limit=float(os.environ.get("RCF_BENCH_RCCL_TIMEOUT", "300"))):
            if not done.wait(limit):
                print("bench.py: rank %d: communicator set-up / proof not finished after %.0f s -- giving up "
                      "(RCF_BENCH_TRANSPORT=host runs without RCCL)" % (rank, limit), file=sys.stderr, flush=True)
                os._exit(3)
        threading.Thread(target=_watchdog, daemon=True).start()
        if self.use_rccl:
            uid = None
            if group.rank == 0 and "RCF_BENCH_DEVICE" not in os.environ:
                try:
                    uid = native.comm_unique_id()
                except Exception as e:
                    print("note: RCCL unavailable on rank 0 (%s): host transport" % e, file=sys.stderr)
            uid = group.broadcast(uid if uid is not None else b"")
            ok = len(uid) == 128
            if ok:
                try:
                    fe.comm_init(group.rank, group.world, uid)
                except Exception as e:
                    print("note: rank %d could not join the RCCL communicator (%s)" % (rank, e), file=sys.stderr)
                    ok = False
            self.use_rccl = all(p == b"1" for p in group.all_gather(b"1" if ok else b"0"))
            if not self.use_rccl:
                fe.comm_destroy()
        if self.use_rccl:
            self.rccl_ranks = fe.comm_size()
            parts = fe.allgather_peaks(np.array([rank], dtype=np.int64), multigpu.PEAK_CAP)   # the real gather's capacity
            seen = [int(p[0]) if len(p) else -1 for p in parts]
            top = fe.allreduce_max(float(rank))
            if self.rccl_ranks != world or seen != list(range(world)) or top != float(world - 1):
                raise RuntimeError("RCCL proof failed on rank %d: comm size %d of %d, all-gather %s, all-reduce max %s"
                                   % (rank, self.rccl_ranks, world, seen, top))
            self.rccl_proof = {"allgather_of_rank_numbers": seen, "allreduce_max_of_rank_numbers": top,
                               "when": "before the warm-up steps"}
        done.set()

    def barrier_max(self, fe, v=0.0):
        """barrier + device sync on every rank, max of v over ranks"""
        fe.sync()
        if self.group is None:
            return v
        return fe.allreduce_max(v) if self.use_rccl else self.group.max(v)

    def gather_json(self, obj):
        return sorted((json.loads(b.decode("utf-8")) for b in self.group.all_gather(json.dumps(obj).encode("utf-8"))),
                      key=lambda r_: r_["rank"])


def rt_k_per_gpu(args, world):
    """N > 1: the front-end count of the one paced point every rank runs AT THE SAME TIME.  All ranks share the host: the
    pumps and the replayed sources of `world` legs must fit the container's CPU quota (about 32 front-ends per core was
    what a 16-core quota carried at N = 1), so K = min(--rt-k-per-gpu, quota x 32 / world)."""
    quota = cgroup_cpu_stat()[3]
    k = args.rt_k_per_gpu
    if quota:
        k = min(k, max(8, int(quota * 32 / world)))
    return k, quota


def run(args, R, native, multigpu, synth, numa):
    """-> (out, ctx): rank 0's record of the timed configuration (None on the other ranks) and what the untimed legs need"""
    rank, world, n_gpus, local_rank = R.rank, R.world, R.n_gpus, R.local_rank
    cfg5 = args.config == "cfg5"
    # cfg5 (BASELINE configs[4]): 8 spectrum slices of 25 Msps, a 512-bin bank each = 4096 channels at 200 Msps
    # aggregate; every rank scans its slice (N = 2^20, 1000 frames, 100-frame average: fft_vector.py:31-60) and
    # contributes <= 1024 peaks (fft_peak_detection.py:38-73) to the all-gather
    fs, nb, n_active = (25e6, 512, 0) if cfg5 else (FS, NB, N_ACTIVE)
    B = args.block
    assert B % nb == 0 and (not cfg5 or B % SCAN_N == 0)
    frames = B // nb
    out_cap = 1
    while out_cap < 2 * frames + 64:                  # two blocks of frames + the stage-2 channels' reach: the stage-2 launch
        out_cap <<= 1                                 # of block n rides in block n + 1's filterbank launch (rcf_set_stage2_lag)
    fe = native.Frontend(fs, 0.0, device=local_rank, block_capacity=B, hist_capacity=SCAN_N if cfg5 else 1 << 16,
                         out_capacity=out_cap)
    R.join(fe, native, multigpu)
    group, use_rccl = R.group, R.use_rccl
    barrier_max = lambda v=0.0: R.barrier_max(fe, v)
    taps = proto_taps(native, fs, nb)
    fe.pfb_open(nb, nb, taps)
    if cfg5:
        # the slice's stream: a 16-frame periodic tile with 12 scan-shaped carriers (SURVEY 8(d) cfg3's recipe at
        # 25 Msps: occupied widths 4-9 kHz = 170-380 bins of 23.8 Hz, inside find_peaks' [126, 1258] window)
        rng = np.random.default_rng(5000 + rank)
        centres = [40000 + 80000 * i + int(rng.integers(-3000, 3000)) for i in range(12)]
        scan_carriers = [(c, float(rng.uniform(4000, 9000)), 45.0) for c in centres]
        tile = synth.scan_stream(fs, SCAN_N, 16, scan_carriers, seed=5000 + rank)
        meta = {"carriers": [{"f_off": (k - nb // 2 + 0.5) * fs / nb * 0.9} for k in range(0, nb, nb // 32)]}
        chans = []
    else:
        tile, meta = synth.cfg2(n=1 << 20, seed=2002 if n_gpus == 1 else 4000 + rank, n_bins=nb, n_active=n_active)
        chans = [fe.pfb_chan_open(c["bin"] % nb, 12500, c["delta"]) for c in meta["carriers"]]

    # make the batch resident in both ping-pong buffers (not timed: "inputs already resident in HBM")
    for _ in range(2):
        for at in range(0, B, len(tile)):
            fe.ingest_write(tile[: min(len(tile), B - at)], at)
        fe.commit(B)
    fe.sync()

    tp = time.perf_counter()
    n_prewarm = 0
    while time.perf_counter() - tp < args.prewarm_seconds:
        for _ in range(32):
            fe.commit(B)
        n_prewarm += 32
    # HIP events only on the kernel the roofline reports, and only on every 4th launch of it.  The two events are
    # ATTACHED to the filterbank's dispatch (hipExtLaunchKernelGGL), not recorded around it: one barrier packet less
    # inside the measured interval.  A timed launch costs the step ~4 us, hence every 4th.  Switched on BEFORE the
    # warm-up steps, so that nothing but the barrier and one counter reset lies between them and the timed region.
    fe.timing_enable(True, classes=[native.T_PFB])
    time_every = args.time_every if args.steps >= 2 * args.time_every else 1     # a short run times every launch
    fe.timing_stride(time_every)
    for _ in range(args.warmup):
        fe.commit(B)
    barrier_max()
    fe.timing_read(native.T_PFB, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.commit(B)
    fe.sync()
    t1 = time.perf_counter()
    elapsed = barrier_max(t1 - t0)
    per_rank_ms = [(t1 - t0) / args.steps * 1e3] if group is None else \
        [struct.unpack("<d", p)[0] / args.steps * 1e3 for p in group.all_gather(struct.pack("<d", t1 - t0))]

    pfb_ms, pfb_n = fe.timing_read(native.T_PFB)
    fe.timing_stride(1)
    # ... and a second pass, NOT timed by the wall clock, in which EVERY launch of the same number of steps (at least
    # 20) carries its two events: the large-sample launch time beside the every-4th one of the timed region
    n_all = max(args.steps, 20)
    for _ in range(n_all):
        fe.commit(B)
    fe.sync()
    pfb_all_ms, pfb_all_n = fe.timing_read(native.T_PFB)
    # the FM channels' newest outputs, for the parity check against the oracle (done in the cpu_baseline leg)
    fm_check = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and chans:
        fm_check = {"taps": taps, "total_in": fe.samples_in, "carriers": meta["carriers"],
                    "fm": [fe.chan_read_fm(c, 1.0, max_samples=out_cap) for c in chans]}
    # per-kernel breakdown of the other launches: a few extra, untimed steps with every class instrumented
    fe.timing_enable(True)
    n_extra = min(args.steps, 5)
    for _ in range(n_extra):
        fe.commit(B)
    fe.sync()
    fe.timing_read(native.T_PFB)
    fir2_ms, _ = fe.timing_read(native.T_FIR_DERIVED)
    disc_ms, _ = fe.timing_read(native.T_DISC)
    hist_ms, _ = fe.timing_read(native.T_HISTORY)
    fe.timing_enable(False)
    if chans:
        assert fe.chan_produced(chans[0]) > 0        # the FM channels really produced output

    alg_bytes_pfb = 16.0 * B                          # 8 B read + 8 B written per input sample (critically sampled)
    # the stage-2 work of the previous block rides in the filterbank's launch (rcf_set_stage2_lag: 256-bin kernel only):
    # the launch then also moves that work's algorithmic bytes -- SURVEY 8(d): per active bin 8 B read per frame of its
    # stream, 8 B (IQ) + 4 B (fused discriminator) written per output at a third of the frame rate
    s2_rides = bool(chans) and nb == 256 and os.environ.get("RCF_S2_LAG", "1") != "0" and hasattr(fe, "set_stage2_lag")
    alg_bytes_s2 = len(chans) * (8.0 * (B // nb) + 12.0 * ((B // nb) // 3)) if s2_rides else 0.0
    alg_bytes = alg_bytes_pfb + alg_bytes_s2
    # ... and the filterbank kernel ALONE (the lag switched off for a pass of its own: every launch timed)
    pfb_alone_ms = pfb_alone_n = None
    if s2_rides:
        fe.set_stage2_lag(False)
        for _ in range(args.alone_warm):              # (the read-backs above idled the queue: ~15 ms of work to settle)
            fe.commit(B)
        fe.timing_enable(True, classes=[native.T_PFB])
        fe.timing_read(native.T_PFB)
        for _ in range(max(args.steps, args.alone_launches)):
            fe.commit(B)
        fe.sync()
        pfb_alone_ms, pfb_alone_n = fe.timing_read(native.T_PFB)
        fe.timing_enable(False)
        fe.set_stage2_lag(True)
    sustained = None
    if not args.no_sustained:                         # every rank runs it (the ranks stay in step); rank 0 reports
        sustained = sustained_leg(fe, native, B, alg_bytes, seconds=args.sustained_seconds)
        if group is not None:
            sustained["kernel_us_last_window_max_over_ranks"] = barrier_max(sustained["kernel_us_last_window"])

    # ---- scan of the rank's slice (cfg5) and the peak-list all-gather, outside the timed region
    allgather_us, gathered_n, scan_out = None, None, None
    freqs = []
    if cfg5:
        fe.timing_enable(True, classes=[native.T_SCAN_FFT, native.T_SCAN_MOVSUM])
        fe.timing_read(native.T_SCAN_FFT)
        fe.timing_read(native.T_SCAN_MOVSUM)
        fe.scan_start(SCAN_N, SCAN_F, SCAN_L)
        ts = time.perf_counter()
        while fe.scan_frames_done() < SCAN_F:
            fe.commit(B)                              # the bank keeps running: scan and channelizer share the stream
        fe.sync()
        scan_wall = time.perf_counter() - ts
        fft_ms, _ = fe.timing_read(native.T_SCAN_FFT)
        mov_ms, _ = fe.timing_read(native.T_SCAN_MOVSUM)
        fe.timing_enable(False)
        tp = time.perf_counter()
        idx, _, _ = fe.scan_find_peaks(cap=1024)
        pick_ms = (time.perf_counter() - tp) * 1e3
        centre = 851e6 + fs * rank                    # slice g is centred fs * g above the first
        freqs = [native.peak_frequency(int(i), fs, SCAN_N, centre) for i in idx]
        scan_out = {"workload": "N=2^20, 1000 frames, 100-frame average over the rank's 25 Msps slice, 12 carriers",
                    "fft_logmag_ms": fft_ms, "moving_sum_ms": mov_ms, "peak_pick_ms_incl_readback": pick_ms,
                    "wall_ms_with_the_bank_running": scan_wall * 1e3, "peaks_found_rank0": int(len(idx)),
                    "scan_ms_max_over_ranks": barrier_max(fft_ms + mov_ms),
                    "realtime_factor_at_25Msps": float(SCAN_N) * SCAN_F / fs / ((fft_ms + mov_ms) * 1e-3)}
    if group is not None:
        if not cfg5:
            fe.scan_start(16384, 8, 4)
            fe.commit(B)
            idx, _, _ = fe.scan_find_peaks(cap=1024)
            freqs = [native.peak_frequency(int(i), fs, 16384, 851e6 + 25e6 * rank) for i in idx]
            if not freqs:                            # the filterbank tile has no scan-shaped carriers: exchange its
                freqs = [int(851e6 + 25e6 * rank + c["f_off"]) for c in meta["carriers"]]   # 32 known ones instead
        gather = (lambda: multigpu.allgather_peaks(fe, freqs)) if use_rccl else \
                 (lambda: multigpu.allgather_peaks_host(group, freqs))
        gather()                                     # warm-up (RCCL ring setup)
        barrier_max()
        ta = time.perf_counter()
        everyone = gather()
        allgather_us = barrier_max((time.perf_counter() - ta) * 1e6)
        gathered_n = len(everyone)
    pfb_avg_ms_max = barrier_max(pfb_ms / max(pfb_n, 1)) if group is not None else pfb_ms / max(pfb_n, 1)
    by_rank = None
    rt_k, rt_quota = rt_k_per_gpu(args, world)
    if group is not None:
        # what the line says about EVERY rank, not only the slowest: launch time / roofline fraction, the sustained
        # leg's last window, the rank's peak count (their sum must be what the gather returned), its NUMA pinning --
        # and, with --rt-seconds > 0, one paced real-time point per GPU (K front-ends on every rank at the same time)
        mine = {"rank": rank, "avg_launch_ms": pfb_ms / max(pfb_n, 1),
                "frac": alg_bytes / (pfb_ms / max(pfb_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if pfb_ms > 0 else None,
                "sustained_frac_last_window": sustained["frac_last_window"] if sustained else None,
                "peaks": len(freqs), "numa": numa}
        if args.rt_seconds > 0 and not args.no_extras:
            from .realtime import realtime_point
            try:
                barrier_max()
                blk = int(round(FS * args.rt_block_ms * 1e-3))
                raw = native.PinnedArray(2 * blk * 2 * rt_k, np.uint8)
                t8 = np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255).astype(np.uint8)
                for b_ in range(2 * rt_k):
                    at = 2 * ((b_ * 40961) % (len(tile) - blk))
                    raw.array[2 * blk * b_: 2 * blk * (b_ + 1)] = t8[at: at + 2 * blk]
                pool = {"fes": [], "chans": []}
                p = realtime_point(native, pool, rt_k, "pfb256", {"array": raw.array},
                                   meta["carriers"] if not cfg5 else synth.cfg2(n=1 << 16, seed=2002)[1]["carriers"],
                                   local_rank, min(4.0, args.rt_seconds), args.rt_block_ms, args.rt_pumps or 4,
                                   not args.rt_burst, args.rt_window_ms)
                for f_ in pool["fes"]:
                    f_.close()
                raw.free()
                mine["realtime"] = {k_: p[k_] for k_ in ("front_ends", "ok", "deadline_misses", "ring_overruns",
                                                         "latency_ms_p50", "latency_ms_p99", "latency_ms_max",
                                                         "output_samples_lost", "errors")}
            except Exception as e:
                mine["realtime"] = {"front_ends": rt_k, "ok": False, "errors": ["%s: %s" % (type(e).__name__, e)]}
        by_rank = R.gather_json(mine)

    out = None
    if rank == 0:
        total_samples = float(B) * args.steps * n_gpus
        value = total_samples / elapsed / 1e6
        avg_pfb_s = (pfb_ms / max(pfb_n, 1)) * 1e-3
        achieved = alg_bytes / avg_pfb_s / 1e9 if avg_pfb_s > 0 else 0.0
        traffic, traffic_src = None, None
        live = None
        if n_gpus == 1 and not args.no_live_traffic and B == 1 << 25:
            fe.sync()
            live = measure_traffic_live(cfg5, B, "pfb_kernel_2b<256" if cfg5 else "pfb_kernel_os<256")
        tname = "pfb512_traffic.json" if cfg5 else "pfb_traffic.json"
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("block") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %s (not measured " \
                                  "in this run)" % (tname, tj.get("measured", "an earlier run of this configuration"))
            except Exception:
                traffic = None
        traffic_file = traffic
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_src = ("measured in this run on this box: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; "
                           "--kernel-trace only) over `bench.py --steps 5 --no-extras --no-cpu-baseline --no-sustained`, "
                           "%d dispatches of the kernel averaged; KiB x 1024, FETCH x 2 (gfx950, MI355X_MICROARCH.md)"
                           % live["dispatches_averaged"])
        if cfg5:
            workload = ("BASELINE configs[4], per-GPU shape: 512-bin critically-sampled PFB (6981-tap prototype) over "
                        "one 25 Msps cf32 spectrum slice per GPU (x8 = 4096 channels at 200 Msps), N=2^20 scan of the "
                        "slice + <=1024 peaks per rank into the all-gather outside the timed region")
            kname = "pfb_kernel_2b<256, 14, 3, false> (512 bins: 256-thread workgroups, two branches per thread)"
        else:
            workload = ("BASELINE configs[1]: 256-bin critically-sampled PFB (3491-tap prototype) over one "
                        "20 Msps cf32 stream per GPU, stage-2 xlating FIR /3 + FM discriminator on 32 active bins")
            kname = "pfb_kernel_os<256,1,14,4,false>"
        frac_of = lambda ms: alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms and ms > 0 else None
        out = {
            "metric": "input IQ Msamples/s + concurrent 12.5 kHz FM channels sustained",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm": {"seconds": args.prewarm_seconds, "untimed_steps": n_prewarm,
                        "why": "steady state before the W warm-up steps (metric: sustained); see `sustained`"},
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_by_rank": per_rank_ms,
            "by_rank": by_rank,
            "ranks_started_by": ("bench.py itself (one process per GPU)" if os.environ.get("RCF_BENCH_SPAWNED")
                                 else "the launcher's environment (RANK / WORLD_SIZE)") if world > 1 else "single process",
            "rccl_ranks": R.rccl_ranks if world > 1 else 1,
            "transport": ("rccl" if use_rccl else "host-tcp") if world > 1 else "none (one rank)",
            "rccl_proof": R.rccl_proof,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "samp_rate": fs, "pfb_bins": nb, "fm_channels_per_gpu": n_active,
                "block_samples": B, "parallelism": "1 front-end per GPU x%d" % n_gpus,
            },
            "numa": numa,
            "channels": {"pfb_bins_total": nb * n_gpus, "fm_demod_total": n_active * n_gpus,
                         "realtime_factor_at_%dMsps" % int(fs / 1e6): value / n_gpus / (fs / 1e6)},
            "roofline": {
                "bound": "hbm", "kernel": kname,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_fetch_x2_bytes": live["fetch_bytes_corrected_x2"] if live else None,
                "traffic_write_bytes": live["write_bytes"] if live else None,
                "traffic_from_tracked_file": traffic_file if live is not None else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_filterbank": alg_bytes_pfb, "algorithmic_bytes_stage2_rider": alg_bytes_s2,
                "stage2_rides_in_this_launch": s2_rides,
                "filterbank_alone": ({"avg_launch_ms": pfb_alone_ms / max(pfb_alone_n, 1), "launches": pfb_alone_n,
                                      "frac": alg_bytes_pfb / (pfb_alone_ms / max(pfb_alone_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "what": "the same kernel without the rider (rcf_set_stage2_lag off for a pass of its "
                                              "own, every launch timed): 16 B x block / launch time"}
                                     if pfb_alone_n else None),
                "avg_launch_ms": avg_pfb_s * 1e3, "launches": pfb_n, "timed_every": time_every,
                "avg_launch_ms_every_launch_pass": pfb_all_ms / max(pfb_all_n, 1), "launches_every_launch_pass": pfb_all_n,
                "frac_every_launch_pass": frac_of(pfb_all_ms / max(pfb_all_n, 1)),
                "every_launch_pass_note": "a second, untimed pass of max(steps, 20) commits right behind the timed region "
                                          "with events on EVERY filterbank launch (each costs the step ~4 us, which is why "
                                          "the timed region itself times every 4th)",
                "timed_how": "bracket of two hipEventRecord (RCF_TIMING_BRACKET)"
                if os.environ.get("RCF_TIMING_BRACKET", "0") not in ("", "0") else
                "HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL start / stop events), on the launch stream",
                "avg_launch_ms_slowest_rank": pfb_avg_ms_max,
                "frac_slowest_rank": frac_of(pfb_avg_ms_max) or 0.0,
            },
            "kernel_ms_per_step": {
                "pfb": pfb_ms / max(pfb_n, 1),
                "stage2_fir_with_fused_discriminator": fir2_ms / max(n_extra, 1),
                "stage2_note": ("rides in the NEXT block's filterbank launch (its first workgroups): no launch of its own in "
                                "steady state -- the figure above is the one flush the timing read forced, spread over the "
                                "steps") if s2_rides else None,
                "separate_discriminator_launches": disc_ms / max(n_extra, 1),
                "launch_records_and_history_copy": hist_ms / max(n_extra, 1),
                "launch_records_and_history_copy_note": "0 = no launch of its own: the filterbank kernel's first workgroups "
                                                        "do both copies on the way in (PfbLaunch::rider_*)",
            },
        }
        if by_rank is not None:
            out["roofline"]["frac_by_rank"] = [r_["frac"] for r_ in by_rank]
            out["numa_by_rank"] = [r_["numa"] for r_ in by_rank]
            if any("realtime" in r_ for r_ in by_rank):
                rts = [r_.get("realtime", {"ok": False}) for r_ in by_rank]
                out["realtime_per_gpu"] = {
                    "what": "one paced point per GPU, all GPUs at the same time: K 20 Msps u8 front-ends per GPU (256-bin "
                            "bank + 32 FM channels each) in %.0f ms blocks through native pumps; no search at N > 1; "
                            "K = min(--rt-k-per-gpu, CPU quota x 32 / ranks)" % args.rt_block_ms,
                    "front_ends_per_gpu": rt_k, "front_ends_per_gpu_asked": args.rt_k_per_gpu, "cpu_quota_cores": rt_quota,
                    "ok_by_rank": [bool(r_.get("ok")) for r_ in rts],
                    "latency_ms_p99_by_rank": [r_.get("latency_ms_p99") for r_ in rts],
                    "deadline_misses_by_rank": [r_.get("deadline_misses") for r_ in rts],
                    "front_ends_sustained_total": sum(rt_k for r_ in rts if r_.get("ok")),
                    "fm_channels_sustained_total": sum(rt_k * 32 for r_ in rts if r_.get("ok")),
                    "input_Msps_sustained_total": sum(rt_k * FS / 1e6 for r_ in rts if r_.get("ok")),
                    "errors": [e_ for r_ in rts for e_ in (r_.get("errors") or [])]}
        if sustained is not None:
            if by_rank is not None:
                sustained["frac_last_window_by_rank"] = [r_["sustained_frac_last_window"] for r_ in by_rank]
            out["sustained"] = sustained
        if scan_out is not None:
            out["scan"] = scan_out
        if allgather_us is not None:
            out["peaks_allgather_us"] = allgather_us
            out["peaks_allgather"] = {"transport": "ncclAllGather via rcf_allgather_peaks" if use_rccl else "host TCP",
                                      "values_gathered": gathered_n, "ranks": world,
                                      "peaks_by_rank": [r_["peaks"] for r_ in by_rank] if by_rank else None,
                                      "values_expected": sum(r_["peaks"] for r_ in by_rank) if by_rank else None,
                                      "ok": (gathered_n == sum(r_["peaks"] for r_ in by_rank)) if by_rank else None,
                                      "peaks_from": "N=2^20 scan of each rank's slice" if cfg5 else
                                                    "16384-point scan / the tile's known carriers"}
    fe.close()
    if group is not None:
        group.close()
    return out, {"tile": tile, "meta": meta, "fm_check": fm_check, "fs": fs, "cfg5": cfg5}
