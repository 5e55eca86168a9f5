"""Randomised channel lifecycles against the oracle: random source rate, random ragged pushes, channels opened and
closed at random block boundaries, retuned at random -- every channel that lived is compared over its whole life with
the oracle's stream for a channel that starts with zero history at its opening sample (GNU Radio's own start-up) and
keeps rotator phase and FIR history across a retune (freq_xlating_fir_filter_ccc::set_center_freq).  Three seeds in
the suite; RCF_FUZZ_SEEDS=a:b runs a range (tools: gpurun -- 'RCF_FUZZ_SEEDS=0:200 pytest tests/test_gpu_fuzz.py')."""
import os

import numpy as np
import pytest

from oracle import cbind as OC
from oracle import grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _seeds():
    spec = os.environ.get("RCF_FUZZ_SEEDS")
    if not spec:
        return [11, 12, 13]
    a, b = spec.split(":")
    return list(range(int(a), int(b)))


def rel_rms(a, b):
    d = np.mean(np.abs(b) ** 2)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / d)) if d > 0 else float(np.max(np.abs(a), initial=0.0))


def _fm_rms(fm, fo, gain=1.0):
    """rms of the discriminator difference as an ANGLE: +pi and -pi are the same phase step (a product that lands on the
    negative real axis takes its sign from the last bit of its imaginary part)"""
    d = (np.asarray(fm, np.float64) - np.asarray(fo, np.float64)) / gain
    d = (d + np.pi) % (2 * np.pi) - np.pi
    return float(np.sqrt(np.mean(d ** 2))) * gain if len(d) else 0.0


def _oracle_life(x, fs, cr, segments, start, stop, filt=None):
    """segments: [(first_sample, offset_hz)] -- the offset in force from that input sample on (retunes land on block
    boundaries).  Zero history before `start`; outputs on the absolute decimation grid k D >= start, k D < stop.
    The rotator keeps running across a retune with the NEW increment (rotator::set_phase_incr), the taps switch to
    the new composite set at the boundary while the delay line keeps the samples."""
    D, taps = filt if filt is not None else G.channel_params(fs, cr)
    xz = x[:stop].copy()
    xz[:start] = 0
    k0 = -(-start // D)
    k1 = (stop - 1) // D + 1
    out = np.zeros(max(k1 - k0, 0), dtype=np.complex64)
    state = ()                                   # (phase, call counter) of GNU Radio's rotator, carried across retunes
    for i, (s0, f0) in enumerate(segments):
        s1 = segments[i + 1][0] if i + 1 < len(segments) else stop
        ka, kb = max(-(-s0 // D), k0), min((s1 - 1) // D + 1, k1)
        if kb <= ka:
            continue
        ct, incr = OC.xlating_composite(taps, D, f0, fs)
        v = G.fir_decim_cc(xz, ct, D)[ka:kb]
        ph, p_after, c_after = G.rotator_phases(incr, len(v), *state)       # continues with this segment's increment
        state = (p_after, c_after)
        out[ka - k0:kb - k0] = (v * ph).astype(np.complex64)
    return out


@pytest.mark.parametrize("seed", _seeds())
def test_random_channel_lifecycles_equal_the_oracle(gpu_required, seed):
    nat = gpu_required
    rng = np.random.default_rng(1000 + seed)
    crowd = rng.random() < 0.12                  # now and then a crowd: several matrix-core groups of 32 with churn in them
    fs = 2.4e6 if crowd else float(rng.choice([2.4e6, 8e6, 10e6, 20e6]))
    rates = [r for r in (6250, 12500, 25000, 50000) if (fs / r) == int(fs / r) and int(fs / r) % 2 == 0]
    cr = 12500                                    # the reference's rate; a third of the slots take another one, so that
    D, taps = G.channel_params(fs, cr)            # several (D, T) classes are scheduled side by side
    T = len(taps)
    n_blocks = int(rng.integers(6, 14))
    sizes = [int(rng.integers(1, 6 * D)) if rng.random() < 0.3 else int(rng.integers(T, T + 60 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    # a carrier per potential channel so that the comparison is not noise against noise
    n_slots = int(rng.integers(70, 140)) if crowd else int(rng.integers(3, 40))
    grid = 6250.0
    offs = [float(np.round(o / grid) * grid) for o in rng.uniform(-0.45, 0.45, n_slots) * fs]
    crs = [int(rng.choice(rates)) if rng.random() < 0.33 else cr for _ in range(n_slots)]
    t = np.arange(len(x)) / fs
    for f in offs[:8]:
        x = x + (0.5 * np.exp(2j * np.pi * (f + 500.0) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    lives = []                                   # dict(id, start, stop, segments, reads)
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 12) as fe:
        if getattr(fe, "set_rotator", None) and rng.random() < 0.5:
            fe.set_rotator(True)
        live = {}
        shift = 0.0                              # rcf_source_shift so far: every channel's NCO sits that much higher
        for b in range(n_blocks):
            s0 = int(cuts[b])
            # events at this block boundary
            if b and rng.random() < 0.15:        # receiver.source_offset (receiver.py:436-475): all channels move together
                d_hz = 25.0 * float(rng.integers(-8, 9))
                fe.source_shift(d_hz)
                shift += d_hz
                for L in live.values():
                    L["segments"].append((s0, L["nominal"] + shift))
            for slot in range(n_slots):
                r = rng.random()
                if slot not in live and r < (0.6 if b == 0 else 0.12):
                    cid = fe.chan_open(crs[slot], offs[slot])
                    live[slot] = dict(id=cid, cr=crs[slot], start=s0, stop=None, nominal=offs[slot],
                                      segments=[(s0, offs[slot] + shift)], reads=[], fm=[])
                elif slot in live and r < 0.06:
                    L = live.pop(slot)
                    L["reads"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["stop"] = s0
                    lives.append(L)
                elif slot in live and r < 0.14:
                    f_new = offs[slot] + grid * float(rng.integers(-3, 4))
                    fe.chan_set_offset(live[slot]["id"], f_new)
                    live[slot]["nominal"] = f_new
                    live[slot]["segments"].append((s0, f_new + shift))
            fe.push(x[s0:int(cuts[b + 1])])
            for L in live.values():
                if rng.random() < 0.3:
                    L["reads"].append(fe.chan_read_iq(L["id"]))
                if rng.random() < 0.3:
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))      # (its own cursor)
        for L in live.values():
            L["reads"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["stop"] = int(cuts[-1])
            lives.append(L)
    if not lives:
        pytest.skip("this seed's draws open no channel at all (seed 80167: three slots, every open draw above its threshold)")
    worst = 0.0
    for L in lives:
        y = np.concatenate(L["reads"]) if L["reads"] else np.zeros(0, np.complex64)
        yo = _oracle_life(x, fs, L["cr"], L["segments"], L["start"], L["stop"])
        assert len(y) == len(yo), (seed, L["cr"], L["start"], L["stop"], len(y), len(yo))
        if len(yo) == 0:
            continue
        e = rel_rms(y, yo)
        worst = max(worst, e)
        assert e < 2e-5, (seed, fs, L["segments"], L["start"], L["stop"], e)
        # the discriminator over the emitted stream (retunes included); where two consecutive outputs are both deep in a
        # fade the angle of their product is ill-conditioned: the 1e-4 bar is taken over the rest
        fm = np.concatenate(L["fm"])
        fo = G.quadrature_demod_cf(yo, 1.0)
        assert len(fm) == len(fo)
        mag = np.abs(yo)
        ok = np.ones(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        if ok.sum() > 8:
            efm = _fm_rms(fm[ok], fo[ok])
            if efm >= 1e-4 and os.environ.get("RCF_FUZZ_DEBUG"):
                bad = np.nonzero((np.abs(fm - fo) > 1e-3) & ok)[0]
                Dd = G.channel_params(fs, L["cr"])[0]
                print("DBG", seed, fs, L["cr"], "D", Dd, "T", len(G.channel_params(fs, L["cr"])[1]), L["segments"], L["start"], L["stop"],
                      "k0", -(-L["start"] // Dd), "cuts(out)", [-(-int(c) // Dd) for c in cuts], "bad", bad[:20].tolist(), len(bad),
                      "fm", [float(fm[i]) for i in bad[:4]], "fo", [float(fo[i]) for i in bad[:4]], "reads_fm", [len(a) for a in L["fm"]])
            assert efm < 1e-4, (seed, fs, L["cr"], L["segments"], L["start"], L["stop"], efm)


@pytest.mark.parametrize("seed", _seeds())
def test_random_filterbank_tap_lifecycles(gpu_required, seed):
    """Bins of a frame-major bank opened and closed as channels at random block boundaries, in dense stretches (complete
    aligned runs of 16: read from the bank's ring) and scattered (tap matrix), with duplicates, through ragged pushes:
    every tap's stream is its bin over the tap's life, bit for bit, and its discriminator stream the discriminator of
    that; two of the bins are also checked against the float64 exact-phase bank."""
    nat = gpu_required
    rng = np.random.default_rng(5000 + seed)
    fs, nb = (5e6, 400) if rng.random() < 0.6 else (20e6, 1600)
    D, taps = G.channel_params(fs, 12500)
    assert nb == 2 * D
    n_blocks = int(rng.integers(4, 9))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.25 else int(rng.integers(2 * D, 90 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    # what gets opened when is drawn up front, so that the bins that are ever tapped can be followed from frame 0
    plan = []
    for b in range(n_blocks):
        opens = []
        for _ in range(int(rng.integers(0, 4))):
            if rng.random() < 0.5:               # a dense stretch
                lo = int(rng.integers(0, nb - 40))
                bins = list(range(lo, lo + int(rng.integers(10, 40))))
                if rng.random() < 0.3:
                    rng.shuffle(bins)
            else:
                bins = [int(v) for v in rng.integers(0, nb, int(rng.integers(1, 12)))]
            opens += bins
        plan.append(opens)
    ever = sorted({k for opens in plan for k in opens})
    lives = []
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 11) as fe:
        fe.pfb_open(nb, D, taps)
        live = []                                # dicts: id, bin, first (bank frame count at opening), iq, fm
        ring = {k: [] for k in ever}
        for b in range(n_blocks):
            produced = fe.pfb_produced()
            for k in plan[b]:
                live.append(dict(id=fe.pfb_tap_open(k, gr_phase=False), bin=k, first=produced, iq=[], fm=[]))
            for L in list(live):
                if rng.random() < 0.08:
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["last"] = produced
                    live.remove(L)
                    lives.append(L)
            fe.push(x[int(cuts[b]):int(cuts[b + 1])])
            for k in ever:
                ring[k].append(fe.pfb_read_bin(k))               # (a cursor: whatever is new since the last read)
            for L in live:
                if rng.random() < 0.4:
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
        n_out = fe.pfb_produced()
        for L in live:
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["last"] = n_out
            lives.append(L)
    if not lives:
        return
    assert n_out == (len(x) - 1) // D + 1
    checked = 0
    for L in lives:
        y = np.concatenate(L["iq"]) if L["iq"] else np.zeros(0, np.complex64)
        fm = np.concatenate(L["fm"]) if L["fm"] else np.zeros(0, np.float32)
        full = np.concatenate(ring[L["bin"]])
        assert len(full) == n_out
        want = full[L["first"]:L["last"]]
        assert len(y) == len(want), (seed, L["bin"], L["first"], L["last"], len(y), len(want))
        np.testing.assert_array_equal(y, want, err_msg="seed %d bin %d" % (seed, L["bin"]))
        if len(want):
            np.testing.assert_array_equal(fm, G.quadrature_demod_cf(want, 1.0), err_msg="fm, seed %d bin %d" % (seed, L["bin"]))
        checked += 1
    assert checked == len(lives)
    for k in ever[:2]:                            # and the bins themselves against the float64 exact-phase bank
        f_k = (k if k <= nb // 2 else k - nb) * fs / nb
        ref = G.xlating_fir_exact(x, D, taps, f_k, fs).astype(np.complex64)
        assert rel_rms(np.concatenate(ring[k]), ref) < 3e-5


@pytest.mark.parametrize("seed", _seeds())
def test_random_scans_equal_the_oracle_chain(gpu_required, seed):
    """fft_vector.py's chain at random lengths (single-pass and four-step FFTs), frame counts and averaging lengths,
    fed in ragged pushes (frames straddle blocks, some pushes shorter than a frame, a scan restarted in mid-stream):
    the emitted vector against the oracle's chain, and the device peak picker against scipy's on the same vector."""
    from oracle import peaks as P
    nat = gpu_required
    rng = np.random.default_rng(9000 + seed)
    N = 1 << int(rng.integers(8, 18))                        # 256 .. 131072
    F = int(rng.integers(3, 30 if N <= 16384 else 8))
    L = int(rng.integers(1, F + 1))
    fs = 2.4e6 if N <= 16384 else 100e6
    lead = int(rng.integers(0, 3 * N))                       # samples before the scan starts
    x = synth.awgn(rng, lead + N * F + int(rng.integers(0, N)))
    n = np.arange(len(x))
    for _ in range(int(rng.integers(1, 5))):
        x = x + (float(rng.uniform(1, 6)) * np.exp(2j * np.pi * float(rng.uniform(-0.45, 0.45)) * n)).astype(np.complex64)
    x = x.astype(np.complex64)
    cap = int(max(2 * N, 4096))
    with nat.Frontend(fs, block_capacity=cap, hist_capacity=max(N, 1 << 12)) as fe:
        if lead:
            pos = 0
            while pos < lead:
                step = min(int(rng.integers(1, cap)), lead - pos)
                fe.push(x[pos:pos + step])
                pos += step
        fe.scan_start(N, F, L)
        pos = lead
        while pos < len(x):
            step = min(int(rng.integers(1, N // 2)) if rng.random() < 0.3 else int(rng.integers(N // 2, cap)), len(x) - pos)
            fe.push(x[pos:pos + step])
            pos += step
        assert fe.scan_frames_done() == F
        got = fe.scan_result()
        idx, mean, _ = fe.scan_find_peaks(cap=4096)
    want = OC.scan_chain(x[lead:], N, F, L)
    assert got is not None and got.shape == (N,)
    # a sum of L log-magnitudes: float32 FFT error shows where a frame's bin happens to be nearly empty (the log of a
    # small difference) -- the bound on the worst bin is looser than the bound on the rest
    err = np.abs(got - want)
    assert err.max() < 0.5 and np.sort(err)[int(0.999 * N)] < 2e-3 and err.mean() < 1e-4, (seed, N, F, L, float(err.max()), float(err.mean()))
    l_got, _ = P.peak_detect_scipy(got, fs, 0.0)
    np.testing.assert_array_equal(idx, l_got, err_msg="seed %d N %d F %d L %d" % (seed, N, F, L))


@pytest.mark.parametrize("seed", _seeds())
def test_random_stage2_channel_lifecycles_on_a_bank(gpu_required, seed):
    """The BASELINE throughput shape under churn: a critically sampled power-of-two bank (64 .. 512 bins, prototype by
    the low_pass_2 rule) with stage-2 channels (channel.py's rule at the bin rate + discriminator) opened, retuned and
    closed at random block boundaries through ragged pushes -- each against the two-stage oracle: the float64
    exact-phase bank's bin, zeroed before the channel's opening frame, through GNU Radio's xlating FIR and rotator."""
    nat = gpu_required
    rng = np.random.default_rng(7000 + seed)
    if rng.random() < 0.75:
        nb = int(rng.choice([64, 128, 256, 512]))
        fs = nb * 78125.0                                    # bin rate 78.125 kHz: channel.py gives D2 = 3, T2 = 11
        Db = nb
        bw = fs / nb
        proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    else:                                                    # a frame-major bank (bins read with stride n_bins): 400 bins at
        fs, nb = 5e6, 400                                    # 5 Msps from the reference's channel filter, 25 kS/s per bin
        Db, proto = G.channel_params(fs, 12500)              # -> stage 2 at decimation 1
        bw = fs / Db
    D2, taps2 = G.channel_params(bw, 12500)
    grid_hz = fs / nb
    n_blocks = int(rng.integers(5, 11))
    sizes = [int(rng.integers(1, 4 * Db)) if rng.random() < 0.25 else int(rng.integers(20 * Db, 300 * Db)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    slots = [(int(rng.integers(0, nb)), float(rng.integers(-4, 5)) * 1562.5) for _ in range(int(rng.integers(2, 9)))]
    t = np.arange(len(x)) / fs
    for k, d in slots:
        f = (k if k <= nb // 2 else k - nb) * grid_hz + d + 700.0
        x = x + (0.7 * np.exp(2j * np.pi * f * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    lives = []
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 12) as fe:
        fe.pfb_open(nb, Db, proto)
        live = {}
        shift = 0.0                              # rcf_source_shift: the stage-2 NCOs follow it (the bank's grid does not move)
        for b in range(n_blocks):
            frames = fe.pfb_produced()
            if b and rng.random() < 0.15:
                d_hz = 25.0 * float(rng.integers(-8, 9))
                fe.source_shift(d_hz)
                shift += d_hz
                for L in live.values():
                    L["segments"].append((frames, L["nominal"] + shift))
            for i, (k, d) in enumerate(slots):
                r = rng.random()
                if i not in live and r < (0.7 if b == 0 else 0.2):
                    live[i] = dict(id=fe.pfb_chan_open(k, 12500, d), bin=k, first=frames, nominal=d,
                                   segments=[(frames, d + shift)], iq=[], fm=[])
                elif i in live and r < 0.07:
                    L = live.pop(i)
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["last"] = frames
                    lives.append(L)
                elif i in live and r < 0.17:
                    d_new = float(rng.integers(-4, 5)) * 1562.5
                    fe.chan_set_offset(live[i]["id"], d_new)
                    live[i]["nominal"] = d_new
                    live[i]["segments"].append((frames, d_new + shift))
            fe.push(x[int(cuts[b]):int(cuts[b + 1])])
        frames = fe.pfb_produced()
        for L in live.values():
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["last"] = frames
            lives.append(L)
    assert frames == (len(x) - 1) // Db + 1
    stage1 = {}
    for L in lives:
        k = L["bin"]
        if k not in stage1:
            stage1[k] = G.xlating_fir_exact(x, Db, proto, (k if k <= nb // 2 else k - nb) * grid_hz, fs).astype(np.complex64)
        s1 = stage1[k][:L["last"]].copy()
        s1[:L["first"]] = 0
        k0, k1 = -(-L["first"] // D2), ((L["last"] - 1) // D2 + 1 if L["last"] > 0 else 0)
        want = np.zeros(max(k1 - k0, 0), dtype=np.complex64)
        state = ()
        for j, (f0, d) in enumerate(L["segments"]):
            f1 = L["segments"][j + 1][0] if j + 1 < len(L["segments"]) else L["last"]
            ka, kb = max(-(-f0 // D2), k0), min((f1 - 1) // D2 + 1 if f1 > 0 else 0, k1)
            if kb <= ka:
                continue
            ct, incr = OC.xlating_composite(taps2, D2, d, bw)
            v = G.fir_decim_cc(s1, ct, D2)[ka:kb]
            ph, p_after, c_after = G.rotator_phases(incr, len(v), *state)
            state = (p_after, c_after)
            want[ka - k0:kb - k0] = (v * ph).astype(np.complex64)
        y = np.concatenate(L["iq"])
        fm = np.concatenate(L["fm"])
        assert len(y) == len(want) == len(fm), (seed, nb, L["first"], L["last"], len(y), len(want))
        if len(want) < 4:
            continue
        assert rel_rms(y, want) < 3e-5, (seed, nb, L["bin"], L["segments"], L["first"], L["last"], rel_rms(y, want))
        fo = G.quadrature_demod_cf(want, 1.0)
        bad = np.nonzero(np.abs(fm - fo) > 1e-3)[0]
        if len(bad) and os.environ.get("RCF_FUZZ_DEBUG"):
            print("DBG", seed, nb, L["bin"], L["segments"], L["first"], L["last"], "k0", k0, "bad", bad[:16].tolist(), len(bad),
                  "|want|", [float(abs(want[i])) for i in bad[:4]], "fm", [float(fm[i]) for i in bad[:4]],
                  "fo", [float(fo[i]) for i in bad[:4]], "rms", float(np.sqrt(np.mean(np.abs(want) ** 2))),
                  "frames at cuts", [int((c - 1) // nb + 1) if c else 0 for c in cuts])
        assert _fm_rms(fm[2:], fo[2:]) < 1e-4, (
            seed, nb, L["bin"], L["segments"], L["first"], L["last"], k0, bad[:12].tolist(), len(bad),
            [float(abs(want[i])) for i in bad[:4]], [float(fm[i]) for i in bad[:4]], [float(fo[i]) for i in bad[:4]],
            float(np.sqrt(np.mean(np.abs(want) ** 2))), [int(c) for c in cuts[:12]])


@pytest.mark.parametrize("seed", _seeds())
def test_random_voice_chain_cuts_and_silences(gpu_required, seed):
    """The analog voice chain (pwr_squelch_cc with gating -> fm demod -> de-emphasis -> 300 Hz high-pass -> 25:8
    resampler, logging_receiver.py:211-222) is stateful in every stage: an NBFM carrier with random stretches of exact
    silence (the squelch closes and REMOVES samples), fed in random ragged pushes -- audio against oracle/audio.py on
    the oracle's channel stream, sample counts included."""
    from oracle import audio as A
    from rcf import audio as host_audio
    nat = gpu_required
    rng = np.random.default_rng(3000 + seed)
    fs, cr = 2.4e6, 12500
    D = 96
    f0 = float(np.round(rng.uniform(-0.4, 0.4) * fs / 6250) * 6250)
    n_out = int(rng.integers(3000, 9000))
    n = D * n_out + int(rng.integers(0, D))
    x = synth.nbfm_carrier(n, fs, f0, float(rng.uniform(300, 2500)), float(rng.uniform(1000, 3000)), float(rng.uniform(0.1, 0.6)))
    x = x.astype(np.complex64)
    if rng.random() < 0.5:
        x = (x + 1e-3 * synth.awgn(rng, n)).astype(np.complex64)       # a noise floor keeps the squelch open through the gaps
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n_out - 600)) * D
        x[a:a + int(rng.integers(200, 2500)) * D] = 0
    cuts = sorted({0, n} | {int(v) for v in rng.integers(1, n, int(rng.integers(1, 9)))})
    with nat.Frontend(fs, block_capacity=n) as fe:
        cid = fe.chan_open(cr, f0)
        host_audio.open_analog_voice(fe, cid, 25000)
        got = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            fe.push(x[a:b])
            if rng.random() < 0.4:
                got.append(fe.chan_read_audio(cid))
        n_audio, n_ungated = fe.chan_audio_produced(cid)
        got.append(fe.chan_read_audio(cid))
        y_dev = fe.chan_read_iq(cid)                        # the channel stream the device's squelch saw (the ring holds all of it)
    audio = np.concatenate(got)
    Dd, taps = G.channel_params(fs, cr)
    ct, incr = OC.xlating_composite(taps, Dd, f0, fs)
    y, _ = OC.channel_bank(x, Dd, ct[None, :], np.array([incr]), acc_double=True)
    st = A.analog_chain(y[0], 25000.0, stages=True)
    if n_ungated != len(st["gated"]):
        # The gate is a threshold on a power estimate: when a crossing lands within float32 accuracy of the threshold, the
        # device's channel stream and the oracle's (equal to ~1e-7) may gate one sample apart (seed 78068: the estimate passes
        # the threshold at 1e-6 of it).  Only then -- the oracle says how close it came -- the chain is checked on the
        # samples the device's squelch saw; a gate that disagrees on a well-conditioned stream still fails here.
        margin = A.pwr_squelch_margin(y[0])
        assert margin < 1e-4 and len(y_dev) == len(y[0]), (seed, n_ungated, len(st["gated"]), margin)
        st = A.analog_chain(y_dev, 25000.0, stages=True)
    assert n_ungated == len(st["gated"]), (seed, n_ungated, len(st["gated"]))
    assert len(audio) == n_audio == len(st["audio"]), (seed, len(audio), n_audio, len(st["audio"]))
    if len(audio):
        e = float(np.sqrt(np.mean((audio.astype(np.float64) - st["audio"]) ** 2)))
        assert e < 1e-4, (seed, e)


@pytest.mark.parametrize("seed", _seeds())
def test_random_p25_front_half_chains(gpu_required, seed):
    """p25_control_demod.py:105-137 under random cuts: wideband channel -> chained pre-filter channel (fir on the channel's
    own output ring, opened at the start or in mid-stream) -> discriminator -> symbol filter, the parent retuned at random;
    every stage against the oracle's chain (a chained channel opened later starts with zero history of ITS input)."""
    nat = gpu_required
    rng = np.random.default_rng(4000 + seed)
    fs, cr = float(rng.choice([2.4e6, 8e6])), 12500
    D, taps = G.channel_params(fs, cr)
    f0 = float(np.round(rng.uniform(-0.4, 0.4) * fs / 6250) * 6250)
    n_out = int(rng.integers(1500, 5000))
    n = D * n_out + int(rng.integers(0, D))
    x = (synth.nbfm_carrier(n, fs, f0 + float(rng.uniform(-300, 300)), 600.0, 1800.0, 0.5) + 0.02 * synth.awgn(rng, n)).astype(np.complex64)
    pre = G.low_pass_2(1.0, 25000.0, 6250.0, float(rng.choice([500.0, 1500.0])), 30.0, G.WIN_BLACKMAN)
    gain = G.p25_fm_gain(25000.0)
    coeffs = np.full(5, 0.2, dtype=np.float32)
    cuts = sorted({0, n} | {int(v) for v in rng.integers(1, n, int(rng.integers(2, 9)))})
    open_at = int(rng.integers(0, len(cuts) - 1))            # block boundary at which the chained channel is opened
    retune_at = int(rng.integers(1, len(cuts) - 1)) if rng.random() < 0.5 else None
    f1 = f0 + 6250.0 * float(rng.integers(-1, 2))
    got_sym, got_fm, got_iq = [], [], []
    with nat.Frontend(fs, block_capacity=n) as fe:
        c1 = fe.chan_open(cr, f0)
        c2 = None
        first2 = None
        for b, (a, e) in enumerate(zip(cuts[:-1], cuts[1:])):
            if b == open_at:
                first2 = fe.chan_produced(c1)
                c2 = fe.chan_open_taps(c1, 1, pre, 0.0)
                fe.chan_fm_filter(c2, gain, coeffs)
            if retune_at is not None and b == retune_at:
                fe.chan_set_offset(c1, f1)
            fe.push(x[a:e])
            if c2 is not None and rng.random() < 0.4:
                got_sym.append(fe.chan_read_sym(c2))
                got_fm.append(fe.chan_read_fm(c2, gain))
                got_iq.append(fe.chan_read_iq(c2))
        got_sym.append(fe.chan_read_sym(c2))
        got_fm.append(fe.chan_read_fm(c2, gain))
        got_iq.append(fe.chan_read_iq(c2))
    segs = [(0, f0)] + ([(cuts[retune_at], f1)] if retune_at is not None else [])
    yo = _oracle_life(x, fs, cr, segs, 0, n)
    y1 = yo.copy()
    y1[:first2] = 0                                          # the chained channel's zero history
    y2o = G.xlating_fir_ccc(y1, 1, pre, 0.0, 25000.0)[first2:]
    fo = G.quadrature_demod_cf(y2o, gain)
    so = np.convolve(fo.astype(np.float64), coeffs.astype(np.float64))[: len(fo)].astype(np.float32) if len(fo) else fo
    iq, fm, sym = np.concatenate(got_iq), np.concatenate(got_fm), np.concatenate(got_sym)
    assert len(iq) == len(y2o) and len(fm) == len(fo) and len(sym) == len(so), (seed, len(iq), len(y2o), len(fm), len(sym))
    if len(iq) > 16:
        assert rel_rms(iq, y2o) < 2e-5, (seed, rel_rms(iq, y2o))
        # the discriminator bar is taken where the angle is well conditioned: not where two consecutive outputs sit in a
        # deep fade, nor in the first outputs after the opening (the pre-filter filling up from zero)
        mag = np.abs(y2o)
        ok = np.zeros(len(mag), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        ok5 = ok.copy()
        for d in range(1, 5):
            ok5[d:] &= ok[:-d]
        ok5[:4] = False
        if os.environ.get("RCF_FUZZ_DEBUG"):
            bad = np.nonzero((np.abs(fm - fo) > 1e-3) & ok)[0]
            print("DBG", seed, fs, "n", len(fm), "first2", first2, "bad", bad[:20].tolist(), len(bad))
        if ok.sum() > 16:
            assert _fm_rms(fm[ok], fo[ok], gain) < 1e-4, seed
        if ok5.sum() > 16:
            assert float(np.sqrt(np.mean((sym[ok5] - so[ok5]) ** 2))) < 1e-4, seed


@pytest.mark.parametrize("seed", _seeds())
def test_random_gr_phase_taps_equal_gnuradio_channels_started_at_their_opening(gpu_required, seed):
    """What frontend_mode = 'pfb' hands out: a bin of the 400-bin bank (5 Msps, the reference's D = 200 / T = 727 channel
    filter) opened WITH GNU Radio's phase convention at a random block boundary must be the stream of a
    freq_xlating_fir_filter_ccc started at that sample -- zero history, rotator at 1 -- up to the tap-phase leakage the
    parity budget allows (bins within +-100 of the centre: float32 tap phases below 1200 rad).  (No source shifts here: a
    bin's filter cannot follow one, only the tap's rotator does -- test_source_shift_reaches_filterbank_taps.)"""
    nat = gpu_required
    rng = np.random.default_rng(8000 + seed)
    cr = 12500
    if rng.random() < 0.6:
        fs, nb = 5e6, 400
        D, taps = G.channel_params(fs, cr)
        assert D == 200 and nb == 2 * D
    else:                                        # an oversampled power-of-two bank: its taps are ordinary channels on a bin's ring
        nb = int(rng.choice([128, 256]))
        fs, D = nb * 25000.0, nb // 2
        taps = G.low_pass_2(1.0, fs, fs / nb / 4, fs / nb / 4, 20.0)
    grid = fs / nb
    n_blocks = int(rng.integers(3, 8))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.2 else int(rng.integers(4 * D, 120 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    ks = [int(v) for v in rng.integers(-(nb // 4), nb // 4 + 1, int(rng.integers(2, 10)))]
    t = np.arange(len(x)) / fs
    for k in ks:
        x = x + (0.8 * np.exp(2j * np.pi * (k * grid + float(rng.uniform(-2000, 2000))) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    lives = []
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 11) as fe:
        fe.pfb_open(nb, D, taps)
        live = {}
        shift = 0.0
        for b in range(n_blocks):
            s0 = int(cuts[b])
            for i, k in enumerate(ks):
                r = rng.random()
                if i not in live and r < (0.6 if b == 0 else 0.25):
                    cid = fe.pfb_tap_open(k % nb, gr_phase=True)
                    live[i] = dict(id=cid, start=s0, nominal=k * grid, segments=[(s0, k * grid + shift)], iq=[], fm=[])
                elif i in live and r < 0.08:
                    L = live.pop(i)
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["stop"] = s0
                    lives.append(L)
            fe.push(x[s0:int(cuts[b + 1])])
        for L in live.values():
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["stop"] = int(cuts[-1])
            lives.append(L)
    for L in lives:
        y, fm = np.concatenate(L["iq"]), np.concatenate(L["fm"])
        yo = _oracle_life(x, fs, cr, L["segments"], L["start"], L["stop"], filt=(D, taps))
        assert len(y) == len(yo) == len(fm), (seed, L["start"], L["stop"], len(y), len(yo))
        # the one difference that is meant: the bank has been running, so a bin opened in mid-stream comes with the
        # filter's history in it, where GNU Radio's new flowgraph starts from zeros -- the first (T - 1) / D outputs
        warm = (len(taps) - 1) // D + 1 if L["start"] else 0
        if len(yo) < warm + 16:
            continue
        e = rel_rms(y[warm:], yo[warm:])
        assert e < 5e-4, (seed, L["segments"], L["start"], L["stop"], e)
        fo = G.quadrature_demod_cf(yo, 1.0)
        mag = np.abs(yo)
        ok = np.zeros(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        ok[:warm + 1] = False
        if ok.sum() > 8:
            efm = _fm_rms(fm[ok], fo[ok])
            assert efm < 1e-4, (seed, L["segments"], L["start"], L["stop"], efm)


@pytest.mark.parametrize("seed", _seeds())
def test_random_discriminator_only_taps_with_flips_equal_gnuradio_discriminators(gpu_required, seed):
    """rcf_chan_set_fm_only on taps of the reference-grid bank, switched on and off at random block boundaries (the
    discriminator-only path rotates nothing: it turns bin[n] conj(bin[n-1]) by the rotator's increment, and the call
    converts the one ring sample a switch straddles): the discriminator stream must stay that of quadrature_demod_cf behind
    a freq_xlating_fir_filter_ccc started at the tap's opening sample, through every flip; IQ is refused while the flag
    is on"""
    nat = gpu_required
    rng = np.random.default_rng(8800 + seed)
    cr, fs, nb = 12500, 5e6, 400
    D, taps = G.channel_params(fs, cr)
    grid = fs / nb
    n_blocks = int(rng.integers(3, 8))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.2 else int(rng.integers(4 * D, 120 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    ks = [int(v) for v in rng.integers(-(nb // 4), nb // 4 + 1, int(rng.integers(2, 9)))]
    t = np.arange(len(x)) / fs
    for k in ks:
        x = x + (0.8 * np.exp(2j * np.pi * (k * grid + float(rng.uniform(-2000, 2000))) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    lives = []
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 11) as fe:
        fe.pfb_open(nb, D, taps)
        live = {}
        for b in range(n_blocks):
            s0 = int(cuts[b])
            for i, k in enumerate(ks):
                r = rng.random()
                if i not in live and r < (0.6 if b == 0 else 0.25):
                    cid = fe.pfb_tap_open(k % nb, gr_phase=True)
                    live[i] = dict(id=cid, start=s0, segments=[(s0, k * grid)], fm=[], only=False, flips=0)
                    if rng.random() < 0.5:
                        fe.chan_set_fm_only(cid, True)
                        live[i]["only"] = True
                elif i in live and r < 0.35:
                    L = live[i]
                    L["only"] = not L["only"]
                    L["flips"] += 1
                    fe.chan_set_fm_only(L["id"], L["only"])
            fe.push(x[s0:int(cuts[b + 1])])
            for L in live.values():
                if L["only"] and rng.random() < 0.3:
                    with pytest.raises(nat.RcfError):
                        fe.chan_read_iq(L["id"])
                if rng.random() < 0.5:
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
        for L in live.values():
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["stop"] = int(cuts[-1])
            lives.append(L)
    for L in lives:
        fm = np.concatenate(L["fm"])
        yo = _oracle_life(x, fs, cr, L["segments"], L["start"], L["stop"], filt=(D, taps))
        assert len(yo) == len(fm), (seed, L["start"], L["stop"], len(yo), len(fm))
        warm = (len(taps) - 1) // D + 1 if L["start"] else 0
        if len(yo) < warm + 16:
            continue
        fo = G.quadrature_demod_cf(yo, 1.0)
        mag = np.abs(yo)
        ok = np.zeros(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        ok[:warm + 1] = False
        if ok.sum() > 8:
            efm = _fm_rms(fm[ok], fo[ok])
            assert efm < 1e-4, (seed, L["segments"], L["start"], L["stop"], L["flips"], efm)


def _hip_memcpy_h2d(dst_ptr, arr):
    """test plumbing: overwrite device memory the library handed out (hipMemcpy through the runtime librcf loaded)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    a = np.ascontiguousarray(arr)
    assert hip.hipMemcpy(C.c_void_p(dst_ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("seed", _seeds())
def test_random_quantised_spectra_through_the_device_picker(gpu_required, seed):
    """The device peak picker on spectra a scan never produces: values quantised so that plateaus, equal neighbours and
    ties at the prominence walks' bases occur all over (scipy's plateau midpoints, left-to-right tie rules), carriers of
    every width around the [3 kHz, 30 kHz] window, offsets that make the minimum negative -- injected into the finished
    scan's device buffer, indices against scipy.signal.find_peaks through the reference's prologue."""
    import ctypes as C
    from oracle import peaks as P
    nat = gpu_required
    rng = np.random.default_rng(6000 + seed)
    N = 1 << int(rng.integers(10, 18))
    fs = float(rng.choice([2.4e6, 12.5e6, 100e6]))
    hz = fs / N
    x = rng.normal(100.0, float(rng.uniform(0.5, 6.0)), N)
    for _ in range(int(rng.integers(0, 12))):
        c = int(rng.integers(0, N))
        w = float(rng.uniform(500.0, 60000.0)) / hz              # occupied width in bins, around the 3-30 kHz window
        shape = rng.random()
        d = (np.arange(N) - c) / max(w / 2.355, 0.5)
        bump = np.exp(-0.5 * d ** 2) if shape < 0.6 else (np.abs(np.arange(N) - c) < w / 2).astype(np.float64)
        x += float(rng.uniform(5, 60)) * bump
    quant = float(rng.choice([0.0, 0.25, 1.0, 4.0, 16.0]))
    if quant:
        x = np.round(x / quant) * quant
    x = (x - float(rng.choice([0.0, 480.0]))).astype(np.float32)
    with nat.Frontend(fs, 855e6, block_capacity=max(2 * N, 4096), hist_capacity=max(N, 1 << 12)) as fe:
        fe.scan_start(N, 1, 1)
        fe.push(synth.awgn(rng, N))
        assert fe.scan_result() is not None
        dev = C.c_void_p()
        assert nat.lib().rcf_scan_result_device(fe._h, C.byref(dev)) == 0 and dev.value
        _hip_memcpy_h2d(dev.value, x)
        np.testing.assert_array_equal(fe.scan_result(), x)       # the injected vector is what the picker will see
        idx, mean, _ = fe.scan_find_peaks(cap=4096)
    want, _ = P.peak_detect_scipy(x, fs, 855e6)
    if len(want) <= 4096:
        np.testing.assert_array_equal(idx, want, err_msg="seed %d N %d quant %s" % (seed, N, quant))
    else:
        np.testing.assert_array_equal(idx, want[:4096])


@pytest.mark.parametrize("seed", _seeds())
def test_random_wire_format_pushes_equal_host_conversion(gpu_required, seed):
    """rcf_push_raw (u8 / s8 / s16 straight off the SDR, converted on the device) mixed with cf32 pushes in random ragged
    pieces -- including single samples and pieces that are not a multiple of anything -- must give a channel and a
    filterbank bin the same bits as the host conversion (float(raw) - offset) * scale pushed as cf32 in one piece."""
    nat = gpu_required
    rng = np.random.default_rng(2000 + seed)
    fmt_name, dtype = [("FMT_U8", np.uint8), ("FMT_S8", np.int8), ("FMT_S16", np.int16)][int(rng.integers(0, 3))]
    info = np.iinfo(dtype)
    scale = float(np.float32(1.0 / (info.max + 1)))
    offset = 127.4 if dtype == np.uint8 else 0.0
    fs, nb = 2.4e6, 64
    n = int(rng.integers(3000, 60000))
    raw = rng.integers(info.min, info.max + 1, size=2 * n).astype(dtype)
    x = ((raw.astype(np.float32) - np.float32(offset)) * np.float32(scale)).view(np.complex64)
    proto = G.low_pass_2(1.0, fs, fs / nb * 0.4, fs / nb * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    f0 = float(np.round(rng.uniform(-0.4, 0.4) * fs / 6250) * 6250)
    cuts = sorted({0, n} | {int(v) for v in rng.integers(1, n, int(rng.integers(1, 12)))} | {int(rng.integers(1, n - 1)) + d for d in (0, 1)})
    cuts = [c for c in cuts if 0 <= c <= n]

    def run(pieces_raw):
        with nat.Frontend(fs, block_capacity=n + 16) as fe:
            fe.set_rotator(True)         # GNU Radio's iterated phase: the same bits however the stream is cut (the closed form
            fe.pfb_open(nb, nb, proto)   # is rebased every block and may round the last bit of a phase differently)
            cid = fe.chan_open(12500, f0)
            if pieces_raw is None:
                fe.push(x)
            else:
                for a, b in zip(cuts[:-1], cuts[1:]):
                    if pieces_raw[a]:
                        fe.push_raw(raw[2 * a:2 * b], getattr(nat, fmt_name), scale, offset)
                    else:
                        fe.push(x[a:b])
            return fe.chan_read_iq(cid), fe.chan_read_fm(cid, 1.0), fe.pfb_read_bin(5)

    ref = run(None)
    how = {a: bool(rng.random() < 0.7) for a in cuts}
    got = run(how)
    for g, r in zip(got, ref):
        assert len(g) == len(r)
        np.testing.assert_array_equal(g.view(np.float32) if g.dtype == np.complex64 else g,
                                      r.view(np.float32) if r.dtype == np.complex64 else r)


@pytest.mark.parametrize("seed", _seeds())
def test_random_split2_chains(gpu_required, seed):
    """receiver_split2 (receiver.py:205-237) under churn: the two half-band sub-sources (decimation 2 at -/+ fs/4) stay open;
    channels at the half rate (channel.py's rule: D = 48, T = 175 at 2.4 Msps -- the vector / matrix-core bank kernels on a
    CHANNEL's ring as their source) are opened, retuned and closed on either half at random block boundaries through
    ragged pushes.  Oracle: the two GNU Radio blocks in series, the child starting with zero history of the half-band
    stream at its opening."""
    nat = gpu_required
    rng = np.random.default_rng(1500 + seed)
    fs = float(rng.choice([2.4e6, 8e6]))
    t1 = G.low_pass_2(1.0, fs, fs / 4, fs / 8, 53.0)
    D2, t2 = G.channel_params(fs / 2, 12500)
    n_blocks = int(rng.integers(4, 10))
    T2 = len(t2)
    sizes = [int(rng.integers(1, 8 * D2)) if rng.random() < 0.25 else int(rng.integers(2 * T2, 2 * T2 + 80 * 2 * D2)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    n_slots = int(rng.integers(2, 14))             # >= 8 live children of one half ride the matrix cores
    slots = [(int(rng.integers(0, 2)), float(np.round(rng.uniform(-0.4, 0.4) * fs / 2 / 6250) * 6250)) for _ in range(n_slots)]
    t = np.arange(len(x)) / fs
    for half, f in slots[:6]:
        x = x + (0.5 * np.exp(2j * np.pi * ((-1.0 if half == 0 else 1.0) * fs / 4 + f + 400.0) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    lives = []
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 14) as fe:
        halves = [fe.chan_open_taps(-1, 2, t1, -fs / 4), fe.chan_open_taps(-1, 2, t1, fs / 4)]
        live = {}
        for b in range(n_blocks):
            prod = [fe.chan_produced(h) for h in halves]
            for i, (half, f) in enumerate(slots):
                r = rng.random()
                if i not in live and r < (0.6 if b == 0 else 0.2):
                    cid = fe.chan_open_taps(halves[half], D2, t2, f)
                    live[i] = dict(id=cid, half=half, start=prod[half], segments=[(prod[half], f)], iq=[], fm=[])
                elif i in live and r < 0.07:
                    L = live.pop(i)
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["stop"] = prod[L["half"]]
                    lives.append(L)
                elif i in live and r < 0.17:
                    f_new = f + 6250.0 * float(rng.integers(-3, 4))
                    fe.chan_set_offset(live[i]["id"], f_new)
                    live[i]["segments"].append((prod[live[i]["half"]], f_new))
            fe.push(x[int(cuts[b]):int(cuts[b + 1])])
            for L in live.values():
                if rng.random() < 0.3:
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
        prod = [fe.chan_produced(h) for h in halves]
        for L in live.values():
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["stop"] = prod[L["half"]]
            lives.append(L)
    hb = []
    for sgn in (-1.0, 1.0):
        ct1, incr1 = OC.xlating_composite(t1, 2, sgn * fs / 4, fs)
        v1 = G.fir_decim_cc(x, ct1, 2)
        ph1, _, _ = G.rotator_phases(incr1, len(v1))
        hb.append((v1 * ph1).astype(np.complex64))
    assert prod == [len(hb[0]), len(hb[1])]
    for L in lives:
        y, fm = np.concatenate(L["iq"]), np.concatenate(L["fm"])
        yo = _oracle_life(hb[L["half"]], fs / 2, 12500, L["segments"], L["start"], L["stop"], filt=(D2, t2))
        assert len(y) == len(yo) == len(fm), (seed, L["start"], L["stop"], len(y), len(yo))
        if len(yo) < 8:
            continue
        e = rel_rms(y, yo)
        assert e < 3e-5, (seed, fs, L["half"], L["segments"], L["start"], L["stop"], e)
        fo = G.quadrature_demod_cf(yo, 1.0)
        mag = np.abs(yo)
        ok = np.zeros(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        if ok.sum() > 8:
            assert _fm_rms(fm[ok], fo[ok]) < 1e-4, (seed, L["segments"])


@pytest.mark.parametrize("seed", _seeds())
def test_random_everything_on_one_front_end(gpu_required, seed):
    """One source carrying it all at once, as a deployment does: direct channels (open / close / retune), a 64-bin bank
    with stage-2 channels (open / close / retune) and a scan that is started somewhere in the stream -- ragged pushes, the
    direct channels coming and going (which moves the launch records and the history copy into and out of the bank's
    launch).  Every stream against its oracle."""
    nat = gpu_required
    rng = np.random.default_rng(12000 + seed)
    nb = 64
    fs = nb * 78125.0                                        # 5 Msps
    bw = fs / nb
    cr = 12500
    D, taps = G.channel_params(fs, cr)                       # 200 / 727
    proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    D2, taps2 = G.channel_params(bw, cr)
    N = 1 << int(rng.integers(9, 14))
    F = int(rng.integers(2, 9))
    Lavg = int(rng.integers(1, F + 1))
    n_blocks = int(rng.integers(6, 12))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.25 else int(rng.integers(800, 40 * D)) for _ in range(n_blocks)]
    scan_at = int(rng.integers(0, n_blocks - 1))
    need = N * F + 16
    while sum(sizes[scan_at:]) < need:                       # enough stream behind the scan's start
        sizes.append(int(rng.integers(2000, 40 * D)))
    n_blocks = len(sizes)
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    d_slots = [float(np.round(v / 6250) * 6250) for v in rng.uniform(-0.4, 0.4, int(rng.integers(1, 12))) * fs]
    s_slots = [(int(rng.integers(0, nb)), float(rng.integers(-4, 5)) * 1562.5) for _ in range(int(rng.integers(1, 7)))]
    t = np.arange(len(x)) / fs
    for f in d_slots[:4]:
        x = x + (0.4 * np.exp(2j * np.pi * (f + 300.0) * t)).astype(np.complex64)
    for k, d in s_slots[:4]:
        x = x + (0.4 * np.exp(2j * np.pi * ((k if k <= nb // 2 else k - nb) * bw + d + 500.0) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    d_lives, s_lives = [], []
    spec = None
    feed = str(rng.choice(["pageable", "pinned", "in_place"]))
    pins = [nat.PinnedArray(int(max(sizes)) + 16, np.complex64) for _ in range(2)] if feed == "pinned" else None
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, hist_capacity=max(N, 1 << 12), out_capacity=1 << 12) as fe:
        if rng.random() < 0.5:
            fe.set_rotator(True)
        fe.pfb_open(nb, nb, proto)
        d_live, s_live = {}, {}
        for b in range(n_blocks):
            s0 = int(cuts[b])
            frames = fe.pfb_produced()
            if b == scan_at:
                fe.scan_start(N, F, Lavg)
            for i, f in enumerate(d_slots):
                r = rng.random()
                if i not in d_live and r < (0.4 if b == 0 else 0.15):
                    d_live[i] = dict(id=fe.chan_open(cr, f), start=s0, segments=[(s0, f)], iq=[], fm=[])
                elif i in d_live and r < 0.12:
                    L = d_live.pop(i)
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["stop"] = s0
                    d_lives.append(L)
                elif i in d_live and r < 0.2:
                    f_new = f + 6250.0 * float(rng.integers(-3, 4))
                    fe.chan_set_offset(d_live[i]["id"], f_new)
                    d_live[i]["segments"].append((s0, f_new))
            for i, (k, d) in enumerate(s_slots):
                r = rng.random()
                if i not in s_live and r < (0.6 if b == 0 else 0.2):
                    s_live[i] = dict(id=fe.pfb_chan_open(k, cr, d), bin=k, first=frames, segments=[(frames, d)], iq=[], fm=[])
                elif i in s_live and r < 0.08:
                    L = s_live.pop(i)
                    L["iq"].append(fe.chan_read_iq(L["id"]))
                    L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
                    fe.chan_close(L["id"])
                    L["last"] = frames
                    s_lives.append(L)
                elif i in s_live and r < 0.18:
                    d_new = float(rng.integers(-4, 5)) * 1562.5
                    fe.chan_set_offset(s_live[i]["id"], d_new)
                    s_live[i]["segments"].append((frames, d_new))
            seg = x[s0:int(cuts[b + 1])]
            if feed == "pinned":                             # rcf_push_iq from rcf_host_alloc memory: the copy of this block
                pin = pins[b & 1]                            # overlaps the kernels of the one before
                pin.array[:len(seg)] = seg
                fe.push(pin.array[:len(seg)])
            elif feed == "in_place":                         # rcf_ingest_ptr / rcf_commit: no buffer events at all
                fe.ingest_write(seg, 0)
                fe.commit(len(seg))
            else:
                fe.push(seg)
            if spec is None and b >= scan_at and fe.scan_frames_done() == F:
                spec = fe.scan_result()
        frames = fe.pfb_produced()
        for L in d_live.values():
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["stop"] = int(cuts[-1])
            d_lives.append(L)
        for L in s_live.values():
            L["iq"].append(fe.chan_read_iq(L["id"]))
            L["fm"].append(fe.chan_read_fm(L["id"], 1.0))
            L["last"] = frames
            s_lives.append(L)
    # the scan
    assert spec is not None
    want = OC.scan_chain(x[int(cuts[scan_at]):], N, F, Lavg)
    err = np.abs(spec - want)
    assert err.max() < 0.5 and np.sort(err)[int(0.999 * N)] < 2e-3 and err.mean() < 1e-4, (seed, N, F, Lavg)

    def check(y, fm, yo, what):
        assert len(y) == len(yo) == len(fm), (seed, what, len(y), len(yo), len(fm))
        if len(yo) < 8:
            return
        e = rel_rms(y, yo)
        assert e < 3e-5, (seed, what, e)
        fo = G.quadrature_demod_cf(yo, 1.0)
        mag = np.abs(yo)
        ok = np.zeros(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        if ok.sum() > 8:
            assert _fm_rms(fm[ok], fo[ok]) < 1e-4, (seed, what)

    for L in d_lives:
        check(np.concatenate(L["iq"]), np.concatenate(L["fm"]),
              _oracle_life(x, fs, cr, L["segments"], L["start"], L["stop"]), ("direct", L["segments"], L["start"], L["stop"]))
    stage1 = {}
    for L in s_lives:
        k = L["bin"]
        if k not in stage1:
            stage1[k] = G.xlating_fir_exact(x, nb, proto, (k if k <= nb // 2 else k - nb) * bw, fs).astype(np.complex64)
        yo = _oracle_life(stage1[k], bw, cr, L["segments"], L["first"], L["last"], filt=(D2, taps2))
        check(np.concatenate(L["iq"]), np.concatenate(L["fm"]), yo, ("stage2", k, L["segments"], L["first"], L["last"]))


@pytest.mark.parametrize("seed", _seeds())
def test_random_receiver_sessions_in_pfb_mode(gpu_required, seed):
    """The reference's control plane over the device path, frontend_mode = 'pfb' (5 Msps, 400 bins): connect_channel /
    release_channel / idle re-use (a released channel is RETUNED for the next request: channel.set_offset, which keeps the
    native channel when the serving path stays and replaces it when a bin is involved) / the idle sweep, on-grid and
    off-grid requests, ragged feeds.  Every native channel that lived -- idle ones keep producing, like the reference's
    flowgraphs -- against GNU Radio's channel at the requested offset: direct-served to 3e-5 IQ, bank-served to the
    parity budget (discriminator <= 1e-4, IQ <= 5e-4 behind the filter's warm-up)."""
    import types
    from rcf import receiver
    nat = gpu_required
    rng = np.random.default_rng(15000 + seed)
    fs, fc, cr = 5e6, 850e6, 12500
    D, taps = G.channel_params(fs, cr)
    grid = 12500.0
    n_blocks = int(rng.integers(4, 10))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.2 else int(rng.integers(4 * D, 100 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    wanted = []
    for _ in range(int(rng.integers(2, 9))):
        k = int(rng.integers(-150, 151))
        wanted.append(k * grid if rng.random() < 0.7 else k * grid + float(rng.choice([6250.0, 3125.0, -1250.0])))
    t = np.arange(len(x)) / fs
    for f in wanted:
        x = x + (0.8 * np.exp(2j * np.pi * (f + float(rng.uniform(-1500, 1500))) * t)).astype(np.complex64)
    x = x.astype(np.complex64)
    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=fc, samp_rate=int(fs))}, frontend_mode="pfb")
    tb = receiver.receiver(cfg, frontend_factory=lambda sr, cf, dev: nat.Frontend(
        sr, cf, device=dev, block_capacity=int(max(sizes)) + 16, out_capacity=1 << 11))
    lives = []                                   # finished native lives
    cur = {}                                     # block_id -> life of the native channel it holds now

    def begin(bid, s0):
        ch = tb.channels[bid]
        cur[bid] = dict(chan_id=ch.chan_id, bin=ch.pfb_bin, start=s0, segments=[(s0, ch.offset)], iq=[], fm=[])

    def drain(bid):
        ch = tb.channels[bid]
        cur[bid]["iq"].append(ch.read_iq())
        cur[bid]["fm"].append(ch.read_fm(1.0))

    try:
        assert tb.sources[0].get("pfb") is not None
        in_use = []
        n_bank = 0
        for b in range(n_blocks):
            s0 = int(cuts[b])
            for _ in range(int(rng.integers(0, 4))):
                r = rng.random()
                if r < 0.55 or not in_use:
                    f = float(wanted[int(rng.integers(0, len(wanted)))])
                    before = {bid: tb.channels[bid].chan_id for bid in tb.channels}
                    bid, _ = tb.connect_channel(cr, fc + f)
                    ch = tb.channels[bid]
                    assert ch.offset == f
                    if bid not in before:
                        begin(bid, s0)                                   # a new channel object
                    elif ch.chan_id != before[bid]:                      # re-used, and the native channel was replaced
                        L = cur.pop(bid)                                 # (what it had produced went with the old id)
                        L["stop"] = s0
                        L["lost_tail"] = True
                        lives.append(L)
                        begin(bid, s0)
                    else:                                                # re-used in place: a retune (or the same bin)
                        if cur[bid]["bin"] is None:
                            cur[bid]["segments"].append((s0, f))
                    in_use.append(bid)
                elif r < 0.85:
                    bid = in_use.pop(int(rng.integers(0, len(in_use))))
                    tb.release_channel(bid)
                else:
                    for bid in tb.sweep_idle_channels(now=1e12):         # everything idle goes
                        L = cur.pop(bid)
                        L["stop"] = s0
                        L["lost_tail"] = True
                        lives.append(L)
            # what idle and busy channels produced so far belongs to the life they are in
            tb.feed(0, x[s0:int(cuts[b + 1])])
            for bid in list(cur):
                drain(bid)
        for bid, L in cur.items():
            L["stop"] = int(cuts[-1])
            lives.append(L)
        n_bank = sum(1 for L in lives if L["bin"] is not None)
    finally:
        tb.close()
    for L in lives:                              # (a seed may draw no request at all)
        y = np.concatenate(L["iq"]) if L["iq"] else np.zeros(0, np.complex64)
        fm = np.concatenate(L["fm"]) if L["fm"] else np.zeros(0, np.float32)
        yo = _oracle_life(x, fs, cr, L["segments"], L["start"], L["stop"])
        assert len(y) == len(yo) == len(fm), (seed, L["bin"], L["start"], L["stop"], len(y), len(yo))
        bank = L["bin"] is not None
        warm = ((len(taps) - 1) // D + 1) if (bank and L["start"]) else 0
        if len(yo) < warm + 16:
            continue
        e = rel_rms(y[warm:], yo[warm:])
        assert e < (5e-4 if bank else 3e-5), (seed, "bank" if bank else "direct", L["segments"], L["start"], L["stop"], e)
        fo = G.quadrature_demod_cf(yo, 1.0)
        mag = np.abs(yo)
        ok = np.zeros(len(yo), dtype=bool)
        ok[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
        ok[:warm + 1] = False
        if ok.sum() > 8:
            efm = _fm_rms(fm[ok], fo[ok])
            assert efm < 1e-4, (seed, "bank" if bank else "direct", L["segments"], efm)


@pytest.mark.parametrize("seed", _seeds())
def test_random_threads_on_one_handle(gpu_required, seed):
    """Feeder, two control threads and two readers on ONE handle (librcf serialises on the handle's mutex; a control call
    lands between two blocks wherever the scheduler puts it): direct channels, stage-2 channels, bank taps and scans come
    and go at random while three keepers -- a direct channel, a stage-2 channel, a bin tap -- live through it all and
    must equal their oracles sample for sample; no call may fail."""
    import threading
    nat = gpu_required
    rng = np.random.default_rng(17000 + seed)
    nb = 64
    fs = nb * 78125.0
    bw = fs / nb
    cr = 12500
    D, taps = G.channel_params(fs, cr)
    proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    D2, taps2 = G.channel_params(bw, cr)
    n_blocks = int(rng.integers(30, 80))
    sizes = [int(rng.integers(1, 2 * D)) if rng.random() < 0.2 else int(rng.integers(2000, 30 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    f_keep, k_keep, d_keep, k_tap = 262500.0, 9, 3125.0, 21
    t = np.arange(len(x)) / fs
    x = (x + 0.5 * np.exp(2j * np.pi * (f_keep + 300.0) * t) + 0.5 * np.exp(2j * np.pi * (k_keep * bw + d_keep + 400.0) * t)).astype(np.complex64)
    errors = []
    kept = {"d": [], "s": [], "t": []}
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, hist_capacity=1 << 13, out_capacity=1 << 14) as fe:
        fe.pfb_open(nb, nb, proto)
        keep_d = fe.chan_open(cr, f_keep)
        keep_s = fe.pfb_chan_open(k_keep, cr, d_keep)
        keep_t = fe.pfb_tap_open(k_tap, gr_phase=False)
        stop = threading.Event()

        def feeder():
            try:
                for b in range(n_blocks):
                    fe.push(x[int(cuts[b]):int(cuts[b + 1])])
            except Exception as e:               # pragma: no cover
                errors.append(("feeder", e))
            finally:
                stop.set()

        def control(s):
            r = np.random.default_rng(s)
            live = []
            try:
                while not stop.is_set():
                    u = r.random()
                    if u < 0.25 and len(live) < 20:
                        live.append(fe.chan_open(cr, float(r.integers(-150, 150)) * 6250.0))
                    elif u < 0.4 and len(live) < 20:
                        live.append(fe.pfb_chan_open(int(r.integers(0, nb)), cr, float(r.integers(-4, 5)) * 1562.5))
                    elif u < 0.5 and len(live) < 20:
                        live.append(fe.pfb_tap_open(int(r.integers(0, nb)), gr_phase=bool(r.integers(0, 2))))
                    elif u < 0.7 and live:
                        fe.chan_set_offset(live[int(r.integers(len(live)))], float(r.integers(-100, 100)) * 1562.5)
                    elif u < 0.9 and live:
                        fe.chan_close(live.pop(int(r.integers(len(live)))))
                    elif u < 0.93:
                        fe.scan_start(1 << int(r.integers(8, 13)), int(r.integers(1, 6)), 1)
                    else:
                        fe.scan_result()
                for c in live:
                    fe.chan_close(c)
            except Exception as e:               # pragma: no cover
                errors.append(("control", e))

        def reader(which, cid):
            try:
                while not stop.is_set():
                    kept[which].append(fe.chan_read_iq(cid))
            except Exception as e:               # pragma: no cover
                errors.append(("reader", e))

        th = [threading.Thread(target=feeder), threading.Thread(target=control, args=(seed * 2 + 1,)),
              threading.Thread(target=control, args=(seed * 2 + 2,)), threading.Thread(target=reader, args=("d", keep_d)),
              threading.Thread(target=reader, args=("s", keep_s))]
        for q in th:
            q.start()
        for q in th:
            q.join(timeout=300)
        assert not any(q.is_alive() for q in th), "a thread is stuck"
        kept["d"].append(fe.chan_read_iq(keep_d))
        kept["s"].append(fe.chan_read_iq(keep_s))
        kept["t"].append(fe.chan_read_iq(keep_t))
        bin_t = fe.pfb_read_bin(k_tap)
    assert not errors, errors
    y = np.concatenate(kept["d"])
    yo = _oracle_life(x, fs, cr, [(0, f_keep)], 0, len(x))
    assert len(y) == len(yo) and rel_rms(y, yo) < 3e-5
    s1 = G.xlating_fir_exact(x, nb, proto, k_keep * bw, fs).astype(np.complex64)
    ys = np.concatenate(kept["s"])
    yso = _oracle_life(s1, bw, cr, [(0, d_keep)], 0, len(s1), filt=(D2, taps2))
    assert len(ys) == len(yso) and rel_rms(ys, yso) < 3e-5
    yt = np.concatenate(kept["t"])
    n_ring = min(len(bin_t), len(yt))             # the bin's own ring may have wrapped under the reader; its newest part
    np.testing.assert_array_equal(yt[len(yt) - n_ring:], bin_t[len(bin_t) - n_ring:])
    assert len(yt) == (len(x) - 1) // nb + 1


@pytest.mark.parametrize("seed", _seeds())
def test_random_lagging_readers_get_the_newest_ring_full(gpu_required, seed):
    """Small output rings and readers that come at random: a reader that fell behind by more than the ring holds gets the
    newest ring-full, oldest first, and carries on from there (a PUB socket at its high-water mark drops the same way).
    Channel IQ, its discriminator (own cursor), a stage-2 channel and a filterbank bin, each cursor modelled, every read
    against the oracle's stream at the positions the model says."""
    nat = gpu_required
    rng = np.random.default_rng(19000 + seed)
    nb = 64
    fs = nb * 78125.0
    bw = fs / nb
    cr = 12500
    D, taps = G.channel_params(fs, cr)
    proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    D2, taps2 = G.channel_params(bw, cr)
    cap = 1 << int(rng.integers(6, 10))                      # 64 .. 512 outputs
    n_blocks = int(rng.integers(8, 30))
    max_block = (cap - 16) * nb                              # a block's frames + the history its stage-2 readers reach back must fit the ring (librcf refuses otherwise)
    sizes = [int(rng.integers(1, max_block)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    f0, k2, d2, kb = 187500.0, 7, -1562.5, 11
    t = np.arange(len(x)) / fs
    x = (x + 0.5 * np.exp(2j * np.pi * (f0 + 250.0) * t) + 0.5 * np.exp(2j * np.pi * (k2 * bw + d2 + 350.0) * t)).astype(np.complex64)
    streams = ["iq", "fm", "s2", "bin"]
    cursor = {k: 0 for k in streams}
    produced = {k: 0 for k in streams}
    reads = {k: [] for k in streams}                         # (first index, array)
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, hist_capacity=1 << 13, out_capacity=cap) as fe:
        fe.pfb_open(nb, nb, proto)
        cid = fe.chan_open(cr, f0)
        c2 = fe.pfb_chan_open(k2, cr, d2)
        for b in range(n_blocks):
            fe.push(x[int(cuts[b]):int(cuts[b + 1])])
            produced["iq"] = produced["fm"] = fe.chan_produced(cid)
            produced["s2"] = fe.chan_produced(c2)
            produced["bin"] = fe.pfb_produced()
            for k in streams:
                if rng.random() < 0.25 or b == n_blocks - 1:
                    got = {"iq": lambda: fe.chan_read_iq(cid), "fm": lambda: fe.chan_read_fm(cid, 1.0),
                           "s2": lambda: fe.chan_read_iq(c2), "bin": lambda: fe.pfb_read_bin(kb)}[k]()
                    first = max(cursor[k], produced[k] - cap)
                    assert len(got) == produced[k] - first, (seed, k, b, len(got), produced[k], first, cap)
                    reads[k].append((first, got))
                    cursor[k] = produced[k]
    yo = _oracle_life(x, fs, cr, [(0, f0)], 0, len(x))
    fo = G.quadrature_demod_cf(yo, 1.0)
    s1 = G.xlating_fir_exact(x, nb, proto, k2 * bw, fs).astype(np.complex64)
    s2o = _oracle_life(s1, bw, cr, [(0, d2)], 0, len(s1), filt=(D2, taps2))
    bo = G.xlating_fir_exact(x, nb, proto, kb * bw, fs).astype(np.complex64)
    want = {"iq": yo, "fm": fo, "s2": s2o, "bin": bo}
    for k in streams:
        assert produced[k] == len(want[k]), (seed, k)
        for first, got in reads[k]:
            if len(got) < 4:
                continue
            w = want[k][first:first + len(got)]
            if k == "fm":
                mag = np.abs(yo)
                good = np.zeros(len(yo), dtype=bool)
                good[1:] = (mag[1:] > 0.05 * mag.mean()) & (mag[:-1] > 0.05 * mag.mean())
                ok = good[first:first + len(got)]
                if ok.sum() > 4:
                    assert _fm_rms(got[ok], w[ok]) < 1e-4, (seed, k, first)
            else:
                assert rel_rms(got, w) < 3e-5, (seed, k, first, len(got), rel_rms(got, w))


@pytest.mark.parametrize("seed", _seeds())
def test_random_filterbank_shapes_and_cuts(gpu_required, seed):
    """Every filterbank kernel family under random cuts: power-of-two banks (64 .. 1024 bins, critically sampled and
    oversampled by 2, prototypes of 4 .. 16 taps per branch) and the frame-major 400 / 800 / 1600-bin banks built from the
    reference's channel filter (12.5 and 6.25 kHz rasters: oversampling 1, 2, 4), opened at the start or in mid-stream
    (zero history from there), fed in ragged pushes: bins against the float64 exact-phase bank, and bit for bit against
    the same stream in one push."""
    nat = gpu_required
    rng = np.random.default_rng(21000 + seed)
    if rng.random() < 0.5:
        nb = int(rng.choice([64, 128, 256, 512, 1024]))
        osf = int(rng.choice([1, 2]))
        D = nb // osf
        fs = nb * 25000.0
        P = int(rng.choice([3, 4, 9, 14, 16])) if osf == 1 else int(rng.choice([3, 4, 12, 16]))
        T = nb * P - int(rng.integers(0, nb // 2))
        bw = fs / nb
        proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
        # stretch / cut the design to the tap count drawn (any low-pass will do: the comparison is against the same taps)
        proto = np.interp(np.linspace(0, len(proto) - 1, T), np.arange(len(proto)), proto).astype(np.float32)
    else:
        fs, nb, cr = [(5e6, 400, 12500), (10e6, 800, 12500), (5e6, 800, 12500), (20e6, 1600, 12500), (5e6, 800, 6250),
                      (10e6, 800, 25000), (5e6, 400, 25000)][int(rng.integers(0, 7))]
        D, proto = G.channel_params(fs, cr)
        if nb % D:
            pytest.skip("not a bank shape")
    if not nat.pfb_shape_supported(nb, D, len(proto)):
        pytest.skip("no kernel for bins=%d decim=%d taps=%d" % (nb, D, len(proto)))
    n_frames = int(rng.integers(40, 400 if nb <= 512 else 120))
    lead = int(rng.integers(0, 3 * D)) if rng.random() < 0.5 else 0       # samples before the bank is opened
    n = lead + D * n_frames + int(rng.integers(0, D))
    x = synth.awgn(rng, n)
    cuts = sorted({lead, n} | {int(v) for v in rng.integers(lead + 1, n, int(rng.integers(0, 8)))})
    ks = sorted({int(v) for v in rng.integers(0, nb, 3)})        # (pfb_read_bin is a cursor: each bin once)

    def run(pieces):
        with nat.Frontend(fs, block_capacity=n + 16, hist_capacity=max(1 << 14, len(proto) + 2 * nb), out_capacity=1 << 10) as fe:
            if lead:
                fe.push(x[:lead])
            fe.pfb_open(nb, D, proto)
            for a, b in pieces:
                fe.push(x[a:b])
            return fe.pfb_produced(), [fe.pfb_read_bin(k) for k in ks]

    n_one, one = run([(lead, n)])
    n_cut, cut = run(list(zip(cuts[:-1], cuts[1:])))
    k0 = -(-lead // D)
    assert n_one == n_cut == (n - 1) // D + 1 - k0
    xz = x.copy()
    xz[:lead] = 0
    for k, a, b in zip(ks, one, cut):
        np.testing.assert_array_equal(a, b, err_msg="cut invariance, seed %d bins %d decim %d bin %d" % (seed, nb, D, k))
        ref = G.xlating_fir_exact(xz, D, proto, (k if k <= nb // 2 else k - nb) * fs / nb, fs).astype(np.complex64)[k0:]
        assert len(a) == len(ref), (seed, nb, D, len(a), len(ref))
        assert rel_rms(a, ref) < 3e-5, (seed, nb, D, len(proto), k, rel_rms(a, ref))


@pytest.mark.parametrize("seed", _seeds())
def test_random_structural_churn(gpu_required, seed):
    """Things that go away under their users: the filterbank closed and opened again in another shape in mid-stream (its
    taps and stage-2 channels go with it; channels chained to THOSE starve), a parent channel closed under its chained
    child, ids used after their channel is gone (ENOCHAN, nothing else).  A keeper channel and the last bank's bins must
    come through untouched: the keeper equals the oracle over the whole stream, the bank's bin the oracle's bin with
    zero history from the sample the bank was opened at."""
    nat = gpu_required
    rng = np.random.default_rng(23000 + seed)
    fs, cr = 5e6, 12500
    D, taps = G.channel_params(fs, cr)                       # 200 / 727
    pre = G.low_pass_2(1.0, 25000.0, 6250.0, 1500.0, 30.0, G.WIN_BLACKMAN)
    shapes = [(64, 64), (128, 64), (256, 256), (400, 200), (800, 200)]

    def proto_of(nb, Db):
        if nb in (400, 800):                                 # the reference's 12.5 kHz channel filter on a 12.5 / 6.25 kHz raster
            return G.channel_params(fs, 12500)[1]
        return G.low_pass_2(1.0, fs, fs / nb * 0.4, fs / nb * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)

    n_blocks = int(rng.integers(8, 18))
    sizes = [int(rng.integers(1, 2 * D)) if rng.random() < 0.2 else int(rng.integers(1500, 40 * D)) for _ in range(n_blocks)]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    x = synth.awgn(rng, int(cuts[-1]))
    f_keep = 312500.0
    x = (x + 0.5 * np.exp(2j * np.pi * (f_keep + 200.0) * np.arange(len(x)) / fs)).astype(np.complex64)
    kept = []
    bank = None                                  # (nb, Db, proto, opened_at_sample, bin followed, reads)
    with nat.Frontend(fs, block_capacity=int(max(sizes)) + 16, hist_capacity=1 << 14, out_capacity=1 << 12) as fe:
        keeper = fe.chan_open(cr, f_keep)
        on_bank, chained, parents = [], [], []
        for b in range(n_blocks):
            s0 = int(cuts[b])
            for _ in range(int(rng.integers(0, 4))):
                u = rng.random()
                try:
                    if u < 0.15:
                        if bank is not None:
                            fe.pfb_close()
                            bank = None
                            on_bank = []
                        nb, Db = shapes[int(rng.integers(0, len(shapes)))]
                        if nat.pfb_shape_supported(nb, Db, len(proto_of(nb, Db))):
                            fe.pfb_open(nb, Db, proto_of(nb, Db))
                            bank = dict(nb=nb, Db=Db, proto=proto_of(nb, Db), at=s0, bin=int(rng.integers(0, nb)), reads=[])
                    elif u < 0.22 and bank is not None:
                        fe.pfb_close()
                        bank = None
                        on_bank = []                         # their ids are gone with the bank
                    elif u < 0.4 and bank is not None:
                        k = int(rng.integers(0, bank["nb"]))
                        on_bank.append(fe.pfb_tap_open(k, gr_phase=bool(rng.integers(0, 2))) if rng.random() < 0.5
                                       else fe.pfb_chan_open(k, cr, 0.0))
                    elif u < 0.5 and on_bank:
                        chained.append(fe.chan_open_taps(on_bank[int(rng.integers(len(on_bank)))], 1, pre, 0.0))
                    elif u < 0.6:
                        parents.append(fe.chan_open(cr, float(rng.integers(-150, 150)) * 6250.0))
                    elif u < 0.7 and parents:
                        chained.append(fe.chan_open_taps(parents[int(rng.integers(len(parents)))], 1, pre, 0.0))
                    elif u < 0.8 and parents:
                        fe.chan_close(parents.pop(int(rng.integers(len(parents)))))      # children keep their ids and starve
                    elif u < 0.9 and chained:
                        cid = chained[int(rng.integers(len(chained)))]
                        fe.chan_read_iq(cid)
                        fe.chan_read_fm(cid, 1.0)
                    elif chained:
                        fe.chan_close(chained.pop(int(rng.integers(len(chained)))))
                except nat.RcfError as e:
                    assert e.code in (nat.RCF_ENOCHAN, nat.RCF_ECAP, nat.RCF_EINVAL, nat.RCF_ERANGE), (seed, str(e))
            fe.push(x[s0:int(cuts[b + 1])])
            kept.append(fe.chan_read_iq(keeper))
            if bank is not None:
                bank["reads"].append(fe.pfb_read_bin(bank["bin"]))
    y = np.concatenate(kept)
    yo = _oracle_life(x, fs, cr, [(0, f_keep)], 0, len(x))
    assert len(y) == len(yo) and rel_rms(y, yo) < 2e-5, seed
    if bank is not None and bank["reads"]:
        got = np.concatenate(bank["reads"])
        xz = x.copy()
        xz[:bank["at"]] = 0
        k, nb, Db = bank["bin"], bank["nb"], bank["Db"]
        ref = G.xlating_fir_exact(xz, Db, bank["proto"], (k if k <= nb // 2 else k - nb) * fs / nb, fs).astype(np.complex64)
        ref = ref[-(-bank["at"] // Db):]
        assert len(got) == len(ref), (seed, nb, Db, len(got), len(ref))
        if len(ref) > 4:
            assert rel_rms(got, ref) < 3e-5, (seed, nb, Db, k)


@pytest.mark.parametrize("seed", _seeds())
def test_random_voice_chains_on_every_kind_of_channel(gpu_required, seed):
    """The voice chain on a direct channel, on a stage-2 channel behind a 64-bin bank and on a tapped bin of the 400-bin
    bank, attached at the start or at a random block boundary (a flowgraph started at that channel sample: zero state in
    every stage), small rings that wrap, ragged pushes, random reads -- audio against oracle/audio.py run on the ORACLE's
    channel stream from the attachment on."""
    from oracle import audio as A
    from rcf import audio as host_audio
    nat = gpu_required
    rng = np.random.default_rng(25000 + seed)
    kind = str(rng.choice(["direct", "stage2", "tap"]))
    cr = 12500
    if kind == "stage2":
        nb = 64
        fs = nb * 75000.0                                    # bin rate 75 kHz: stage 2 at /3 gives the chain's 25 kS/s
        Db = nb
        proto = G.low_pass_2(1.0, fs, fs / nb * 0.4, fs / nb * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
        k, delta = int(rng.integers(1, nb // 2)), float(rng.integers(-3, 4)) * 1562.5
        f_sig = k * fs / nb + delta
        out_rate = 25000.0
    elif kind == "tap":
        fs, nb = 5e6, 400
        Db, proto = G.channel_params(fs, cr)
        k = int(rng.integers(1, nb // 2))
        f_sig = k * fs / nb
        out_rate = 25000.0
    else:
        fs = float(rng.choice([2.4e6, 5e6]))
        f_sig = float(np.round(rng.uniform(-0.4, 0.4) * fs / 6250) * 6250)
        out_rate = 25000.0
    D, taps = G.channel_params(fs, cr)
    n_ch = int(rng.integers(2500, 7000))                    # channel samples
    dec = (Db * 3 if kind == "stage2" else Db) if kind != "direct" else D
    n = dec * n_ch + int(rng.integers(0, dec))
    x = synth.nbfm_carrier(n, fs, f_sig, float(rng.uniform(300, 2500)), float(rng.uniform(1000, 3000)), float(rng.uniform(0.1, 0.6)))
    x = (x + 1e-3 * synth.awgn(rng, n)).astype(np.complex64)
    for _ in range(int(rng.integers(0, 3))):
        a = int(rng.integers(0, n_ch - 600)) * dec
        x[a:a + int(rng.integers(200, 1500)) * dec] = 0
    out_cap = 1 << int(rng.integers(12, 15))
    max_step = (out_cap // 4) * dec                          # a block's channel samples + the chain's reach must fit its rings
    cuts = [0]
    while cuts[-1] < n:
        cuts.append(min(n, cuts[-1] + int(rng.integers(1, max_step))))
    attach_at = int(rng.integers(0, len(cuts) - 1)) if rng.random() < 0.5 else 0
    got = []
    with nat.Frontend(fs, block_capacity=max_step + 16, out_capacity=out_cap) as fe:
        if kind == "direct":
            cid = fe.chan_open(cr, f_sig)
        else:
            fe.pfb_open(nb, Db, proto)
            cid = fe.pfb_chan_open(k, cr, delta) if kind == "stage2" else fe.pfb_tap_open(k, gr_phase=False)
        k_attach = None
        for b, (a, e) in enumerate(zip(cuts[:-1], cuts[1:])):
            if b == attach_at:
                k_attach = fe.chan_produced(cid)
                host_audio.open_analog_voice(fe, cid, out_rate)
            fe.push(x[a:e])
            if k_attach is not None and rng.random() < 0.5:
                got.append(fe.chan_read_audio(cid))
        n_audio, n_ungated = fe.chan_audio_produced(cid)
        got.append(fe.chan_read_audio(cid))
    audio = np.concatenate(got)
    if kind == "direct":
        y = _oracle_life(x, fs, cr, [(0, f_sig)], 0, n)
    elif kind == "tap":
        y = G.xlating_fir_exact(x, Db, proto, f_sig, fs).astype(np.complex64)
    else:
        s1 = G.xlating_fir_exact(x, Db, proto, k * fs / nb, fs).astype(np.complex64)
        y = _oracle_life(s1, fs / nb, cr, [(0, delta)], 0, len(s1))
    if len(y) - k_attach < 64:                               # attached too late for the oracle's filters to have anything to do
        assert len(audio) == n_audio <= 32
        return
    try:
        st = A.analog_chain(y[k_attach:], out_rate, stages=True)
    except ValueError:                                       # the squelch let nothing through: the oracle's filters have no input
        assert n_ungated == 0 and len(audio) == n_audio == 0, (seed, kind, n_ungated, n_audio)
        return
    assert n_ungated == len(st["gated"]), (seed, kind, n_ungated, len(st["gated"]))
    assert len(audio) == n_audio == len(st["audio"]), (seed, kind, len(audio), n_audio, len(st["audio"]))
    if len(audio):
        e = float(np.sqrt(np.mean((audio.astype(np.float64) - st["audio"]) ** 2)))
        assert e < 1e-4, (seed, kind, e)


def test_device_picker_names_the_frequencies_the_reference_found(gpu_required):
    """tests/golden/peaks.npz: what /root/reference/fft_peak_detection.py:44-73 itself found on 16 quantised spectra (the
    statements run as they stand, tests/golden/make_peak_goldens.py).  The device picker on the same vectors, injected
    into a finished scan's buffer, through rcf_peak_frequency."""
    import ctypes as C
    nat = gpu_required
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks.npz"))
    for i in range(len(g["meta"])):
        x, want = g["spectrum_%02d" % i], [int(v) for v in g["freqs_%02d" % i]]
        fs, fc = int(g["meta"][i][0]), int(g["meta"][i][1])
        N = len(x)
        with nat.Frontend(float(fs), float(fc), block_capacity=2 * N, hist_capacity=N) as fe:
            fe.scan_start(N, 1, 1)
            fe.push(synth.awgn(np.random.default_rng(i), N))
            dev = C.c_void_p()
            assert nat.lib().rcf_scan_result_device(fe._h, C.byref(dev)) == 0 and dev.value
            _hip_memcpy_h2d(dev.value, x)
            idx, _, _ = fe.scan_find_peaks(cap=4096)
        assert [nat.peak_frequency(int(l), fs, N, fc) for l in idx] == want, i


@pytest.mark.parametrize("seed", _seeds())
def test_random_batched_reads_equal_single_reads_under_churn(gpu_required, seed):
    """rcf_chan_read_many (round 4: the egress pump's read -- one gather launch into pinned memory, one sync) against
    rcf_chan_read_iq / _fm channel by channel on a twin front-end fed the same stream: random channel sets (direct,
    filterbank taps, stage-2 channels), opened and closed at random block boundaries, ragged pushes, small rings (wraps
    and lagging readers), random capacities per call, ids of closed channels left in the list.  Bit for bit."""
    nat = gpu_required
    rng = np.random.default_rng(31000 + seed)
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    n_blocks = int(rng.integers(6, 14))
    sizes = [int(rng.integers(1, 3 * D)) if rng.random() < 0.2 else int(rng.integers(2000, 30 * D)) for _ in range(n_blocks)]
    x = synth.awgn(rng, int(np.sum(sizes)))
    cap_log2 = int(rng.integers(8, 11))
    with nat.Frontend(fs, block_capacity=max(sizes) + 16, hist_capacity=1 << 14, out_capacity=1 << cap_log2) as a, \
            nat.Frontend(fs, block_capacity=max(sizes) + 16, hist_capacity=1 << 14, out_capacity=1 << cap_log2) as b:
        for fe in (a, b):
            fe.pfb_open(400, D, taps)
        ia, ib, dead = [], [], []

        at = 0
        for blk, n in enumerate(sizes):
            for _ in range(int(rng.integers(0, 4))):
                u = rng.random()
                if u < 0.6 or not ia:
                    r = rng.random()
                    if r < 0.5:
                        f = float(rng.integers(-190, 190)) * 12500.0 + float(rng.choice([0.0, 300.0]))
                        ia.append(a.chan_open(12500, f)); ib.append(b.chan_open(12500, f))
                    elif r < 0.8:
                        k, g = int(rng.integers(0, 400)), bool(rng.integers(0, 2))
                        ia.append(a.pfb_tap_open(k, gr_phase=g)); ib.append(b.pfb_tap_open(k, gr_phase=g))
                    else:
                        k = int(rng.integers(0, 400))
                        ia.append(a.pfb_chan_open(k, 6250, 100.0)); ib.append(b.pfb_chan_open(k, 6250, 100.0))
                else:
                    j = int(rng.integers(0, len(ia)))
                    a.chan_close(ia[j]); b.chan_close(ib[j])
                    dead.append(ib[j])
                    del ia[j]; del ib[j]
            a.push(x[at:at + n]); b.push(x[at:at + n])
            at += n
            if rng.random() < 0.3:
                continue                                        # nobody reads this block: lag, perhaps past the ring
            cap = int(rng.choice([64, 257, 1 << cap_log2, 1 << 12]))
            what = "iq" if rng.random() < 0.6 else "fm"
            ids = list(ib) + ([dead[-1]] if dead and rng.random() < 0.5 else [])
            order = rng.permutation(len(ids))
            got = b.chan_read_many([ids[i] for i in order], what, gain=2.5, cap_each=cap)
            for pos, i in enumerate(order):
                if i >= len(ib):
                    assert got[pos] is None
                    continue
                one = a.chan_read_iq(ia[i], cap) if what == "iq" else a.chan_read_fm(ia[i], 2.5, cap)
                assert len(one) == len(got[pos]) and np.array_equal(one, got[pos]), (seed, blk, what, cap, i, len(one), len(got[pos]))
