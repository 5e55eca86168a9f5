"""Round-4 parity cases: the two-branch filterbank kernel (pfb_kernel_2b, 512 / 1024 critically sampled bins) against a
float64 filterbank over EVERY bin -- its decimation-in-time combine writes bins k and k + NB/2 from one lane, so every
bin is checked, not a sample of them."""
import numpy as np
import pytest

from oracle import grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _bank_f64(x, nb, taps, n_frames, start=0):
    """out[n, k] = sum_i h[i] e^{+2 pi j k i / nb} x[n nb - i] (== freq_xlating_fir_filter_ccc(nb, h, k fs / nb, fs) for
    every on-grid k: rc_frontend/channel.py:35), samples before `start` zero; float64."""
    P = -(-len(taps) // nb)
    h = np.zeros(P * nb)
    h[: len(taps)] = taps
    xz = x.astype(np.complex128).copy()
    xz[:start] = 0
    n_first = -(-start // nb)
    pad = np.concatenate([np.zeros(P * nb, dtype=np.complex128), xz])
    out = np.empty((n_frames, nb), dtype=np.complex128)
    rho = np.arange(nb)
    for f in range(n_frames):
        n = n_first + f
        u = np.zeros(nb, dtype=np.complex128)
        for p in range(P):
            u += h[p * nb + rho] * pad[P * nb + n * nb - rho - p * nb]
        out[f] = np.fft.ifft(u) * nb
    return out


@pytest.mark.parametrize("nb,P,lead", [(512, 14, 0), (512, 4, 700), (512, 16, 0), (1024, 14, 1500), (1024, 4, 0)])
def test_two_branch_bank_every_bin(gpu_required, nb, P, lead):
    nat = gpu_required
    fs = nb * 25000.0
    bw = fs / nb
    proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    T = nb * P - nb // 3
    proto = np.interp(np.linspace(0, len(proto) - 1, T), np.arange(len(proto)), proto).astype(np.float32)
    assert nat.pfb_shape_supported(nb, nb, T)
    n_frames = 16 * 9 + 5                                  # nine whole chunks and a ragged one
    n = lead + nb * n_frames
    rng = np.random.default_rng(nb + P)
    x = synth.awgn(rng, n)
    cut = lead + nb * 37 + 11                              # the second launch has history, the first does not
    with nat.Frontend(fs, block_capacity=n + 16, hist_capacity=max(1 << 15, 2 * T + 2 * nb), out_capacity=1 << 9) as fe:
        if lead:
            fe.push(x[:lead])
        fe.pfb_open(nb, nb, proto)
        fe.push(x[lead:cut])
        fe.push(x[cut:])
        produced = fe.pfb_produced()
        got = np.stack([fe.pfb_read_bin(k) for k in range(nb)], axis=1)
    n_first = -(-lead // nb)
    want_frames = (n - 1) // nb + 1 - n_first
    assert produced == want_frames == got.shape[0]
    want = _bank_f64(x, nb, proto, want_frames, start=lead)
    scale = np.sqrt(np.mean(np.abs(want) ** 2))
    err_bin = np.sqrt(np.mean(np.abs(got - want) ** 2, axis=0)) / scale
    assert err_bin.max() < 2e-5, (int(err_bin.argmax()), float(err_bin.max()))


def test_batched_read_equals_channel_by_channel_reads(gpu_required):
    """rcf_chan_read_many (one gather launch + one synchronisation for all channels: the egress pump's read) hands
    out exactly what rcf_chan_read_iq / rcf_chan_read_fm do channel by channel -- across ring wrap, ragged pushes, a
    closed channel in the list and a reader that lagged more than a ring."""
    nat = gpu_required
    fs = 2.4e6
    rng = np.random.default_rng(44)
    x = synth.awgn(rng, 96 * 9000)
    offs = [-300e3, -62500.0, 0.0, 125e3, 412.5e3]

    def open_all(fe):
        return [fe.chan_open(12500, f) for f in offs]

    with nat.Frontend(fs, out_capacity=1 << 11) as a, nat.Frontend(fs, out_capacity=1 << 11) as b:
        ia, ib = open_all(a), open_all(b)
        one = {i: ([], []) for i in range(len(offs))}
        many = {i: ([], []) for i in range(len(offs))}
        cuts = [0, 96 * 700 + 5, 96 * 1900, 96 * 2000 + 17, 96 * 3900, 96 * 5500, 96 * 7000, len(x)]
        unread = {3, 4}                                    # 1900 + 1600 outputs pile up unread: more than the 2048 ring
        for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            a.push(x[lo:hi])
            b.push(x[lo:hi])
            if k == 2:
                a.chan_close(ia[1])
                b.chan_close(ib[1])
            if k in unread:
                continue
            for i, c in enumerate(ia):
                if k >= 2 and i == 1:
                    continue
                one[i][0].append(a.chan_read_iq(c))
                one[i][1].append(a.chan_read_fm(c, 5.0))
            iq = b.chan_read_many(ib, "iq", cap_each=4096)
            fm = b.chan_read_many(ib, "fm", gain=5.0, cap_each=4096)
            for i in range(len(offs)):
                if k >= 2 and i == 1:
                    assert iq[i] is None and fm[i] is None
                    continue
                many[i][0].append(iq[i].copy())
                many[i][1].append(fm[i].copy())
        for i in range(len(offs)):
            for w in (0, 1):
                p, q = np.concatenate(one[i][w]), np.concatenate(many[i][w])
                assert len(p) == len(q) > 0 and np.array_equal(p, q), (i, w, len(p), len(q))
        # cap_each smaller than what is waiting: the rest stays for the next call
        a.push(x[: 96 * 500])
        b.push(x[: 96 * 500])
        first = b.chan_read_many([ib[0]], "iq", cap_each=100)[0].copy()
        rest = b.chan_read_many([ib[0]], "iq", cap_each=4096)[0].copy()
        assert len(first) == 100 and np.array_equal(np.concatenate([first, rest]), a.chan_read_iq(ia[0]))
        # a channel listed twice has one reader position: the second mention is refused, nothing is skipped or repeated
        a.push(x[: 96 * 300])
        b.push(x[: 96 * 300])
        twice = b.chan_read_many([ib[0], ib[2], ib[0]], "iq", cap_each=4096)
        assert twice[2] is None and np.array_equal(twice[0], a.chan_read_iq(ia[0])) and np.array_equal(twice[1], a.chan_read_iq(ia[2]))
        a.push(x[: 96 * 300])
        b.push(x[: 96 * 300])
        assert np.array_equal(b.chan_read_many([ib[0]], "iq", cap_each=4096)[0], a.chan_read_iq(ia[0]))


def test_python2_decimation_rule_opens_the_10p67_msps_channels(gpu_required):
    """configs/config_denver_massive_p25.py:20,31 (10 666 666 sps, receiver_split2 = False): int(fs/cr)/2 = 426.5 under
    Python 3 -- refused by default -- and 853 // 2 = 426 under the Python 2 the line was written for.  With
    rcf_set_decim_rule(RCF_DECIM_FLOOR) the channel opens at D = 426, 25 039 S/s, and equals the oracle's channel."""
    from oracle import cbind as OC
    nat = gpu_required
    fs, cr, f0 = 10666666.0, 12500, 771106250 - 771500000      # a control channel of that config's system 0
    rng = np.random.default_rng(426)
    D, taps = G.channel_params(fs, cr, py2_floor=True)
    x = synth.awgn(rng, D * 900 + 77)
    x = (x + synth.nbfm_carrier(len(x), fs, f0, 1000.0, 2500.0, synth.snr_amp(30.0, 12500.0, fs))).astype(np.complex64)
    with nat.Frontend(fs) as fe:
        with pytest.raises(nat.RcfError) as e:
            fe.chan_open(cr, f0)
        assert e.value.code == nat.RCF_ERANGE
        fe.set_decim_rule(nat.DECIM_FLOOR)
        cid = fe.chan_open(cr, f0)
        info = fe.chan_info(cid)
        assert info["decim"] == 426 and info["ntaps"] == len(taps) == 1551 and abs(info["out_rate"] - fs / 426) < 1e-6
        fe.push(x[: D * 333 + 5])
        fe.push(x[D * 333 + 5:])
        y, fm = fe.chan_read_iq(cid), fe.chan_read_fm(cid, 5.0)
    ct, inc = OC.xlating_composite(taps, D, float(f0), fs)
    yo, fo = OC.channel_bank(x, D, ct[None, :], np.array([inc]), gains=[5.0])
    assert len(y) == len(yo[0])
    assert np.sqrt(np.mean(np.abs(y - yo[0]) ** 2) / np.mean(np.abs(yo[0]) ** 2)) < 1e-5
    assert np.sqrt(np.mean((fm - fo[0]) ** 2)) < 1e-4


def test_a_block_whose_planning_fails_leaves_no_trace(gpu_required, tmp_path):
    """planning advances every channel's counters and can still fail afterwards (allocation, launch arena).
    Run in a child process with RCF_FAIL_PLAN_AT=3: the third block is refused -- and refused cleanly: counters where
    they were, and the same block pushed again continues the streams as if nothing had happened (direct channels with
    the exact rotator, a filterbank with a tapped bin and a stage-2 channel: all against the uninterrupted run)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path[:0] = ["radiocapture-rf_amd", "."]
from rcf import native, synth
from oracle import grspec as G
fs = 5e6
rng = np.random.default_rng(9)
D, taps = G.channel_params(fs, 12500)
x = synth.awgn(rng, D * 600 + 13)
cuts = [0, D * 100 + 7, D * 250, D * 251 + 3, D * 420, len(x)]
def run(fail):
    out = {}
    with native.Frontend(fs, hist_capacity=1 << 14, out_capacity=1 << 11) as fe:
        fe.set_rotator(True)
        a = fe.chan_open(12500, 312500.0)
        fe.pfb_open(400, D, taps)
        b = fe.pfb_tap_open(21, gr_phase=True)
        refused = 0
        for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            before = (fe.chan_produced(a), fe.chan_produced(b), fe.pfb_produced(), fe.samples_in)
            try:
                fe.push(x[lo:hi])
            except native.RcfError as e:
                assert fail and e.code == native.RCF_ENOMEM, e
                refused += 1
                assert (fe.chan_produced(a), fe.chan_produced(b), fe.pfb_produced(), fe.samples_in) == before
                fe.push(x[lo:hi])
        out = (fe.chan_read_iq(a), fe.chan_read_fm(a, 5.0), fe.chan_read_iq(b), fe.pfb_read_bin(3))
    return refused, out
import os
refused, got = run(True)
assert refused == 1, refused
np.savez(sys.argv[1], *got)
"""
    env = dict(os.environ, RCF_FAIL_PLAN_AT="3")
    out_fail = str(tmp_path / "fail.npz")
    r = subprocess.run([sys.executable, "-c", code, out_fail], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    env.pop("RCF_FAIL_PLAN_AT")
    out_ok = str(tmp_path / "ok.npz")
    r = subprocess.run([sys.executable, "-c", code.replace("run(True)", "run(False)").replace("assert refused == 1, refused", "assert refused == 0"),
                        out_ok], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(out_fail), np.load(out_ok)
    for k in a.files:
        assert len(a[k]) == len(b[k]) > 0 and np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("fmt_name", ["u8", "s16"])
def test_raw_blocks_from_pinned_memory_are_converted_in_place_and_equal_the_staged_copy(gpu_required, fmt_name):
    """rcf_push_raw has two ingest paths since round 4: a pinned block of <= 4 MiB is converted straight out of host
    memory (one launch, no staging copy), anything else goes through the staged copy.  Same samples, three ways --
    pinned (direct), pageable (staged), and cf32 converted on the host with the same float32 operations -- mixed with
    cf32 pushes on the same handle: channel and filterbank outputs bit for bit the same."""
    nat = gpu_required
    from rcf import sources
    fs = 2.4e6
    rng = np.random.default_rng(808)
    n = 96 * 700 + 33
    x = synth.awgn(rng, n) * np.float32(1.5)
    raw = sources.to_wire(x, fmt_name)
    scale, off = sources.WIRE_SCALE[fmt_name]
    fmt = {"u8": nat.FMT_U8, "s16": nat.FMT_S16}[fmt_name]
    host = ((raw.astype(np.float32) - np.float32(off)) * np.float32(scale)).view(np.complex64)
    cuts = [0, 96 * 100 + 7, 96 * 101, 96 * 350 + 1, 96 * 351, n]
    taps = G.low_pass_2(1.0, fs, fs / 64 * 0.4, fs / 64 * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)

    def run(mode):
        pin = nat.PinnedArray(2 * n, raw.dtype)
        pin.array[:] = raw
        with nat.Frontend(fs) as fe:
            c = fe.chan_open(12500, -62500.0)
            fe.pfb_open(64, 64, taps)
            for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                if k == 2:                                      # one block as cf32 in between: the paths interleave
                    if mode == "pinned":                        # ... from pinned memory too (rcf_push_iq's in-place copy)
                        pc = nat.PinnedArray(hi - lo, np.complex64)
                        pc.array[:] = host[lo:hi]
                        fe.push(pc.array)
                        pc.free()
                    else:
                        fe.push(host[lo:hi])
                elif mode == "pinned":
                    fe.push_raw(pin.array[2 * lo:2 * hi], fmt, scale, off)
                elif mode == "pageable":
                    fe.push_raw(raw[2 * lo:2 * hi].copy(), fmt, scale, off)
                else:
                    fe.push(host[lo:hi])
            out = (fe.chan_read_iq(c), fe.chan_read_fm(c, 5.0), fe.pfb_read_bin(5))
        pin.free()
        return out

    a, b, c = run("pinned"), run("pageable"), run("host")
    for u, v, w in zip(a, b, c):
        assert len(u) == len(v) == len(w) > 0
        assert np.array_equal(u, v) and np.array_equal(u, w)


def test_filterbank_launch_timing_with_attached_events(gpu_required):
    """rcf_timing_*: the filterbank launch carries its two HIP events attached to the dispatch (PfbLaunch::ev_start /
    ev_stop): every timed launch is counted once, the time is a kernel's (microseconds, not zero, not the wall clock of
    the loop), the stride picks every n-th launch, and outputs are the same bits with timing on or off -- for a
    power-of-two bank and for a frame-major one."""
    nat = gpu_required
    rng = np.random.default_rng(44)
    for nb, fs in ((256, 20e6), (400, 5e6)):
        if nb == 256:
            D = nb
            taps = nat.design_low_pass_2(1.0, fs, 0.4 * fs / nb, 0.2 * fs / nb, 60.0, nat.WIN_BLACKMAN_HARRIS)
        else:
            D, _ = nat.channel_params(fs, 12500)
            taps = nat.design_low_pass_2(1.0, fs, 6250.0, 6250.0, 20.0)
        x = (rng.standard_normal(D * 640) + 1j * rng.standard_normal(D * 640)).astype(np.complex64)
        outs = []
        for timed in (False, True):
            with nat.Frontend(fs, block_capacity=len(x), out_capacity=1 << 12) as fe:
                fe.pfb_open(nb, D, taps)
                if timed:
                    fe.timing_enable(True, classes=[nat.T_PFB])
                for _ in range(6):
                    fe.push(x)
                outs.append(fe.pfb_read_bin(3, 1 << 12))
                if timed:
                    ms, n = fe.timing_read(nat.T_PFB, reset=True)
                    assert n == 6 and 0.0 < ms / n < 5.0, (nb, ms, n)
                    fe.timing_stride(3)
                    for _ in range(6):
                        fe.push(x)
                    ms, n = fe.timing_read(nat.T_PFB)
                    assert n == 2 and 0.0 < ms / n < 5.0, (nb, ms, n)
        assert len(outs[0]) > 0 and np.array_equal(outs[0], outs[1])
