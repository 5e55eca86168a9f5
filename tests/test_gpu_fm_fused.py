"""rcf_pfb_fm_enable: the discriminator of EVERY bin of a reference-grid filterbank (bin k = the reference's channel at
offset k fs / NB: rc_frontend/channel.py:31-35, demodulated by quadrature_demod_cf in p25_control_demod.py:120-121 /
logging_receiver.py:214) computed inside the bank's own kernel and written to a frame-major ring -- 8 + 8 bytes per input
sample at OS = 2 instead of the 8 + 16 of the bank plus the 16 + 8 of a tap_finalize pass behind it.  The bits are those of a
discriminator-only tap (rcf_chan_set_fm_only) on the same bin, however the stream is cut and whichever mode (beside the bins
ring / instead of it); against the GNU-Radio-faithful oracle the north-star budget (1e-4 rms) holds with two orders to spare."""
import numpy as np
import pytest

from oracle import cbind as OC, grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _signal(rng, fs, n, nb, carriers):
    x = synth.awgn(rng, n).astype(np.complex128)
    t = np.arange(n) / fs
    for i, k in enumerate(carriers):
        f0 = (k if k < nb // 2 else k - nb) * fs / nb
        x += 6.0 * np.exp(2j * np.pi * (f0 * t + (0.2 + 0.05 * i) * np.sin(2 * np.pi * (500 + 130 * i) * t)))
    return x.astype(np.complex64)


def _fused(nat, fs, nb, D, taps, x, cuts, bins, mode, gr_phase=True, span=None, tap_bins=()):
    with nat.Frontend(fs, 0.0, device=0, block_capacity=max(cuts), hist_capacity=1 << 15, out_capacity=1 << 12) as fe:
        fe.pfb_open(nb, D, taps)
        ids = [fe.pfb_tap_open(b, gr_phase=gr_phase) for b in tap_bins]
        fe.pfb_fm_enable(mode, gr_phase=gr_phase)
        fm = [[] for _ in bins]
        tap_iq = [[] for _ in ids]
        at = 0
        for n in cuts:
            fe.push(x[at:at + n])
            at += n
            for i, b in enumerate(bins):
                fm[i].append(fe.pfb_read_fm(b, 1.0))
            for i, c in enumerate(ids):
                tap_iq[i].append(fe.chan_read_iq(c))
        if mode == 2:
            with pytest.raises(nat.RcfError):
                fe.pfb_read_bin(bins[0])
        if mode:
            assert fe.pfb_fm_lost() == 0                           # every chunk's predecessor frame arrived
        return [np.concatenate(f) for f in fm], [np.concatenate(f) for f in tap_iq]


def _fm_only_taps(nat, fs, nb, D, taps, x, cuts, bins, gr_phase=True):
    with nat.Frontend(fs, 0.0, device=0, block_capacity=max(cuts), hist_capacity=1 << 15, out_capacity=1 << 12) as fe:
        fe.pfb_open(nb, D, taps)
        ids = [fe.pfb_tap_open(b, gr_phase=gr_phase) for b in bins]
        for c in ids:
            fe.chan_set_fm_only(c, True)
        fm = [[] for _ in ids]
        at = 0
        for n in cuts:
            fe.push(x[at:at + n])
            at += n
            for i, c in enumerate(ids):
                fm[i].append(fe.chan_read_fm(c, 1.0))
        return [np.concatenate(f) for f in fm]


SHAPES = [
    # fs, channel rate -> (NB, D): the reference's own rule, 12.5 kHz raster (OS = 2) ...
    (20e6, 12500, 2),
    (10e6, 12500, 2),
    (5e6, 12500, 2),
    # ... and the 6.25 kHz raster (OS = 4, one tap per branch)
    (20e6, 12500, 4),
]


@pytest.mark.parametrize("fs,cr,os_", SHAPES)
def test_fused_discriminator_has_the_bits_of_a_discriminator_only_tap(gpu_required, fs, cr, os_):
    nat = gpu_required
    D, taps = G.channel_params(fs, cr)
    nb = os_ * D
    if os_ == 4:
        taps = taps[: nb]                                          # one tap per branch (the kernel instantiated for OS = 4)
    rng = np.random.default_rng(int(fs / 1e6) * 10 + os_)
    frames = 150
    bins = [0, 1, 5, 17, nb // 2 - 1, nb // 2, nb - 3, nb - 1, 321 % nb, 640 % nb, 959 % nb]
    x = _signal(rng, fs, D * frames + 11, nb, [5, 17, nb - 3])
    cuts_a = [len(x)]
    cuts_b = [D * 40 + 5, D * 3, 1, D * 57 - 6, len(x) - (D * 100)]
    assert sum(cuts_b) == len(x)
    ref = _fm_only_taps(nat, fs, nb, D, taps, x, cuts_a, bins)
    both, _ = _fused(nat, fs, nb, D, taps, x, cuts_a, bins, 1)
    only, _ = _fused(nat, fs, nb, D, taps, x, cuts_b, bins, 2)
    for b, r, a_, o_ in zip(bins, ref, both, only):
        assert len(r) >= frames - 1 and len(a_) == len(r) == len(o_), (b, len(r), len(a_), len(o_))
        # the bits of the discriminator-only tap (tap_finalize's arithmetic), whichever mode and however the stream is cut
        assert _same_bits(a_, r), (b, float(np.max(np.abs(a_ - r))), int(np.argmax(np.abs(a_ - r))))
        assert _same_bits(o_, r), (b, float(np.max(np.abs(o_ - r))), int(np.argmax(np.abs(o_ - r))))


def test_fused_discriminator_within_budget_of_the_gr_faithful_oracle(gpu_required):
    nat = gpu_required
    fs, cr = 5e6, 12500
    D, taps = G.channel_params(fs, cr)
    nb = 2 * D
    rng = np.random.default_rng(4242)
    carriers = [7, 33, nb - 12]
    x = _signal(rng, fs, D * 400, nb, carriers)
    fm, _ = _fused(nat, fs, nb, D, taps, x, [D * 150, D * 250], carriers, 2)
    for k, got in zip(carriers, fm):
        f0 = (k if k < nb // 2 else k - nb) * fs / nb
        ct, incr = OC.xlating_composite(taps, D, f0, fs)
        _, fo = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[1.0])
        n = min(len(got), len(fo[0]))
        assert n >= 399
        d = np.angle(np.exp(1j * (got[:n].astype(np.float64) - fo[0][:n])))
        rms = float(np.sqrt(np.mean(d[2:] ** 2)))
        assert rms < 1e-4, (k, rms)


def test_fused_discriminator_every_span_the_same_bits(gpu_required, monkeypatch):
    """the chunks one workgroup walks (RCF_PFB5_FM_SPAN) change which chunk is somebody's halo, never the result; and the
    look-back form (one chunk per workgroup, the predecessor frame handed over through global memory) has the same bits"""
    import os
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys, hashlib, numpy as np
        sys.path[:0] = [%r, %r]
        from rcf import native as nat, synth
        from oracle import grspec as G
        fs = 20e6
        D, taps = G.channel_params(fs, 12500)
        # 7000 frames = 1750 chunks in the second launch: more workgroups than the chip holds at once (768)
        x = synth.awgn(np.random.default_rng(5), D * 7300 + 7)
        h = hashlib.sha256()
        with nat.Frontend(fs, 0.0, device=0, block_capacity=len(x), hist_capacity=1 << 15, out_capacity=1 << 13) as fe:
            fe.pfb_open(2 * D, D, taps)
            fe.pfb_fm_enable(2, gr_phase=True)
            fe.push(x[: D * 300 + 3]); fe.push(x[D * 300 + 3:])
            for b in range(0, 2 * D, 37):
                h.update(fe.pfb_read_fm(b, 1.0).tobytes())
            assert fe.pfb_fm_lost() == 0
        print(h.hexdigest())
    """) % (os.path.join(os.path.dirname(__file__), ".."), os.path.join(os.path.dirname(__file__), "..", "radiocapture-rf_amd"))
    digests = set()
    for span in ("1", "2", "5", "16", "", "lookback"):
        env = dict(os.environ)
        env.pop("RCF_PFB5_FM_SPAN", None)
        env["RCF_PFB5_FM_LOOKBACK"] = "1" if span == "lookback" else "0"      # (the look-back form has no spans)
        if span and span != "lookback":
            env["RCF_PFB5_FM_SPAN"] = span
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.add(out.stdout.strip().splitlines()[-1])
    assert len(digests) == 1, digests


def test_fused_discriminator_beside_taps_and_switching(gpu_required):
    """taps of the same bank keep their IQ streams (tap matrix) while the bank demodulates every bin; switching the fused
    discriminator off and on again restarts the ring at the frame of the switch"""
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    nb = 2 * D
    rng = np.random.default_rng(91)
    x = _signal(rng, fs, D * 300, nb, [21])
    with nat.Frontend(fs, 0.0, device=0, block_capacity=D * 100, hist_capacity=1 << 15, out_capacity=1 << 12) as fa, \
            nat.Frontend(fs, 0.0, device=0, block_capacity=D * 100, hist_capacity=1 << 15, out_capacity=1 << 12) as fb:
        for f in (fa, fb):
            f.pfb_open(nb, D, taps)
        ta = [fa.pfb_tap_open(b, gr_phase=True) for b in (21, 140)]
        tb = [fb.pfb_tap_open(b, gr_phase=True) for b in (21, 140)]
        fb.pfb_fm_enable(2, gr_phase=True)
        got_fm, want_fm = [], []
        for i in range(3):
            if i == 1:
                fb.pfb_fm_enable(0)
            if i == 2:
                fb.pfb_fm_enable(1, gr_phase=True)
            fa.push(x[i * D * 100:(i + 1) * D * 100])
            fb.push(x[i * D * 100:(i + 1) * D * 100])
            for ca, cb in zip(ta, tb):
                assert _same_bits(fa.chan_read_iq(ca), fb.chan_read_iq(cb))
            want_fm.append(fa.chan_read_fm(ta[0], 1.0))
            got_fm.append(fb.pfb_read_fm(21, 1.0))
        assert len(got_fm[0]) == 100 and len(got_fm[1]) == 0 and len(got_fm[2]) == 100
        for i in (0, 2):
            d = np.angle(np.exp(1j * (got_fm[i].astype(np.float64) - want_fm[i])))
            assert np.max(np.abs(d)) < 2e-6, (i, float(np.max(np.abs(d))))
        # mode 1 keeps the bins ring: the bin read back is the tap's bare stream
        assert len(fb.pfb_read_bin(21)) > 0


def test_fused_discriminator_in_a_grouped_launch_has_the_bits_of_one_by_one(gpu_required):
    """G front-ends whose banks demodulate every bin, pushed as group blocks (rcf_group_push: ONE bank launch for all of
    them, pfb5_fmlb_group_kernel) -- ragged rounds, a member skipped in one of them: the discriminator rings hold the
    bits of the same front-ends run one by one."""
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    nb = 2 * D
    G_ = 3
    rng = np.random.default_rng(313)
    xs = [_signal(rng, fs, D * 900 + 50, nb, [11 + 40 * m, nb - 7 - m]) for m in range(G_)]
    rounds = [[D * 100 + 3, D * 80, D * 120], [D * 200, D * 150 + 1, 0], [D * 300 - 3, D * 250, D * 260 + 9], [D * 90, 0, D * 100]]
    bins = [0, 11, 51, 91, nb - 7, nb - 8, nb - 9, nb // 2, 333]
    out = {}
    for which in ("grouped", "one by one"):
        fes = []
        for m in range(G_):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=D * 300, hist_capacity=1 << 15, out_capacity=1 << 12)
            fe.pfb_open(nb, D, taps)
            fe.pfb_fm_enable(2, gr_phase=True)
            fes.append(fe)
        grp = nat.Group(fes) if which == "grouped" else None
        at = [0] * G_
        got = [[[] for _ in bins] for _ in range(G_)]
        for r in rounds:
            blocks = [xs[m][at[m]:at[m] + r[m]] if r[m] else None for m in range(G_)]
            for m in range(G_):
                at[m] += r[m]
            if grp is not None:
                grp.push(blocks, nat.FMT_CF32)
            else:
                for m in range(G_):
                    if blocks[m] is not None:
                        fes[m].push(blocks[m])
            for m in range(G_):
                for i, b in enumerate(bins):
                    got[m][i].append(fes[m].pfb_read_fm(b, 1.0))
        for fe in fes:
            assert fe.pfb_fm_lost() == 0
        if grp is not None:
            grp.close()
        for fe in fes:
            fe.close()
        out[which] = [[np.concatenate(g_) for g_ in gm] for gm in got]
    for m in range(G_):
        for i, b in enumerate(bins):
            a_, o_ = out["grouped"][m][i], out["one by one"][m][i]
            assert len(a_) == len(o_) > 400, (m, b, len(a_), len(o_))
            assert _same_bits(a_, o_), (m, b, float(np.max(np.abs(a_ - o_))), int(np.argmax(np.abs(a_ - o_))))


def test_pump_delivers_bins_of_the_fused_discriminator_ring(gpu_required):
    """rcf_pump_subscribe(member, RCF_SRC_PFB_BIN0 + bin, RCF_READ_FM): bins of the bank's own discriminator ring delivered
    into host rings by the native pump (no tap, no tap matrix, no tap_finalize) -- two front-ends in one group, some bins
    subscribed at the start, one while it runs; what arrives is, bit for bit, gain x what rcf_pfb_read_fm hands out for
    the same blocks pushed one front-end at a time."""
    import time
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    nb = 2 * D
    blk, n_blocks = 100000, 12
    rng = np.random.default_rng(515)
    xs = [_signal(rng, fs, blk * n_blocks, nb, [21 + 9 * m, nb - 30 - m]) for m in range(2)]
    u8 = [np.clip(np.round(x.view(np.float32) * 32.0 + 127.4), 0, 255).astype(np.uint8) for x in xs]
    gain = 3.5

    def open_all():
        fes = []
        for m in range(2):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=blk, hist_capacity=1 << 15, out_capacity=1 << 13)
            fe.pfb_open(nb, D, taps)
            fe.pfb_fm_enable(2, gr_phase=True)
            fes.append(fe)
        return fes

    fes = open_all()
    rings = []
    for m in range(2):
        r = nat.PinnedArray(len(u8[m]), np.uint8)
        r.array[:] = u8[m]
        rings.append(r)
    subs = [(0, nat.SRC_PFB_BIN0 + 21), (0, nat.SRC_PFB_BIN0 + nb - 30), (1, nat.SRC_PFB_BIN0 + 30)]
    grp = nat.Group(fes)
    pump = nat.Pump(grp, rings, blk, fs, subs, fmt=nat.FMT_U8, scale=1.0 / 32, offset=127.4, what="fm", gain=gain,
                    phase_s=[0.0, 0.009], out_ring_samples=1 << 14, n_blocks=n_blocks, max_read=6, start_delay_s=0.05)
    with pytest.raises(nat.RcfError):
        pump.subscribe(0, nat.SRC_PFB_BIN0 + 5, "iq")             # the fused ring holds discriminator samples only
    with pytest.raises(nat.RcfError):
        pump.subscribe_bin(1, nb)                                  # no such bin
    late = None
    t0 = time.perf_counter()
    while pump.running():
        if late is None and time.perf_counter() - t0 > 0.05 + 4.3 * blk / fs:
            late = pump.subscribe_bin(1, nb - 31, gain)
        time.sleep(0.003)
    st = pump.stats()
    assert st["error"] == 0 and st["blocks_done"] == 2 * n_blocks, st
    if late is None:                                              # (this thread was starved of CPU for the pump's whole run)
        late = pump.subscribe_bin(1, nb - 31, gain)
    got = [pump.read(e) for e in range(3)] + [pump.read(late)]
    pump.stop()
    grp.close()
    for fe in fes:
        assert fe.pfb_fm_lost() == 0
        fe.close()
    fes = open_all()
    want = {k: [] for k in range(4)}
    for b in range(n_blocks):
        for m in range(2):
            fes[m].push_raw(u8[m][2 * b * blk: 2 * (b + 1) * blk], nat.FMT_U8, 1.0 / 32, 127.4)
        want[0].append(fes[0].pfb_read_fm(21, gain))
        want[1].append(fes[0].pfb_read_fm(nb - 30, gain))
        want[2].append(fes[1].pfb_read_fm(30, gain))
        want[3].append(fes[1].pfb_read_fm(nb - 31, gain))
    for fe in fes:
        fe.close()
    for r in rings:
        r.free()
    for k in range(3):
        w = np.concatenate(want[k])
        assert len(w) == len(got[k]) > 5000 and _same_bits(got[k], w), (k, len(w), len(got[k]))
    w = np.concatenate(want[3])
    n = len(got[3])
    assert n < len(w) and n % (blk // D) == 0 and _same_bits(got[3], w[len(w) - n:]), (n, len(w))   # from the block it was subscribed in on
