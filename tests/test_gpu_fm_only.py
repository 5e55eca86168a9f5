"""rcf_chan_set_fm_only: a tap of the reference-grid filterbank that is only ever demodulated (rc_frontend/channel.py:35 +
p25_control_demod.py:120-121 -- every channel of the reference is demodulated, in a process of its own) has tap_finalize
write its discriminator ring alone, and without rotating anything: arg(y[n] conj(y[n-1])) of the rotated stream is
arg(bin[n] conj(bin[n-1]) x incr).  The discriminator is that of an ordinary tap to float32 rounding (a few 1e-7 rad), the
SAME BITS however the stream is cut, continuous across a switch of the flag in mid-stream; the IQ stream is refused while
the flag is on and comes back, from the next block on, when it goes off."""
import numpy as np
import pytest

from oracle import grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _taps_fm(nat, x, cuts, bins, fm_only, fs=5e6, cr=12500):
    D, taps = G.channel_params(fs, cr)
    nb = 2 * D
    with nat.Frontend(fs, 0.0, device=0, block_capacity=max(cuts), hist_capacity=1 << 14, out_capacity=1 << 12) as fe:
        fe.pfb_open(nb, D, taps)
        ids = [fe.pfb_tap_open(b, gr_phase=True) for b in bins]
        for c in ids:
            if fm_only:
                fe.chan_set_fm_only(c, True)
        fm = [[] for _ in ids]
        at = 0
        for n in cuts:
            fe.push(x[at:at + n])
            at += n
            for i, c in enumerate(ids):
                fm[i].append(fe.chan_read_fm(c, 3.0))
        return [np.concatenate(f) for f in fm]


def test_fm_only_taps_same_discriminator_bits_whatever_the_cuts(gpu_required):
    nat = gpu_required
    rng = np.random.default_rng(77)
    fs = 5e6
    D, _ = G.channel_params(fs, 12500)
    x = synth.awgn(rng, D * 700 + 13)
    t = np.arange(len(x)) / fs
    x = (x + 8 * np.exp(2j * np.pi * (12500.0 * 21 * t + 0.3 * np.sin(2 * np.pi * 900 * t)))).astype(np.complex64)
    # a complete aligned run of 16 bins (read from the bank's ring), scattered bins (through the tap matrix), a duplicate
    bins = list(range(16, 32)) + [21, 3, 77, 140, 21, 399]
    cuts_a = [len(x)]
    cuts_b = [D * 100 + 5, D * 3, 1, D * 250 - 6, len(x) - (D * 353)]
    assert sum(cuts_b) == len(x)
    ref = _taps_fm(nat, x, cuts_a, bins, False)
    one = _taps_fm(nat, x, cuts_a, bins, True)
    cut = _taps_fm(nat, x, cuts_b, bins, True)
    for r, a_, b_ in zip(ref, one, cut):
        assert len(r) > 600 and _same_bits(a_, b_)               # cut invariance, bit for bit
        d = np.angle(np.exp(1j * (a_.astype(np.float64) - r) / 3.0))   # (gain 3; +-pi wraps of a noise bin are the same angle)
        assert np.max(np.abs(d)) < 2e-6, float(np.max(np.abs(d)))


def test_fm_only_switched_in_mid_stream_keeps_the_discriminator_continuous(gpu_required):
    """the ring sample that is the next block's "output before" is kept rotated by an ordinary tap and bare by a
    discriminator-only one: rcf_chan_set_fm_only converts it, so the sample straddling a switch is right too"""
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    rng = np.random.default_rng(79)
    x = synth.awgn(rng, D * 400)
    t = np.arange(len(x)) / fs
    x = (x + 8 * np.exp(2j * np.pi * (12500.0 * 40 * t + 0.4 * np.sin(2 * np.pi * 700 * t)))).astype(np.complex64)
    cuts = [D * 100, D * 100, D * 100, D * 100]
    out = {}
    for mode in ("plain", "switched"):
        with nat.Frontend(fs, 0.0, device=0, block_capacity=D * 100, hist_capacity=1 << 14, out_capacity=1 << 12) as fe:
            fe.pfb_open(2 * D, D, taps)
            c = fe.pfb_tap_open(40, gr_phase=True)
            fm, at = [], 0
            for i, n in enumerate(cuts):
                if mode == "switched" and i in (1, 3):
                    fe.chan_set_fm_only(c, True)
                if mode == "switched" and i == 2:
                    fe.chan_set_fm_only(c, False)
                fe.push(x[at:at + n])
                at += n
                fm.append(fe.chan_read_fm(c, 1.0))
            out[mode] = np.concatenate(fm)
    assert len(out["plain"]) == 400 == len(out["switched"])
    d = np.angle(np.exp(1j * (out["switched"].astype(np.float64) - out["plain"])))
    assert np.max(np.abs(d)) < 2e-6, (int(np.argmax(np.abs(d))), float(np.max(np.abs(d))))


def test_fm_only_refuses_the_iq_stream_and_gives_it_back(gpu_required):
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    rng = np.random.default_rng(78)
    x = synth.awgn(rng, D * 300)
    with nat.Frontend(fs, 0.0, device=0, block_capacity=len(x), hist_capacity=1 << 14, out_capacity=1 << 12) as fe, \
            nat.Frontend(fs, 0.0, device=0, block_capacity=len(x), hist_capacity=1 << 14, out_capacity=1 << 12) as fr:
        for f in (fe, fr):
            f.pfb_open(2 * D, D, taps)
        a, b = fe.pfb_tap_open(40, gr_phase=True), fe.pfb_tap_open(41, gr_phase=True)
        ra = fr.pfb_tap_open(40, gr_phase=True)
        direct = fe.chan_open(12500, 100e3)
        with pytest.raises(nat.RcfError):
            fe.chan_set_fm_only(direct, True)                     # not a filterbank tap
        fe.chan_open_taps(b, 1, np.ones(3, np.float32), 0.0)
        with pytest.raises(nat.RcfError):
            fe.chan_set_fm_only(b, True)                          # something reads b's IQ stream
        fe.chan_set_fm_only(a, True)
        fe.push(x[: D * 100])
        fr.push(x[: D * 100])
        with pytest.raises(nat.RcfError):
            fe.chan_read_iq(a)
        many = fe.chan_read_many([a, b], what="iq")
        assert many[0] is None and len(many[1]) == 100            # the batched read: refused for a, b's samples delivered
        with pytest.raises(nat.RcfError):
            fe.chan_open_taps(a, 1, np.ones(3, np.float32), 0.0)  # nothing chains on a discriminator-only channel
        fe.chan_set_fm_only(a, False)                             # IQ from the next block on
        assert len(fe.chan_read_iq(a)) == 0
        fe.push(x[D * 100:])
        fr.push(x[D * 100:])
        iq = fe.chan_read_iq(a)
        want = fr.chan_read_iq(ra)
        assert len(iq) == 200 and _same_bits(iq, want[-200:])
        fa, fb = fe.chan_read_fm(a, 1.0), fr.chan_read_fm(ra, 1.0)
        assert len(fa) == len(fb) == 300 and np.max(np.abs(np.angle(np.exp(1j * (fa.astype(np.float64) - fb))))) < 2e-6


def test_fm_only_taps_in_a_group_same_bits_as_alone(gpu_required):
    """discriminator-only and ordinary taps side by side on the members of a group (rcf_group_*: one tap_finalize launch over
    the members): the same bits as the members fed one by one"""
    nat = gpu_required
    fs = 5e6
    D, _ = G.channel_params(fs, 12500)
    t = G.low_pass_2(1.0, fs, 6250.0, 6250.0, 20.0, G.WIN_HAMMING)
    tap_sets = [list(range(32, 48)) + [3, 77, 399], list(range(0, 16)) + [200, 201]]
    blk = D * 200
    rng = np.random.default_rng(80)
    xs = [synth.awgn(rng, 3 * blk) for _ in tap_sets]
    res = []
    for grouped in (True, False):
        fes, ids = [], []
        for m, ts in enumerate(tap_sets):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=blk, hist_capacity=1 << 13, out_capacity=1 << 11)
            fe.pfb_open(400, D, t)
            ids.append([fe.pfb_tap_open(b, gr_phase=True) for b in ts])
            for i, c in enumerate(ids[-1]):
                if i % 2 == 0 or m == 1:
                    fe.chan_set_fm_only(c, True)
            fes.append(fe)
        grp = nat.Group(fes) if grouped else None
        for r in range(3):
            blocks = [xs[m][r * blk:(r + 1) * blk] for m in range(2)]
            if grp is not None:
                grp.push(blocks, nat.FMT_CF32)
            else:
                for m in range(2):
                    fes[m].push(blocks[m])
        out = [fes[m].chan_read_fm(c, 6.6315) for m in range(2) for c in ids[m]]
        if grp is not None:
            grp.close()
        for fe in fes:
            fe.close()
        res.append(out)
    for a_, b_ in zip(*res):
        assert len(a_) == 600 and _same_bits(a_, b_)
