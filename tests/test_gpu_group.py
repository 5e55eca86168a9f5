"""Grouped launches over front-ends (rcf_group_*, rcf_pump_*; the reference's receiver with ALL its sources in one top
block, rc_frontend/receiver.py:67-70,170-204): the outputs of G front-ends processed by one launch per stage are the SAME
BITS as those of the same front-ends run one by one, and inside the bars against the oracle."""
import time

import numpy as np
import pytest

from oracle import cbind as OC, grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _u8(x, scale=32.0, offset=127.4):
    v = np.clip(np.round(x.view(np.float32) * scale + offset), 0, 255).astype(np.uint8)
    return v


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _proto(fs, nb, P=14):
    bw = fs / nb
    t = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    assert len(t) <= nb * 16
    return t


def _open_pfb256(nat, fs, seed, n_ch, block_cap, out_cap=1 << 11):
    """one front-end: 256-bin bank + n_ch stage-2 channels (the BASELINE configs[1] shape)"""
    rng = np.random.default_rng(seed)
    fe = nat.Frontend(fs, 0.0, device=0, block_capacity=block_cap, hist_capacity=1 << 14, out_capacity=out_cap)
    fe.pfb_open(256, 256, _proto(fs, 256))
    bins = rng.permutation(np.arange(-100, 101))[:n_ch]
    deltas = rng.choice(np.array([-25000.0, -12500.0, 0.0, 12500.0, 25000.0]), size=n_ch)
    ids = [fe.pfb_chan_open(int(b) % 256, 12500, float(d)) for b, d in zip(bins, deltas)]
    return fe, ids, list(zip(bins.tolist(), deltas.tolist()))


def test_group_pfb256_stage2_same_bits_as_one_by_one_and_inside_the_fm_bar(gpu_required):
    nat = gpu_required
    fs = 20e6
    n_ch = [6, 32, 1, 9]
    blk = 200_000
    G_ = len(n_ch)
    x = []
    for m in range(G_):
        xm, meta = synth.cfg2(n=4 * blk, seed=2100 + m)
        x.append(xm)
    # ragged rounds: different block sizes per member, one member skipped in one round; the first round still sees
    # zero history (those banks are launched on their own), the later ones go out grouped
    rounds = [[blk, blk // 2 + 77, blk, 1000], [blk, blk, 0, blk], [blk - 3, blk, blk, blk], [blk, 0, 123, blk // 3]]
    grouped, single = [], []
    for which in (0, 1):
        fes = [_open_pfb256(nat, fs, 500 + m, n_ch[m], blk) for m in range(G_)]
        at = [0] * G_
        grp = nat.Group([f[0] for f in fes]) if which == 0 else None
        for r in rounds:
            blocks = []
            for m in range(G_):
                blocks.append(_u8(x[m][at[m]: at[m] + r[m]]) if r[m] else None)
                at[m] += r[m]
            if grp is not None:
                grp.push(blocks, nat.FMT_U8, 1.0 / 32, 127.4)
            else:
                for m in range(G_):
                    if blocks[m] is not None:
                        fes[m][0].push_raw(blocks[m], nat.FMT_U8, 1.0 / 32, 127.4)
        out = []
        if grp is not None:
            pairs = [(m, c) for m in range(G_) for c in fes[m][1]]
            iq = grp.read_many(pairs, "iq", cap_each=1 << 11)
            fm = grp.read_many(pairs, "fm", gain=5.0, cap_each=1 << 11)
            out = list(zip(iq, fm))
            grp.close()
        else:
            for m in range(G_):
                for c in fes[m][1]:
                    out.append((fes[m][0].chan_read_iq(c), fes[m][0].chan_read_fm(c, 5.0)))
        (grouped if which == 0 else single).extend(out)
        bins0 = fes[1][0].pfb_read_bin(7)
        (grouped if which == 0 else single).append((bins0, None))
        for f in fes:
            f[0].close()
    assert len(grouped) == len(single) == sum(n_ch) + 1
    for (gi, gf), (si, sf) in zip(grouped, single):
        assert len(gi) > 0 and _same_bits(gi, si)
        if gf is not None:
            assert _same_bits(gf, sf)
    # ... and one member against the oracle chain on the same (quantised) stream: discriminator <= 1e-4 rms
    m = 1
    xq = ((_u8(x[m][: sum(r[m] for r in rounds)]).astype(np.float32) - np.float32(127.4)) * np.float32(1.0 / 32)).view(np.complex64)
    proto = _proto(fs, 256)
    fe, ids, plan = _open_pfb256(nat, fs, 500 + m, n_ch[m], blk)
    fe.close()
    D2, t2 = G.channel_params(fs / 256, 12500)
    first = sum(n_ch[:m])
    for j in (0, 5, 31):
        b, d = plan[j]
        want_bin = G.xlating_fir_exact(xq, 256, proto, b * fs / 256, fs).astype(np.complex64)
        ct, incr = OC.xlating_composite(t2, D2, d, fs / 256)
        yo, fo = OC.channel_bank(want_bin, D2, ct[None, :], np.array([incr]), gains=[5.0])
        got = grouped[first + j][1]
        n = min(len(got), len(fo[0]))
        assert n > 500
        assert np.sqrt(np.mean((got[:n] - fo[0][:n]) ** 2)) < 1e-4


def test_group_reference_grid_bank_taps_same_bits(gpu_required):
    """the bank whose bins ARE the reference's channels (5 Msps: 400 bins, D = 200, channel.py's own filter) with tapped
    bins -- a complete aligned run of 16 (read from the ring), scattered ones (through the tap matrix), gr_phase on and
    off -- in a group of three against the same three alone"""
    nat = gpu_required
    fs = 5e6
    D, taps = G.channel_params(fs, 12500)
    t = G.low_pass_2(1.0, fs, 6250.0, 6250.0, 20.0, G.WIN_HAMMING)
    tap_sets = [list(range(32, 48)) + [3, 77, 200, 399], [5, 6, 7], list(range(0, 16)) + list(range(384, 400))]
    blk = D * 300
    rng = np.random.default_rng(42)
    xs = [synth.awgn(rng, 3 * blk + 500) for _ in tap_sets]
    res = []
    for which in (0, 1):
        fes, ids = [], []
        for m, ts in enumerate(tap_sets):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=blk + 500, hist_capacity=1 << 13, out_capacity=1 << 11)
            fe.pfb_open(400, D, t)
            ids.append([fe.pfb_tap_open(b, gr_phase=(m != 1)) for b in ts])
            fes.append(fe)
        grp = nat.Group(fes) if which == 0 else None
        cuts = [[blk, blk + 11, blk - 200], [blk + 500, blk, blk], [77, blk, blk]]
        at = [0, 0, 0]
        for r in range(3):
            blocks = []
            for m in range(3):
                n = cuts[m][r]
                blocks.append(xs[m][at[m]: at[m] + n])
                at[m] += n
            if grp is not None:
                grp.push(blocks, nat.FMT_CF32)
            else:
                for m in range(3):
                    fes[m].push(blocks[m])
        out = []
        for m in range(3):
            for c in ids[m]:
                out.append((fes[m].chan_read_iq(c), fes[m].chan_read_fm(c, 6.6315)))
            out.append((fes[m].pfb_read_bin(201), None))
        if grp is not None:
            grp.close()
        for fe in fes:
            fe.close()
        res.append(out)
    assert len(res[0]) == len(res[1])
    for (gi, gf), (si, sf) in zip(*res):
        assert len(gi) > 100 and _same_bits(gi, si)
        if gf is not None:
            assert _same_bits(gf, sf)


def test_group_with_direct_channels_scan_and_exact_rotator_members(gpu_required):
    """what a group cannot concatenate -- a matrix-core bank (>= 8 direct channels), a shared-source vector bank, an armed
    scan, channels on the exact rotator -- follows per member on the group's stream: same bits as alone"""
    nat = gpu_required
    fs = 2.4e6
    x, meta = synth.cfg1(seconds=0.25)
    n = len(x)
    res = []
    for which in (0, 1):
        fes = [nat.Frontend(fs, 0.0, device=0, block_capacity=n, hist_capacity=1 << 15, out_capacity=1 << 13) for _ in range(3)]
        fes[2].set_rotator(True)
        ids = [[fes[0].chan_open(12500, -62500.0 + 12500.0 * k) for k in range(10)],
               [fes[1].chan_open(12500, -62500.0), fes[1].chan_open(12500, 50000.0)],
               [fes[2].chan_open(12500, -62500.0), fes[2].chan_open(25000, 100000.0)]]
        fes[1].scan_start(4096, 40, 10)
        grp = nat.Group(fes) if which == 0 else None
        cuts = [0, n // 3 + 5, 2 * n // 3, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if grp is not None:
                grp.push([x[a:b]] * 3, nat.FMT_CF32)
            else:
                for fe in fes:
                    fe.push(x[a:b])
        out = []
        for m in range(3):
            for c in ids[m]:
                out.append(fes[m].chan_read_iq(c))
                out.append(fes[m].chan_read_fm(c, 5.0))
        out.append(fes[1].scan_result())
        if grp is not None:
            grp.close()
        for fe in fes:
            fe.close()
        res.append(out)
    for a, b in zip(*res):
        assert a is not None and len(a) > 0 and _same_bits(a, b)


def test_group_refusals_leave_every_member_untouched(gpu_required):
    nat = gpu_required
    fs = 20e6
    blk = 100_000
    fes = [_open_pfb256(nat, fs, 900 + m, 3, 4 * blk, 1 << 10) for m in range(3)]
    rng = np.random.default_rng(9)
    x = [synth.awgn(rng, 5 * blk) for _ in range(3)]
    with pytest.raises(nat.RcfError):
        nat.Group([fes[0][0], fes[0][0]])
    grp = nat.Group([f[0] for f in fes])
    with pytest.raises(nat.RcfError):
        nat.Group([fes[1][0]])                               # already in a group
    assert fes[0][0]._h and nat.lib().rcf_close(fes[0][0]._h) == nat.RCF_ESTATE
    grp.push([x[m][:blk] for m in range(3)], nat.FMT_CF32)
    before = [[f[0].chan_produced(c) for c in f[1]] + [f[0].pfb_produced(), f[0].samples_in] for f in fes]
    # member 2's block yields more frames than its ring holds: refused as a whole, nobody advances
    with pytest.raises(nat.RcfError) as e:
        grp.push([x[0][blk:2 * blk], x[1][blk:2 * blk], x[2][blk:5 * blk]], nat.FMT_CF32)
    assert e.value.code == nat.RCF_ECAP
    after = [[f[0].chan_produced(c) for c in f[1]] + [f[0].pfb_produced(), f[0].samples_in] for f in fes]
    assert before == after
    grp.push([x[m][blk:2 * blk] for m in range(3)], nat.FMT_CF32)
    got = grp.read_many([(m, c) for m in range(3) for c in fes[m][1]] + [(0, fes[0][1][0]), (7, 1), (1, 9999)], "iq", cap_each=1 << 11)
    assert got[-1] is None and got[-2] is None and got[-3] is None          # listed twice / no such member / no such channel
    grp.close()
    ref = [_open_pfb256(nat, fs, 900 + m, 3, 4 * blk, 1 << 10) for m in range(3)]
    k = 0
    for m in range(3):
        ref[m][0].push(x[m][:blk])
        ref[m][0].push(x[m][blk:2 * blk])
        for c in ref[m][1]:
            assert _same_bits(got[k], ref[m][0].chan_read_iq(c))
            k += 1
    for f in fes + ref:
        f[0].close()


def test_pump_feeds_a_group_in_real_time_and_delivers_the_same_bits(gpu_required):
    """the native pump: three 3.2 Msps u8 sources replayed from pinned rings at wall-clock rate in 20 ms blocks, the
    subscribed channels' discriminator output in per-channel host rings -- equal, bit for bit, to the same blocks pushed
    one front-end at a time"""
    nat = gpu_required
    fs, blk, n_blocks = 3.2e6, 64000, 12
    srcs = []
    for m in range(3):
        rng = np.random.default_rng(1001 + m)
        xm = synth.awgn(rng, n_blocks * blk).astype(np.complex128)
        xm += synth.nbfm_carrier(len(xm), fs, -2 * fs / 64, 700.0 + 100 * m, 2500.0, synth.snr_amp(30.0, 12500.0, fs))
        srcs.append(_u8(xm.astype(np.complex64)))

    def open_all():
        fes, ids = [], []
        for m in range(3):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=blk, hist_capacity=1 << 13, out_capacity=1 << 12)
            fe.pfb_open(64, 64, G.low_pass_2(1.0, fs, fs / 64 * 0.4, fs / 64 * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS))
            ids.append([fe.pfb_chan_open((b - 2) % 64, 12500, 0.0) for b in range(m + 2)])
            fes.append(fe)
        return fes, ids

    fes, ids = open_all()
    rings = []
    for m in range(3):
        r = nat.PinnedArray(len(srcs[m]), np.uint8)
        r.array[:] = srcs[m]
        rings.append(r)
    subs = [(m, c) for m in range(3) for c in ids[m]]
    grp = nat.Group(fes)
    pump = nat.Pump(grp, rings, blk, fs, subs, fmt=nat.FMT_U8, scale=1.0 / 32, offset=127.4, what="fm", gain=5.0,
                    phase_s=[0.0, 0.007, 0.013], out_ring_samples=1 << 14, n_blocks=n_blocks, warm_blocks=2)
    with pytest.raises(nat.RcfError):
        grp.push([None] * 3)                                 # the group is fed by its pump
    st = pump.wait(timeout_s=30.0)
    assert not st["running"] and st["error"] == 0, st
    assert st["blocks_done"] == 3 * n_blocks and st["blocks_judged"] == 3 * (n_blocks - 2)
    assert 0.2 < st["elapsed_s"] < 2.0                        # paced: twelve 20 ms blocks, not as fast as they go
    got = [pump.read(e) for e in range(len(subs))]
    assert sum(len(g_) for g_ in got) == st["samples_out"]
    pump.stop()
    grp.close()
    for fe in fes:
        fe.close()
    fes, ids = open_all()
    k = 0
    for m in range(3):
        want = [[] for _ in ids[m]]
        for b in range(n_blocks):
            fes[m].push_raw(srcs[m][2 * b * blk: 2 * (b + 1) * blk], nat.FMT_U8, 1.0 / 32, 127.4)
            for j, c in enumerate(ids[m]):
                want[j].append(fes[m].chan_read_fm(c, 5.0))
        for j in range(len(ids[m])):
            w = np.concatenate(want[j])
            assert len(w) > 5000 and _same_bits(got[k], w)
            k += 1
        fes[m].close()
    for r in rings:
        r.free()


def test_receiver_feed_all_is_the_sources_fed_one_by_one(gpu_required):
    """rcf.receiver with three sources (no -i: the reference's whole-host receiver, rc_frontend/receiver.py:67-70): the
    channels its clients hold deliver the same bits whether the sources' blocks arrive together (feed_all: one group
    block) or source by source (feed)"""
    import types
    from rcf import receiver
    fs = 2.4e6
    xs = [synth.cfg1(seconds=0.1, seed=1001 + m)[0] for m in range(3)]
    res = []
    for which in (0, 1):
        cfg = types.SimpleNamespace(
            sources={m: dict(type="synthetic", center_freq=855050000 + 3000000 * m, samp_rate=int(fs)) for m in range(3)},
            frontend_mode="xlat")
        tb = receiver.receiver(cfg)
        try:
            ids = [tb.connect_channel(12500, 854987500 + 3000000 * m)[0] for m in range(3)]
            ids.append(tb.connect_channel(12500, 855100000)[0])
            assert [tb.channels[c].source_id for c in ids] == [0, 1, 2, 0]
            n = len(xs[0])
            for a, b in ((0, n // 2 + 17), (n // 2 + 17, n)):
                if which == 0:
                    tb.feed_all({m: xs[m][a:b] for m in range(3)})
                else:
                    for m in range(3):
                        tb.feed(m, xs[m][a:b])
            assert (getattr(tb, "_group", None) is not None) == (which == 0)
            res.append([(tb.channels[c].read_iq(), tb.channels[c].read_fm(5.0)) for c in ids])
            assert tb.metrics()["rcf_samples_in"] == 3 * n
        finally:
            tb.close()
    for (gi, gf), (si, sf) in zip(*res):
        assert len(gi) > 2000 and _same_bits(gi, si) and _same_bits(gf, sf)


def test_pump_subscriptions_come_and_go_and_a_counter_fed_member(gpu_required):
    """channels are subscribed and unsubscribed under the RUNNING pump (rcf_pump_subscribe / rcf_pump_unsubscribe: the
    reference creates and destroys channel flowgraphs under its running top block, rc_frontend/receiver.py:296-342,
    :635-648): an IQ slot and a discriminator slot of the same channel side by side, a slot reused by another channel
    (its new tenant's stream starts at the returned cursor), a slot whose channel is closed under the pump (it starves,
    nothing else does); member 1 is fed by a producer that counts its blocks complete (rcf_pump_config_t.written)
    instead of the pump's clock.  Everything delivered equals the same blocks pushed one front-end at a time."""
    nat = gpu_required
    fs, blk, n_blocks = 3.2e6, 64000, 14
    srcs = []
    for m in range(2):
        rng = np.random.default_rng(2001 + m)
        xm = synth.awgn(rng, n_blocks * blk).astype(np.complex128)
        xm += synth.nbfm_carrier(len(xm), fs, -2 * fs / 64, 650.0 + 100 * m, 2500.0, synth.snr_amp(30.0, 12500.0, fs))
        srcs.append(_u8(xm.astype(np.complex64)))

    def open_all():
        fes, ids = [], []
        for m in range(2):
            fe = nat.Frontend(fs, 0.0, device=0, block_capacity=blk, hist_capacity=1 << 13, out_capacity=1 << 13)
            fe.pfb_open(64, 64, G.low_pass_2(1.0, fs, fs / 64 * 0.4, fs / 64 * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS))
            ids.append([fe.pfb_chan_open((b - 2) % 64, 12500, 0.0) for b in range(3)])
            fes.append(fe)
        return fes, ids

    fes, ids = open_all()
    ring0 = nat.PinnedArray(len(srcs[0]), np.uint8)
    ring0.array[:] = srcs[0]                                  # member 0: the whole stream, replayed by the pump's clock
    ring1 = nat.PinnedArray(4 * blk * 2, np.uint8)            # member 1: a ring of four blocks, filled by a producer
    ring1.array[:] = 0
    counter = np.zeros(1, dtype=np.uint64)
    grp = nat.Group(fes)
    pump = nat.Pump(grp, [ring0, ring1], blk, fs, (), fmt=nat.FMT_U8, scale=1.0 / 32, offset=127.4, what="iq",
                    out_ring_samples=1 << 14, n_blocks=n_blocks, max_read=4, written=[None, counter], start_delay_s=0.05)
    period = blk / fs
    got = {}

    def drain(name, slot):
        got.setdefault(name, []).append(pump.read(slot))

    # before the first block: member 0's channel 0 as IQ AND as discriminator x 5, member 1's channel 1 as IQ
    s_iq = pump.subscribe(0, ids[0][0], "iq")
    s_fm = pump.subscribe(0, ids[0][0], "fm", 5.0)
    s_b = pump.subscribe(1, ids[1][1], "iq")
    assert len({s_iq, s_fm, s_b}) == 3
    s_c = pump.subscribe(0, ids[0][2], "iq")
    with pytest.raises(nat.RcfError):
        pump.subscribe(1, ids[1][0], "iq")                    # max_read = 4 slots, all taken
    with pytest.raises(nat.RcfError):
        pump.unsubscribe(17)
    t0 = time.perf_counter() + 0.05
    fed = 0
    swapped = closed = False
    while pump.running():
        now = time.perf_counter() - t0
        while fed < n_blocks and now >= (fed + 1) * period:   # the producer of member 1: block `fed` is complete
            at = (fed % 4) * blk * 2
            ring1.array[at:at + 2 * blk] = srcs[1][2 * fed * blk: 2 * (fed + 1) * blk]
            fed += 1
            counter[0] = fed
        if not swapped and now > 5.2 * period:
            # the slot of member 0's channel 2 goes to member 1's channel 2: what it delivered so far is the old tenant's
            drain("c", s_c)
            pump.unsubscribe(s_c)
            s_d = pump.subscribe(1, ids[1][2], "iq")
            assert s_d == s_c                                  # the freed slot is the one handed out
            swapped = True
        if not closed and now > 9.2 * period:
            fes[1].chan_close(ids[1][1])                       # closed under the pump: its slot starves, nobody else
            closed = True
        drain("iq", s_iq), drain("fm", s_fm), drain("b", s_b)
        if swapped:
            drain("d", s_d)
        time.sleep(0.002)
    st = pump.stats()
    assert st["error"] == 0 and st["blocks_done"] == 2 * n_blocks, st
    if not (swapped and closed):
        # (this thread did not get to run for a third of a second while the pump did: a host so busy decides nothing here)
        pump.stop(); grp.close()
        for fe in fes:
            fe.close()
        ring0.free(); ring1.free()
        pytest.skip("the test's own thread was starved of CPU: the pump finished before the subscriptions were changed")
    drain("iq", s_iq), drain("fm", s_fm), drain("b", s_b), drain("d", s_d)
    pump.stop()
    grp.close()
    produced_b = None
    for fe in fes:
        fe.close()
    g = {k: np.concatenate(v) for k, v in got.items()}
    # the same blocks one front-end at a time
    fes, ids2 = open_all()
    want = {}
    for b in range(n_blocks):
        for m in range(2):
            fes[m].push_raw(srcs[m][2 * b * blk: 2 * (b + 1) * blk], nat.FMT_U8, 1.0 / 32, 127.4)
        want.setdefault("iq", []).append(fes[0].chan_read_iq(ids2[0][0]))
        want.setdefault("fm", []).append(fes[0].chan_read_fm(ids2[0][0], 5.0))
        want.setdefault("b", []).append(fes[1].chan_read_iq(ids2[1][1]))
        want.setdefault("c", []).append(fes[0].chan_read_iq(ids2[0][2]))
        want.setdefault("d", []).append(fes[1].chan_read_iq(ids2[1][2]))
    w = {k: np.concatenate(v) for k, v in want.items()}
    for fe in fes:
        fe.close()
    ring0.free()
    ring1.free()
    assert len(g["iq"]) == len(w["iq"]) > 3000 and _same_bits(g["iq"], w["iq"])
    assert len(g["fm"]) == len(w["fm"]) and _same_bits(g["fm"], w["fm"])
    # the channel that was closed under the pump delivered a whole number of blocks' worth of its stream, then nothing
    nb_ = len(g["b"])
    assert 0 < nb_ < len(w["b"]) and _same_bits(g["b"], w["b"][:nb_])
    # the slot's first tenant up to the hand-over, its second from the hand-over on: the channel existed all along, so its
    # reader stood at its first output and the pump delivers it from there as far as the device ring reaches (all of it here)
    nc = len(g["c"])
    assert 0 < nc < len(w["c"]) and _same_bits(g["c"], w["c"][:nc])
    assert len(g["d"]) == len(w["d"]) and _same_bits(g["d"], w["d"])
