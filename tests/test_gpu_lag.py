"""The stage-2 lag (rcf_set_stage2_lag): on a power-of-two bank with short stage-2 channels on its bins (BASELINE configs[1]:
rcf_pfb_chan_open == channel.py's rule at the bin rate + quadrature_demod_cf), the stage-2 launch of block n rides in block
n + 1's filterbank launch instead of trailing block n's.  Whatever the schedule -- riding, flushed by a read, refused for
want of ring room, interrupted by a retune -- the outputs are the SAME BITS as with the lag switched off."""
import numpy as np
import pytest

from oracle import grspec as G
from rcf import synth

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _run(nat, lag, x, cuts, script, out_cap, nb=256, os_=1):
    fs = 20e6
    bw = fs / nb
    proto = G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS) if os_ == 1 else \
        G.low_pass_2(1.0, fs, bw / 4, bw / 4, 20.0, G.WIN_HAMMING)
    out = []
    with nat.Frontend(fs, 0.0, device=0, block_capacity=max(cuts), hist_capacity=1 << 14, out_capacity=out_cap) as fe:
        fe.set_stage2_lag(lag)
        fe.pfb_open(nb, nb // os_, proto)
        ids = [fe.pfb_chan_open((5 + 17 * i) % nb, 12500, 12500.0 * ((i % 5) - 2)) for i in range(7)]
        at = 0
        for step, n in enumerate(cuts):
            for what, arg in script.get(step, []):
                if what == "retune":
                    fe.chan_set_offset(ids[arg], 6250.0)
                elif what == "open":
                    ids.append(fe.pfb_chan_open(arg, 12500, 0.0))
                elif what == "close":
                    fe.chan_close(ids.pop(arg))
                elif what == "read":
                    out.append(fe.chan_read_fm(ids[arg], 5.0))
                elif what == "shift":
                    fe.source_shift(arg)
                elif what == "sync":
                    fe.sync()
            fe.push(x[at:at + n])
            at += n
        for c in ids:
            out.append(fe.chan_read_iq(c))
            out.append(fe.chan_read_fm(c, 5.0))
        out.append(fe.pfb_read_bin(5))
    return out


@pytest.mark.parametrize("case", ["steady", "churn", "tight_ring", "oversampled"])
def test_lagging_stage2_is_bit_identical_to_the_trailing_launch(gpu_required, case):
    nat = gpu_required
    blk = 256 * 16 * 40
    rng = np.random.default_rng(11)
    x, _ = synth.cfg2(n=8 * blk, seed=2200)
    if case == "steady":                         # equal blocks, nothing in between: every stage-2 launch but the last rides
        cuts, script, cap = [blk] * 7, {}, 1 << 12
    elif case == "churn":                        # ragged blocks with reads, a retune, a source shift, a channel opened and one
        cuts = [blk, blk // 2 + 48, blk, 300, blk, blk - 256, blk]          # closed in mid-stream: rides and flushes mixed
        script = {1: [("read", 0)], 2: [("retune", 3)], 3: [("open", 77), ("sync", None)], 4: [("shift", 40.0), ("read", 2)],
                  5: [("close", 1)], 6: [("read", 0)]}
        cap = 1 << 12
    elif case == "tight_ring":                   # a ring that holds ONE block's frames and its reach: no room for two, no lag
        cuts, script, cap = [blk] * 5, {2: [("read", 1)]}, 1 << 10
    else:                                        # the oversampled 256-bin bank (decim 128): the same kernel family
        cuts, script, cap = [blk // 2] * 6, {3: [("retune", 2)]}, 1 << 12
    os_ = 2 if case == "oversampled" else 1
    on = _run(nat, True, x, cuts, script, cap, os_=os_)
    off = _run(nat, False, x, cuts, script, cap, os_=os_)
    assert len(on) == len(off) and len(on) > 10
    for a, b in zip(on, off):
        assert len(a) > 0 and _same_bits(a, b)
