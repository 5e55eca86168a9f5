#!/usr/bin/env python3
"""Generate tests/golden/demod_params.json: the arguments the reference's demodulator front halves hand to GNU Radio.

/root/reference/p25_control_demod.py (C4FM control channel: pre-filter, discriminator gain, drift probe, symbol filter;
:105-137) and /root/reference/logging_receiver.py (protocol 'analog': squelch, fm_demod_cf, 300 Hz high-pass, 8 kHz
resampler; :211-222, and 'p25': pre-filter + discriminator) are constructed here with EVERYTHING outside the standard
library replaced by MagicMock stand-ins (build container only: needs /root/reference); the block constructors' call
arguments are read off the stand-ins.  Written: numbers only.
"""
import json
import os
import sys
import types
from unittest import mock

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demod_params.json")


class _TopBlock:
    def __init__(self, *a, **k): pass
    def __getattr__(self, name):                    # connect / disconnect / start / stop / lock / unlock / ...
        return lambda *a, **k: None


gr = mock.MagicMock()
gr.top_block = _TopBlock
gr.sizeof_gr_complex, gr.sizeof_char, gr.sizeof_float, gr.sizeof_short = 8, 1, 4, 2
gnuradio = types.ModuleType("gnuradio")
gnuradio.gr = gr
subs = ("filter", "blocks", "zeromq", "uhd", "analog", "digital", "fft", "audio", "vocoder")
for name in subs:
    setattr(gnuradio, name, mock.MagicMock())
sys.modules["gnuradio"] = gnuradio
sys.modules["gnuradio.gr"] = gr
for name in subs:
    sys.modules["gnuradio." + name] = getattr(gnuradio, name)
sys.modules["gnuradio.filter.firdes"] = gnuradio.filter.firdes
sys.modules["gnuradio.filter.pfb"] = gnuradio.filter.pfb
sys.modules["gnuradio.filter.optfir"] = gnuradio.filter.optfir


class _Finder:                                       # every other import that is not the standard library: a MagicMock
    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        top = name.split(".")[0]
        if top in sys.builtin_module_names or top in sys.stdlib_module_names or top in ("numpy", "scipy"):
            return None
        if os.path.exists(os.path.join(REF, top + ".py")) and top in ("p25_control_demod", "logging_receiver", "fft_vector"):
            return None
        return importlib.machinery.ModuleSpec(name, self)

    def create_module(self, spec):
        return mock.MagicMock()

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Finder())
# logging_receiver's constructor makes its audio/<y>/<m>/<d>/... directory tree relative to the working directory
# (logging_receiver.py: os.makedirs): run in a scratch directory so that nothing is left in the repository
import tempfile
os.chdir(tempfile.mkdtemp(prefix="rcf_goldens_"))
sys.path.insert(0, REF)


def args_of(m):
    """positional + keyword arguments of the last call of a stand-in, JSON-able"""
    if not m.call_args:
        return None
    a, k = m.call_args

    def conv(v):
        if isinstance(v, (int, float, str, bool)) or v is None:
            return v
        if isinstance(v, (tuple, list)):
            return [conv(x) for x in v]
        return "<%s>" % type(v).__name__
    return {"args": [conv(v) for v in a], "kwargs": {kk: conv(vv) for kk, vv in k.items()}}


golden = {}

import p25_control_demod as P25       # noqa: E402
P25.threading = mock.MagicMock()                     # no quality-check thread
P25.p25_control_demod.tune_next_control_channel = lambda self: None
for m in (gnuradio.filter.firdes.low_pass_2, gnuradio.filter.freq_xlating_fir_filter_ccc, gnuradio.analog.quadrature_demod_cf,
          gnuradio.blocks.moving_average_ff, gnuradio.blocks.multiply_const_vff, gnuradio.filter.fir_filter_fff):
    m.reset_mock()
P25.p25_control_demod({"type": "p25", "id": "p25", "modulation": "C4FM", "default_control_channel": 0,
                       "channels": {0: 855000000}}, "site", "overseer", rcm=mock.MagicMock())
golden["p25_control_demod_c4fm"] = {
    "low_pass_2": args_of(gnuradio.filter.firdes.low_pass_2),
    "low_pass_2_window_is_blackman": gnuradio.filter.firdes.low_pass_2.call_args[0][5] is gnuradio.filter.firdes.WIN_BLACKMAN,
    "freq_xlating_fir_filter_ccc": args_of(gnuradio.filter.freq_xlating_fir_filter_ccc),
    "quadrature_demod_cf": args_of(gnuradio.analog.quadrature_demod_cf),
    "moving_average_ff": args_of(gnuradio.blocks.moving_average_ff),
    "multiply_const_vff": args_of(gnuradio.blocks.multiply_const_vff),
    "fir_filter_fff": args_of(gnuradio.filter.fir_filter_fff),
}

import logging_receiver as LR          # noqa: E402  (it takes firdes from gnuradio.gr when that import works: LR.firdes)
for proto in ("analog", "p25"):
    for m in (gnuradio.analog.pwr_squelch_cc, gnuradio.analog.fm_demod_cf, LR.firdes.high_pass,
              gnuradio.filter.rational_resampler_fff, LR.firdes.low_pass_2, gnuradio.analog.quadrature_demod_cf,
              gnuradio.filter.freq_xlating_fir_filter_ccc, gnuradio.filter.fir_filter_fff):
        m.reset_mock()
    connector = mock.MagicMock()
    connector.create_channel.return_value = ("chan", 12345)
    LR.frontend_connector = lambda *a, **k: connector
    class CDR(dict):                                 # the call record: whatever else the constructor looks up is "0"
        def __missing__(self, key):
            return "0"

    cdr = CDR({"instance_uuid": "u", "channel_bandwidth": 12500, "frequency": 855000000, "modulation_type": proto,
               "p25_nac": 1, "p25_system_id": "0x1", "p25_wacn": "0x1", "system_id": 1, "system_type": "x", "type": "group"})
    try:
        LR.logging_receiver(cdr, mock.MagicMock(), mock.MagicMock(), mock.MagicMock())
    except Exception as e:
        print("logging_receiver(%s) raised after its blocks were made: %s: %s" % (proto, type(e).__name__, e))
    if proto == "analog":
        golden["logging_receiver_analog"] = {
            "pwr_squelch_cc": args_of(gnuradio.analog.pwr_squelch_cc),
            "fm_demod_cf": args_of(gnuradio.analog.fm_demod_cf),
            "high_pass": args_of(LR.firdes.high_pass),
            "high_pass_window_is_hamming": LR.firdes.high_pass.call_args[0][4] is LR.firdes.WIN_HAMMING,
            "rational_resampler_fff": args_of(gnuradio.filter.rational_resampler_fff),
        }
    else:
        golden["logging_receiver_p25"] = {
            "low_pass_2": args_of(LR.firdes.low_pass_2),
            "freq_xlating_fir_filter_ccc": args_of(gnuradio.filter.freq_xlating_fir_filter_ccc),
            "quadrature_demod_cf": args_of(gnuradio.analog.quadrature_demod_cf),
        }

# ---------------------------------------------------------------- the scan flowgraph (fft_vector.py:31-60)
sys.modules["gnuradio.fft.window"] = gnuradio.fft.window
import fft_vector as FV                # noqa: E402
edges = []
_TopBlock.connect = lambda self, *ends: edges.append(ends)
tbv = FV.fft_vector(0)
names = {id(getattr(tbv, n)): n for n in dir(tbv) if n.startswith(("blocks_", "fft_", "zeromq_"))}
golden["fft_vector"] = {
    "samp_rate": tbv.samp_rate, "length": tbv.length,
    "fft_vcc": args_of(gnuradio.fft.fft_vcc), "window_blackmanharris": args_of(gnuradio.fft.window.blackmanharris),
    "window_passed_is_that_one": gnuradio.fft.fft_vcc.call_args[0][2] is gnuradio.fft.window.blackmanharris.return_value,
    "stream_to_vector": args_of(gnuradio.blocks.stream_to_vector), "complex_to_mag_squared": args_of(gnuradio.blocks.complex_to_mag_squared),
    "nlog10_ff": args_of(gnuradio.blocks.nlog10_ff), "moving_average_ff": args_of(gnuradio.blocks.moving_average_ff),
    "head": args_of(gnuradio.blocks.head), "skiphead": args_of(gnuradio.blocks.skiphead),
    "edges": sorted([names[id(a[0])], names[id(b[0])]] for a, b in edges),
}

with open(OUT, "w") as f:
    json.dump(golden, f, indent=1, sort_keys=True)
print(json.dumps(golden, indent=1, sort_keys=True))
os._exit(0)
