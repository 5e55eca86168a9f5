"""Drop-in for /root/reference/rc_frontend/channel.py: one narrowband channel of a front-end.

The reference builds a GNU Radio top_block per channel (sub_source -> freq_xlating_fir_filter_ccc ->
pub_sink, channel.py:29-38).  Here a channel is a slot in the batched HIP FIR bank of a
`native.Frontend`; this class keeps the reference's attributes and methods so that `receiver` (and
anything poking at `receiver.channels`) reads the same.
"""
from __future__ import annotations

import time


class channel:
    def __init__(self, frontend, port, channel_rate, samp_rate, offset, parent_chan=None, pfb=None, decim_rule=0):
        """frontend: rcf.native.Frontend (the HBM-resident source that replaces `parent_zmq_address`).
        parent_chan: channel id of a receiver_split2 half-band source (receiver.py:205-237) this channel
        reads instead of the wideband stream; samp_rate is then that half's rate.
        pfb: the source's filterbank plan (receiver._open_pfb) when frontend_mode == 'pfb': requests that fall on
        its grid are served by a bin of the bank, everything else by the direct kernel."""
        self.frontend = frontend
        self.samp_rate = samp_rate
        self.channel_rate = channel_rate
        self.port = port
        self.offset = offset
        self.in_use = False
        self.source_id = None
        self.block_id = None
        self.parent_chan = parent_chan
        self.pfb = pfb
        self.decim_rule = int(decim_rule)    # native.DECIM_EXACT / DECIM_FLOOR (config.py2_decim): what THIS object does with an
                                             # odd int(fs / cr) where it derives the decimation itself (split2 halves)
        self.pfb_bin = None
        self.chan_id = None
        self.route = "direct"
        self.start_sample = None
        self._build(offset)
        self.init_time = time.time()
        self.channel_close_time = 0
        self._started = False

    def _target_bin(self, offset):
        """the bank bin that serves `offset`, or None for the direct kernel (the intent of connect_channel_pfb,
        receiver.py:343-383: bin = round(offset / grid); a non-zero residual simply takes the direct path, and so
        does a bin whose exact phases are too far from GNU Radio's float32 ones for the discriminator budget --
        receiver._open_pfb)"""
        pfb = self.pfb
        self.route = "direct"
        if pfb is None or self.parent_chan is not None or self.channel_rate != pfb["channel_rate"]:
            return None
        k = int(round(offset / pfb["grid"]))
        if offset != k * pfb["grid"] or abs(k) > pfb["n_bins"] // 2 or not abs(offset) < self.samp_rate / 2:
            self.route = "direct (off the bank's grid)"
            return None
        from .receiver import receiver
        if not receiver.pfb_serves_bin(pfb, k):
            # on the grid, but the bank's exact phases are further from GNU Radio's float32 ones than the budget allows
            # at this offset: a 2909-tap direct channel instead.  Counted (receiver.metrics) -- it costs what `xlat`
            # mode costs, and a deployment whose carriers are weaker or stronger than the default environment
            # (pfb_parity_env_db) wants to know how its requests were routed
            self.route = "direct (parity budget: predicted fm error %.1e > %.1e)" % (
                receiver.pfb_predicted_fm_error(pfb, k), pfb["parity"]["budget"])
            return None
        self.route = "bank bin %d" % (k % pfb["n_bins"])
        return k % pfb["n_bins"]                               # receiver.py:373-375: wrap negative bins

    def _build(self, offset):
        """(re)create the native channel for `offset`: a filterbank bin when _target_bin() names one, else the direct
        xlating FIR.  The object's state changes only after the native open succeeded."""
        frontend, channel_rate, samp_rate = self.frontend, self.channel_rate, self.samp_rate
        pfb = self.pfb
        bin_ = self._target_bin(offset)
        # rc_frontend/channel.py:31-35: decim = int(fs/cr)/2, low_pass_2(1.0, fs, cr/2, cr/2, 20, HAMMING);
        # the C ABI derives both (rcf_chan_open) and rejects non-integral decimations
        if bin_ is not None:
            chan_id = frontend.pfb_tap_open(bin_, gr_phase=True)
            start = self._start_of(chan_id)
            self.chan_id, self.pfb_bin = chan_id, bin_
            self.decim, self.ntaps = pfb["decim"], pfb["ntaps"]
            self.out_rate = samp_rate / pfb["decim"]
            self.start_sample = start
            return
        if self.parent_chan is None:
            chan_id = frontend.chan_open(channel_rate, offset)
        else:
            from . import native
            decim, ntaps = native.channel_params(samp_rate, channel_rate, self.decim_rule)
            taps = native.design_low_pass_2(1.0, samp_rate, channel_rate / 2, channel_rate / 2, 20.0)
            assert len(taps) == ntaps
            chan_id = frontend.chan_open_taps(self.parent_chan, decim, taps, offset)
        try:
            info = frontend.chan_info(chan_id)
            start = self._start_of(chan_id)
        except Exception:
            # nothing holds chan_id yet: close it, or the native channel leaks with no owner
            try:
                frontend.chan_close(chan_id)
            except Exception:
                pass
            raise
        self.chan_id, self.pfb_bin = chan_id, None
        self.decim = info["decim"]
        self.ntaps = info["ntaps"]
        self.out_rate = info["out_rate"]
        self.start_sample = start

    def _start_of(self, chan_id):
        """index in the channel's source stream of the first sample it sees (rcf_chan_start); None for front-ends
        that cannot tell (stubs)"""
        f = getattr(self.frontend, "chan_start", None)
        return f(chan_id) if f is not None else None

    def __str__(self):
        return "Channel: port:%s channel_rate:%s samp_rate:%s offset:%s init_time:%s" % (
            self.port, self.channel_rate, self.samp_rate, self.offset, self.init_time)

    def __repr__(self):
        return "<Channel port:%s channel_rate:%s samp_rate:%s offset:%s init_time:%s>" % (
            self.port, self.channel_rate, self.samp_rate, self.offset, self.init_time)

    # gr.top_block surface used by the reference's receiver
    def start(self):
        self._started = True

    def stop(self):
        self._started = False

    def get_samp_rate(self):
        return self.samp_rate

    def get_channel_rate(self):
        return self.channel_rate

    def get_offset(self):
        return self.offset

    def set_offset(self, offset):
        """channel.py:61-63 -> prefilter.set_center_freq.  A direct channel is retuned in place: rotator phase and
        FIR history kept, as GNU Radio does.  In filterbank mode a retune that changes the serving path (another bin,
        bin -> direct, direct -> bin) replaces the native channel: the new one starts with zero history and a fresh
        rotator, and symbol-filter / voice-chain attachments of the old id are gone (the reference only retunes
        channels it is re-using after they sat idle, receiver.py:311-319)."""
        new_bin = self._target_bin(offset)
        if new_bin is None and self.pfb_bin is None:
            self.frontend.chan_set_offset(self.chan_id, offset)        # direct stays direct
            self.offset = offset
            return
        if new_bin is not None and new_bin == self.pfb_bin:
            self.offset = offset                                       # same bin: nothing to do
            return
        old = self.chan_id
        self._build(offset)                                            # raises before any state changed
        self.offset = offset
        try:
            self.frontend.chan_close(old)
        except Exception as e:
            # the object already points at the new native channel; the old one is unreferenced from here on
            import logging
            logging.getLogger("frontend").error("closing replaced native channel %s failed: %s" % (old, e))

    def destroy(self):
        """channel.py:64-67"""
        if self.chan_id is not None:
            self.frontend.chan_close(self.chan_id)
            self.chan_id = None
        self.stop()

    # data plane (what the reference's zeromq.pub_sink at channel.py:36 would carry)
    def read_iq(self, max_samples=1 << 20):
        return self.frontend.chan_read_iq(self.chan_id, max_samples)

    def attach_analog_voice(self, **kw):
        """Run the reference's analog voice chain (logging_receiver.py:211-222: pwr_squelch_cc -> fm_demod_cf ->
        300 Hz high-pass -> rational_resampler to 8 kHz) on this channel's stream on the GPU; the consumer then
        reads finished 8 kHz float audio instead of 2 x channel_rate complex samples."""
        from . import audio
        audio.open_analog_voice(self.frontend, self.chan_id, self.out_rate, **kw)

    def read_audio(self, max_samples=1 << 20):
        return self.frontend.chan_read_audio(self.chan_id, max_samples)

    def read_fm(self, gain, max_samples=1 << 20):
        """analog.quadrature_demod_cf(gain) of the channel stream (p25_control_demod.py:120-121)."""
        return self.frontend.chan_read_fm(self.chan_id, gain, max_samples)
