"""ctypes binding of librcf.so (include/rcf.h).  Thin: numpy arrays in/out, no arithmetic here.

There is no CPU fallback: if librcf.so is missing the import of any compute entry point raises
RuntimeError, and rcf_open() fails without a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCF_LIBRCF: another build of the same library (the AddressSanitizer build of the host layer, tools/asan_gpu.sh)
_SO = os.environ.get("RCF_LIBRCF") or os.path.join(_HERE, "librcf.so")
_lib = None

RCF_OK, RCF_EINVAL, RCF_ENOMEM, RCF_EHIP, RCF_ENOCHAN = 0, -1, -2, -3, -4
RCF_ECAP, RCF_ESTATE, RCF_EAGAIN, RCF_ERANGE = -5, -6, -7, -8
WIN_HAMMING, WIN_BLACKMAN, WIN_KAISER, WIN_BLACKMAN_HARRIS = 0, 2, 4, 5
FIR_LOW_PASS, FIR_HIGH_PASS = 0, 1
SRC_PFB_BIN0 = 0x40000000

# every symbol include/rcf.h declares (tests check the library exports all of them)
SYMBOLS = [
    "rcf_version", "rcf_last_error", "rcf_device_count", "rcf_device_pci_bus_id", "rcf_design_low_pass_2", "rcf_design_window",
    "rcf_channel_params", "rcf_channel_params_ex", "rcf_set_decim_rule", "rcf_open", "rcf_open_ex", "rcf_close", "rcf_sync", "rcf_stream", "rcf_device",
    "rcf_push_iq", "rcf_ingest_ptr", "rcf_commit", "rcf_samples_in", "rcf_chan_open", "rcf_chan_open_taps",
    "rcf_chan_set_offset", "rcf_chan_close", "rcf_chan_info", "rcf_chan_produced", "rcf_chan_start", "rcf_chan_read_many", "rcf_chan_read_iq",
    "rcf_chan_read_fm", "rcf_chan_rings", "rcf_source_shift", "rcf_pfb_open", "rcf_pfb_close",
    "rcf_pfb_produced", "rcf_pfb_read_bin", "rcf_pfb_rings", "rcf_pfb_fm_enable", "rcf_pfb_read_fm", "rcf_pfb_fm_ring", "rcf_pfb_fm_lost", "rcf_pfb_chan_open", "rcf_scan_start",
    "rcf_scan_result", "rcf_scan_frames_done", "rcf_scan_result_device", "rcf_find_peaks",
    "rcf_peak_frequency", "rcf_scan_find_peaks", "rcf_timing_enable", "rcf_timing_read",
    "rcf_ingest_write", "rcf_push_raw", "rcf_chan_fm_filter", "rcf_chan_read_sym", "rcf_chan_fm_level",
    "rcf_design_firdes", "rcf_design_optfir_low_pass", "rcf_design_fm_deemph", "rcf_design_resampler", "rcf_chan_audio_open",
    "rcf_chan_audio_close", "rcf_chan_audio_produced", "rcf_chan_read_audio",
    "rcf_host_alloc", "rcf_host_free", "rcf_comm_unique_id", "rcf_comm_init", "rcf_comm_destroy", "rcf_comm_size",
    "rcf_allgather_peaks", "rcf_allreduce_max", "rcf_pfb_tap_open", "rcf_chan_set_fm_only", "rcf_pfb_shape_supported",
    "rcf_pfb_tap_leakage", "rcf_set_rotator", "rcf_timing_stride", "rcf_set_stage2_lag",
    "rcf_group_open", "rcf_group_close", "rcf_group_size", "rcf_group_push", "rcf_group_commit", "rcf_group_read_many",
    "rcf_group_sync", "rcf_pump_start", "rcf_pump_stats", "rcf_pump_written", "rcf_pump_read", "rcf_pump_read_many", "rcf_pump_subscribe", "rcf_pump_unsubscribe", "rcf_pump_stop",
]
FMT_CF32, FMT_U8, FMT_S8, FMT_S16 = 0, 1, 2, 3
READ_IQ, READ_FM = 0, 1
T_FIR, T_PFB, T_FIR_DERIVED, T_DISC, T_SCAN_FFT, T_SCAN_MOVSUM, T_HISTORY, T_FIR_MFMA, T_AUDIO, T_TAPS = range(10)


class AudioParams(C.Structure):
    """rcf_audio_params_t (include/rcf.h)"""
    _fields_ = [("squelch_db", C.c_double), ("squelch_alpha", C.c_double), ("quad_gain", C.c_float),
                ("reserved_", C.c_int), ("deemph_b", C.c_double * 2), ("deemph_a", C.c_double * 2),
                ("lpf_taps", C.POINTER(C.c_float)), ("n_lpf", C.c_int), ("n_hpf", C.c_int),
                ("hpf_taps", C.POINTER(C.c_float)), ("rs_taps", C.POINTER(C.c_float)), ("n_rs", C.c_int),
                ("interpolation", C.c_int), ("decimation", C.c_int), ("reserved2_", C.c_int)]


class PumpConfig(C.Structure):
    """rcf_pump_config_t (include/rcf.h)"""
    _fields_ = [("block_samples", C.c_size_t), ("fmt", C.c_int), ("scale", C.c_float), ("offset", C.c_float),
                ("samp_rate", C.c_double), ("rings", C.POINTER(C.c_void_p)), ("ring_blocks", C.POINTER(C.c_size_t)),
                ("phase_s", C.POINTER(C.c_double)), ("written", C.POINTER(C.c_void_p)), ("what", C.c_int),
                ("gain", C.c_float), ("read_members", C.POINTER(C.c_int)), ("read_chans", C.POINTER(C.c_int)),
                ("n_read", C.c_int), ("out_ring_samples", C.c_size_t), ("n_blocks", C.c_int64),
                ("warm_blocks", C.c_int64), ("max_batch", C.c_int), ("cpu", C.c_int), ("start_delay_s", C.c_double),
                ("batch_window_s", C.c_double), ("rt_priority", C.c_int), ("spin_us", C.c_int), ("max_read", C.c_int)]


class PumpStats(C.Structure):
    """rcf_pump_stats_t (include/rcf.h)"""
    _fields_ = [("blocks_done", C.c_int64), ("blocks_judged", C.c_int64), ("late", C.c_int64), ("overruns", C.c_int64),
                ("group_blocks", C.c_int64), ("max_batch", C.c_int64), ("samples_out", C.c_int64),
                ("latency_ms_p50", C.c_double), ("latency_ms_p99", C.c_double), ("latency_ms_max", C.c_double),
                ("host_plan_ms", C.c_double), ("host_wait_ms", C.c_double), ("elapsed_s", C.c_double),
                ("max_plan_ms", C.c_double), ("max_wait_ms", C.c_double), ("max_sleep_overshoot_ms", C.c_double),
                ("slow_plans", C.c_int64), ("slow_waits", C.c_int64), ("slow_sleeps", C.c_int64),
                ("rt_priority_granted", C.c_int),
                ("running", C.c_int), ("error", C.c_int),
                ("cpu", C.c_int), ("runq_ms_total", C.c_double), ("slow_wait_ms_total", C.c_double),
                ("runq_ms_in_slow_waits", C.c_double), ("slow_sleep_ms_total", C.c_double),
                ("runq_ms_in_slow_sleeps", C.c_double), ("involuntary_switches", C.c_int64)]


class RcfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librcf error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError("librcf.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % _SO)
    L = C.CDLL(_SO)
    fp, vp, i64, sz = C.POINTER(C.c_float), C.c_void_p, C.c_int64, C.c_size_t
    ip = C.POINTER(C.c_int)
    sig = {
        "rcf_version": (C.c_char_p, []),
        "rcf_last_error": (C.c_char_p, []),
        "rcf_device_count": (C.c_int, []),
        "rcf_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
        "rcf_design_low_pass_2": (C.c_int, [C.c_double] * 5 + [C.c_int, fp, C.c_int]),
        "rcf_design_window": (C.c_int, [C.c_int, C.c_int, fp]),
        "rcf_channel_params": (C.c_int, [C.c_double, C.c_int, ip, ip]),
        "rcf_channel_params_ex": (C.c_int, [C.c_double, C.c_int, C.c_int, ip, ip, C.POINTER(C.c_double)]),
        "rcf_set_decim_rule": (C.c_int, [vp, C.c_int]),
        "rcf_open": (C.c_int, [C.c_int, C.c_double, C.c_double, C.POINTER(vp)]),
        "rcf_open_ex": (C.c_int, [C.c_int, C.c_double, C.c_double, sz, sz, sz, C.POINTER(vp)]),
        "rcf_close": (C.c_int, [vp]),
        "rcf_set_rotator": (C.c_int, [vp, C.c_int]),
        "rcf_set_stage2_lag": (C.c_int, [vp, C.c_int]),
        "rcf_sync": (C.c_int, [vp]),
        "rcf_stream": (vp, [vp]),
        "rcf_device": (C.c_int, [vp]),
        "rcf_push_iq": (C.c_int, [vp, fp, sz]),
        "rcf_ingest_ptr": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "rcf_commit": (C.c_int, [vp, sz]),
        "rcf_ingest_write": (C.c_int, [vp, fp, sz, sz]),
        "rcf_push_raw": (C.c_int, [vp, vp, sz, C.c_int, C.c_float, C.c_float]),
        "rcf_chan_fm_filter": (C.c_int, [vp, C.c_int, C.c_float, fp, C.c_int]),
        "rcf_chan_read_sym": (i64, [vp, C.c_int, fp, sz]),
        "rcf_chan_fm_level": (C.c_int, [vp, C.c_int, C.c_float, C.c_int, fp]),
        "rcf_design_firdes": (C.c_int, [C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_double, fp, C.c_int]),
        "rcf_design_optfir_low_pass": (C.c_int, [C.c_double] * 6 + [fp, C.c_int]),
        "rcf_design_fm_deemph": (C.c_int, [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rcf_design_resampler": (C.c_int, [C.c_int, C.c_int, ip, ip, fp, C.c_int]),
        "rcf_chan_audio_open": (C.c_int, [vp, C.c_int, C.POINTER(AudioParams)]),
        "rcf_chan_audio_close": (C.c_int, [vp, C.c_int]),
        "rcf_chan_audio_produced": (C.c_int, [vp, C.c_int, C.POINTER(i64), C.POINTER(i64)]),
        "rcf_chan_read_audio": (i64, [vp, C.c_int, fp, sz]),
        "rcf_samples_in": (i64, [vp]),
        "rcf_chan_open": (C.c_int, [vp, C.c_int, C.c_double, ip]),
        "rcf_chan_open_taps": (C.c_int, [vp, C.c_int, C.c_int, fp, C.c_int, C.c_double, ip]),
        "rcf_chan_set_offset": (C.c_int, [vp, C.c_int, C.c_double]),
        "rcf_chan_close": (C.c_int, [vp, C.c_int]),
        "rcf_chan_info": (C.c_int, [vp, C.c_int, ip, ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rcf_chan_produced": (i64, [vp, C.c_int]),
        "rcf_chan_start": (i64, [vp, C.c_int]),
        "rcf_chan_read_many": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float, vp, C.c_size_t,
                                         C.POINTER(C.c_int64)]),
        "rcf_chan_read_iq": (i64, [vp, C.c_int, fp, sz]),
        "rcf_chan_read_fm": (i64, [vp, C.c_int, C.c_float, fp, sz]),
        "rcf_chan_rings": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]),
        "rcf_source_shift": (C.c_int, [vp, C.c_double]),
        "rcf_pfb_open": (C.c_int, [vp, C.c_int, C.c_int, fp, C.c_int]),
        "rcf_pfb_close": (C.c_int, [vp]),
        "rcf_pfb_produced": (i64, [vp]),
        "rcf_pfb_read_bin": (i64, [vp, C.c_int, fp, sz]),
        "rcf_pfb_fm_enable": (C.c_int, [vp, C.c_int, C.c_int]),
        "rcf_pfb_read_fm": (i64, [vp, C.c_int, C.c_float, fp, sz]),
        "rcf_pfb_fm_ring": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(i64)]),
        "rcf_pfb_fm_lost": (i64, [vp]),
        "rcf_pfb_rings": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]),
        "rcf_pfb_chan_open": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, ip]),
        "rcf_pfb_tap_open": (C.c_int, [vp, C.c_int, C.c_int, ip]),
        "rcf_chan_set_fm_only": (C.c_int, [vp, C.c_int, C.c_int]),
        "rcf_pfb_shape_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "rcf_pfb_tap_leakage": (C.c_int, [C.c_double, C.c_int, fp, C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]),
        "rcf_scan_start": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "rcf_scan_result": (C.c_int, [vp, fp]),
        "rcf_scan_frames_done": (C.c_int, [vp]),
        "rcf_scan_result_device": (C.c_int, [vp, C.POINTER(vp)]),
        "rcf_find_peaks": (C.c_int, [fp, i64, C.c_double, C.c_double, C.c_double, C.POINTER(i64), i64,
                                     C.POINTER(i64), C.POINTER(C.c_double)]),
        "rcf_peak_frequency": (i64, [i64, C.c_double, i64, C.c_double]),
        "rcf_timing_enable": (C.c_int, [vp, C.c_int]),
        "rcf_timing_stride": (C.c_int, [vp, C.c_int]),
        "rcf_timing_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(i64), C.c_int]),
        "rcf_scan_find_peaks": (C.c_int, [vp, C.c_double, C.POINTER(i64), i64, C.POINTER(i64),
                                          C.POINTER(C.c_double), C.POINTER(vp)]),
        "rcf_host_alloc": (vp, [sz]),
        "rcf_host_free": (None, [vp]),
        "rcf_comm_unique_id": (C.c_int, [vp]),
        "rcf_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "rcf_comm_destroy": (C.c_int, [vp]),
        "rcf_comm_size": (C.c_int, [vp]),
        "rcf_allgather_peaks": (C.c_int, [vp, C.POINTER(i64), C.c_int, C.POINTER(i64), C.c_int, ip]),
        "rcf_allreduce_max": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "rcf_group_open": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
        "rcf_group_close": (C.c_int, [vp]),
        "rcf_group_size": (C.c_int, [vp]),
        "rcf_group_push": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_float, C.c_float]),
        "rcf_group_commit": (C.c_int, [vp, C.POINTER(sz)]),
        "rcf_group_read_many": (C.c_int, [vp, C.c_int, ip, ip, C.c_int, C.c_float, vp, sz, C.POINTER(i64)]),
        "rcf_group_sync": (C.c_int, [vp]),
        "rcf_pump_start": (C.c_int, [vp, C.POINTER(PumpConfig), C.POINTER(vp)]),
        "rcf_pump_stats": (C.c_int, [vp, C.POINTER(PumpStats)]),
        "rcf_pump_written": (i64, [vp, C.c_int]),
        "rcf_pump_read": (i64, [vp, C.c_int, C.POINTER(i64), vp, sz]),
        "rcf_pump_read_many": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(i64), C.c_int, vp, sz, C.POINTER(i64)]),
        "rcf_pump_subscribe": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(i64)]),
        "rcf_pump_unsubscribe": (C.c_int, [vp, C.c_int]),
        "rcf_pump_stop": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise RcfError(rc, lib().rcf_last_error().decode("utf-8", "replace"))
    return rc


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def device_count() -> int:
    return lib().rcf_device_count()


def device_pci_bus_id(device: int):
    """'0000:75:00.0' of HIP device `device` (sysfs: /sys/bus/pci/devices/<that>/numa_node), None if unknown"""
    buf = C.create_string_buffer(32)
    return buf.value.decode() if lib().rcf_device_pci_bus_id(device, buf, 32) == 0 else None


def design_low_pass_2(gain, fs, fc, tw, att_db, window=WIN_HAMMING) -> np.ndarray:
    L = lib()
    n = -L.rcf_design_low_pass_2(gain, fs, fc, tw, att_db, window, None, 0)
    if n <= 0:
        _check(-n if n else RCF_EINVAL)
    taps = np.empty(n, dtype=np.float32)
    _check(L.rcf_design_low_pass_2(gain, fs, fc, tw, att_db, window, _fp(taps), n))
    return taps


def design_firdes(kind, gain, fs, fc, tw, window=WIN_HAMMING, beta=6.76) -> np.ndarray:
    """firdes.low_pass / firdes.high_pass(gain, fs, fc, tw, window, beta)"""
    L = lib()
    n = -L.rcf_design_firdes(kind, gain, fs, fc, tw, window, beta, None, 0)
    if n <= 0:
        _check(-n if n else RCF_EINVAL)
    taps = np.empty(n, dtype=np.float32)
    _check(L.rcf_design_firdes(kind, gain, fs, fc, tw, window, beta, _fp(taps), n))
    return taps


def design_optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db) -> np.ndarray:
    """gr-filter optfir.low_pass(): remezord + Parks-McClellan"""
    L = lib()
    n = -L.rcf_design_optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, None, 0)
    if n <= 0:
        _check(-n if n else RCF_EINVAL)
    taps = np.empty(n, dtype=np.float32)
    _check(L.rcf_design_optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, _fp(taps), n))
    return taps


def design_fm_deemph(fs, tau=75e-6):
    b, a = (C.c_double * 2)(), (C.c_double * 2)()
    _check(lib().rcf_design_fm_deemph(fs, tau, b, a))
    return [b[0], b[1]], [a[0], a[1]]


def design_resampler(interpolation, decimation):
    """-> (interp, decim, taps) of rational_resampler_fff(interpolation, decimation) with default taps"""
    L = lib()
    i, d = C.c_int(), C.c_int()
    n = -L.rcf_design_resampler(int(interpolation), int(decimation), C.byref(i), C.byref(d), None, 0)
    if n <= 0:
        _check(-n if n else RCF_EINVAL)
    taps = np.empty(n, dtype=np.float32)
    _check(L.rcf_design_resampler(int(interpolation), int(decimation), C.byref(i), C.byref(d), _fp(taps), n))
    return i.value, d.value, taps


def design_window(window, n) -> np.ndarray:
    w = np.empty(n, dtype=np.float32)
    _check(lib().rcf_design_window(window, n, _fp(w)))
    return w


DECIM_EXACT, DECIM_FLOOR = 0, 1


def channel_params(samp_rate, channel_rate, decim_rule=DECIM_EXACT):
    """(decim, ntaps) of rc_frontend/channel.py:31-33; decim_rule=DECIM_FLOOR: the Python-2 reading int(fs/cr) // 2"""
    d, t = C.c_int(), C.c_int()
    _check(lib().rcf_channel_params_ex(samp_rate, int(channel_rate), int(decim_rule), C.byref(d), C.byref(t), None))
    return d.value, t.value


def pfb_shape_supported(n_bins, decim, ntaps) -> bool:
    return bool(lib().rcf_pfb_shape_supported(int(n_bins), int(decim), int(ntaps)))


def pfb_tap_leakage(samp_rate, n_bins, taps, bin):
    """(leak_l2, const_phase) of rcf_pfb_tap_leakage: how far bin `bin` of an exact-phase bank is from GNU Radio's
    float32-phase channel at the same offset."""
    t = np.ascontiguousarray(taps, dtype=np.float32)
    l2, cp = C.c_double(), C.c_double()
    _check(lib().rcf_pfb_tap_leakage(float(samp_rate), int(n_bins), _fp(t), len(t), int(bin), C.byref(l2),
                                     C.byref(cp)))
    return l2.value, cp.value


def find_peaks(spectrum, min_w, max_w, prominence=1.0, cap=4096):
    s = np.ascontiguousarray(spectrum, dtype=np.float32)
    idx = np.empty(cap, dtype=np.int64)
    cnt, mean = C.c_int64(), C.c_double()
    _check(lib().rcf_find_peaks(_fp(s), len(s), min_w, max_w, prominence,
                                idx.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(cnt), C.byref(mean)))
    return idx[:min(cnt.value, cap)].copy(), mean.value, cnt.value


def peak_frequency(line, samp_rate, fft_len, center_freq) -> int:
    return lib().rcf_peak_frequency(int(line), samp_rate, int(fft_len), center_freq)


class PinnedArray:
    """numpy view of page-locked host memory (rcf_host_alloc): the buffer an SDR driver fills and hands to
    Frontend.push / push_raw; freed when the object goes away"""

    def __init__(self, n, dtype):
        self.dtype = np.dtype(dtype)
        self.nbytes = int(n) * self.dtype.itemsize
        self._p = lib().rcf_host_alloc(self.nbytes)
        if not self._p:
            raise RcfError(RCF_ENOMEM, lib().rcf_last_error().decode("utf-8", "replace"))
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p), dtype=self.dtype)

    def free(self):
        if self._p:
            self.array = None
            lib().rcf_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    """rank 0: the 128-byte RCCL id every rank passes to Frontend.comm_init"""
    buf = C.create_string_buffer(128)
    _check(lib().rcf_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


class Frontend:
    """One SDR source: owner of the HBM wideband buffer, channels, PFB and scanner (rcf_t)."""

    def __init__(self, samp_rate, center_freq=0.0, device=0, block_capacity=0, hist_capacity=0,
                 out_capacity=0):
        self._h = C.c_void_p()
        self.samp_rate = float(samp_rate)
        self.center_freq = float(center_freq)
        # the scan's length sizes scan_result()'s buffer: a scan_start() on another thread must not change it between
        # the allocation and the copy
        self._scan_len = 0
        self._scan_lock = threading.Lock()
        _check(lib().rcf_open_ex(device, samp_rate, center_freq, block_capacity, hist_capacity,
                                 out_capacity, C.byref(self._h)))

    # -- lifecycle
    def close(self):
        if self._h:
            lib().rcf_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_rotator(self, exact=True):
        """exact: iterate GNU Radio's float32 rotator per channel (rcf_set_rotator); before the first channel"""
        _check(lib().rcf_set_rotator(self._h, 1 if exact else 0))

    def set_stage2_lag(self, on=True):
        """stage-2 channels of a power-of-two bank ride in the NEXT block's filterbank launch (rcf_set_stage2_lag; default on)"""
        _check(lib().rcf_set_stage2_lag(self._h, 1 if on else 0))

    def set_decim_rule(self, rule):
        """DECIM_EXACT (default) | DECIM_FLOOR (rcf_set_decim_rule): what rcf_chan_open does with an odd int(fs/cr)"""
        _check(lib().rcf_set_decim_rule(self._h, int(rule)))

    def sync(self):
        _check(lib().rcf_sync(self._h))

    @property
    def stream(self):
        return lib().rcf_stream(self._h)

    @property
    def samples_in(self):
        return lib().rcf_samples_in(self._h)

    # -- multi-GPU (one front-end per GPU; the only exchange is the detected-peak lists)
    def comm_init(self, rank, n_ranks, unique_id=None):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        _check(lib().rcf_comm_init(self._h, int(rank), int(n_ranks), C.cast(buf, C.c_void_p) if buf is not None else None))

    def comm_destroy(self):
        _check(lib().rcf_comm_destroy(self._h))

    def comm_size(self) -> int:
        """ranks of the handle's RCCL communicator (1 without one)"""
        return int(lib().rcf_comm_size(self._h))

    def allgather_peaks(self, mine, cap=1024):
        """-> list (one int64 array per rank) of every rank's values, via ncclAllGather on the handle's device"""
        mine = np.ascontiguousarray(mine, dtype=np.int64)
        w = lib().rcf_comm_size(self._h)
        out = np.empty(w * cap, dtype=np.int64)
        counts = (C.c_int * w)()
        _check(lib().rcf_allgather_peaks(self._h, mine.ctypes.data_as(C.POINTER(C.c_int64)), len(mine),
                                         out.ctypes.data_as(C.POINTER(C.c_int64)), int(cap), counts))
        return [out[r * cap: r * cap + counts[r]].copy() for r in range(w)]

    def allreduce_max(self, value: float) -> float:
        """barrier + max over ranks (syncs the stream first); identity without a communicator"""
        v = C.c_double(float(value))
        _check(lib().rcf_allreduce_max(self._h, C.byref(v)))
        return v.value

    # -- measurement
    def timing_enable(self, on=True, classes=None):
        """classes: iterable of T_* to time only those (each timed launch costs two event records)"""
        v = 1 if on else 0
        if on and classes is not None:
            v = 0
            for c in classes:
                v |= 1 << (int(c) + 1)
        _check(lib().rcf_timing_enable(self._h, v))

    def timing_stride(self, every=1):
        """time only every `every`-th launch of each timed class (an event pair costs ~12 us of queue gap)"""
        _check(lib().rcf_timing_stride(self._h, int(every)))

    def timing_read(self, what, reset=True):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().rcf_timing_read(self._h, int(what), C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    # -- ingest
    def push(self, iq: np.ndarray):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        _check(lib().rcf_push_iq(self._h, _fp(iq.view(np.float32)), len(iq)))

    def ingest_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().rcf_ingest_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def push_raw(self, raw: np.ndarray, fmt, scale, offset=0.0):
        """raw: interleaved I,Q in the SDR's wire type (uint8 / int8 / int16), 2 values per sample."""
        dt = {FMT_U8: np.uint8, FMT_S8: np.int8, FMT_S16: np.int16}[fmt]
        raw = np.ascontiguousarray(raw, dtype=dt)
        _check(lib().rcf_push_raw(self._h, raw.ctypes.data_as(C.c_void_p), raw.size // 2, int(fmt),
                                  float(scale), float(offset)))

    def ingest_write(self, iq: np.ndarray, at=0):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        _check(lib().rcf_ingest_write(self._h, _fp(iq.view(np.float32)), len(iq), int(at)))

    def commit(self, n):
        _check(lib().rcf_commit(self._h, int(n)))

    # -- channels
    def chan_open(self, channel_rate, offset_hz) -> int:
        cid = C.c_int()
        _check(lib().rcf_chan_open(self._h, int(channel_rate), float(offset_hz), C.byref(cid)))
        return cid.value

    def chan_open_taps(self, src, decim, taps, offset_hz) -> int:
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        cid = C.c_int()
        _check(lib().rcf_chan_open_taps(self._h, int(src), int(decim), _fp(taps), len(taps),
                                        float(offset_hz), C.byref(cid)))
        return cid.value

    def pfb_chan_open(self, bin_, channel_rate, delta_hz) -> int:
        cid = C.c_int()
        _check(lib().rcf_pfb_chan_open(self._h, int(bin_), int(channel_rate), float(delta_hz), C.byref(cid)))
        return cid.value

    def pfb_tap_open(self, bin_, gr_phase=True) -> int:
        """bin `bin_` of the open filterbank as a channel id (rcf_pfb_tap_open)"""
        cid = C.c_int()
        _check(lib().rcf_pfb_tap_open(self._h, int(bin_), 1 if gr_phase else 0, C.byref(cid)))
        return cid.value

    def chan_set_fm_only(self, cid, on=True):
        """a filterbank tap that is only demodulated: its discriminator ring alone is written (rcf_chan_set_fm_only)"""
        _check(lib().rcf_chan_set_fm_only(self._h, cid, 1 if on else 0))

    def chan_set_offset(self, cid, offset_hz):
        _check(lib().rcf_chan_set_offset(self._h, cid, float(offset_hz)))

    def chan_close(self, cid):
        _check(lib().rcf_chan_close(self._h, cid))

    def chan_info(self, cid):
        d, t, r, o = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        _check(lib().rcf_chan_info(self._h, cid, C.byref(d), C.byref(t), C.byref(r), C.byref(o)))
        return dict(decim=d.value, ntaps=t.value, out_rate=r.value, offset_hz=o.value)

    def chan_start(self, cid):
        """first source-stream sample the channel sees (zero history before it)"""
        return _check(lib().rcf_chan_start(self._h, cid))

    def chan_produced(self, cid):
        return _check(lib().rcf_chan_produced(self._h, cid))

    def chan_read_iq(self, cid, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.complex64)
        n = _check(lib().rcf_chan_read_iq(self._h, cid, _fp(out.view(np.float32)), max_samples))
        return out[:n].copy()

    def chan_read_fm(self, cid, gain, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.float32)
        n = _check(lib().rcf_chan_read_fm(self._h, cid, float(gain), _fp(out), max_samples))
        return out[:n].copy()

    def chan_read_many(self, cids, what="iq", gain=1.0, cap_each=1 << 14, out=None):
        """new samples of many channels behind ONE device synchronisation (rcf_chan_read_many) -> list of arrays
        (views into `out`, a [len(cids), cap_each] complex64 / float32 array -- pass a PinnedArray's for overlapping
        copies), None where the channel no longer exists"""
        n = len(cids)
        dt = np.complex64 if what == "iq" else np.float32
        if out is None:
            out = np.empty((max(n, 1), cap_each), dtype=dt)
        out = out.reshape(-1, cap_each)
        ids = (C.c_int * max(n, 1))(*[int(c) for c in cids])
        counts = (C.c_int64 * max(n, 1))()
        _check(lib().rcf_chan_read_many(self._h, 0 if what == "iq" else 1, ids, n, float(gain),
                                        out.ctypes.data_as(C.c_void_p), int(cap_each), counts))
        return [None if counts[i] < 0 else out[i, :counts[i]] for i in range(n)]

    def chan_read_many_plan(self, cids, what="iq", gain=1.0, cap_each=1 << 14, out=None):
        """chan_read_many for a FIXED channel list called over and over (an egress pump, a real-time loop): the ctypes
        arguments are built once; the returned callable performs one rcf_chan_read_many and returns (counts, out) --
        counts an int64 array (negative: that channel is gone), out the [len(cids), cap_each] array the samples are in.
        Costs ~10 us of interpreter time per call instead of ~1 us per channel."""
        n = len(cids)
        dt = np.complex64 if what == "iq" else np.float32
        if out is None:
            out = np.empty((max(n, 1), cap_each), dtype=dt)
        out2 = out.reshape(-1, cap_each)
        ids = (C.c_int * max(n, 1))(*[int(c) for c in cids])
        counts = np.zeros(max(n, 1), dtype=np.int64)
        f = lib().rcf_chan_read_many
        args = (self._h, 0 if what == "iq" else 1, ids, n, C.c_float(float(gain)), out2.ctypes.data_as(C.c_void_p),
                C.c_size_t(int(cap_each)), counts.ctypes.data_as(C.POINTER(C.c_int64)))

        def call():
            _check(f(*args))
            return counts, out2
        call.keep = (ids, counts, out2)                     # the buffers the C call writes into live as long as the plan
        return call

    def chan_fm_filter(self, cid, gain, taps):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        _check(lib().rcf_chan_fm_filter(self._h, cid, float(gain), _fp(taps), len(taps)))

    def chan_read_sym(self, cid, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.float32)
        n = _check(lib().rcf_chan_read_sym(self._h, cid, _fp(out), max_samples))
        return out[:n].copy()

    def chan_fm_level(self, cid, gain, window=10000) -> float:
        v = C.c_float()
        _check(lib().rcf_chan_fm_level(self._h, cid, float(gain), int(window), C.byref(v)))
        return v.value

    def chan_audio_open(self, cid, squelch_db, squelch_alpha, quad_gain, deemph_b, deemph_a, lpf_taps, hpf_taps,
                        interpolation, decimation, rs_taps):
        """analog voice chain behind a channel (rcf_chan_audio_open); rcf.audio.analog_chain_params builds the
        arguments the way the reference's GNU Radio calls do"""
        lpf = np.ascontiguousarray(lpf_taps, dtype=np.float32)
        hpf = np.ascontiguousarray(hpf_taps, dtype=np.float32)
        rs = np.ascontiguousarray(rs_taps, dtype=np.float32)
        p = AudioParams()
        p.squelch_db, p.squelch_alpha, p.quad_gain = float(squelch_db), float(squelch_alpha), float(quad_gain)
        p.deemph_b[0], p.deemph_b[1] = float(deemph_b[0]), float(deemph_b[1])
        p.deemph_a[0], p.deemph_a[1] = float(deemph_a[0]), float(deemph_a[1])
        p.lpf_taps, p.n_lpf = _fp(lpf), len(lpf)
        p.hpf_taps, p.n_hpf = _fp(hpf), len(hpf)
        p.rs_taps, p.n_rs = _fp(rs), len(rs)
        p.interpolation, p.decimation = int(interpolation), int(decimation)
        _check(lib().rcf_chan_audio_open(self._h, cid, C.byref(p)))

    def chan_audio_close(self, cid):
        _check(lib().rcf_chan_audio_close(self._h, cid))

    def chan_audio_produced(self, cid):
        a, u = C.c_int64(), C.c_int64()
        _check(lib().rcf_chan_audio_produced(self._h, cid, C.byref(a), C.byref(u)))
        return a.value, u.value

    def chan_read_audio(self, cid, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.float32)
        n = _check(lib().rcf_chan_read_audio(self._h, cid, _fp(out), max_samples))
        return out[:n].copy()

    def source_shift(self, delta_hz):
        _check(lib().rcf_source_shift(self._h, float(delta_hz)))

    # -- PFB
    def pfb_open(self, n_bins, decim, taps):
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        _check(lib().rcf_pfb_open(self._h, int(n_bins), int(decim), _fp(taps), len(taps)))

    def pfb_close(self):
        _check(lib().rcf_pfb_close(self._h))

    def pfb_produced(self):
        return _check(lib().rcf_pfb_produced(self._h))

    def pfb_read_bin(self, bin_, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.complex64)
        n = _check(lib().rcf_pfb_read_bin(self._h, int(bin_), _fp(out.view(np.float32)), max_samples))
        return out[:n].copy()

    def pfb_fm_enable(self, mode=1, gr_phase=True):
        """the discriminator of EVERY bin in the bank's own kernel (rcf_pfb_fm_enable): mode 1 beside the bins ring, 2
        instead of it, 0 off"""
        _check(lib().rcf_pfb_fm_enable(self._h, int(mode), 1 if gr_phase else 0))

    def pfb_fm_lost(self):
        return _check(lib().rcf_pfb_fm_lost(self._h))

    def pfb_read_fm(self, bin_, gain=1.0, max_samples=1 << 20) -> np.ndarray:
        out = np.empty(max_samples, dtype=np.float32)
        n = _check(lib().rcf_pfb_read_fm(self._h, int(bin_), float(gain), _fp(out), max_samples))
        return out[:n].copy()

    # -- scan
    def scan_start(self, fft_len, n_frames=1000, avg_len=100):
        with self._scan_lock:
            _check(lib().rcf_scan_start(self._h, int(fft_len), int(n_frames), int(avg_len)))
            self._scan_len = int(fft_len)            # (only once librcf has accepted the scan)

    def scan_frames_done(self):
        return _check(lib().rcf_scan_frames_done(self._h))

    def scan_result(self):
        with self._scan_lock:
            if not self._scan_len:
                return None                          # no scan was ever started
            out = np.empty(self._scan_len, dtype=np.float32)
            rc = lib().rcf_scan_result(self._h, _fp(out))
        if rc == RCF_EAGAIN:
            return None
        _check(rc)
        return out

    def scan_find_peaks(self, prominence=1.0, cap=1024, want_device=False):
        idx = np.empty(cap, dtype=np.int64)
        cnt, mean, dev = C.c_int64(), C.c_double(), C.c_void_p()
        _check(lib().rcf_scan_find_peaks(self._h, prominence, idx.ctypes.data_as(C.POINTER(C.c_int64)), cap,
                                         C.byref(cnt), C.byref(mean), C.byref(dev) if want_device else None))
        return idx[:min(cnt.value, cap)].copy(), mean.value, dev.value


_WIRE_DTYPE = {FMT_CF32: np.complex64, FMT_U8: np.uint8, FMT_S8: np.int8, FMT_S16: np.int16}


class Group:
    """G front-ends of one device whose blocks go out as ONE launch per stage (rcf_group_t): the reference's receiver with
    all its sources in one top block (rc_frontend/receiver.py:67-70,170-204).  The members stay usable one by one (channel
    control, reads); close the group before closing them."""

    def __init__(self, frontends):
        self.frontends = list(frontends)
        self._g = C.c_void_p()
        arr = (C.c_void_p * len(self.frontends))(*[fe._h for fe in self.frontends])
        _check(lib().rcf_group_open(arr, len(self.frontends), C.byref(self._g)))

    def close(self):
        if self._g:
            _check(lib().rcf_group_close(self._g))
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.frontends)

    def push(self, blocks, fmt=FMT_CF32, scale=1.0, offset=0.0):
        """blocks[i]: member i's next block (complex64 for FMT_CF32, interleaved I,Q in the wire type otherwise), or
        None / empty to skip the member this time."""
        n = len(self.frontends)
        if len(blocks) != n:
            raise ValueError("one block (or None) per member")
        dt = _WIRE_DTYPE[fmt]
        keep, ptrs, cnt = [], (C.c_void_p * n)(), (C.c_size_t * n)()
        for i, b in enumerate(blocks):
            if b is None or len(b) == 0:
                ptrs[i], cnt[i] = None, 0
                continue
            b = np.ascontiguousarray(b, dtype=dt)
            keep.append(b)
            ptrs[i] = b.ctypes.data
            cnt[i] = len(b) if fmt == FMT_CF32 else b.size // 2
        _check(lib().rcf_group_push(self._g, ptrs, cnt, int(fmt), float(scale), float(offset)))

    def commit(self, counts):
        n = len(self.frontends)
        cnt = (C.c_size_t * n)(*[int(c) for c in counts])
        _check(lib().rcf_group_commit(self._g, cnt))

    def sync(self):
        _check(lib().rcf_group_sync(self._g))

    def read_many(self, pairs, what="iq", gain=1.0, cap_each=1 << 14):
        """pairs: [(member index, channel id), ...] -> list of arrays (None where the channel is gone), ONE gather launch
        and one synchronisation for all of them (rcf_group_read_many)"""
        n = len(pairs)
        dt = np.complex64 if what == "iq" else np.float32
        out = np.empty((max(n, 1), cap_each), dtype=dt)
        ms = (C.c_int * max(n, 1))(*[int(m) for m, _ in pairs])
        cs = (C.c_int * max(n, 1))(*[int(c) for _, c in pairs])
        counts = (C.c_int64 * max(n, 1))()
        _check(lib().rcf_group_read_many(self._g, READ_IQ if what == "iq" else READ_FM, ms, cs, n, float(gain),
                                         out.ctypes.data_as(C.c_void_p), int(cap_each), counts))
        return [None if counts[i] < 0 else out[i, :counts[i]].copy() for i in range(n)]


class Pump:
    """The native real-time loop of a group (rcf_pump_t): one C++ thread takes whichever members' blocks are complete,
    pushes them as one group block, gathers the subscribed channels' new output into pinned host rings.  `rings[i]`: a
    PinnedArray (or any pinned uint8 / int8 / int16 / complex64 array) holding whole blocks of member i -- replayed
    cyclically at `samp_rate` of wall-clock time, block boundaries shifted by phase_s[i], or, where `written[i]` is a
    one-element uint64 array, taken block by block as its producer counts them complete.  Subscriptions may be given at
    start ((member, channel id) pairs delivered as `what` x `gain`) and change while it runs (subscribe / unsubscribe, up to
    `max_read` at a time)."""

    def __init__(self, group, rings, block_samples, samp_rate, subscriptions=(), fmt=FMT_U8, scale=1.0, offset=0.0,
                 what="fm", gain=1.0, phase_s=None, out_ring_samples=4096, n_blocks=0, warm_blocks=0, max_batch=0,
                 cpu=-1, start_delay_s=0.05, batch_window_s=0.0, rt_priority=0, spin_us=0, written=None, max_read=0):
        self.group = group
        n = len(group)
        if len(rings) != n:
            raise ValueError("one source ring per member")
        self._keep = []
        bps = {FMT_CF32: 8, FMT_U8: 2, FMT_S8: 2, FMT_S16: 4}[fmt]
        rp, rb = (C.c_void_p * n)(), (C.c_size_t * n)()
        for i, r in enumerate(rings):
            a = r.array if isinstance(r, PinnedArray) else r
            self._keep.append(r)
            rp[i] = a.ctypes.data
            rb[i] = a.nbytes // (bps * int(block_samples))
            if rb[i] < 1:
                raise ValueError("source ring %d is shorter than one block" % i)
        ph = (C.c_double * n)(*([0.0] * n if phase_s is None else [float(x) for x in phase_s]))
        wr = None
        if written is not None and any(w is not None for w in written):
            wr = (C.c_void_p * n)()
            for i, w in enumerate(written):
                if w is None:
                    wr[i] = None
                    continue
                if w.dtype != np.uint64 or w.size < 1:
                    raise ValueError("written[%d] must be a uint64 array" % i)
                self._keep.append(w)
                wr[i] = w.ctypes.data
        ne = len(subscriptions)
        ms = (C.c_int * max(ne, 1))(*[int(m) for m, _ in subscriptions])
        cs = (C.c_int * max(ne, 1))(*[int(c) for _, c in subscriptions])
        cfg = PumpConfig()
        cfg.block_samples, cfg.fmt, cfg.scale, cfg.offset = int(block_samples), int(fmt), float(scale), float(offset)
        cfg.samp_rate = float(samp_rate)
        cfg.rings, cfg.ring_blocks, cfg.phase_s, cfg.written = rp, rb, ph, wr
        cfg.what, cfg.gain = (READ_IQ if what == "iq" else READ_FM), float(gain)
        cfg.read_members, cfg.read_chans, cfg.n_read = ms, cs, ne
        cfg.out_ring_samples, cfg.n_blocks, cfg.warm_blocks = int(out_ring_samples), int(n_blocks), int(warm_blocks)
        cfg.max_batch, cfg.cpu, cfg.start_delay_s = int(max_batch), int(cpu), float(start_delay_s)
        cfg.batch_window_s = float(batch_window_s)
        cfg.rt_priority = int(rt_priority)
        cfg.spin_us = int(spin_us)
        cfg.max_read = int(max_read)
        self.what = what
        self.n_entries = ne
        self.n_slots = max(ne, int(max_read), 1)
        self._cursors = [C.c_int64(0) for _ in range(self.n_slots)]
        self._what = [what] * self.n_slots
        for i, (_, c) in enumerate(subscriptions):
            if int(c) >= SRC_PFB_BIN0:                 # a bin of the member's fused discriminator ring: floats whatever `what` says
                self._what[i] = "fm"
        self._p = C.c_void_p()
        _check(lib().rcf_pump_start(group._g, C.byref(cfg), C.byref(self._p)))

    def stats(self):
        st = PumpStats()
        lib().rcf_pump_stats(self._p, C.byref(st))
        d = {k: getattr(st, k) for k, _ in PumpStats._fields_}
        if st.error:
            d["error_text"] = lib().rcf_last_error().decode("utf-8", "replace")
        return d

    def running(self):
        return bool(self.stats()["running"])

    def wait(self, timeout_s=60.0, poll_s=0.02):
        """until the pump has finished its n_blocks (or timeout) -> stats"""
        import time
        t_end = time.monotonic() + timeout_s
        while time.monotonic() < t_end:
            st = self.stats()
            if not st["running"]:
                return st
            time.sleep(poll_s)
        return self.stats()

    def subscribe(self, member, chan_id, what="iq", gain=1.0):
        """channel `chan_id` of member `member` from its next output on -> the slot (rcf_pump_subscribe)"""
        cur = C.c_int64(0)
        e = _check(lib().rcf_pump_subscribe(self._p, int(member), int(chan_id), READ_IQ if what == "iq" else READ_FM,
                                            float(gain), C.byref(cur)))
        self._cursors[e] = cur
        self._what[e] = what
        return e

    def subscribe_bin(self, member, bin_, gain=1.0):
        """bin `bin_` of member `member`'s fused discriminator ring (Frontend.pfb_fm_enable) from the bank's next frame on"""
        return self.subscribe(member, SRC_PFB_BIN0 + int(bin_), "fm", gain)

    def unsubscribe(self, entry):
        _check(lib().rcf_pump_unsubscribe(self._p, int(entry)))

    def written(self, entry):
        return _check(lib().rcf_pump_written(self._p, int(entry)))

    def read(self, entry, max_items=1 << 16):
        dt = np.complex64 if self._what[entry] == "iq" else np.float32
        out = np.empty(max_items, dtype=dt)
        n = _check(lib().rcf_pump_read(self._p, int(entry), C.byref(self._cursors[entry]),
                                       out.ctypes.data_as(C.c_void_p), int(max_items)))
        return out[:n].copy()

    def read_many_plan(self, entries, cap_each=1 << 13):
        """-> call() -> list of arrays (views into one buffer that the next call overwrites; None: no such slot): the new
        items of every listed slot with ONE native call (rcf_pump_read_many).  The call's arguments are built once."""
        n = len(entries)
        ents = (C.c_int * max(n, 1))(*[int(e) for e in entries])
        curs = (C.c_int64 * max(n, 1))(*[self._cursors[e].value for e in entries])
        counts = (C.c_int64 * max(n, 1))()
        out = np.empty((max(n, 1), cap_each), dtype=np.complex64)
        outf = out.view(np.float32)
        is_iq = [self._what[e] == "iq" for e in entries]
        fn, pp, op = lib().rcf_pump_read_many, self._p, out.ctypes.data_as(C.c_void_p)

        def call():
            _check(fn(pp, ents, curs, n, op, int(cap_each), counts))
            res = []
            for i in range(n):
                c = counts[i]
                self._cursors[entries[i]].value = curs[i]
                res.append(None if c < 0 else (out[i, :c] if is_iq[i] else outf[i, :c]))
            return res
        return call

    def stop(self):
        if self._p:
            lib().rcf_pump_stop(self._p)
            self._p = C.c_void_p()
            self._keep = []

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass
