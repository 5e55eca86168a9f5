"""ctypes binding of oracle/_build/librcf_oracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/grspec.py header for who may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librcf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rcf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class RotState(C.Structure):
    _fields_ = [("phase_re", C.c_float), ("phase_im", C.c_float), ("counter", C.c_uint32)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        L.ro_low_pass_2.argtypes = [C.c_double] * 5 + [C.c_int, fp, C.c_int]
        L.ro_low_pass_2.restype = C.c_int
        L.ro_window_f32.argtypes = [C.c_int, C.c_int, fp]
        L.ro_xlating_composite.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, fp, fp]
        L.ro_xlating_composite.restype = None
        L.ro_xlating_fir_ccc.argtypes = [fp, C.c_int64, C.c_int64, C.c_int, fp, C.c_int, fp,
                                         C.POINTER(RotState), fp, C.c_int]
        L.ro_xlating_fir_ccc.restype = None
        L.ro_quad_demod_cf.argtypes = [fp, C.c_int64, C.c_float, fp, fp]
        L.ro_quad_demod_cf.restype = None
        L.ro_channel_bank.argtypes = [fp, C.c_int64, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp,
                                      C.c_int, C.c_int]
        L.ro_channel_bank.restype = C.c_int
        L.ro_max_threads.restype = C.c_int
        ip = C.POINTER(C.c_int)
        L.ro_bank_bench.argtypes = [fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, ip, C.c_int,
                                    C.c_int64, fp, fp]
        L.ro_bank_bench.restype = C.c_double
        L.ro_read_bandwidth.argtypes = [C.c_int64, C.c_int, C.c_int, ip, fp]
        L.ro_read_bandwidth.restype = C.c_double
        L.ro_fir_summation.argtypes = [fp, C.c_int64, C.c_int, fp, C.c_int, C.c_int, fp]
        L.ro_fir_summation.restype = C.c_int
        L.ro_rotator_phases.argtypes = [fp, C.c_int64, C.c_int, fp]
        L.ro_rotator_phases.restype = None
        L.ro_scan_chain.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.ro_scan_chain.restype = C.c_int
        L.ro_peak_detect.argtypes = [fp, C.c_int64, C.c_double, C.c_double, C.c_double,
                                     C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_double)]
        L.ro_peak_detect.restype = C.c_int64
        L.ro_fast_atan2f.argtypes = [C.c_float, C.c_float]
        L.ro_fast_atan2f.restype = C.c_float
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def low_pass_2(gain, fs, fc, tw, att_db, wintype=0):
    L = lib()
    n = -L.ro_low_pass_2(gain, fs, fc, tw, att_db, wintype, None, 0)
    taps = np.empty(n, dtype=np.float32)
    assert L.ro_low_pass_2(gain, fs, fc, tw, att_db, wintype, _fp(taps), n) == n
    return taps


def window(wintype, n):
    w = np.empty(n, dtype=np.float32)
    lib().ro_window_f32(wintype, n, _fp(w))
    return w


def xlating_composite(taps, D, f0, fs):
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    ct = np.empty(len(taps), dtype=np.complex64)
    incr = np.empty(1, dtype=np.complex64)
    lib().ro_xlating_composite(_fp(taps), len(taps), D, f0, fs, _fp(ct.view(np.float32)),
                               _fp(incr.view(np.float32)))
    return ct, incr[0]


def channel_bank(x, D, ctaps, incr, gains=None, acc_double=True, nthreads=0):
    """ctaps: (C,T) complex64; incr: (C,) complex64 -> (y (C,n_out) complex64, fm (C,n_out) f32|None)"""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    ctaps = np.ascontiguousarray(ctaps, dtype=np.complex64)
    incr = np.ascontiguousarray(incr, dtype=np.complex64)
    Cn, T = ctaps.shape
    n_out = (len(x) - 1) // D + 1 if len(x) else 0
    y = np.empty((Cn, n_out), dtype=np.complex64)
    fm = None
    g = None
    if gains is not None:
        g = np.ascontiguousarray(gains, dtype=np.float32)
        fm = np.empty((Cn, n_out), dtype=np.float32)
    rc = lib().ro_channel_bank(_fp(x.view(np.float32)), len(x), D, T, Cn,
                               _fp(ctaps.view(np.float32)), _fp(incr.view(np.float32)),
                               _fp(g) if g is not None else None,
                               _fp(y.view(np.float32)), _fp(fm) if fm is not None else None,
                               1 if acc_double else 0, nthreads)
    assert rc == 0
    return y, fm


def quad_demod(x, gain):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty(len(x), dtype=np.float32)
    prev = np.zeros(2, dtype=np.float32)
    lib().ro_quad_demod_cf(_fp(x.view(np.float32)), len(x), gain, _fp(prev), _fp(out))
    return out


def scan_chain(x, N, n_frames, avg_len):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    assert len(x) >= N * n_frames
    out = np.empty(N, dtype=np.float32)
    assert lib().ro_scan_chain(_fp(x.view(np.float32)), N, n_frames, avg_len, _fp(out)) == 0
    return out


def peak_detect(spectrum, samp_rate, fft_width=None, cap=4096):
    s = np.ascontiguousarray(spectrum, dtype=np.float32)
    fft_width = len(s) if fft_width is None else fft_width
    hz = samp_rate / fft_width
    lines = np.empty(cap, dtype=np.int64)
    mean = C.c_double()
    n = lib().ro_peak_detect(_fp(s), len(s), 3000 / hz, 30000 / hz, 1.0,
                             lines.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(mean))
    return lines[:min(n, cap)].copy(), mean.value


def max_threads():
    return lib().ro_max_threads()


def physical_cores():
    """one logical CPU id per physical core this process may run on (sysfs thread_siblings_list); [] if unknown"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return []
    seen, out = set(), []
    for cpu in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                sib = f.read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            out.append(cpu)
    return out


def bank_bench(tile, passes, D, ctaps, incr, gains, n_threads, cpt, cpu_ids=None, tiled=False, tile_block=0,
               want_y0=False):
    """seconds of the timed region of ro_bank_bench (see rcf_oracle.c): n_threads x cpt channels, each over
    passes x len(tile) samples.  ctaps: (n_threads * cpt, T)"""
    tile = np.ascontiguousarray(tile, dtype=np.complex64)
    ctaps = np.ascontiguousarray(ctaps, dtype=np.complex64)
    incr = np.ascontiguousarray(incr, dtype=np.complex64)
    gains = np.ascontiguousarray(gains, dtype=np.float32)
    assert ctaps.shape[0] == n_threads * cpt == len(incr) == len(gains)
    ids = (C.c_int * n_threads)(*cpu_ids[:n_threads]) if cpu_ids else None
    chk = np.zeros(n_threads, dtype=np.float32)
    y0 = np.zeros(len(tile) // D, dtype=np.complex64) if want_y0 else None
    t = lib().ro_bank_bench(_fp(tile.view(np.float32)), len(tile), int(passes), int(D), ctaps.shape[1], int(n_threads),
                            int(cpt), _fp(ctaps.view(np.float32)), _fp(incr.view(np.float32)), _fp(gains), ids,
                            1 if tiled else 0, int(tile_block), _fp(chk),
                            _fp(y0.view(np.float32)) if y0 is not None else None)
    if t < 0:
        raise RuntimeError("ro_bank_bench failed (arguments / memory)")
    return (t, chk, y0) if want_y0 else (t, chk)


def read_bandwidth(bytes_per_thread, reps, n_threads, cpu_ids=None):
    ids = (C.c_int * n_threads)(*cpu_ids[:n_threads]) if cpu_ids else None
    sink = np.zeros(n_threads, dtype=np.float32)
    return lib().ro_read_bandwidth(int(bytes_per_thread), int(reps), int(n_threads), ids, _fp(sink))


SUM_8_LANES, SUM_SEQUENTIAL, SUM_PAIRWISE, SUM_16_LANES, SUM_FLOAT64 = 0, 1, 2, 3, 4


def fir_summation(x, D, ctaps, mode):
    """v[k] = sum_i ctaps[i] x[kD - i] with the dot product summed in `mode` (SUM_*): the orders a VOLK build may use"""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    ctaps = np.ascontiguousarray(ctaps, dtype=np.complex64)
    n_out = (len(x) - 1) // D + 1 if len(x) else 0
    v = np.empty(n_out, dtype=np.complex64)
    assert lib().ro_fir_summation(_fp(x.view(np.float32)), len(x), D, _fp(ctaps.view(np.float32)), len(ctaps), mode,
                                  _fp(v.view(np.float32))) == 0
    return v


def rotator_phases(incr, n, fma=False):
    """gr::blocks::rotator's phase for outputs 0 .. n-1, with or without FMA contraction of the complex multiply"""
    inc = np.array([incr], dtype=np.complex64)
    out = np.empty(n, dtype=np.complex64)
    lib().ro_rotator_phases(_fp(inc.view(np.float32)), n, 1 if fma else 0, _fp(out.view(np.float32)))
    return out
