"""The C-ABI library loads on a GPU-less box and exports every symbol include/rcf.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rcf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rcf import native
    lib = native.lib()
    syms = header_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(native.SYMBOLS) == syms          # the binding tracks the header


def test_no_torch_or_hip_types_in_signatures():
    text = open(os.path.join(ROOT, "include", "rcf.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "hipStream_t" not in code and "hip" not in code.lower().replace("rcf_ehip", "")
    assert "torch" not in code.lower() and "tensor" not in code.lower()


def test_host_entry_points_work_without_gpu():
    from rcf import native
    from oracle import grspec as G
    assert native.lib().rcf_version().startswith(b"rcf-mi355x")
    assert native.channel_params(2.4e6, 12500) == (96, 349)
    assert native.channel_params(20e6, 12500) == (800, 2909)
    with pytest.raises(native.RcfError) as e:
        native.channel_params(10666666, 12500)      # 426.5: rejected (documented divergence)
    assert e.value.code == native.RCF_ERANGE
    # ... unless the Python-2 reading of channel.py:31 is asked for (config_denver_massive_p25.py:20: 853 // 2)
    assert native.channel_params(10666666, 12500, native.DECIM_FLOOR) == (426, 1551)
    assert native.channel_params(20e6, 12500, native.DECIM_FLOOR) == (800, 2909)      # integral cases do not change
    D, taps = G.channel_params(10666666, 12500, py2_floor=True)
    assert (D, len(taps)) == (426, 1551)
    # product design code == oracle restatement (both restate firdes.low_pass_2 / windows)
    for fs, cr in ((2.4e6, 12500), (20e6, 12500), (2.4e6, 25000)):
        a = native.design_low_pass_2(1.0, fs, cr / 2, cr / 2, 20.0)
        b = G.low_pass_2(1.0, fs, cr / 2, cr / 2, 20.0)
        assert len(a) == len(b)
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-9)
    np.testing.assert_allclose(native.design_window(native.WIN_BLACKMAN_HARRIS, 16384),
                               G.blackman_harris(16384), rtol=0, atol=1.2e-7)
    assert native.peak_frequency(2004, 2.4e6, 16384, 855.05e6) == int(2004 * (2.4e6 / 16384) - 1.2e6 + 855.05e6)


def test_open_fails_loudly_without_device():
    from rcf import native
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.RcfError) as e:
        native.Frontend(2.4e6)
    assert e.value.code == native.RCF_EHIP
    assert "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("seed", range(6))
def test_product_peak_picker_equals_scipy(seed):
    """rcf_find_peaks (host C++) against the live oracle scipy.signal.find_peaks -- bit-exact."""
    from rcf import native, scan
    from oracle import peaks as P
    rng = np.random.default_rng(500 + seed)
    n = 16384
    x = rng.normal(100.0, 5.5, n)
    for _ in range(7):
        c = rng.integers(300, n - 300)
        w = rng.uniform(15, 120)
        x += rng.uniform(40, 90) * np.exp(-0.5 * ((np.arange(n) - c) / (w / 2.355)) ** 2)
    if seed % 2:
        x = np.round(x * 2) / 2                           # plateaus / ties
    x = (x - 130.0).astype(np.float32)
    l0, f0 = P.peak_detect_scipy(x, 2.4e6, 855e6)
    l1, f1 = scan.peak_detect(x, 2.4e6, 855e6)
    np.testing.assert_array_equal(l1, l0)
    assert f1 == f0
    _, mean, _ = native.find_peaks(x, 20.48, 204.8)
    assert mean == P.prologue(x, 2.4e6, n)[1]


def test_pinned_host_alloc_roundtrip_without_gpu():
    """rcf_host_alloc needs the HIP runtime, not a device: on a GPU-less box it returns NULL and sets the error
    text; with a device the numpy view is writable.  Either way no crash."""
    import numpy as np
    from rcf import native
    try:
        p = native.PinnedArray(1024, np.complex64)
    except native.RcfError as e:
        assert "pinned" in str(e)
        return
    p.array[:] = 1 + 2j
    assert p.array[5] == 1 + 2j
    p.free()
