"""Host logic of the native data plane (rcf/dataplane.py) that needs no GPU: how a clock-paced ring is cut into blocks, the
producer side of a counter-fed source ring, what the host-information record holds; and that a daemon whose front-ends
are not native ones (or whose sources are receiver_split2 halves) takes the Python data plane instead of failing."""
import numpy as np
import pytest

from rcf import dataplane, hostinfo


@pytest.mark.parametrize("tile,fs,ms", [(1 << 20, 2.4e6, 20.0), (1 << 20, 20e6, 20.0), (1 << 22, 20e6, 20.0),
                                        (1000000, 2.4e6, 20.0), (48000 * 7, 2.4e6, 20.0), (17, 1e3, 5.0)])
def test_block_of_a_clock_paced_ring_is_a_whole_fraction_of_the_tile(tile, fs, ms):
    b = dataplane.block_for(tile, fs, ms)
    assert b >= 1 and tile % b == 0
    assert b <= max(1, round(fs * ms * 1e-3))                    # never longer than asked for ...
    # ... and the longest such fraction: the next larger divisor of the tile would be too long
    larger = [d for d in range(b + 1, min(tile, int(fs * ms * 1e-3)) + 1) if tile % d == 0] if tile <= 1 << 20 else []
    assert not larger, larger[:3]


def test_ring_writer_counts_whole_blocks_and_wraps():
    ring = np.zeros(4 * 10, dtype=np.int16)
    w = dataplane.RingWriter(ring, block_items=10, ring_blocks=4)
    data = np.arange(1, 200, dtype=np.int16)
    w.write(data[:7])
    assert int(w.counter[0]) == 0                                 # block 0 is not whole yet
    w.write(data[7:25])
    assert int(w.counter[0]) == 2 and np.array_equal(ring[:25], data[:25])
    w.write(data[25:65])                                          # wraps: blocks 4, 5 land in slots 0, 1
    assert int(w.counter[0]) == 6
    assert np.array_equal(ring[:20], data[40:60])
    assert np.array_equal(ring[20:25], data[60:65]) and np.array_equal(ring[25:40], data[25:40])   # block 6 is half written
    w.write(data[65:70])
    assert int(w.counter[0]) == 7 and np.array_equal(ring[20:30], data[60:70])


def test_host_record_fields():
    m = hostinfo.CpuMeter()
    x = sum(i * i for i in range(200000))                         # some CPU time
    out = m.sample()
    assert x > 0 and out["rcf_host_process_cores"] >= 0 and out["rcf_host_cpus_allowed"] >= 1
    c = hostinfo.cgroup_cpu()
    if "quota_cores" in c and c["quota_cores"] is not None:
        assert c["quota_cores"] > 0 and out["rcf_host_cpu_quota_cores"] == c["quota_cores"]
    if "throttled_ms" in c:
        assert out["rcf_host_throttled_ms"] >= 0


def test_daemon_falls_back_to_the_python_data_plane_without_native_front_ends(tmp_path):
    from test_daemon import OracleFrontend
    from rcf import frontend

    class Cfg:
        receiver_split2 = False
        frontend_mode = "xlat"
        sources = {0: {"type": "synthetic", "center_freq": 100e6, "samp_rate": 250000.0, "tile_samples": 1 << 14}}

    d = frontend.Daemon(Cfg(), transport="tcp", registry="none", bind="127.0.0.1", start_sources=False,
                        frontend_factory=OracleFrontend)
    try:
        assert d.dataplane == "python"
        with pytest.raises(ValueError):
            frontend.Daemon(Cfg(), transport="tcp", registry="none", bind="127.0.0.1", start_sources=False, dataplane="pump",
                            frontend_factory=OracleFrontend)
    finally:
        d.close()
