"""Untimed GPU legs: PCIe-inclusive ingest, control plane."""
import time

import numpy as np

from .common import FS, NB, proto_taps

def end_to_end_leg(native, tile, device, B=1 << 24):
    """PCIe-inclusive ingest with the filterbank and the FM channels running: pinned host buffers handed to
    rcf_push_iq (cf32, 8 B/sample) and rcf_push_raw (u8, 2 B/sample); block n+1 is copied while block n runs."""
    fe = native.Frontend(FS, 0.0, device=device, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << 18)
    fe.pfb_open(NB, NB, proto_taps(native))
    out = {"block_samples": B}
    pin = native.PinnedArray(B, np.complex64)
    pin.array[:] = np.tile(tile, B // len(tile))
    for _ in range(2):
        fe.push(pin.array)
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        fe.push(pin.array)
    fe.sync()
    out["pinned_cf32_push_iq_Msps"] = 8 * B / (time.perf_counter() - t0) / 1e6
    pin.free()
    raw = native.PinnedArray(2 * B, np.uint8)
    raw.array[:] = np.tile((np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255)).astype(np.uint8),
                           B // len(tile))
    for _ in range(2):
        fe.push_raw(raw.array, native.FMT_U8, 1.0 / 128, 127.4)
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        fe.push_raw(raw.array, native.FMT_U8, 1.0 / 128, 127.4)
    fe.sync()
    out["pinned_u8_push_raw_Msps"] = 8 * B / (time.perf_counter() - t0) / 1e6
    raw.free()
    fe.close()
    out["note"] = "host -> HBM over PCIe Gen5 x16 (63 GB/s spec) + 256-bin PFB per block; never `value`"
    return out


def control_plane_leg(device):
    """100 x create / release through the reference's client (frontend_connector.py:242-251 times exactly this)."""
    import types
    from rcf import frontend_connector as FC, protocol, receiver

    class OneChannelizer:
        def get_channelizer_for_frequency(self, f):
            return ("127.0.0.1", 0)

    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=20000000)},
                                frontend_mode="xlat")
    tb = receiver.receiver(cfg, device=device)
    srv = protocol.FrontendServer(tb)
    fc = FC.frontend_connector("bench", OneChannelizer(), heartbeat=False,
                               transport_factory=lambda h, p: protocol.LoopbackTransport(srv))
    t_create, t_release = [], []
    for i in range(100):
        t0 = time.perf_counter()
        cid, port = fc.create_channel(12500, int(855e6 + 12500 * (i - 50)))
        t1 = time.perf_counter()
        assert cid, "create_channel failed"
        fc.release_channel()
        t2 = time.perf_counter()
        t_create.append(t1 - t0)
        t_release.append(t2 - t1)
    # and 100 distinct channels held at once (no idle reuse): what a busy trunked system asks for
    t0 = time.perf_counter()
    held = [tb.connect_channel(12500, int(855e6 + 12500 * (i - 50)))[0] for i in range(100)]
    t_hold = (time.perf_counter() - t0) / 100
    for b in held:
        tb.release_channel(b)
    tb.sweep_idle_channels(now=time.time() + 60)
    tb.close()
    return {"n": 100, "create_ms_median": sorted(t_create)[50] * 1e3, "create_ms_max": max(t_create) * 1e3,
            "release_ms_median": sorted(t_release)[50] * 1e3,
            "connect_channel_new_ms_mean": t_hold * 1e3,
            "note": "create = connect_channel (reuses an idle channel after the first, as receiver.py:311-319 does) "
                    "+ protocol; channel buffers come from the handle's slab pool"}
