"""rcf -- MI355X-native channelizer / NBFM discriminator / FFT peak scanner behind the
radiocapture-rf front-end API.  Compute lives in librcf.so (hand-written HIP for gfx950, C ABI in
include/rcf.h); this package is the ctypes binding plus the Python mirror of the reference's
rc_frontend interface (receiver / channel / frontend_connector / redis_channel_publisher)."""
from . import native  # noqa: F401

__all__ = ["native"]
