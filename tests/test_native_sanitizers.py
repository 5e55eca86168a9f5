"""SURVEY.md 5 "sanitizers": the CPU-visible part of librcf's host layer (filter designs, Parks-McClellan, peak
picker -- everything that runs without a device) built with AddressSanitizer + UBSan and with ThreadSanitizer and
hammered from eight threads (tests/native/host_stress.cpp).  The GPU-side counterpart -- the whole C ABI under
ASan over the channel-churn tests -- is tools/asan_gpu.sh (needs the device; its log is profiles/r02_asan_gpu.txt)."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_layer_under_sanitizers(kind):
    subprocess.check_call(["make", "-C", HERE, "-s", kind])
    exe = os.path.join(HERE, "_build", "host_stress_%s" % kind)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0, out
    assert "identical to the single-threaded results" in out
    assert "Sanitizer" not in out, out
