"""The discriminator kernels' gr::fast_atan2f (radiocapture-rf_amd/csrc/fast_atan2f_gr.hpp) is written as selects, not as GNU
Radio's nested branches (gr-runtime/lib/math/fast_atan2f.cc; restated branch for branch in oracle/rcf_oracle.c:
ro_fast_atan2f): left as branches, the device compiler ran a division on both sides of `|y| < |x|` and both octant arms
under execution masks.  The header has no HIP types, so the SAME SOURCE is compiled here for the host (g++,
-ffp-contract=off) and held bit for bit to the oracle over ~24 million arguments: every octant, signed zeros, equal
magnitudes, denormals, the TAN_MAP_RES edge, every table interval edge (tests/native/device_atan_check.cpp)."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_device_fast_atan2f_source_equals_oracle_bit_for_bit():
    out = os.path.join(HERE, "native", "_build")
    os.makedirs(out, exist_ok=True)
    obj, exe = os.path.join(out, "rcf_oracle_for_atan.o"), os.path.join(out, "device_atan_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-math-errno", "-fopenmp", "-c",
                           os.path.join(ROOT, "oracle", "rcf_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           os.path.join(HERE, "native", "device_atan_check.cpp"), obj, "-o", exe, "-lm", "-fopenmp"])
    p = subprocess.run([exe, "20000000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = p.stdout.decode()
    m = re.search(r"checked (\d+) mismatches (\d+)", text)
    assert p.returncode == 0 and m, text
    assert int(m.group(1)) > 20000000 and int(m.group(2)) == 0, text


def test_every_discriminator_kernel_uses_the_shared_header():
    """No second copy of the function in the kernel sources: fir.hip / pfb.hip (through fir_small.hpp), tapfin.hip and audio.hip
    all include fast_atan2f_gr.hpp."""
    csrc = os.path.join(ROOT, "radiocapture-rf_amd", "csrc")
    defs = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            text = open(os.path.join(csrc, f)).read()
            if re.search(r"float\s+fast_atan2f_gr\s*\(", text):
                defs.append(f)
    assert defs == ["fast_atan2f_gr.hpp"], defs
    assert '#include "fast_atan2f_gr.hpp"' in open(os.path.join(csrc, "fir_small.hpp")).read()
    assert '#include "fast_atan2f_gr.hpp"' in open(os.path.join(csrc, "audio.hip")).read()
    assert '#include "fast_atan2f_gr.hpp"' in open(os.path.join(csrc, "tapfin.hip")).read()
