"""tools/tile_shape_probe.hip -- the tile kernel's stage loop by itself, the in-run ceiling of bench.py (roofline.stage_loop_probe_frac_of_peak).  The probe shares
the kernels' device headers (LDS layout, FP4 expansion, MFMA wrapper), so a change there that breaks it must show: every shape's checksum -- the sum of G over all
pairs of its plan, accumulated on the matrix pipe -- against the one a plain kernel sums from the same rows."""
import json
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(REPO, "tools", "_bin", "tile_shape_probe")
SRC = os.path.join(REPO, "tools", "tile_shape_probe.hip")


def build_probe():
    if os.path.exists(PROBE) and os.path.getmtime(PROBE) >= os.path.getmtime(SRC):
        return
    os.makedirs(os.path.dirname(PROBE), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(REPO, "plink-ng_amd", "csrc"), SRC, "-o", PROBE])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_probe_builds_for_gfx950():
    build_probe()
    assert os.access(PROBE, os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["2x4", "2x4pf", "2x4pfh", "2x4pf2", "2x4r4", "2x4p2", "4x4", "4x4pf", "2x6", "4x6"])
def test_every_shape_sums_what_a_plain_kernel_sums(gpu_pkg, shape):
    build_probe()
    # 10,240 samples (20 stages), 16 J tiles, 48 row-blocks of reach: 6,144 block products, a few milliseconds
    cp = subprocess.run([PROBE, shape, "0.05", "10240", "16", "48"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert cp.returncode == 0, cp.stdout[-1500:]
    d = json.loads(cp.stdout.strip().splitlines()[-1])
    assert d["shape"] == shape and d["checksum_ok"] is True and d["checksum"] == d["checksum_plain_kernel"] > 0
    assert d["block_products"] == 16 * 8 * 48
