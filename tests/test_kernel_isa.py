"""Compile-time properties of the matrix-pipe pair kernel that its speed depends on and that hipcc silently loses
(DESIGN.md 4.1b): no scratch (a spill reload's vmcnt wait drains the LDS-DMA ring), and no `s_waitcnt vmcnt(0)` in
front of the k-loop's LDS reads (hipcc adds one as soon as it cannot prove the DMA in flight does not alias them).
Needs hipcc only (it cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_pair_mfma_kernel_keeps_its_dma_ring_running(tmp_path):
    src = os.path.join(REPO, "plink-ng_amd", "csrc", "ldp_pair_mfma.hip")
    out = tmp_path / "mf.s"
    cp = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S",
                         src, "-o", str(out), "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cp.returncode == 0, cp.stdout[-2000:]
    kernels = re.findall(r"Function Name: (\S*pair_mfma_kernel\S*)", cp.stdout)
    assert len(kernels) >= 2
    occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", cp.stdout)]
    assert occ and min(occ) >= 2
    lines = open(out).read().splitlines()
    n_reads = 0
    for k, ln in enumerate(lines):
        if "ds_read_b128" in ln:
            n_reads += 1
            before = [x for x in lines[max(0, k - 6):k] if not x.strip().startswith(";")]
            window = "\n".join(before[-3:])
            assert "vmcnt(0)" not in window, "s_waitcnt vmcnt(0) in front of a stage read:\n" + "\n".join(lines[k - 6:k + 1])
    assert n_reads >= 20
    # scratch traffic (register shuffles between the two forms of the stage loop, at checkpoints) never inside a stage: no
    # scratch instruction between two MFMAs that are less than 300 lines apart
    mf = [k for k, ln in enumerate(lines) if "v_mfma_scale" in ln]
    for a, b in zip(mf, mf[1:]):
        if b - a < 300:
            assert not any("scratch_" in ln for ln in lines[a:b]), "scratch access inside a stage near line %d" % a
    assert "v_mfma_scale_f32_32x32x64_f8f6f4" in "\n".join(lines)
