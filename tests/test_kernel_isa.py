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
    is_mfma = [("v_mfma_scale" in ln) for ln in lines]
    for k, ln in enumerate(lines):
        # a stage read: a 16-byte LDS read with matrix instructions right behind it (the checkpoints read their statistics with
        # the ring drained; a wait there costs nothing)
        if ("ds_read_b128" in ln) and any(is_mfma[k:k + 150]):
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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_wide_band_kernel_has_one_branch_free_stage_loop_and_no_scratch(tmp_path):
    """pair_mfma_wide_kernel (ldp_pair_wide.hip): every body's stage loop exists in ONE form -- 64 MFMAs behind 24 LDS reads (2 x 4 rectangles), 48
    behind 20 (the diagonal tiles' 2 x 3), no branch, no accumulator copy -- and no scratch access anywhere inside a k-loop (a masked form made hipcc
    spill inside the loop: DESIGN.md 4.1e).  The SPARSE instantiation uses no scratch at all; the complete-data one, whose stages make the second
    half-stage's J fragments during the first (32 more registers, round 6), parks a dozen set-up values that only its checkpoints read."""
    src = os.path.join(REPO, "plink-ng_amd", "csrc", "ldp_pair_wide.hip")
    out = tmp_path / "wd.s"
    cp = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S",
                         src, "-o", str(out), "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cp.returncode == 0, cp.stdout[-2000:]
    # (the file also holds the measurement-only ablations pair_mfma_wide_kernel<1 | 2 | 4 | 7>; the product is <0>)
    usage = {}
    for chunk in cp.stdout.split("Function Name: ")[1:]:
        usage[chunk.split()[0]] = (int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", chunk).group(1)), int(re.search(r"VGPRs Spill: (\d+)", chunk).group(1)))
    assert usage, cp.stdout[-2000:]
    for name, (scr, spill) in usage.items():
        if "pair_mfma_wide_kernelILi0ELb0E" in name:
            assert (scr <= 64) and (spill <= 16), (name, scr, spill)   # (outside the k-loops: checked below)
        else:
            assert (scr, spill) == (0, 0), (name, scr, spill)
    assert set(int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", cp.stdout)) == {2}
    text = open(out).read()
    # <0, false>: complete data, the allele-count coding; <0, true>: the SPARSE instantiation (rows with a few missing calls: the +-2 coding,
    # an interval checkpoint and epilogue with FP64 interval arithmetic and a whole-wave recount around the SAME stage loop)
    for sparse in (False, True):
        m = re.search(r"^(_ZN3ldp\w*pair_mfma_wide_kernelILi0ELb%dE\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel" % int(sparse), text, re.S | re.M)
        assert m, "pair_mfma_wide_kernel<0, %s> not found" % sparse
        lines = m.group(2).splitlines()
        mf = [k for k, ln in enumerate(lines) if "v_mfma_scale_f32_32x32x64_f8f6f4" in ln]
        # the copies of the stage body: runs of matrix instructions less than 300 lines apart.  SPARSE: one (two half-stages of 32).  Complete data: the
        # 2 x 4 body (64) and the diagonal tiles' 2 x 3 body (48), picked per workgroup -- two whole bodies side by side, not two forms inside a loop
        runs = [[mf[0]]]
        for k in mf[1:]:
            if k - runs[-1][-1] < 300:
                runs[-1].append(k)
            else:
                runs.append([k])
        assert sorted(len(r) for r in runs) == ([64] if sparse else [48, 64])
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1)) <= 256
        labels = {mm.group(1): k for k, ln in enumerate(lines) for mm in [re.match(r"(\.LBB\d+_\d+):", ln)] if mm}
        loops = []
        for run in runs:
            body = lines[run[0]:run[-1] + 1]
            assert not any(("s_cbranch" in ln) or ("scratch_" in ln) or ("v_accvgpr" in ln) for ln in body)
            # the k-loop around the body: from the earliest label a branch behind the body jumps back to, to the last such branch
            back = [(labels[mm.group(1)], k) for k, ln in enumerate(lines) if run[-1] < k < run[-1] + 120
                    for mm in [re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", ln)] if mm and labels.get(mm.group(1), k) < run[0]]
            assert back, "no loop around the stage body"
            top, bottom = min(b[0] for b in back), max(b[1] for b in back)
            assert run[0] - top < 800
            assert not any("scratch_" in ln for ln in lines[top:bottom + 1]), "scratch access inside the k-loop"
            loops.append((top, bottom))
            window = lines[run[0] - 80:run[-1]]
            blocks = 6 if len(run) == 64 else 5   # row-blocks a wave expands per half-stage: 2 J + 4 V, or 2 J + 3 V
            if sparse:
                # the +-2 expansion: one v_bitop3_b32 per fragment dword and a shift for every second one -- three VALU per 16 samples, as below
                assert sum("v_bitop3_b32" in ln for ln in window) == 192
                assert sum("v_lshlrev_b32" in ln for ln in window) == 96
            else:
                # the allele-count expansion (ldp_mfma_device.h): 2 half-stages x row-blocks x 4 k-steps x 4 fragment dwords, one v_and each, and a
                # shift for every second one (the odd samples of a code dword) -- three VALU per 16 samples
                assert sum(("v_and_b32" in ln) and ("0x33333333" in ln) for ln in window) == 2 * blocks * 16
                assert sum("v_lshrrev_b32" in ln for ln in window) == 2 * blocks * 8
        for k, ln in enumerate(lines):
            if ("ds_read_b128" in ln) and any(top <= k <= bottom for top, bottom in loops):   # (a checkpoint has drained the ring anyway)
                before = [x for x in lines[max(0, k - 6):k] if not x.strip().startswith(";")]
                assert "vmcnt(0)" not in "\n".join(before[-3:]), "s_waitcnt vmcnt(0) in front of a stage read:\n" + "\n".join(lines[k - 6:k + 1])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_barrier_free_tile_kernel_polls_lds_counters_and_never_drains_its_ring(tmp_path):
    """pair_mfma_wide_async_kernel (ldp_pair_wide.hip, round 5's experiment): one copy of the 256-sample stage (32 MFMAs), the waves'
    counters read by hand-placed ds_read_b32 -- no flat access (a volatile pointer made hipcc take the generic address space, which counts
    against vmcnt AND lgkmcnt) and no `s_waitcnt vmcnt(0)` in front of a poll or a stage read (it would drain the four-stage ring) --, spins
    that sleep, a trap behind them, no s_barrier between the polls and the matrix instructions, no scratch."""
    src = os.path.join(REPO, "plink-ng_amd", "csrc", "ldp_pair_wide.hip")
    out = tmp_path / "wa.s"
    cp = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", str(out)],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cp.returncode == 0, cp.stdout[-2000:]
    m = re.search(r"^(_ZN3ldp\w*pair_mfma_wide_async_kernelILi0E\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", open(out).read(), re.S | re.M)
    assert m, "pair_mfma_wide_async_kernel<0> not found"
    lines = m.group(2).splitlines()
    mf = [k for k, ln in enumerate(lines) if "v_mfma_scale_f32_32x32x64_f8f6f4" in ln]
    assert len(mf) == 32
    assert not any(("flat_load" in ln) or ("flat_store" in ln) or ("scratch_" in ln) for ln in lines)
    polls = [k for k, ln in enumerate(lines) if "ds_read_b32" in ln]
    assert len(polls) >= 3 and any("s_sleep" in ln for ln in lines) and any("s_trap" in ln for ln in lines)
    for k in polls + [k for k, ln in enumerate(lines) if ("ds_read_b128" in ln) and (mf[0] - 60 < k < mf[-1])]:
        before = [x for x in lines[max(0, k - 6):k] if not x.strip().startswith(";")]
        assert "vmcnt(0)" not in "\n".join(before[-3:]), "the ring is drained in front of line %d:\n%s" % (k, "\n".join(lines[k - 6:k + 1]))
    # from the first poll in front of the stage to its last matrix instruction: no workgroup barrier
    first_poll = max(k for k in polls if k < mf[0])
    assert not any("s_barrier" in ln for ln in lines[first_poll:mf[-1] + 1])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_quarter_tile_kernel_has_one_stage_loop_without_scratch(tmp_path):
    """pair_mfma_tile4_kernel: ONE form of the stage loop -- two 256-sample stages, 2 products x 4 sums x 4 k-steps x 2
    = 64 MFMAs behind 12 LDS reads -- with no scratch access and no accumulator copy inside (three forms chosen per segment made
    hipcc spill an address inside the loop, whose reload drained the DMA ring every stage), two waves per SIMD."""
    src = os.path.join(REPO, "plink-ng_amd", "csrc", "ldp_pair_mfma.hip")
    out = tmp_path / "mf.s"
    cp = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S",
                         src, "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert cp.returncode == 0, cp.stdout[-2000:]
    text = open(out).read()
    m = re.search(r"^(_ZN3ldp22pair_mfma_tile4_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "kernel not found"
    lines = m.group(2).splitlines()
    assert re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)) and int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1)) <= 256
    labels = {mm.group(1): k for k, ln in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", ln)] if mm}
    loops = []
    for k, ln in enumerate(lines):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            body = lines[labels[mm.group(1)]:k]
            loops.append((sum("v_mfma_scale_f32_32x32x64_f8f6f4" in x for x in body), body))
    inner = min((lp for lp in loops if lp[0] > 0), key=lambda lp: len(lp[1]))
    assert inner[0] == 64
    assert not any(("scratch_" in x) or ("v_accvgpr" in x) for x in inner[1])
    assert sum("ds_read_b128" in x for x in inner[1]) == 12
    assert sum("v_mfma_scale_f32_32x32x64_f8f6f4" in x for x in lines) == 64   # no second copy of the stage anywhere
