"""Parity at BASELINE configs 3 and 5's sample count (N = 500,000), where the covariance numerators reach ~6e22 and the
FP64 predicate (plink2_ld.cc:1085-1090) works on 2^53-scale integers: `plink2-hip --indep-pairwise 500kb 0.2` against the
reference binary on a 20,020-variant x 500,000-sample slice of the benchmark generator (22 chromosomes, 300 bp spacing, the
window reaches ~1,667 variants: the same band density as config 3), complete data, 0.1 % and 5 % missing calls
(configs 3 / 5).  Files must be byte-identical.  Also: bench.py's own step under torch.distributed.run with one rank, so
that init_process_group("nccl") and the RCCL all_gather of the prune bitmask run on whatever GPU box runs the suite."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(REPO, "oracle", "_ref", "plink2")


@pytest.mark.gpu
@pytest.mark.parametrize("miss", [0.0, 0.001, 0.05])
def test_config3_sample_count_files_identical_to_reference(gpu_pkg, tmp_path, miss):
    import torch
    import bench
    if not os.path.exists(REF):
        pytest.fail("oracle/_ref/plink2 is missing (build() makes it where /root/reference exists; it travels with the snapshot)")
    cli = gpu_pkg.build_cli()
    n, m = 500000, 20020
    chr_idx, bps = bench.genome_layout(m, 1, 300)
    stride = (n + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    gpu_pkg.synth_genotypes_device(bench.SEED + 1, 0, m, n, miss, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    prefix = str(tmp_path / "c3")
    bench.write_plink1_fileset(prefix, host, n, chr_idx, bps)
    del host
    cores = os.cpu_count() or 1
    ref = subprocess.run([REF, "--bfile", "c3", "--indep-pairwise", "500kb", "0.2", "--threads", str(cores), "--out", "ref"], cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert ref.returncode == 0, ref.stdout[-1500:]
    hip = subprocess.run([cli, "--bfile", "c3", "--indep-pairwise", "500kb", "0.2", "--out", "hip", "--timing"], cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert hip.returncode == 0, hip.stdout[-1500:]
    # the kernels that produced these bytes: the 8 x 8 tiles for complete data AND for 0.1 % missing calls (their SPARSE
    # instantiation), quarter tiles for 5 %
    import re
    mm = re.search(r"pair launches by route: complete data (\d+) \| a few missing calls (\d+) \((\d+) on the 8 x 8 tiles\) \| missing calls (\d+) \((\d+) on quarter tiles\) \| (\d+) tiles planned",
                   hip.stdout)
    assert mm, hip.stdout[-1500:]
    complete, sparse, sparse_tiles, general, quarter, tiles = (int(x) for x in mm.groups())
    assert tiles > 0
    if miss == 0.0:
        assert complete > 0 and sparse == 0 and general == 0
    elif miss == 0.001:
        assert sparse > 0 and sparse_tiles == sparse and complete == 0 and general == 0
    else:
        assert general > 0 and quarter == general and complete == 0 and sparse == 0
    for ext in (".prune.in", ".prune.out"):
        a = open(os.path.join(str(tmp_path), "ref" + ext), "rb").read()
        b = open(os.path.join(str(tmp_path), "hip" + ext), "rb").read()
        assert a == b, "%s differs (%d vs %d bytes)" % (ext, len(a), len(b))
    removed = len(open(os.path.join(str(tmp_path), "ref.prune.out")).read().split())
    assert 0.1 * m < removed < 0.9 * m  # the slice really is pruned (LD chains of the generator)
    os.remove(prefix + ".bed")


@pytest.mark.gpu
def test_bench_step_under_torchrun_initialises_rccl(gpu_pkg, tmp_path):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "config2", "--variants", "60000", "--samples", "20000",
           "--no-cpu-baseline", "--no-legs"]
    cp = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-2000:]
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["variants_removed"] > 0
    # the same workload without a process group must prune the same variants (the exchange is an identity at one rank)
    cp2 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", "config2", "--variants", "60000", "--samples", "20000",
                          "--no-cpu-baseline", "--no-legs"], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp2.returncode == 0, cp2.stderr[-2000:]
    j2 = json.loads([ln for ln in cp2.stdout.splitlines() if ln.startswith("{")][-1])
    assert j2["config"]["variants_removed"] == j["config"]["variants_removed"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_n_ranks_on_one_device(gpu_pkg, tmp_path, world):
    """bench.py's own N-rank line on a one-GPU box (LDP_BENCH_ALIAS_DEVICES=1: the ranks share the device, gloo carries the exchange): the
    rank / LPT shard / all-gather code of `--gpus N` runs under torch.distributed.run, the line is labelled a model, and the stitched
    prune set has as many variants as the single-rank run over the same total workload (strong scaling: the same genome)."""
    env = dict(os.environ, LDP_BENCH_ALIAS_DEVICES="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--steps", "2", "--warmup", "1", "--workload", "config2", "--variants", "90000", "--samples", "20000", "--strong", "--no-cpu-baseline", "--no-legs"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29590 + world),
           os.path.join(REPO, "bench.py"), "--gpus", str(world)] + common
    cp = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-2000:]
    j = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == world and j["scaling"] == "strong" and j["value"] > 0 and j["data"].startswith("model (LDP_BENCH_ALIAS_DEVICES")
    cp1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp1.returncode == 0, cp1.stderr[-2000:]
    j1 = json.loads([ln for ln in cp1.stdout.splitlines() if ln.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j1["config"]["variants_removed"] == j["config"]["variants_removed"] > 0
    assert j["config"]["candidate_pairs_total"] == j1["config"]["candidate_pairs_total"] and len(j["config"]["candidate_pairs_per_rank"]) == world
    # the N-rank line checks itself: world size, ranks seen by the collective backend, and the stitched bitmap against ONE engine on chr20-22
    sc = j["multi_rank_selfcheck"]
    assert sc["world_size"] == world and sc["world_size_is_n_gpus"] and sc["rccl_ranks_seen"] == world and len(sc["ranks"]) == world
    assert sc["three_chromosome_subset"]["identical_to_single_engine"] is True and sc["three_chromosome_subset"]["variants"] > 1000
    assert sc["devices_distinct"] is False   # (this box: the ranks share its one device, and the line says so)


@pytest.mark.gpu
def test_bench_n_ranks_time_plink2_hip_end_to_end(gpu_pkg):
    """`bench.py --gpus N` also reports the wall-clock half of the metric at N GPUs (round 6): rank 0 writes a small fileset, runs `plink2-hip --gpus N`
    (every engine fed by a thread of its own) and `--gpus 1` on it, and the line carries both walls, the phase split and whether the outputs agree --
    as flat scalars inside `roofline` too, where the driver's record keeps them."""
    world = 2
    env = dict(os.environ, LDP_BENCH_ALIAS_DEVICES="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = ["--steps", "1", "--warmup", "1", "--workload", "config2", "--variants", "40000", "--samples", "100032", "--strong", "--no-cpu-baseline", "--no-legs",
            "--e2e-variants", "3000"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", "29617",
           os.path.join(REPO, "bench.py"), "--gpus", str(world)] + args
    cp = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-2000:]
    j = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    e = j["e2e_n_gpus"]
    assert "error" not in e, e
    assert e["gpus"] == world and e["plink2_hip"]["rc"] == 0 and e["plink2_hip_one_gpu"]["rc"] == 0 and e["identical_to_one_gpu"] is True
    assert any("engines fed concurrently" in ln for ln in e["plink2_hip"]["timing_lines"]), e["plink2_hip"]["timing_lines"]
    r = j["roofline"]
    assert r["e2e_plink2_hip_gpus"] == world and r["e2e_plink2_hip_gpus_wall_s"] > 0 and r["e2e_plink2_hip_gpus_identical_to_one_gpu"] is True


@pytest.mark.gpu
def test_bench_share_that_does_not_fit_hbm(gpu_pkg):
    """bench.py's non-resident mode (a rank's share of config 3 at N = 2 / 4 exceeds HBM): forced here with a small HBM limit.
    One engine per chromosome, rows copied from one resident chromosome's worth of generated rows inside the step; every chromosome
    then prunes like the master chromosome does on its own."""
    env = dict(os.environ, LDP_BENCH_HBM_GB="6.2")
    args = ["--workload", "config3", "--samples", "20000", "--variants", "44000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs"]
    cp = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-2000:]
    j = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["config"]["resident"] is False and j["value"] > 0
    assert j["data"].startswith("model")   # a share that does not fit is a model of the workload and the line says so
    assert 0.1 * 44000 < j["config"]["variants_removed"] < 0.9 * 44000
    cp2 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert cp2.returncode == 0, cp2.stderr[-2000:]
    j2 = json.loads([ln for ln in cp2.stdout.splitlines() if ln.startswith("{")][-1])
    assert j2["config"]["resident"] is True and j2["config"]["candidate_pairs_total"] == j["config"]["candidate_pairs_total"]
    assert j2["data"] == "synthetic" and j2["scaling"] == "weak"
