#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the MI355X-native --indep-pairwise hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 50,000 samples x 1,000,000 biallelic variants per GPU, 22
autosomes with variant counts proportional to GRCh38 lengths at uniform 2,875 bp spacing,
`--indep-pairwise 200kb 0.5`.  Weak scaling: with N GPUs the variant table is N such genomes
(22*N chromosomes); subcontigs are LPT-sharded over ranks, there is no data-path collective, and the
prune bitmask is exchanged once per step with an RCCL all_gather of the per-rank segments.

A "step" is one pass of the hot path over the HBM-resident packed 2-bit genotype matrix:
bit-plane split + per-variant aggregates + allele counts (prepare_kernel), the banded pair-statistics /
prune-predicate kernel (pair_tiles_kernel), the host replay of the greedy scan, and the bitmask exchange.
`value` = candidate variant pairs decided per second over all ranks.

One JSON line is printed by rank 0.  `roofline` describes pair_tiles_kernel<false> (the dominant kernel):
achieved = candidate pairs per launch x N/2 algorithmic bytes / mean launch time (HIP events on the
engine's stream).  `cpu_baseline` (rank 0, N=1 only) times the REFERENCE plink2 binary (oracle/_ref/plink2)
on a bounded sample of the same workload on this host's cores, and checks that its prune set is
identical to the HIP path's on that sample.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GRCH38_MB = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28,
             114.36, 107.04, 101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82]
SEED = 20260925 + 2
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def genome_layout(variants_per_genome, genomes, spacing):
    """chr_idx / bp arrays for `genomes` copies of a 22-autosome genome."""
    tot = sum(GRCH38_MB)
    counts = [int(variants_per_genome * mb / tot) for mb in GRCH38_MB]
    counts[0] += variants_per_genome - sum(counts)
    chr_idx = np.empty(variants_per_genome * genomes, dtype=np.uint32)
    bps = np.empty(variants_per_genome * genomes, dtype=np.uint32)
    pos = 0
    for g in range(genomes):
        for c, n in enumerate(counts):
            chr_idx[pos:pos + n] = g * 22 + c
            bps[pos:pos + n] = 10000 + spacing * np.arange(n, dtype=np.uint32)
            pos += n
    return chr_idx, bps


def cpu_baseline(pkg, torch, args, founder_ct, spacing, window_bp, r2):
    """Reference plink2 (all host cores) on a bounded sample of the same generator; also a parity check."""
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    if not (os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK)):
        return {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/plink2 not built"}
    if "avx2" not in open("/proc/cpuinfo").read():
        return {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference", "sample": "host CPU lacks AVX2"}
    m = args.cpu_sample_variants
    chr_idx, bps = genome_layout(m, 1, spacing)
    stride = (founder_ct + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(SEED, 0, m, founder_ct, 0.0, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    # HIP path on the sample
    eng = pkg.LdPruneEngine(founder_ct, window_bp, 1, True, r2, device=torch.cuda.current_device())
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_device(0, m, buf.data_ptr(), stride, pkg.LDP_GENO_REF)
    removed_hip = eng.run()
    cand = eng.counters()["candidate_pairs"]
    eng.close()
    # the sample as PLINK 1 files (.bed: 00 hom-ALT, 01 missing, 10 het, 11 hom-REF)
    host = buf.cpu().numpy()
    del buf
    lut = np.zeros(256, dtype=np.uint8)
    conv = [3, 2, 0, 1]
    for b in range(256):
        lut[b] = sum(conv[(b >> (2 * k)) & 3] << (2 * k) for k in range(4))
    tmp = tempfile.mkdtemp(prefix="ldbench_")
    try:
        prefix = os.path.join(tmp, "sample")
        with open(prefix + ".bed", "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            bed = lut[host]
            pad = (4 - founder_ct % 4) % 4
            if pad:
                bed[:, -1] &= np.uint8((1 << (2 * (4 - pad))) - 1)  # keep trailing bits zero
            f.write(bed.tobytes())
        with open(prefix + ".bim", "w") as f:
            f.write("".join("%d\tsnp%d\t0\t%d\tC\tA\n" % (chr_idx[i] + 1, i, bps[i]) for i in range(m)))
        with open(prefix + ".fam", "w") as f:
            f.write("".join("s%d s%d 0 0 2 -9\n" % (s, s) for s in range(founder_ct)))
        cores = os.cpu_count() or 1
        cmd = [ref_bin, "--bfile", "sample", "--indep-pairwise", "%gkb" % (window_bp / 1000.0), repr(r2), "--threads", str(cores),
               "--out", "ref"]
        t0 = time.perf_counter()
        cp = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
        wall = time.perf_counter() - t0
        if cp.returncode != 0:
            return {"value": None, "unit": "variant-pairs/s", "cores": cores, "kind": "reference", "sample": "reference failed: " + cp.stdout[-300:]}
        removed_ids = set(ln.strip() for ln in open(os.path.join(tmp, "ref.prune.out")) if ln.strip())
        removed_ref = np.array([("snp%d" % i) in removed_ids for i in range(m)])
        threads_line = [ln for ln in cp.stdout.splitlines() if "compute thread" in ln]
        cli = {}
        cli_bin = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
        if args.cli_compare and os.path.exists(cli_bin):
            # the process-level drop-in on the same files (file mapping + H2D + kernels + replay + writer)
            t1 = time.perf_counter()
            cc = subprocess.run([cli_bin, "--bfile", "sample", "--indep-pairwise", "%gkb" % (window_bp / 1000.0), repr(r2), "--out", "hip"],
                                cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
            cli_wall = time.perf_counter() - t1
            same = (cc.returncode == 0 and open(os.path.join(tmp, "hip.prune.out")).read() == open(os.path.join(tmp, "ref.prune.out")).read()
                    and open(os.path.join(tmp, "hip.prune.in")).read() == open(os.path.join(tmp, "ref.prune.in")).read())
            cli = {"plink2_hip_wall_s": cli_wall, "plink2_hip_files_identical": bool(same), "plink2_hip_rc": cc.returncode}
        mt = re.search(r"\((\d+) compute thread", cp.stdout)
        used = (int(mt.group(1)) + 1) if mt else cores  # LD compute threads + the decode/main thread (plink2_ld.cc:2599-2604)
        return {**cli, "value": cand / wall, "unit": "variant-pairs/s", "cores": used, "threads_requested": cores, "kind": "reference",
                "sample": "%d variants x %d samples of the same generator (22 chromosomes, %d bp spacing, %d candidate pairs), "
                          "reference plink2 AVX2 end-to-end wall %.2f s incl. file load + freq pass; %s" %
                          (m, founder_ct, spacing, cand, wall, (threads_line[-1].split(":")[0].strip() if threads_line else "")),
                "wall_s": wall, "prune_set_identical_to_hip": bool(np.array_equal(removed_ref, removed_hip)),
                "removed": int(removed_ref.sum())}
    finally:
        subprocess.call(["rm", "-rf", tmp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=1000000, help="variants per GPU")
    ap.add_argument("--window-kb", type=float, default=200.0)
    ap.add_argument("--r2", type=float, default=0.5)
    ap.add_argument("--missing-rate", type=float, default=0.0)
    ap.add_argument("--spacing", type=int, default=2875, help="bp between consecutive variants")
    ap.add_argument("--cpu-sample-variants", type=int, default=440000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resident-planes", action="store_true",
                    help="for shapes whose 2-bit input and bit-planes do not fit HBM together (config 3: 156 GB each): convert once, "
                         "chunk by chunk, outside the timed region; a step is then the pair kernel + replay only (said so in config.workload)")
    ap.add_argument("--cli-compare", action="store_true", help="also time plink2-hip end-to-end on the CPU-baseline sample files")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the hot path)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the RCCL path is used even for a single rank, so the exchange code is
    # exercised on a 1-GPU box as well
    use_dist = (world > 1) or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    founder_ct = args.samples
    spacing = args.spacing
    window_bp = pkg.kb_window(args.window_kb)
    chr_idx, bps = genome_layout(args.variants, world, spacing)
    m_total = len(chr_idx)

    eng = pkg.LdPruneEngine(founder_ct, window_bp, 1, True, args.r2, device=local_rank)
    eng.set_variants(chr_idx, bps)
    subs = eng.subcontigs()
    owner = eng.set_shard(rank, world) if world > 1 else np.zeros(len(subs), dtype=np.uint32)
    owned = [(ln, first) for (ln, first), o in zip(subs, owner) if o == rank]
    local_ct = sum(ln for ln, _ in owned)

    # synthetic REF-coded genotypes of the owned subcontigs, resident in HBM before timing starts
    stride = (founder_ct + 3) // 4
    seg = []
    if args.resident_planes:
        chunk_rows = max(1, (8 << 30) // stride)
        geno = torch.empty((chunk_rows, stride), dtype=torch.uint8, device="cuda")
        for ln, first in owned:
            for c0 in range(0, ln, chunk_rows):
                cnt = min(chunk_rows, ln - c0)
                pkg.synth_genotypes_device(SEED, first + c0, cnt, founder_ct, args.missing_rate, geno.data_ptr(), stride)
                torch.cuda.synchronize()  # the generator runs on the null stream, the engine on its own: order them
                eng.load_genotypes_device(first + c0, cnt, geno.data_ptr(), stride, pkg.LDP_GENO_REF)
                torch.cuda.synchronize()  # ... and the chunk buffer is reused
        del geno
        torch.cuda.empty_cache()
    else:
        geno = torch.empty((max(local_ct, 1), stride), dtype=torch.uint8, device="cuda")
        off = 0
        for ln, first in owned:
            pkg.synth_genotypes_device(SEED, first, ln, founder_ct, args.missing_rate, geno.data_ptr() + off * stride, stride)
            seg.append((first, ln, off))
            off += ln
    torch.cuda.synchronize()

    import importlib
    distmod = importlib.import_module("plink_ng_amd.dist")

    def step():
        for first, ln, o in seg:
            eng.load_genotypes_device(first, ln, geno.data_ptr() + o * stride, stride, pkg.LDP_GENO_REF)
        bm = eng.run_bitmap()  # uint64 words over all variants; only this rank's bits are set
        if use_dist:
            # the one exchange step: all_gather of the per-rank removed bitmaps (RCCL over xGMI), OR-ed on the device
            return distmod.allgather_bitmaps(bm, world, device="cuda")
        return bm

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kernel_ms, prep_ms, replay_ms, mfma_ms = [], [], [], []
    removed = None
    for _ in range(args.steps):
        removed = step()
        c = eng.counters()
        kernel_ms.append(c["ms_pair_kernel"])
        mfma_ms.append(c["ms_pair_mfma"])
        prep_ms.append(c["ms_prepare"])
        replay_ms.append(c["ms_replay"])
    sync()
    elapsed = time.perf_counter() - t0
    ctr = eng.counters()
    removed = distmod.bitmap_to_mask(removed.cpu().numpy() if use_dist else removed, m_total)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        pairs_t = torch.tensor([ctr["candidate_pairs"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(pairs_t, op=dist.ReduceOp.SUM)
        total_pairs = int(pairs_t.item())
    else:
        total_pairs = ctr["candidate_pairs"]

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
        value = total_pairs * args.steps / elapsed
        # the pair kernel runs as a few launches per step (groups of J-tiles, back to back on one stream):
        # kms = their summed duration per step, HIP events around every launch on the stream it is launched on
        kms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
        launches = max(int(ctr["pair_kernel_launches"]), 1)
        alg_bytes_per_pair = founder_ct / 2.0
        achieved = (ctr["candidate_pairs"] * alg_bytes_per_pair / (kms * 1e-3)) / 1e9 if kms > 0 else 0.0
        # integer-VALU side of the same kernel: 4 lane-ops per pair per 32 samples (and, bitop3, 2 x bcnt) on the pair
        # slots it really walks (early termination skips the rest), against the measured issue ceiling of exactly
        # this op mix (tools/ubench_valu.hip, profiles/r01_ubench_valu.txt: 4.118e13 lane-ops/s)
        plane_dwords = (founder_ct + 31) // 32
        skipped = (ctr["early_exit_unit_chunks"] / ctr["tile_unit_chunks"]) if ctr["tile_unit_chunks"] else 0.0
        # With missing calls (--missing-rate > 0: every row has some, so every tile takes pair_tiles_kernel<true>) the
        # mix is 7 bcnt + ~7.5 and/or/bitop3 per pair-dword (ISA of the built kernel), ceiling 4.229e13 by the same ubench
        general = args.missing_rate > 0
        ops_per_pair_dword = 14.5 if general else 4.0
        executed_lane_ops = ctr["computed_pairs"] * plane_dwords * ops_per_pair_dword * (1.0 - skipped)
        valu_mix_peak = 4.229e13 if general else 4.118e13
        traffic = None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("samples") == founder_ct and tj.get("variants") == args.variants and tj.get("window_kb") == args.window_kb:
                    traffic = tj.get("hbm_bytes_per_step", tj.get("hbm_bytes_per_launch"))
            except Exception:
                traffic = None
        out = {
            "metric": "variant-pairs/s (--indep-pairwise, whole job)", "value": value, "unit": "variant-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 popcount + f64 predicate",
            "data": "synthetic",
            "config": {"workload": "synthetic %d samples x %d biallelic variants per GPU, 22 autosomes/GPU at %d bp spacing, "
                                   "--indep-pairwise %gkb %g, missing rate %g, subcontig-sharded%s" %
                                   (founder_ct, args.variants, spacing, args.window_kb, args.r2, args.missing_rate,
                                    "; bit-planes resident, conversion outside the timed step" if args.resident_planes else ""),
                       "samples": founder_ct, "variants_per_gpu": args.variants, "window_kb": args.window_kb, "r2": args.r2,
                       "candidate_pairs_per_gpu": ctr["candidate_pairs"], "computed_pair_slots_per_gpu": ctr["computed_pairs"],
                       "pairs_above_threshold_per_gpu": ctr["pred_true"], "above_threshold_pairs_consumed_by_replay_per_gpu": ctr["replay_pairs"],
                       "variants_removed": int(removed.sum()), "variants_total": m_total},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "pair_tiles_kernel<%s>" % ("true" if args.missing_rate > 0 else "false"),
                         "kernel_ms_per_launch": kms / launches, "launches_per_step": launches, "kernel_ms_per_step": kms,
                         "algorithmic_bytes_per_pair": alg_bytes_per_pair,
                         "note": "achieved = algorithmic stream rate (candidate pairs x N/2 B / summed kernel time; traffic is per step, "
                                 "too); LDS/register tiling and early termination make it exceed physical HBM traffic, the kernel is "
                                 "integer-VALU bound: valu_frac = executed and/bitop3/bcnt lane-ops per second over the measured "
                                 "ceiling of that op mix",
                         "valu_lane_ops_per_pair_dword": ops_per_pair_dword, "valu_lane_ops_per_s": (executed_lane_ops / (kms * 1e-3)) if kms > 0 else 0.0, "valu_mix_peak": valu_mix_peak,
                         "valu_frac": (executed_lane_ops / (kms * 1e-3)) / valu_mix_peak if kms > 0 else 0.0},
            "stage_ms": {"prepare_kernel": float(np.mean(prep_ms)), "pair_kernel": kms, "pair_mfma_kernel": float(np.mean(mfma_ms)), "pair_mfma_general_kernel": ctr["ms_pair_mfma_general"],
                         "host_replay": float(np.mean(replay_ms))},
            "early_termination": {"tile_unit_chunks": ctr["tile_unit_chunks"], "skipped_unit_chunks": ctr["early_exit_unit_chunks"],
                                  "skipped_frac": (ctr["early_exit_unit_chunks"] / ctr["tile_unit_chunks"]) if ctr["tile_unit_chunks"] else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, torch, args, founder_ct, spacing, window_bp, args.r2)
        else:
            out["cpu_baseline"] = {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference",
                                   "sample": "measured at N=1 only"}
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
