#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the MI355X-native --indep-pairwise hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[1]): synthetic 50,000 samples x 1,000,000 biallelic variants, 22 autosomes with
variant counts proportional to GRCh38 lengths at uniform 2,875 bp spacing, `--indep-pairwise 200kb 0.5`.
With N GPUs the default is weak scaling on ONE genome: 22 autosomes holding N x 1,000,000 variants, the chromosomes (the
subcontigs) LPT-sharded over the ranks exactly as `ldp_set_shard` does it, so the load imbalance of 22 unequal
chromosomes on N ranks is part of the number.  `--strong` keeps the total at --variants whatever N is.  There is no
data-path collective; the prune bitmask is exchanged once per step with an RCCL all_gather.

A "step" is one pass of the hot path over the HBM-resident packed 2-bit genotype matrix: bit-plane split + per-variant
aggregates + allele counts (prepare_kernel), the banded pair statistics / prune predicate (pair_mfma_kernel for complete
data, pair_mfma_general_kernel when rows have missing calls), the host replay of the greedy scan, and the bitmask exchange.
`value` = candidate variant pairs decided per second over all ranks.

One JSON line is printed by rank 0.
  roofline      the pair kernel of the run against BOTH ceilings it can meet, the larger fraction named as `bound`:
                mfma  executed FP4 MFMA flops (instructions the kernel really issued: plan x k-steps - early termination)
                      / summed kernel time, against the guide's dense FP4 peak (10 PFLOP/s) and against the rate
                      tools/mfma_probe.hip measures on this box in this run;
                hbm   compulsory bytes (every owned bit-plane row once: variants x N/4) / summed kernel time against 8 TB/s,
                      and the streaming-read rate tools/ubench_copy.hip measures in this run.
                `traffic` (HBM bytes per step from PMC counters) is replayed from profiles/ when the workload matches
                and says so; the effective stream rate of SURVEY 8(d) (pairs x N/2 bytes) is a named side field.
  legs          the same step with early termination off, and with 0.1 % / 1 % missing calls (interval epilogue / six-product kernel)
  cpu_baseline  reference plink2 (oracle/_ref/plink2, AVX2, all host threads) on a bounded sample of the same generator,
                prune set compared with the HIP path's; plus both binaries end to end on the sample's files.
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
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP4_PEAK_TFLOPS = 10000.0  # MI355X_MICROARCH.md: ~10 PFLOP/s dense FP4 MFMA (AMD's 20 PF figure is 2:1 sparse)


def genome_layout(variants, genomes, spacing):
    """chr_idx / bp arrays for `genomes` copies of a 22-autosome genome of `variants` variants each."""
    tot = sum(GRCH38_MB)
    counts = [int(variants * mb / tot) for mb in GRCH38_MB]
    counts[0] += variants - sum(counts)
    chr_idx = np.empty(variants * genomes, dtype=np.uint32)
    bps = np.empty(variants * genomes, dtype=np.uint32)
    pos = 0
    for g in range(genomes):
        for c, n in enumerate(counts):
            chr_idx[pos:pos + n] = g * 22 + c
            bps[pos:pos + n] = 10000 + spacing * np.arange(n, dtype=np.uint32)
            pos += n
    return chr_idx, bps


def write_plink1_fileset(prefix, host_codes, founder_ct, chr_idx, bps):
    """REF-coded 2-bit rows (0 hom-REF, 1 het, 2 hom-ALT, 3 missing) -> .bed/.bim/.fam (.bed: 00 hom-ALT, 01 missing,
    10 het, 11 hom-REF)."""
    m = host_codes.shape[0]
    lut = np.zeros(256, dtype=np.uint8)
    conv = [3, 2, 0, 1]
    for b in range(256):
        lut[b] = sum(conv[(b >> (2 * k)) & 3] << (2 * k) for k in range(4))
    with open(prefix + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        pad = (4 - founder_ct % 4) % 4
        rows_per = max(1, (256 << 20) // host_codes.shape[1])
        for r0 in range(0, m, rows_per):
            bed = lut[host_codes[r0:r0 + rows_per]]
            if pad:
                bed[:, -1] &= np.uint8((1 << (2 * (4 - pad))) - 1)  # keep trailing bits zero
            f.write(bed.tobytes())
    with open(prefix + ".bim", "w") as f:
        f.write("".join("%d\tsnp%d\t0\t%d\tC\tA\n" % (chr_idx[i] + 1, i, bps[i]) for i in range(m)))
    with open(prefix + ".fam", "w") as f:
        f.write("".join("s%d s%d 0 0 2 -9\n" % (s, s) for s in range(founder_ct)))


def host_description():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count() or 0}


def cpu_baseline(pkg, torch, args, founder_ct, spacing, window_bp, r2, missing_rate):
    """Reference plink2 (all host cores) on a bounded sample of the same generator; also a parity check."""
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    base = {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference", **host_description()}
    if not (os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK)):
        return {**base, "sample": "oracle/_ref/plink2 not built"}
    if "avx2" not in open("/proc/cpuinfo").read():
        return {**base, "sample": "host CPU lacks AVX2"}
    m = args.cpu_sample_variants
    if m <= 0:
        m = 440000 if founder_ct <= 100000 else 22000  # ~2-20 s of reference time either way
    chr_idx, bps = genome_layout(m, 1, spacing)
    stride = (founder_ct + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(SEED, 0, m, founder_ct, missing_rate, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    # HIP path on the sample
    eng = pkg.LdPruneEngine(founder_ct, window_bp, 1, True, r2, device=torch.cuda.current_device())
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_device(0, m, buf.data_ptr(), stride, pkg.LDP_GENO_REF)
    removed_hip = eng.run()
    cand = eng.counters()["candidate_pairs"]
    eng.close()
    host = buf.cpu().numpy()
    del buf
    tmp = tempfile.mkdtemp(prefix="ldbench_")
    try:
        prefix = os.path.join(tmp, "sample")
        write_plink1_fileset(prefix, host, founder_ct, chr_idx, bps)
        cores = os.cpu_count() or 1
        kb = "%gkb" % (window_bp / 1000.0)
        cmd = [ref_bin, "--bfile", "sample", "--indep-pairwise", kb, repr(r2), "--threads", str(cores), "--out", "ref"]
        t0 = time.perf_counter()
        cp = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
        wall = time.perf_counter() - t0
        if cp.returncode != 0:
            return {**base, "cores": cores, "sample": "reference failed: " + cp.stdout[-300:]}
        removed_ids = set(ln.strip() for ln in open(os.path.join(tmp, "ref.prune.out")) if ln.strip())
        removed_ref = np.array([("snp%d" % i) in removed_ids for i in range(m)])
        cli = {}
        cli_bin = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
        if (not args.no_cli_compare) and os.path.exists(cli_bin):
            # the process-level drop-in on the same files (file mapping + H2D + kernels + replay + writer)
            t1 = time.perf_counter()
            cc = subprocess.run([cli_bin, "--bfile", "sample", "--indep-pairwise", kb, repr(r2), "--out", "hip"],
                                cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
            cli_wall = time.perf_counter() - t1
            same = (cc.returncode == 0 and open(os.path.join(tmp, "hip.prune.out")).read() == open(os.path.join(tmp, "ref.prune.out")).read()
                    and open(os.path.join(tmp, "hip.prune.in")).read() == open(os.path.join(tmp, "ref.prune.in")).read())
            cli = {"e2e_wall_s": {"reference_plink2": wall, "plink2_hip": cli_wall, "speedup": wall / cli_wall if cli_wall > 0 else None,
                                  "what": "process start to exit on the sample's .bed/.bim/.fam (page cache warm), same command line"},
                   "plink2_hip_files_identical": bool(same), "plink2_hip_rc": cc.returncode}
        mt = re.search(r"\((\d+) compute thread", cp.stdout)
        compute_threads = int(mt.group(1)) if mt else 0
        used = (compute_threads + 1) if mt else cores  # LD compute threads + the decode/main thread
        return {**base, **cli, "value": cand / wall, "cores": used, "threads_requested": cores,
                "threads_note": "the reference gives one compute thread to each subcontig at most and keeps one thread for decoding "
                                "(plink2_ld.cc:2599-2604): %d compute threads for 22 chromosomes, whatever --threads says" % compute_threads,
                "sample": "%d variants x %d samples of the same generator (22 chromosomes, %d bp spacing, missing rate %g, %d candidate pairs), "
                          "reference plink2 AVX2 end-to-end wall %.2f s incl. file load + allele-frequency pass" %
                          (m, founder_ct, spacing, missing_rate, cand, wall),
                "wall_s": wall, "prune_set_identical_to_hip": bool(np.array_equal(removed_ref, removed_hip)),
                "removed": int(removed_ref.sum())}
    finally:
        subprocess.call(["rm", "-rf", tmp])


def measured_ceilings():
    """The two ceilings, measured on this box in this run by the repo's own microbenchmarks (built by build())."""
    out = {}
    probe = os.path.join(REPO, "tools", "_bin", "mfma_probe")
    if os.path.exists(probe):
        try:
            txt = subprocess.run([probe, "--rates"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120).stdout
            rates = [float(x) for x in re.findall(r"= ([0-9.]+) PFLOP/s", txt)]
            if len(rates) >= 3:
                out["mfma_fp4_tflops_instruction_alone"] = rates[0] * 1000.0
                out["mfma_fp4_tflops_with_plane_expansion"] = rates[2] * 1000.0
                out["mfma_source"] = "tools/mfma_probe.hip --rates (v_mfma_scale_f32_32x32x64_f8f6f4, random operands, 2 waves per SIMD; mode 2 = LDS read + 7 bit-plane expansions per 8 MFMAs)"
        except Exception as e:  # pragma: no cover
            out["mfma_error"] = str(e)
    copy = os.path.join(REPO, "tools", "_bin", "ubench_copy")
    if os.path.exists(copy):
        try:
            txt = subprocess.run([copy], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
            reads = [float(m.group(1)) for m in re.finditer(r"^read only.*?([0-9.]+) TB/s", txt, re.M)]
            copies = [float(m.group(1)) for m in re.finditer(r"^copy.*?([0-9.]+) TB/s", txt, re.M)]
            if reads:
                out["hbm_read_gbs"] = max(reads) * 1000.0
            if copies:
                out["hbm_copy_gbs"] = max(copies) * 1000.0
            out["hbm_source"] = "tools/ubench_copy.hip (12 GiB streams, best grid)"
        except Exception as e:  # pragma: no cover
            out["hbm_error"] = str(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=1000000, help="variants per GPU (total with --strong)")
    ap.add_argument("--window-kb", type=float, default=200.0)
    ap.add_argument("--r2", type=float, default=0.5)
    ap.add_argument("--missing-rate", type=float, default=0.0)
    ap.add_argument("--spacing", type=int, default=2875, help="bp between consecutive variants")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --variants is the whole genome, sharded over the ranks")
    ap.add_argument("--cpu-sample-variants", type=int, default=0, help="0 = 440,000 up to 100k samples, 22,000 beyond")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the exhaustive / missing-calls legs and the ceiling microbenchmarks")
    ap.add_argument("--resident-planes", action="store_true",
                    help="for shapes whose 2-bit input and bit-planes do not fit HBM together (config 3: 156 GB each): convert once, "
                         "chunk by chunk, outside the timed region; a step is then the pair kernel + replay only (said so in config.workload)")
    ap.add_argument("--no-cli-compare", action="store_true", help="do not time plink2-hip end-to-end on the CPU-baseline sample files")
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
    m_total = args.variants if args.strong else args.variants * world
    chr_idx, bps = genome_layout(m_total, 1, spacing)  # ONE genome, whatever the rank count

    def build_engine(missing_rate):
        eng = pkg.LdPruneEngine(founder_ct, window_bp, 1, True, args.r2, device=local_rank)
        eng.set_variants(chr_idx, bps)
        subs = eng.subcontigs()
        owner = eng.set_shard(rank, world) if world > 1 else np.zeros(len(subs), dtype=np.uint32)
        owned = [(ln, first) for (ln, first), o in zip(subs, owner) if o == rank]
        local_ct = sum(ln for ln, _ in owned)
        # synthetic REF-coded genotypes of the owned subcontigs, resident in HBM before timing starts
        stride = (founder_ct + 3) // 4
        seg, geno = [], None
        if args.resident_planes:
            chunk_rows = max(1, (8 << 30) // stride)
            chunk = torch.empty((chunk_rows, stride), dtype=torch.uint8, device="cuda")
            for ln, first in owned:
                for c0 in range(0, ln, chunk_rows):
                    cnt = min(chunk_rows, ln - c0)
                    pkg.synth_genotypes_device(SEED, first + c0, cnt, founder_ct, missing_rate, chunk.data_ptr(), stride)
                    torch.cuda.synchronize()  # the generator runs on the null stream, the engine on its own: order them
                    eng.load_genotypes_device(first + c0, cnt, chunk.data_ptr(), stride, pkg.LDP_GENO_REF)
                    torch.cuda.synchronize()  # ... and the chunk buffer is reused
            del chunk
            torch.cuda.empty_cache()
        else:
            geno = torch.empty((max(local_ct, 1), stride), dtype=torch.uint8, device="cuda")
            off = 0
            for ln, first in owned:
                pkg.synth_genotypes_device(SEED, first, ln, founder_ct, missing_rate, geno.data_ptr() + off * stride, stride)
                seg.append((first, ln, off))
                off += ln
        torch.cuda.synchronize()
        return eng, geno, seg, stride, local_ct, len(subs)

    import importlib
    distmod = importlib.import_module("plink_ng_amd.dist")

    def make_step(eng, geno, seg, stride):
        def step():
            for first, ln, o in seg:
                eng.load_genotypes_device(first, ln, geno.data_ptr() + o * stride, stride, pkg.LDP_GENO_REF)
            bm = eng.run_bitmap()  # uint64 words over all variants; only this rank's bits are set
            if use_dist:
                # the one exchange step: all_gather of the per-rank removed bitmaps (RCCL over xGMI), OR-ed on the device
                return distmod.allgather_bitmaps(bm, world, device="cuda")
            return bm
        return step

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, eng, steps, warmup):
        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        ks, removed = [], None
        for _ in range(steps):
            removed = step()
            ks.append(eng.counters())
        sync()
        return time.perf_counter() - t0, ks, removed

    eng, geno, seg, stride, local_ct, n_subs = build_engine(args.missing_rate)
    step = make_step(eng, geno, seg, stride)
    elapsed, ks, removed = timed(step, eng, args.steps, args.warmup)
    ctr = ks[-1]
    removed = distmod.bitmap_to_mask(removed.cpu().numpy() if use_dist else removed, m_total)
    per_rank_pairs = [ctr["candidate_pairs"]]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([ctr["candidate_pairs"]], dtype=torch.int64, device="cuda")
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        per_rank_pairs = [int(x.item()) for x in allp]
    total_pairs = sum(per_rank_pairs)

    out = None
    if rank == 0:
        mean = lambda key: float(np.mean([k[key] for k in ks])) if ks else 0.0
        ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
        value = total_pairs * args.steps / elapsed
        kms_mfma, kms_mfma_gen = mean("ms_pair_mfma"), mean("ms_pair_mfma_general")
        # (rows with only a few missing calls stay with pair_mfma_kernel and its interval epilogue: DESIGN 4.1d)
        general = (args.missing_rate > 0) and not (kms_mfma > kms_mfma_gen)
        kms_valu = mean("ms_pair_fast") + mean("ms_pair_general")
        on_matrix_pipe = (kms_mfma + kms_mfma_gen) > kms_valu
        kernel = ("pair_mfma_general_kernel" if general else "pair_mfma_kernel") if on_matrix_pipe else \
                 ("pair_tiles_kernel<true>" if general else "pair_tiles_kernel<false>")
        kms = (kms_mfma_gen if general else kms_mfma) if on_matrix_pipe else kms_valu
        launches = max(int(ctr["pair_kernel_launches"]), 1)
        # --- MFMA side: instructions the kernel really issued.  One block product = 32 x 32 pairs; one k-step = one
        # v_mfma_scale_f32_32x32x64_f8f6f4 = 65,536 MACs; the general kernel issues six per block product and k-step.
        executed_ksteps = max(ctr["mfma_product_stages"] - ctr["mfma_skipped_product_stages"], 0) * (6 if general else 1)
        mfma_flops = executed_ksteps * 65536 * 2.0
        mfma_tflops = (mfma_flops / (kms * 1e-3)) / 1e12 if (kms > 0 and on_matrix_pipe) else 0.0
        # --- HBM side: every owned bit-plane row must be read once (N/4 bytes per variant)
        compulsory = local_ct * ((founder_ct + 511) // 512) * 128.0
        hbm_gbs = (compulsory / (kms * 1e-3)) / 1e9 if kms > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("samples") == founder_ct and tj.get("variants") == args.variants and tj.get("window_kb") == args.window_kb
                        and tj.get("kernel", "").startswith(kernel.split("<")[0]) and not general and world == 1):
                    traffic = tj.get("hbm_bytes_per_step")
                    traffic_src = "profiles/pmc_traffic.json (replayed: PMC passes of tools/profile.sh, tag %s; not measured in this run)" % tj.get("tag")
            except Exception:
                traffic = None
        mfma_frac, hbm_frac = mfma_tflops / FP4_PEAK_TFLOPS, hbm_gbs / HBM_PEAK_GBS
        by_mfma = mfma_frac >= hbm_frac
        roofline = {
            "bound": "mfma" if by_mfma else "hbm",
            "achieved": mfma_tflops if by_mfma else hbm_gbs, "peak": FP4_PEAK_TFLOPS if by_mfma else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if by_mfma else "GB/s", "frac": mfma_frac if by_mfma else hbm_frac,
            "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
            "kernel_ms_per_launch": kms / launches, "launches_per_step": launches, "kernel_ms_per_step": kms,
            "mfma": {"executed_tflops": mfma_tflops, "peak_tflops": FP4_PEAK_TFLOPS, "frac_of_peak": mfma_frac,
                     "mfma_instructions_per_step": executed_ksteps, "block_products": ctr["mfma_block_products"],
                     "plan_efficiency": (ctr["candidate_pairs"] / (ctr["mfma_block_products"] * 1024.0)) if ctr["mfma_block_products"] else None,
                     "early_termination_skipped_frac": (ctr["mfma_skipped_product_stages"] / ctr["mfma_product_stages"]) if ctr["mfma_product_stages"] else 0.0},
            "hbm": {"compulsory_bytes_per_step": compulsory, "compulsory_gbs": hbm_gbs, "peak_gbs": HBM_PEAK_GBS, "frac_of_peak": hbm_frac},
            "effective_stream_gbs": (ctr["candidate_pairs"] * (founder_ct / 2.0) / (kms * 1e-3)) / 1e9 if kms > 0 else 0.0,
            "note": "frac is against the datasheet peak of the named bound; measured_ceilings (same run, same box) give the same "
                    "fractions against what this box delivers.  effective_stream_gbs is SURVEY 8(d)'s pairs x N/2 bytes figure: "
                    "tiling makes it exceed any physical rate, it is not a roofline fraction.",
        }
        out = {
            "metric": "variant-pairs/s (--indep-pairwise, whole job)", "value": value, "unit": "variant-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "fp4 (E2M1: exact -1/0/+1) x fp4 -> f32 integer-exact MFMA accumulation + f64 predicate" if on_matrix_pipe
                     else "u32 popcount + f64 predicate",
            "data": "synthetic",
            "config": {"workload": "synthetic %d samples x %d biallelic variants (%s), one genome of 22 autosomes at %d bp spacing, "
                                   "--indep-pairwise %gkb %g, missing rate %g, chromosomes LPT-sharded over %d rank(s)%s" %
                                   (founder_ct, m_total, "total, strong scaling" if args.strong else "%d per GPU, weak scaling" % args.variants,
                                    spacing, args.window_kb, args.r2, args.missing_rate, world,
                                    "; bit-planes resident, conversion outside the timed step" if args.resident_planes else ""),
                       "samples": founder_ct, "variants_total": m_total, "variants_rank0": local_ct, "window_kb": args.window_kb, "r2": args.r2,
                       "subcontigs": n_subs, "candidate_pairs_total": total_pairs, "candidate_pairs_per_rank": per_rank_pairs,
                       "shard_imbalance_max_over_mean": (max(per_rank_pairs) / (total_pairs / world)) if total_pairs else 1.0,
                       "pairs_above_threshold_rank0": ctr["pred_true"], "above_threshold_pairs_consumed_by_replay_rank0": ctr["replay_pairs"],
                       "variants_removed": int(removed.sum())},
            "roofline": roofline,
            "stage_ms": {"prepare_kernel": mean("ms_prepare"), "pair_kernels": mean("ms_pair_kernel"), "pair_mfma_kernel": kms_mfma,
                         "pair_mfma_general_kernel": kms_mfma_gen, "pair_tiles_kernels_popcount": kms_valu, "host_replay": mean("ms_replay"),
                         "note": "prepare_kernel (HBM-bound: reads N/4, writes N/4 bytes per variant) and the pair kernel run back to back"},
        }
    eng.close()
    del geno
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_legs and not args.resident_planes:
        legs = {}
        # (a) the same step with early termination off: every block product walks every sample
        os.environ["LDP_EARLY_EXIT"] = "0"
        e2, g2, s2, st2, _, _ = build_engine(args.missing_rate)
        el, k2, _ = timed(make_step(e2, g2, s2, st2), e2, max(2, args.steps // 2), 1)
        legs["exhaustive"] = {"ms_per_step": 1000.0 * el / max(2, args.steps // 2), "pair_kernels_ms": float(np.mean([k["ms_pair_kernel"] for k in k2])),
                              "what": "LDP_EARLY_EXIT=0"}
        e2.close()
        del g2
        os.environ.pop("LDP_EARLY_EXIT")
        torch.cuda.empty_cache()
        # (b) missing calls in every variant: 0.1 % (the complete-data kernel with the interval epilogue) and
        # 1 % (the six-product kernel)
        if args.missing_rate == 0.0:
            for rate in (0.001, 0.01):
                e3, g3, s3, st3, _, _ = build_engine(rate)
                el, k3, _ = timed(make_step(e3, g3, s3, st3), e3, max(2, args.steps // 2), 1)
                c3 = e3.counters()
                last = k3[-1]
                kern = max((("pair_mfma_kernel", last["ms_pair_mfma"]), ("pair_mfma_general_kernel", last["ms_pair_mfma_general"]),
                            ("pair_tiles_kernel<true>", last["ms_pair_general"])), key=lambda kv: kv[1])[0]
                legs["missing_rate_%g" % rate] = {"ms_per_step": 1000.0 * el / max(2, args.steps // 2),
                                                  "pair_kernels_ms": float(np.mean([k["ms_pair_kernel"] for k in k3])),
                                                  "prepare_ms": float(np.mean([k["ms_prepare"] for k in k3])),
                                                  "kernel": kern,
                                                  "pairs_counted_exactly": int(c3.get("sparse_exact_pairs", 0)),
                                                  "vs_complete_data_step": (1000.0 * el / max(2, args.steps // 2)) / out["ms_per_step"]}
                e3.close()
                del g3
                torch.cuda.empty_cache()
        out["legs"] = legs
        ceil = measured_ceilings()
        out["roofline"]["measured_ceilings"] = ceil
        if ceil.get("mfma_fp4_tflops_instruction_alone"):
            out["roofline"]["mfma"]["frac_of_measured_instruction_rate"] = out["roofline"]["mfma"]["executed_tflops"] / ceil["mfma_fp4_tflops_instruction_alone"]
            out["roofline"]["mfma"]["frac_of_measured_rate_with_expansion"] = out["roofline"]["mfma"]["executed_tflops"] / ceil["mfma_fp4_tflops_with_plane_expansion"]
        if ceil.get("hbm_read_gbs"):
            out["roofline"]["hbm"]["frac_of_measured_read_rate"] = out["roofline"]["hbm"]["compulsory_gbs"] / ceil["hbm_read_gbs"]

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, torch, args, founder_ct, spacing, window_bp, args.r2, args.missing_rate)
        else:
            out["cpu_baseline"] = {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference",
                                   "sample": "measured at N=1 only", **host_description()}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
