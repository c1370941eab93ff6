#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the MI355X-native --indep-pairwise hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

ONE workload family at every N (synthetic; one genome of 22 autosomes with variant counts proportional to GRCh38 lengths, uniform
spacing, chromosomes = subcontigs LPT-sharded over the ranks exactly as `ldp_set_shard` does it):
  default   BASELINE.json configs[2]'s density, WEAK scaling: 500,000 samples, 290 bp spacing, `--indep-pairwise 500kb 0.2`,
            1,250,000 variants PER GPU (156 GB of 2-bit rows, resident in HBM) -- the real per-GPU share of the metric's 500k x 10M at
            N = 8, where this line IS config 3; at N = 1 it is the largest slice of it that fits one GPU.
  --strong  the metric's 10M variants in total whatever N.  A rank's share is resident only at N = 8; at N = 2 / 4 it exceeds HBM, the
            rank works through its chromosomes one engine at a time with every chromosome's rows copied inside the step from ONE
            generated chromosome, and the line says `"data": "model ..."`: a model of the workload, not a measurement of it.
  `--workload config2` (50,000 x 1,000,000 at 2,875 bp, `200kb 0.5`: BASELINE.json configs[1]; weak: that many variants per GPU),
  `--samples/--variants/--spacing/--window-kb/--r2` override all of this.
There is no data-path collective; the prune bitmask is exchanged once per step with an RCCL all_gather.

A "step" is one pass of the hot path over the HBM-resident 2-bit genotype image: the count pass (codes_kernel: per-variant
aggregates, allele counts, major allele, checkpoint statistics -- a read of N/4 bytes per variant, nothing is rewritten: the pair
kernels expand the 2-bit codes themselves), the banded pair statistics / prune predicate on the matrix pipe (pair_mfma_wide_kernel
on 8 x 8 block tiles for wide bands, pair_mfma_kernel for narrow ones, the interval epilogue / four-product kernels for rows with
missing calls), the host replay of the greedy scan, and the bitmask exchange.  The rows are generated straight into the engine's
image (ldp_map_rows), so they are resident when the timed region starts.  `value` = candidate variant pairs decided per second over
all ranks.

One JSON line is printed by rank 0.
  roofline      the pair kernel of the run against BOTH ceilings it can meet, the larger fraction named as `bound`:
                mfma  executed FP4 MFMA flops (instructions the kernel really issued: plan x k-steps - early termination)
                      / summed kernel time, against the guide's dense FP4 peak (10 PFLOP/s);
                hbm   compulsory bytes (every owned row once: variants x N/4) / summed kernel time against 8 TB/s.
                `traffic` (HBM bytes per step from PMC counters) is replayed from profiles/*_pmc_traffic.json ONLY when the workload
                matches AND the kernel sources that file was profiled on hash to what this tree holds (`traffic_source` "stale"
                otherwise); the effective stream rate of SURVEY 8(d) (pairs x N/2 bytes) is a named side field.
  legs          (N = 1) `config2` (BASELINE.json configs[1]: the step, its roofline, early termination off, 0.1 % / 1 % missing calls,
                the reference on a 440,000-variant sample with both binaries end to end), `config5_density` (500,000 x 120,000, 5 %
                missing calls, 2 % multiallelic records decoded inside the step), `config3_density_missing` (the same slice at 0.1 % and
                1 % missing calls: what a real call set looks like -- the tile kernel's SPARSE instantiation / the quarter tiles),
                `config4_tiles` (--r2-unphased inter-chr: a 65,536 x 65,536 cross-chromosome tile set at 500,000 samples, device-side
                filter, with plink2-hip against the reference on a slice), each with kernel time, roofline and a reference comparison.
                The legs timed by the wall run BEFORE the background reference processes of the end-to-end leg start.
                `--workload config5` runs config 5's per-GPU share (config 3's shape, 5 % missing calls, 2 % multiallelic records) as
                the main line instead.
  flat keys     the driver's record keeps `roofline` and `cpu_baseline` but only their SCALAR fields: every leg's kernel / ms per step /
                fraction with its bound / traffic multiple (`roofline.leg_<name>_*`), the power and clock medians of the timed steps
                (`roofline.timed_steps_*`), the bits check, and the end-to-end walls of both binaries on both file formats
                (`cpu_baseline.e2e_fixed_width_*`, `cpu_baseline.e2e_variable_width_*`) are repeated there as flat scalars
                (`flatten_summary`); at N > 1 `roofline.e2e_plink2_hip_gpus_*` = `plink2-hip --gpus N` end to end on rank 0.
  cpu_baseline  reference plink2 (oracle/_ref/plink2, AVX2, all host threads) on a bounded sample of the same generator,
                prune set compared with the HIP path's; and, at the metric's sample count, BOTH binaries end to end on a chr22-sized
                share of the metric's genome (176,765 variants x 500,000 samples: SURVEY 8(d)) as a 22 GB fixed-width .pgen AND as the
                reference's default variable-width .pgen of the same genotypes (`--make-pgen`, 12.6 GB: records decoded on the device)
                -- `cpu_baseline.e2e_wall_s`, measured walls, plink2-hip's phase split and file -> HBM rate beside them
                (tools/bench_support.py: E2EChr22).
  headline_bits_check  one chromosome of the timed share re-run alone on the popcount kernels (no matrix pipe, no early termination):
                its removed bits must equal the timed run's.
  power_and_clock      socket power / shader clock (rocm-smi) sampled during the timed steps.
"""
import argparse
import ctypes
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
sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_support as support  # noqa: E402  (the rocm-smi sampler, the end-to-end runs of both binaries, the in-run PMC passes)

GRCH38_MB = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28,
             114.36, 107.04, 101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82]
SEED = 20260925 + 2
if os.environ.get("LDP_BENCH_ALT_MINOR"):   # measurement aid (profiles/r04_experiments.md): ALT is the minor allele of every synthetic variant
    SEED |= 1 << 63
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP4_PEAK_TFLOPS = 10000.0  # MI355X_MICROARCH.md: ~10 PFLOP/s dense FP4 MFMA (AMD's 20 PF figure is 2:1 sparse)
CONFIGS = {
    "config2": dict(samples=50000, variants=1000000, spacing=2875, window_kb=200.0, r2=0.5),
    "config3": dict(samples=500000, variants=10000000, spacing=290, window_kb=500.0, r2=0.2),
}
# config5 = BASELINE.json configs[4]: config 3's shape with 5 % missing calls in every variant and 2 % of the variants multiallelic (their records
# decoded and collapsed on the device inside every step): `--workload config5` runs its per-GPU share as the main line
CONFIGS["config5"] = dict(CONFIGS["config3"], missing_rate=0.05, multiallelic_frac=0.02)
PER_GPU_VARIANTS = {"config2": 1000000, "config3": 1250000, "config5": 1250000}  # weak scaling: the share of one GPU (config 3 / 5: 10M / 8)
CHR22_FRACTION = GRCH38_MB[21] / sum(GRCH38_MB)               # end-to-end runs materialise <= one chr22-sized chromosome
# the kernel sources a PMC profile is valid for (roofline.traffic is replayed only when their hashes match this tree)
KERNEL_SOURCES = ["plink-ng_amd/csrc/" + f for f in ("ldp_pair_wide.hip", "ldp_pair_mfma.hip", "ldp_mfma_device.h", "ldp_pair_device.h", "ldp_device.h",
                                                      "ldp_codes.hip", "ldp_engine.cpp", "ldp_engine_run.cpp", "ldp_engine_internal.h")]
HBM_BYTES = float(os.environ.get("LDP_BENCH_HBM_GB", "288")) * 1e9  # (the override forces the non-resident mode at small sizes: tests)


def genome_layout(variants, genomes, spacing):
    """chr_idx / bp arrays of one 22-autosome genome holding `variants` variants (`genomes` is 1: one genome whatever the rank
    count; the argument is kept for the tests and tools that share this generator)."""
    assert genomes == 1
    tot = sum(GRCH38_MB)
    counts = [int(variants * mb / tot) for mb in GRCH38_MB]
    counts[0] += variants - sum(counts)
    chr_idx = np.empty(variants, dtype=np.uint32)
    bps = np.empty(variants, dtype=np.uint32)
    pos = 0
    for c, n in enumerate(counts):
        chr_idx[pos:pos + n] = c
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


def cpu_baseline(pkg, torch, founder_ct, m, spacing, window_kb, r2, missing_rate, cli_compare=True, full_variants=0):
    """Reference plink2 (all host cores) on a bounded sample of the same generator; also a parity check."""
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    base = {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference", **host_description()}
    if not (os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK)):
        return {**base, "sample": "oracle/_ref/plink2 not built"}
    if "avx2" not in open("/proc/cpuinfo").read():
        return {**base, "sample": "host CPU lacks AVX2"}
    window_bp = pkg.kb_window(window_kb)
    chr_idx, bps = genome_layout(m, 1, spacing)
    stride = (founder_ct + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(SEED, 0, m, founder_ct, missing_rate, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    # HIP path on the sample (rows from a caller-owned device buffer: copied into the image, then as in the timed step)
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
        kb = "%gkb" % window_kb
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
        if cli_compare and os.path.exists(cli_bin):
            # the process-level drop-in on the same files (file mapping + H2D + kernels + replay + writer)
            # (twice, the faster wall reported, as in the chr22-sized leg: the first plink2-hip process on a fresh box pays for things no later one does --
            # profiles/r06_experiments.md section 7 -- and which process of a bench run that is depends on the order of its legs)
            cli_walls = []
            for _ in range(2):
                t1 = time.perf_counter()
                cc = subprocess.run([cli_bin, "--bfile", "sample", "--indep-pairwise", kb, repr(r2), "--out", "hip"],
                                    cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
                cli_walls.append(time.perf_counter() - t1)
                if cc.returncode != 0:
                    break
            cli_wall = min(cli_walls)
            same = (cc.returncode == 0 and open(os.path.join(tmp, "hip.prune.out")).read() == open(os.path.join(tmp, "ref.prune.out")).read()
                    and open(os.path.join(tmp, "hip.prune.in")).read() == open(os.path.join(tmp, "ref.prune.in")).read())
            cli = {"e2e_wall_s": {"reference_plink2": wall, "plink2_hip": cli_wall, "plink2_hip_wall_s_runs": cli_walls, "speedup": wall / cli_wall if cli_wall > 0 else None,
                                  "what": "process start to exit on the sample's .bed/.bim/.fam (page cache warm), same command line; plink2-hip: the faster of two runs"},
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


def headline_bits_check(pkg, torch, wl, removed_mask):
    """The bits of the headline run, checked: one whole chromosome of the rank's share (the last owned subcontig: chr21 / chr22, ~20,000
    variants) runs ALONE through a second engine that shares nothing with the timed one but the rows -- the popcount kernels on bit-planes
    (no matrix pipe, no FP4 operands), no early termination, its own plan -- and its removed bits must equal the big run's for that
    subcontig.  (Subcontigs are independent: plink2_ld.cc:2686-2694.)"""
    if not (wl.resident and wl.owned_subs and wl.engines):
        return {"checked": False, "why": "share not resident"}
    eng0 = wl.engines[0][0]
    ln, first = wl.owned_subs[-1]
    seg = None
    for sfirst, sln, ptr, stride in wl.segs[id(eng0)]:
        if sfirst <= first and first + ln <= sfirst + sln:
            seg = (ptr + (first - sfirst) * stride, stride)
    if seg is None:
        return {"checked": False, "why": "subcontig not inside one mapped run"}
    t0 = time.perf_counter()
    e = pkg.LdPruneEngine(wl.founder_ct, wl.window_bp, 1, True, wl.cfg["r2"], device=wl.device)
    e.set_option("pair_mfma", 0)
    e.set_option("early_exit", 0)
    sel = slice(first, first + ln)
    e.set_variants(wl.chr_idx[sel], wl.bps[sel])
    e.load_genotypes_device(0, ln, seg[0], seg[1], pkg.LDP_GENO_REF)
    got = e.run()
    c = e.counters()
    e.close()
    torch.cuda.empty_cache()
    want = np.asarray(removed_mask[sel], dtype=bool)
    return {"checked": True, "identical": bool(np.array_equal(np.asarray(got, dtype=bool), want)), "subcontig_first_variant": int(first), "variants": int(ln),
            "removed": int(want.sum()), "candidate_pairs": int(c["candidate_pairs"]), "seconds": time.perf_counter() - t0,
            "independent_engine": "popcount kernels on bit-planes (pair_mfma 0: ms_pair_mfma %.1f, popcount ms %.1f), early termination off, planned alone" %
                                  (c["ms_pair_mfma"], c["ms_pair_fast"] + c["ms_pair_general"])}


def config5_reference_slice(pkg, torch, founder_ct, m, spacing, window_kb, r2, missing_rate, multi_runs=2, multi_len=110):
    """Config 5's two ingredients against the REFERENCE in one fileset: rows with `missing_rate` missing calls AND variants with a second
    ALT allele.  `m` variants of the generator are written as a variable-width .pgen (storage mode 0x10: type-0 records for the
    biallelic variants, type 0 + auxiliary track 1 for `multi_runs` stretches of `multi_len` variants in which a tenth of the het /
    hom-ALT calls carry ALT2) + a .pvar with `C,G` ALT fields; the HIP path decodes the file's records on the device and collapses the
    multiallelic ones major-vs-rest (ldp_load_pgen_records), reference plink2 prunes the same files (PgrGetInv1 -> Get1Multiallelic,
    pgenlib_read.cc:5417), and the removed sets are compared variant by variant."""
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    if not (os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK)):
        return {"skipped": "oracle/_ref/plink2 not built"}
    n = founder_ct
    assert n % 4 == 0
    window_bp = pkg.kb_window(window_kb)
    chr_idx, bps = genome_layout(m, 1, spacing)
    stride = n // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(SEED, 0, m, n, missing_rate, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    del buf
    rng = np.random.default_rng(11)
    multi = np.zeros(m, dtype=bool)
    for k in range(multi_runs):
        st = int((k + 0.5) * m / multi_runs)
        multi[st:st + multi_len] = True
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    lens = np.full(m, stride, dtype=np.int64)
    aux = {}
    for v in np.flatnonzero(multi):
        codes = ((host[v, :, None] >> shifts) & 3).reshape(-1)
        n1, n2 = int((codes == 1).sum()), int((codes == 2).sum())
        b1, b2 = rng.random(n1) < 0.1, rng.random(n2) < 0.1
        both = rng.random(int(b2.sum())) < 0.5
        aux[int(v)] = np.concatenate([np.array([0], dtype=np.uint8), np.packbits(b1, bitorder="little"), np.packbits(b2, bitorder="little"), np.packbits(both, bitorder="little")])
        lens[v] += len(aux[int(v)])
    tmp = tempfile.mkdtemp(prefix="ldbench5_")
    try:
        blocks = (m + 65535) // 65536
        header_len = 12 + 8 * blocks + 4 * m
        with open(os.path.join(tmp, "s.pgen"), "wb") as f:   # (pgen_spec.tex:160-235: 8-bit record types, 3-byte record lengths)
            f.write(bytes([0x6C, 0x1B, 0x10]) + np.uint32(m).tobytes() + np.uint32(n).tobytes() + bytes([0x40 | 6]))
            off = header_len
            for b in range(blocks):
                f.write(np.uint64(off).tobytes())
                off += int(lens[b * 65536:(b + 1) * 65536].sum())
            for b in range(blocks):
                lo, hi = b * 65536, min(m, (b + 1) * 65536)
                f.write(np.where(multi[lo:hi], 0x08, 0x00).astype(np.uint8).tobytes())
                f.write(b"".join(int(x).to_bytes(3, "little") for x in lens[lo:hi]))
            for v in range(m):
                f.write(host[v].tobytes())
                if multi[v]:
                    f.write(aux[v].tobytes())
        with open(os.path.join(tmp, "s.pvar"), "w") as f:
            f.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join("%d\t%d\tsnp%d\tA\t%s\n" % (chr_idx[i] + 1, bps[i], i, "C,G" if multi[i] else "C") for i in range(m)))
        with open(os.path.join(tmp, "s.psam"), "w") as f:
            f.write("#IID\tSEX\n" + "".join("s%d\t2\n" % q for q in range(n)))
        del host
        # HIP path: the file's records decoded (and the multiallelic ones collapsed) on the device
        pg = pkg.PgenFile(os.path.join(tmp, "s.pgen"))
        eng = pkg.LdPruneEngine(n, window_bp, 1, True, r2, device=torch.cuda.current_device())
        eng.set_variants(chr_idx, bps)
        t0 = time.perf_counter()
        maj = eng.load_pgen_records(0, pg, allele_cts=np.where(multi, 3, 2))
        removed_hip = np.asarray(eng.run(), dtype=bool)
        hip_s = time.perf_counter() - t0
        cand = eng.counters()["candidate_pairs"]
        eng.close()
        pg.close()
        cores = os.cpu_count() or 1
        t1 = time.perf_counter()
        cp = subprocess.run([ref_bin, "--pfile", "s", "--indep-pairwise", "%gkb" % window_kb, repr(r2), "--threads", str(cores), "--out", "ref"], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
        wall = time.perf_counter() - t1
        if cp.returncode != 0:
            return {"error": "reference failed: " + cp.stdout[-300:]}
        removed_ids = set(ln.strip() for ln in open(os.path.join(tmp, "ref.prune.out")) if ln.strip())
        removed_ref = np.array([("snp%d" % i) in removed_ids for i in range(m)])
        non_ref_major = int(((maj != 0xffffffff) & (maj != 0)).sum())
        return {"prune_set_identical_to_hip": bool(np.array_equal(removed_ref, removed_hip)), "removed": int(removed_ref.sum()), "multiallelic_variants": int(multi.sum()),
                "multiallelic_variants_with_a_non_ref_major_allele": non_ref_major, "wall_s": wall, "value": cand / wall, "cores": cores, "hip_decode_and_prune_s": hip_s,
                "sample": "%d variants x %d samples of the same generator at %g missing calls as a variable-width .pgen, %d of the variants with a second ALT allele in "
                          "auxiliary track 1 (%d candidate pairs); reference plink2 end-to-end wall %.2f s" % (m, n, missing_rate, int(multi.sum()), cand, wall),
                "note": "the HIP side loads the FILE'S records through ldp_load_pgen_records (device decode + major-vs-rest collapse), the reference reads the same file: the "
                        "comparison of the collapse with the reference that round 4's leg only made against numpy"}
    finally:
        subprocess.call(["rm", "-rf", tmp])


def measured_ceilings():
    """The two ceilings, measured on this box in this run by the repo's own microbenchmarks (built by build())."""
    out = {}
    probe = os.path.join(REPO, "tools", "_bin", "mfma_probe")
    if os.path.exists(probe):
        try:
            txt = subprocess.run([probe, "--rates"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180).stdout
            peak = {}
            for mt in re.finditer(r"^7\. peak \((\S+(?: \+1)?) operands, (\d) waves per SIMD.*?= ([0-9.]+) PFLOP/s; shader clock (\d+) MHz", txt, re.M):
                peak["%s_operands_%sw" % (mt.group(1).replace(" ", ""), mt.group(2))] = {"tflops": float(mt.group(3)) * 1000.0, "shader_mhz": int(mt.group(4))}
            if peak:
                out["mfma_fp4_instruction_alone"] = peak
                best = max(v["tflops"] for v in peak.values())
                rnd = max((v["tflops"] for k, v in peak.items() if k.startswith("random")), default=None)
                out["mfma_fp4_tflops_best"] = best
                out["mfma_fp4_tflops_random_operands"] = rnd
            m2 = re.search(r"^5\. mode 2 .*?= ([0-9.]+) PFLOP/s", txt, re.M)
            if m2:
                out["mfma_fp4_tflops_with_plane_expansion"] = float(m2.group(1)) * 1000.0
            out["mfma_source"] = ("tools/mfma_probe.hip --rates (v_mfma_scale_f32_32x32x64_f8f6f4 alone: constant operands reach the datasheet rate, "
                                  "random operands clock down -- the shader clock of each run is printed beside it)")
        except Exception as e:  # pragma: no cover
            out["mfma_error"] = str(e)
    loop = os.path.join(REPO, "tools", "_bin", "tile_shape_probe")
    if os.path.exists(loop):
        # the tile kernel's OWN stage loop in its own shape and form ("2x4pf": eight waves x 2 x 4 on 8 x 8 tiles, the second half-stage's J fragments made
        # during the first), on full tiles without checkpoints or epilogue, 500,224 samples, 2 s on this box, checksum against a plain kernel
        try:
            txt = subprocess.run([loop, "2x4pf", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180).stdout
            d = json.loads(txt.strip().splitlines()[-1])
            out["stage_loop_probe"] = {k: d.get(k) for k in ("shape", "waves", "products_per_wave", "tile", "vgprs", "samples", "ms_per_launch", "pflops", "of_fp4_peak",
                                                            "shader_clock_mhz_wg0", "l2_to_lds_tb_s", "checksum_ok")}
            out["stage_loop_probe_tflops"] = d["pflops"] * 1000.0 if d.get("checksum_ok") else None
            out["stage_loop_source"] = ("tools/tile_shape_probe.hip 2x4pf: pair_mfma_wide_kernel's stage loop alone (same DMA ring, LDS layout, expansion and MFMA; "
                                        "full tiles, no checkpoints, no epilogue) -- what the kernel could reach if the band were all full tiles")
        except Exception as e:  # pragma: no cover
            out["stage_loop_error"] = str(e)[:200]
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


class Workload:
    """One rank's share of a genome: engines planned up front (host work), rows generated straight into an engine's resident
    image.  `resident`: one engine holds the whole share and its rows stay in HBM across steps.  Otherwise the share does not
    fit HBM and the rank works through its chromosomes one engine at a time, copying each chromosome's rows inside the step from one
    resident chromosome's worth of generated rows."""

    def __init__(self, pkg, torch, cfg, missing_rate, rank, world, device, options=None, multiallelic=None):
        self.pkg, self.torch, self.cfg, self.missing_rate = pkg, torch, cfg, missing_rate
        self.multi = None
        self.rank, self.world, self.device = rank, world, device
        self.options = options or {}
        self.founder_ct = cfg["samples"]
        self.window_bp = pkg.kb_window(cfg["window_kb"])
        self.m_total = cfg["variants"]
        self.chr_idx, self.bps = genome_layout(self.m_total, 1, cfg["spacing"])
        planner = self._engine()
        planner.set_variants(self.chr_idx, self.bps)
        self.subs = planner.subcontigs()
        self.owner = planner.set_shard(rank, world) if world > 1 else np.zeros(len(self.subs), dtype=np.uint32)
        self.owned_subs = [(ln, first) for (ln, first), o in zip(self.subs, self.owner) if o == rank]
        self.owned = []
        for (ln, first), o in zip(self.subs, self.owner):
            if o != rank:
                continue
            if self.owned and (self.owned[-1][1] + self.owned[-1][0] == first):
                self.owned[-1] = (self.owned[-1][0] + ln, self.owned[-1][1])  # adjacent subcontigs: one run of image rows, one load call
            else:
                self.owned.append((ln, first))
        self.local_ct = sum(ln for ln, _ in self.owned)
        row_bytes = ((self.founder_ct + 511) // 512) * 128
        self.image_bytes = self.local_ct * row_bytes
        self.resident = self.image_bytes * 1.06 + 6e9 < HBM_BYTES
        self.engines = []
        if self.resident:
            self.engines.append((planner, self.owned))
        else:
            planner.close()
            for ln, first in self.owned_subs:  # one engine per owned chromosome, planned now, device memory only while it is worked on
                e = self._engine()
                sel = slice(first, first + ln)
                e.set_variants(self.chr_idx[sel], self.bps[sel])
                self.engines.append((e, [(ln, first)]))
        self.segs = {}
        self.master = None
        if self.resident:
            self._generate(planner, self.owned, 0)
            if multiallelic:
                self._make_multiallelic(planner, *multiallelic)
        elif self.owned_subs:
            # The share does not fit: its chromosomes are worked through one engine at a time, and what an engine counts has to be
            # resident when the step begins -- not produced by the synthetic generator inside it (32 GB/s: it would be most of the
            # step).  ONE chromosome's worth of rows is generated here, once, and every chromosome of the share takes its rows from
            # that buffer by a device-to-device copy into its engine's image (the longest owned chromosome's rows; shorter ones take a
            # prefix).  The chromosomes of a non-resident share therefore hold the same genotypes; shapes, windows and work are the
            # genome's.
            ln_max = max(ln for ln, _ in self.owned_subs)
            first = [f for ln, f in self.owned_subs if ln == ln_max][0]
            self.master = torch.empty((ln_max, row_bytes), dtype=torch.uint8, device="cuda")
            pkg.synth_genotypes_device(SEED, first, ln_max, self.founder_ct, self.missing_rate, self.master.data_ptr(), row_bytes)
        torch.cuda.synchronize()

    def _make_multiallelic(self, eng, n_runs, run_len):
        """Config 5's multiallelic sites: n_runs stretches of run_len variants get a second ALT allele and arrive, inside every step,
        as .pgen variant records resident in HBM -- plain main track + auxiliary track 1 (both patch sets as bit arrays: a tenth of the
        het / hom-ALT calls carry ALT2) -- that ldp_load_pgen_records decodes and collapses major-vs-rest on the device.
        The records are built once, here, from the generated rows."""
        pkg, torch, n = self.pkg, self.torch, self.founder_ct
        first0, ln0, ptr0, stride = self.segs[id(eng)][0]
        nb = (n + 3) // 4
        rng = np.random.default_rng(7)
        starts = [first0 + int((k + 0.5) * ln0 / n_runs) for k in range(n_runs)]
        blobs, runs, offset, checks = [], [], 0, []
        shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
        for st in starts:
            rows = torch.empty((run_len, stride), dtype=torch.uint8, device="cuda")
            assert pkg.hip_memcpy_dtod(rows.data_ptr(), ptr0 + (st - first0) * stride, run_len * stride) == 0
            torch.cuda.synchronize()
            host = rows.cpu().numpy()
            recs = (pkg.ldp_pgen_rec * run_len)()
            for q in range(run_len):
                codes = ((host[q, :nb, None] >> shifts) & 3).reshape(-1)[:n]
                main = host[q, :nb].copy()
                if n % 4:
                    main[-1] &= (1 << (2 * (n % 4))) - 1
                i1, i2 = np.flatnonzero(codes == 1), np.flatnonzero(codes == 2)
                b1, b2 = rng.random(len(i1)) < 0.1, rng.random(len(i2)) < 0.1
                both = rng.random(int(b2.sum())) < 0.5                      # ALT2/ALT2 (else ALT1/ALT2)
                rec = np.concatenate([main, np.array([0], dtype=np.uint8), np.packbits(b1, bitorder="little"), np.packbits(b2, bitorder="little"),
                                      np.packbits(both, bitorder="little")])
                recs[q].offset, recs[q].length, recs[q].vrtype, recs[q].allele_ct = offset, len(rec), 0x08, 3
                offset += len(rec)
                blobs.append(rec)
                if q == 0:   # expected record of the collapsed row, for the parity flag below
                    lo = np.where(codes == 3, 255, np.where(codes == 2, 1, 0)).astype(np.int64)
                    hi = np.where(codes == 3, 255, np.where(codes == 0, 0, 1)).astype(np.int64)
                    hi[i1[b1]] = 2
                    k2 = i2[b2]
                    lo[k2] = np.where(both, 2, 1)
                    hi[k2] = 2
                    called = codes != 3
                    cnt = np.bincount(np.concatenate([lo[called], hi[called]]), minlength=3)[:3]
                    f0, f1 = cnt[0] * (1.0 / cnt.sum()), cnt[1] * (1.0 / cnt.sum())
                    maj = 0 if f0 >= 0.5 else (1 if f1 >= 0.5 else (2 if (cnt[2] > max(cnt[0], cnt[1])) else (0 if f0 >= f1 else 1)))
                    nonmaj = (lo[called] != maj).astype(np.int64) + (hi[called] != maj)
                    checks.append((st, int(called.sum()), int((nonmaj == 0).sum()) - int((nonmaj == 2).sum()), int((nonmaj != 1).sum())))
            runs.append((st, run_len, recs))
        data = np.concatenate(blobs)
        self.multi = {"runs": runs, "bytes": torch.from_numpy(data).cuda(), "nbytes": int(len(data)), "variants": n_runs * run_len, "checks": checks,
                      "ms_calls": []}

    def _load_multiallelic(self, eng):
        t0 = time.perf_counter()
        for st, ln, recs in self.multi["runs"]:
            eng._ck(eng._L.ldp_load_pgen_records(eng._h, st, ln, ctypes.c_void_p(self.multi["bytes"].data_ptr()), self.multi["nbytes"], self.pkg.LDP_MEM_DEVICE,
                                                  recs, None, self.founder_ct, None))
        self.multi["ms_calls"].append(1e3 * (time.perf_counter() - t0))

    def _engine(self):
        e = self.pkg.LdPruneEngine(self.founder_ct, self.window_bp, 1, True, self.cfg["r2"], device=self.device)
        for k, v in self.options.items():
            e.set_option(k, v)
        return e

    def _generate(self, eng, runs, base):
        """rows of `runs` (global variant ranges) generated into eng's image; eng's variant 0 = global variant `base`"""
        segs = []
        for ln, first in runs:
            ptr, stride = eng.map_rows(first - base, ln)
            self.pkg.synth_genotypes_device(SEED, first, ln, self.founder_ct, self.missing_rate, ptr, stride)
            segs.append((first - base, ln, ptr, stride))
        self.torch.cuda.synchronize()  # (the generator runs on the null stream, the engine on its own: order them)
        self.segs[id(eng)] = segs

    def step(self):
        """one pass; returns (global removed bitmap as uint64 words, [counters per engine])"""
        words = np.zeros((self.m_total + 63) // 64, dtype=np.uint64)
        ctrs = []
        for eng, runs in self.engines:
            base = 0 if self.resident else runs[0][1]
            if not self.resident:
                segs = []
                for ln, first in runs:
                    ptr, stride = eng.map_rows(first - base, ln)
                    assert stride == self.master.shape[1]
                    assert self.pkg.hip_memcpy_dtod(ptr, self.master.data_ptr(), ln * stride) == 0
                    segs.append((first - base, ln, ptr, stride))
                self.torch.cuda.synchronize()
                self.segs[id(eng)] = segs
            for first, ln, ptr, stride in self.segs[id(eng)]:
                eng.load_genotypes_device(first, ln, ptr, stride, self.pkg.LDP_GENO_REF)
            if self.multi:
                self._load_multiallelic(eng)
            bm = eng.run_bitmap()  # uint64 words over the engine's variants; only this rank's bits are set
            ctrs.append(eng.counters())
            if self.resident:
                words = bm
            else:
                mask = np.unpackbits(bm.view(np.uint8), bitorder="little")[:runs[0][0]]
                full = np.zeros(self.m_total, dtype=np.uint8)
                full[base:base + runs[0][0]] = mask
                words |= np.packbits(np.pad(full, (0, len(words) * 64 - self.m_total)), bitorder="little").view(np.uint64)
                eng.release_device()
        return words, ctrs

    def close(self):
        for eng, _ in self.engines:
            eng.close()
        self.engines = []
        self.torch.cuda.empty_cache()


def sum_counters(ctrs):
    keys = ("candidate_pairs", "pred_true", "replay_pairs", "ms_prepare", "ms_pair_kernel", "ms_pair_mfma", "ms_pair_mfma_general", "ms_pair_fast",
            "ms_pair_general", "ms_replay", "pair_kernel_launches", "mfma_block_products", "mfma_product_stages", "mfma_skipped_product_stages",
            "sparse_exact_pairs", "route_complete_launches", "route_sparse_launches", "route_general_launches", "mfma_extra_product_stages", "wide_tiles", "four_tile_launches", "sparse_tile_launches")
    return {k: sum(c[k] for c in ctrs) for k in keys}


def source_hashes():
    """git blob hashes (sha1 of "blob <len>\\0" + content) of the kernel sources, computed from the working tree"""
    import hashlib
    out = {}
    for rel in KERNEL_SOURCES:
        try:
            data = open(os.path.join(REPO, rel), "rb").read()
        except OSError:
            out[rel] = None
            continue
        out[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def pmc_traffic(founder_ct, variants, window_kb, missing_rate):
    """HBM bytes per step from PMC counters, replayed from profiles/ (collected by tools/profile.sh on the same workload) -- only
    from a file whose recorded kernel-source hashes equal this tree's: a kernel edit makes every older profile "stale"."""
    now = source_hashes()
    stale = None
    for name in sorted(os.listdir(os.path.join(REPO, "profiles")), reverse=True):
        if not name.endswith("pmc_traffic.json"):
            continue
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", name)))
        except Exception:
            continue
        if not (tj.get("samples") == founder_ct and tj.get("variants") == variants and tj.get("window_kb") == window_kb
                and abs(tj.get("missing_rate", 0.0) - missing_rate) < 1e-12):
            continue
        if tj.get("sources") != now:
            stale = stale or name
            continue
        return tj.get("hbm_bytes_per_step"), "profiles/%s (replayed: PMC passes of tools/profile.sh, tag %s, kernel sources identical to this tree's; not measured in this run)" % (name, tj.get("tag")), tj
    if stale:
        return None, "stale (profiles/%s was collected on other kernel sources than this tree's)" % stale, None
    return None, None, None


def pair_roofline(c, founder_ct, local_ct, missing_rate, variants, window_kb, options=None):
    kms_mfma, kms_gen = c["ms_pair_mfma"], c["ms_pair_mfma_general"]
    general = c["route_general_launches"] > 0
    kms_valu = c["ms_pair_fast"] + c["ms_pair_general"]
    on_matrix_pipe = (kms_mfma + kms_gen) > kms_valu
    four_tiles = c.get("four_tile_launches", 0) > 0   # wide bands: the four-product form runs on quarter tiles
    wide = c.get("wide_tiles", 0) > 0 and (c["route_complete_launches"] > 0 or c.get("sparse_tile_launches", 0) > 0)   # the 8 x 8 tile kernel: complete-data launches of wide-band
    # subcontigs, and (its SPARSE instantiation) launches whose rows have a few missing calls
    kernel = (("pair_mfma_tile4_kernel" if four_tiles else "pair_mfma_general_kernel") if general else (("pair_mfma_wide_kernel<SPARSE>" if c.get("sparse_tile_launches", 0) > 0 else "pair_mfma_wide_kernel") if wide else "pair_mfma_kernel")) if on_matrix_pipe else ("pair_tiles_kernel<true>" if general else "pair_tiles_kernel<false>")
    kms = (kms_mfma + kms_gen) if on_matrix_pipe else kms_valu
    launches = max(int(c["pair_kernel_launches"]), 1)
    # MFMA side: instructions the kernel really issued.  One block product = 32 x 32 pairs; one k-step = one
    # v_mfma_scale_f32_32x32x64_f8f6f4 = 65,536 MACs; the missing-call kernels issue four per block product and k-step on prune
    # launches (six with the engine option pair_four = 0).
    per_product = (6 if (options or {}).get("pair_four", 1) == 0 else 4) if general else 1
    executed = (max(c["mfma_product_stages"] - c["mfma_skipped_product_stages"], 0) + c["mfma_extra_product_stages"]) * per_product
    mfma_tflops = (executed * 65536 * 2.0 / (kms * 1e-3)) / 1e12 if (kms > 0 and on_matrix_pipe) else 0.0
    # HBM side: every owned row must be read once (N/4 bytes per variant, rows padded to 64 bytes)
    compulsory = local_ct * ((founder_ct + 511) // 512) * 128.0
    hbm_gbs = (compulsory / (kms * 1e-3)) / 1e9 if kms > 0 else 0.0
    traffic, traffic_src, tj = pmc_traffic(founder_ct, variants, window_kb, missing_rate)
    mfma_frac, hbm_frac = mfma_tflops / FP4_PEAK_TFLOPS, hbm_gbs / HBM_PEAK_GBS
    by_mfma = mfma_frac >= hbm_frac
    return {
        "bound": "mfma" if by_mfma else "hbm",
        "achieved": mfma_tflops if by_mfma else hbm_gbs, "peak": FP4_PEAK_TFLOPS if by_mfma else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if by_mfma else "GB/s", "frac": mfma_frac if by_mfma else hbm_frac,
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_over_compulsory": (traffic / compulsory) if (traffic and compulsory) else None,
        "kernel": kernel, "kernel_ms_per_launch": kms / launches, "launches_per_step": launches, "kernel_ms_per_step": kms,
        "mfma": {"executed_tflops": mfma_tflops, "peak_tflops": FP4_PEAK_TFLOPS, "frac_of_peak": mfma_frac,
                 "mfma_instructions_per_step": executed, "block_products": c["mfma_block_products"],
                 "plan_efficiency": (c["candidate_pairs"] / (c["mfma_block_products"] * 1024.0)) if c["mfma_block_products"] else None,
                 "early_termination_skipped_frac": (c["mfma_skipped_product_stages"] / c["mfma_product_stages"]) if c["mfma_product_stages"] else 0.0,
                 "computed_beyond_plan_frac": (c["mfma_extra_product_stages"] / c["mfma_product_stages"]) if c["mfma_product_stages"] else 0.0,
                 "wide_tiles": c["wide_tiles"]},
        "hbm": {"compulsory_bytes_per_step": compulsory, "compulsory_gbs": hbm_gbs, "peak_gbs": HBM_PEAK_GBS, "frac_of_peak": hbm_frac},
        "effective_stream_gbs": (c["candidate_pairs"] * (founder_ct / 2.0) / (kms * 1e-3)) / 1e9 if kms > 0 else 0.0,
        "routes": {"complete": c["route_complete_launches"], "sparse": c["route_sparse_launches"], "general": c["route_general_launches"]},
        "note": "frac is against the datasheet peak of the named bound.  effective_stream_gbs is SURVEY 8(d)'s pairs x N/2 bytes figure: "
                "tiling makes it exceed any physical rate, it is not a roofline fraction.",
    }


def config4_tiles_leg(pkg, torch, device, samples, tile, ref_slice_variants, no_cpu):
    """BASELINE config 4 (`--r2-unphased inter-chr`, plink2_ld.cc:11082-11116; VcorMatrixThread :9518-9652) as SURVEY 8(d) sizes it: a
    fixed tile x tile CROSS-chromosome tile set -- `tile` variants of one chromosome against `tile` variants of another, every pair
    -- at `samples` founders, through the all-pairs plan's column-block call with the table writer's filter in the kernel epilogue
    (r^2 >= 0.2 (1 - 2^-44), the command's default), so that only the pairs that would be written leave the device."""
    n, m = samples, 2 * tile
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=device)
    eng.set_variants_matrix(m)
    # rows [0, tile): the first `tile` variants of chromosome 1; rows [tile, 2 tile): the first of chromosome 2 (generator indices far
    # apart: the planted LD never links the two sets, as on two real chromosomes)
    far = 5000000
    ptr, stride = eng.map_rows(0, m)
    pkg.synth_genotypes_device(SEED + 2, 0, tile, n, 0.0, ptr, stride)
    pkg.synth_genotypes_device(SEED + 2, far, tile, n, 0.0, ptr + tile * stride, stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.load_genotypes_device(0, m, ptr, stride, pkg.LDP_GENO_REF)
    count_ms = eng.counters()["ms_prepare"]
    thr = 0.2 * (1.0 - 2.0 ** -44)
    rows_per_call = 8192
    eng.r2_unphased_block_hits(thr, tile, 64, 0, tile, capacity=1 << 16)  # warm-up (plan upload, kernel load)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pairs, kms, hits, products = 0, 0.0, 0, 0
    for r0 in range(tile, m, rows_per_call):
        h = eng.r2_unphased_block_hits(thr, r0, min(rows_per_call, m - r0), 0, tile, capacity=1 << 22)
        c = eng.counters()
        pairs += c["candidate_pairs"]
        kms += c["ms_pair_kernel"]
        products += c["computed_pairs"] // 1024
        hits += len(h)
    wall = time.perf_counter() - t1
    ksteps = (n + 63) // 64
    row_bytes = ((n + 511) // 512) * 128
    mfma_tflops = products * ksteps * 65536 * 2.0 / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    compulsory = m * float(row_bytes)
    traffic, traffic_src, _ = pmc_traffic(n, m, 0.0, 0.0)
    res = {"what": "--r2-unphased inter-chr tile set: %d x %d cross-chromosome variant pairs at %d samples (complete data), every pair's r^2 computed, the default "
                   "--ld-window-r2 0.2 filter applied in the kernel epilogue (ldp_r2_unphased_block_hits, %d rows of second variants per call); rows resident, "
                   "count pass %.1f ms outside the timed region" % (tile, tile, n, rows_per_call, count_ms),
           "pairs": int(pairs), "pairs_passing_filter": int(hits), "wall_s": wall, "pairs_per_s_wall": pairs / wall if wall > 0 else None,
           "kernel_ms": kms, "pairs_per_s_kernel": pairs / (kms * 1e-3) if kms > 0 else None,
           "roofline": {"bound": "mfma", "achieved": mfma_tflops, "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_tflops / FP4_PEAK_TFLOPS,
                        "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_over_compulsory": (traffic / compulsory) if traffic else None,
                        "mfma_instructions": int(products * ksteps), "block_products": int(products),
                        "plan_efficiency": pairs / (products * 1024.0) if products else None,
                        "hbm": {"compulsory_bytes": compulsory, "compulsory_gbs": compulsory / (kms * 1e-3) / 1e9 if kms > 0 else 0.0, "peak_gbs": HBM_PEAK_GBS},
                        "effective_stream_gbs": pairs * (n / 2.0) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                        "kernel": "pair_mfma_wide_kernel / pair_mfma_kernel with the r^2 epilogue (one product per pair: complete data)"}}
    eng.close()
    torch.cuda.empty_cache()
    # the same command, both binaries end to end, on a slice the reference finishes in seconds: two chromosomes x ref_slice_variants / 2
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    cli_bin = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
    if (not no_cpu) and os.path.exists(ref_bin) and os.path.exists(cli_bin) and ref_slice_variants >= 4:
        half = ref_slice_variants // 2
        buf = torch.empty((2 * half, (n + 3) // 4), dtype=torch.uint8, device="cuda")
        pkg.synth_genotypes_device(SEED + 2, 0, half, n, 0.0, buf.data_ptr(), buf.shape[1])
        pkg.synth_genotypes_device(SEED + 2, far, half, n, 0.0, buf.data_ptr() + half * buf.shape[1], buf.shape[1])
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
        del buf
        chr_idx = np.repeat(np.arange(2, dtype=np.uint32), half)
        bps = np.tile(10000 + 290 * np.arange(half, dtype=np.uint32), 2)
        tmp = tempfile.mkdtemp(prefix="ldbench4_")
        try:
            write_plink1_fileset(os.path.join(tmp, "sample"), host, n, chr_idx, bps)
            cores = os.cpu_count() or 1
            walls, outs = {}, {}
            for name, cmd in (("reference_plink2", [ref_bin, "--bfile", "sample", "--r2-unphased", "inter-chr", "--threads", str(cores), "--out", "ref"]),
                              ("plink2_hip", [cli_bin, "--bfile", "sample", "--r2-unphased", "inter-chr", "--out", "hip"])):
                t2 = time.perf_counter()
                cp = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
                walls[name] = time.perf_counter() - t2
                outs[name] = cp.returncode
            same = False
            if outs["reference_plink2"] == 0 and outs["plink2_hip"] == 0:
                same = open(os.path.join(tmp, "ref.vcor"), "rb").read() == open(os.path.join(tmp, "hip.vcor"), "rb").read()
            sl_pairs = (2 * half) * (2 * half - 1) // 2
            res["reference_slice"] = {"variants": 2 * half, "samples": n, "pairs": sl_pairs, "vcor_files_identical": bool(same), "rc": outs,
                                      "e2e_wall_s": walls, "reference_pairs_per_s": sl_pairs / walls["reference_plink2"], "cores": cores,
                                      "what": "both binaries, `--r2-unphased inter-chr` (default filter) on two chromosomes x %d variants of the same generator, "
                                              "byte comparison of the .vcor tables" % half}
        finally:
            subprocess.call(["rm", "-rf", tmp])
    return res


def flatten_summary(out):
    """The driver's record keeps `roofline` and `cpu_baseline` but only their SCALAR fields (nested objects and the other top-level keys -- legs,
    stage_ms, power_and_clock, headline_bits_check -- are dropped).  So the numbers a reader of that record needs are repeated here as flat scalar
    keys: per leg the kernel, ms per step, the roofline fraction with its bound, the traffic multiple; the power / clock medians of the timed steps;
    the bits check; the end-to-end walls of both binaries on both file formats."""
    r = out.get("roofline") or {}
    legs = out.get("legs") or {}

    def put(prefix, leg):
        if not isinstance(leg, dict) or "error" in leg:
            r[prefix + "_error"] = (leg or {}).get("error", "missing")[:100] if isinstance(leg, dict) else "missing"
            return
        rf = leg.get("roofline") or {}
        for key, val in (("kernel", leg.get("kernel") or rf.get("kernel")), ("ms_per_step", leg.get("ms_per_step")), ("pair_kernels_ms", leg.get("pair_kernels_ms") or leg.get("kernel_ms")),
                         ("bound", rf.get("bound")), ("frac", rf.get("frac")), ("traffic_x_compulsory", rf.get("traffic_over_compulsory") or leg.get("traffic_over_compulsory"))):
            if val is not None:
                r["%s_%s" % (prefix, key)] = val

    if "config2" in legs:
        L = legs["config2"]
        put("leg_config2", L)
        for rate in ("0.001", "0.01"):
            M = L.get("missing_rate_" + rate) if isinstance(L, dict) else None
            if M:
                r["leg_config2_miss%s_kernel" % rate], r["leg_config2_miss%s_ms_per_step" % rate] = M.get("kernel"), M.get("ms_per_step")
                r["leg_config2_miss%s_x_complete" % rate] = M.get("vs_complete_data_step")
        cbl = (L.get("cpu_baseline") or {}) if isinstance(L, dict) else {}
        if cbl.get("value"):
            r["leg_config2_reference_pairs_per_s"], r["leg_config2_reference_identical"] = cbl.get("value"), cbl.get("prune_set_identical_to_hip")
    if "config5_density" in legs:
        L = legs["config5_density"]
        put("leg_config5_density", L)
        if isinstance(L, dict) and "error" not in L:
            r["leg_config5_density_x_complete"] = L.get("vs_complete_data_step_of_the_same_slice")
            r["leg_config3_density_complete_ms_per_step"] = (L.get("complete_data_step_of_the_same_slice") or {}).get("ms_per_step")
            r["leg_config5_density_reference_identical"] = (L.get("reference_slice") or {}).get("prune_set_identical_to_hip")
    for rate in ("0.001", "0.01"):
        M = (legs.get("config3_density_missing") or {}).get("missing_rate_" + rate)
        if M:
            r["leg_config3_miss%s_kernel" % rate], r["leg_config3_miss%s_ms_per_step" % rate] = M.get("kernel"), M.get("ms_per_step")
            r["leg_config3_miss%s_x_complete" % rate], r["leg_config3_miss%s_recounted_pairs" % rate] = M.get("vs_complete_data_step"), M.get("pairs_counted_exactly")
            r["leg_config3_miss%s_traffic_x_compulsory" % rate] = M.get("traffic_over_compulsory")
    if "config4_tiles" in legs:
        put("leg_config4_tiles", legs["config4_tiles"])
    pc = out.get("power_and_clock") or {}
    for key in ("socket_power_w_median", "socket_power_cap_w", "shader_clock_mhz_median"):
        if pc.get(key) is not None:
            r["timed_steps_" + key] = pc[key]
    hb = out.get("headline_bits_check") or {}
    if hb:
        r["headline_bits_check_identical"] = hb.get("identical", hb.get("checked"))
    cb = out.get("cpu_baseline") or {}
    e = cb.get("e2e_wall_s") or {}
    if e:
        ph = e.get("plink2_hip_phases") or {}
        cb.update({"e2e_fixed_width_reference_s": e.get("reference_plink2"), "e2e_fixed_width_plink2_hip_s": e.get("plink2_hip"), "e2e_fixed_width_speedup": e.get("speedup"),
                   "e2e_fixed_width_files_identical": e.get("files_identical"), "e2e_fixed_width_file_to_hbm_gbs": ph.get("file_to_hbm_gbs"), "e2e_fixed_width_pgen_bytes": e.get("pgen_bytes")})
        v = e.get("variable_width") or {}
        if v:
            vh = v.get("plink2_hip") or {}
            cb.update({"e2e_variable_width_reference_s": v.get("reference_plink2_wall_s"), "e2e_variable_width_plink2_hip_s": vh.get("wall_s"), "e2e_variable_width_speedup": v.get("speedup"),
                       "e2e_variable_width_files_identical": v.get("files_identical_to_reference_on_the_same_file"), "e2e_variable_width_pgen_bytes": v.get("pgen_bytes"),
                       "e2e_variable_width_bytes_vs_fixed": v.get("bytes_vs_fixed_width"), "e2e_variable_width_make_pgen_s": v.get("make_pgen_s"),
                       "e2e_variable_width_file_to_hbm_s": (vh.get("phases") or {}).get("file_to_hbm_s")})
    n = out.get("e2e_n_gpus") or {}
    if n:
        hip = n.get("plink2_hip") or {}
        r["e2e_plink2_hip_gpus"], r["e2e_plink2_hip_gpus_wall_s"], r["e2e_plink2_hip_gpus_rc"] = n.get("gpus"), hip.get("wall_s"), hip.get("rc")
        r["e2e_plink2_hip_gpus_file_to_hbm_s"] = (hip.get("phases") or {}).get("file_to_hbm_s")
        r["e2e_plink2_hip_gpus_identical_to_one_gpu"] = n.get("identical_to_one_gpu")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(CONFIGS), default="config3", help="default: config3's density (the metric's), weak scaling at 1.25M variants per GPU")
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--variants", type=int, default=None, help="variants of the genome (total with --strong, per GPU otherwise)")
    ap.add_argument("--window-kb", type=float, default=None)
    ap.add_argument("--r2", type=float, default=None)
    ap.add_argument("--spacing", type=int, default=None, help="bp between consecutive variants")
    ap.add_argument("--missing-rate", type=float, default=None, help="missing calls (MCAR, every variant); default 0, or the named workload's (config5: 0.05)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the named configuration's genome in total, sharded over the ranks")
    ap.add_argument("--weak", action="store_true", help="weak scaling (the default): --variants per GPU on one genome")
    ap.add_argument("--cpu-sample-variants", type=int, default=0, help="0 = 440,000 up to 100k samples, 11,000 beyond")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the config2 / config5-density / config4-tiles legs and the ceiling microbenchmarks")
    ap.add_argument("--no-cli-compare", action="store_true", help="do not time plink2-hip end-to-end on the CPU-baseline sample files")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes over a one-step child run); replay profiles/ instead")
    ap.add_argument("--no-e2e", action="store_true", help="skip the measured chr22-sized end-to-end run of both binaries (22 GB fileset, minutes of reference time)")
    ap.add_argument("--no-e2e-variable", action="store_true", help="... or only its second half: the same fileset as the reference's default variable-width .pgen")
    ap.add_argument("--e2e-variants", type=int, default=0, help="variants of that fileset (0 = a chr22-sized share of the named genome: 176,765 of 10M)")
    ap.add_argument("--leg-variants", type=int, default=120000, help="variants of the config5-density leg")
    ap.add_argument("--tile", type=int, default=65536, help="side of the config4_tiles leg's cross-chromosome tile set")
    ap.add_argument("--only-config4", action="store_true", help="run the config4_tiles leg alone and print it as the line (the PMC profile of that shape: tools/profile.sh)")
    ap.add_argument("--option", action="append", default=[], help="name=value: a per-engine kernel switch (ldp_debug_set_option) for the main workload, for experiments")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import importlib
    distmod = importlib.import_module("plink_ng_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the hot path)")
    # LDP_BENCH_ALIAS_DEVICES=1 (tests, one-GPU boxes): the N ranks of `--gpus N` share the devices there are, and the exchange goes
    # over gloo (RCCL refuses a device twice) -- the rank / shard / exchange code of the N-GPU line runs, its numbers are NOT a
    # measurement of N GPUs and the line says so
    alias = bool(os.environ.get("LDP_BENCH_ALIAS_DEVICES")) and (world > 1)
    if alias:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the RCCL path is used even for a single rank, so the exchange code is
    # exercised on a 1-GPU box as well
    use_dist = (world > 1) or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    coll_device = "cpu" if alias else "cuda"
    if use_dist:
        if alias:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.only_config4:
        res = config4_tiles_leg(pkg, torch, local_rank, CONFIGS["config3"]["samples"], args.tile, 0, True)
        print(json.dumps({"metric": "variant-pairs/s (--r2-unphased inter-chr tile set, kernel)", "value": res["pairs_per_s_kernel"], "unit": "variant-pairs/s", "n_gpus": 1,
                          "steps": 1, "warmup": 0, "ms_per_step": res["kernel_ms"], "higher_is_better": True, "data": "synthetic",
                          "config": {"workload": "config4_tiles: %s; missing rate 0" % res["what"], "samples": CONFIGS["config3"]["samples"], "variants_rank0": 2 * args.tile,
                                     "window_kb": 0.0},
                          "roofline": res["roofline"], "legs": {"config4_tiles": res}}))
        return
    name = args.workload
    cfg = dict(CONFIGS[name])
    strong = args.strong and not args.weak
    if not strong:
        cfg["variants"] = PER_GPU_VARIANTS[name]
    for k, v in (("samples", args.samples), ("variants", args.variants), ("window_kb", args.window_kb), ("r2", args.r2), ("spacing", args.spacing)):
        if v is not None:
            cfg[k] = v
    if args.missing_rate is None:
        args.missing_rate = cfg.get("missing_rate", 0.0)
    multi_frac = cfg.get("multiallelic_frac", 0.0)
    per_gpu_variants = cfg["variants"] if not strong else None
    if not strong:
        cfg["variants"] = cfg["variants"] * world
    main_options = {}
    for kv in args.option:
        k, v = kv.split("=", 1)
        main_options[k] = float(v)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(wl, steps, warmup, on_start=None):
        def one():
            words, ctrs = wl.step()
            if use_dist:
                # the one exchange step: all_gather of the per-rank removed bitmaps (RCCL over xGMI), OR-ed on the device
                return distmod.allgather_bitmaps(words, world, device=coll_device), ctrs
            return words, ctrs
        for _ in range(warmup):
            one()
        sync()
        if on_start:
            on_start()
        t0 = time.perf_counter()
        ks, removed = [], None
        for _ in range(steps):
            removed, ctrs = one()
            ks.append(sum_counters(ctrs))
        sync()
        return time.perf_counter() - t0, ks, removed

    def dtype_of(roofline):
        return ("fp4 (E2M1 allele counts 0/1/2 at block scale 2; -2/0/+2 at scale 1/2 where rows miss calls) x fp4 -> f32 integer-exact MFMA accumulation + f64 predicate"
                if roofline["kernel"].startswith("pair_mfma")
                else "u32 popcount + f64 predicate")

    def stage_ms(cmean, image_bytes):
        return {"count_pass_codes_kernel": cmean["ms_prepare"], "pair_kernels": cmean["ms_pair_kernel"], "pair_mfma_kernels": cmean["ms_pair_mfma"],
                "pair_mfma_missing_call_kernels": cmean["ms_pair_mfma_general"], "pair_tiles_kernels_popcount": cmean["ms_pair_fast"] + cmean["ms_pair_general"],
                "host_replay": cmean["ms_replay"],
                "note": "the count pass (HBM-bound: reads N/4 bytes per variant, writes only the records) and the pair kernels run back to back; "
                        "count pass: %.0f GB/s" % ((image_bytes / (cmean["ms_prepare"] * 1e-3) / 1e9) if cmean["ms_prepare"] > 0 else 0.0)}

    # (config5: runs of 400 consecutive multiallelic variants, one ldp_load_pgen_records call each -- a loader hands over records in batches --, 2 % of the rank's share)
    main_multi = (max(1, int(round(multi_frac * (per_gpu_variants or cfg["variants"] // world) / 400.0))), 400) if multi_frac > 0 else None
    wl = Workload(pkg, torch, cfg, args.missing_rate, rank, world, local_rank, main_options, main_multi)
    smi = support.SmiSampler() if rank == 0 else None

    def open_window():
        if smi:
            smi.window = "main"   # (the sampler's window covers the timed steps only)
    elapsed, ks, removed = timed(wl, args.steps, args.warmup, open_window)
    if smi:
        smi.window = None
        smi.stop()
    ctr = ks[-1]
    removed = distmod.bitmap_to_mask(removed.cpu().numpy() if use_dist else removed, cfg["variants"])
    per_rank_pairs = [ctr["candidate_pairs"]]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([ctr["candidate_pairs"]], dtype=torch.int64, device=coll_device)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        per_rank_pairs = [int(x.item()) for x in allp]
    total_pairs = sum(per_rank_pairs)
    # ---- self-check of an N-rank run (every rank takes part; rank 0 reports): the world really has N ranks, every rank sits on a
    # device of its own, and the collective backend saw every rank
    selfcheck = None
    if use_dist:
        props = torch.cuda.get_device_properties(local_rank)
        ident = "%s/%s" % (getattr(props, "uuid", None) or getattr(props, "pci_bus_id", local_rank), getattr(props, "pci_device_id", ""))
        ids = [None] * world
        dist.all_gather_object(ids, (rank, local_rank, str(ident)))
        seen = torch.zeros(world, dtype=torch.int64, device=coll_device)
        seen[rank] = 1
        dist.all_reduce(seen)   # (through the collective backend of the timed exchange: RCCL, or gloo under LDP_BENCH_ALIAS_DEVICES)
        selfcheck = {"world_size": dist.get_world_size(), "world_size_is_n_gpus": bool(dist.get_world_size() == args.gpus),
                     "backend": dist.get_backend(), "rccl_ranks_seen": int((seen > 0).sum().item()), "ranks": [list(x) for x in ids],
                     "device_ordinals_distinct": bool(len({x[1] for x in ids}) == world), "devices_distinct": bool(len({x[2] for x in ids}) == world)}
        selfcheck["ok"] = bool(selfcheck["world_size_is_n_gpus"] and selfcheck["rccl_ranks_seen"] == world and (alias or selfcheck["device_ordinals_distinct"]))
        if rank == 0 and not selfcheck["ok"]:
            sys.stderr.write("bench.py: the %d-rank run does not look like %d ranks on %d devices: %r\n" % (world, world, world, selfcheck))   # (recorded in the line, never fatal)

    out = None
    if rank == 0:
        mean = lambda key: float(np.mean([k[key] for k in ks])) if ks else 0.0
        cmean = {k: mean(k) for k in ks[-1]}
        ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
        value = total_pairs * args.steps / elapsed
        roofline = pair_roofline(cmean, cfg["samples"], wl.local_ct, args.missing_rate, wl.local_ct, cfg["window_kb"], main_options)
        out = {
            "metric": "variant-pairs/s (--indep-pairwise, whole job)", "value": value, "unit": "variant-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": dtype_of(roofline),
            "data": ("model (LDP_BENCH_ALIAS_DEVICES: %d ranks share %d device(s), gloo exchange: the N-rank code path runs, its numbers are NOT a measurement of %d GPUs)"
                     % (world, torch.cuda.device_count(), world)) if alias else
                    ("synthetic" if wl.resident else "model (synthetic shapes; a rank's share exceeds HBM, so every chromosome of it is a copy of ONE generated chromosome: NOT a "
                                                     "measurement of the named workload)"),
            "config": {"workload": "%s: synthetic %d samples x %d biallelic variants (%s), one genome of 22 autosomes at %d bp spacing, "
                                   "--indep-pairwise %gkb %g, missing rate %g, chromosomes LPT-sharded over %d rank(s); %s" %
                                   (name, cfg["samples"], cfg["variants"], "total, strong scaling" if strong else "%d per GPU, weak scaling" % per_gpu_variants,
                                    cfg["spacing"], cfg["window_kb"], cfg["r2"], args.missing_rate, world,
                                    "2-bit rows resident in HBM, counted in place each step (no conversion pass, no second image)" if wl.resident else
                                    "a rank's share (%.0f GB) exceeds HBM: one engine per chromosome, each chromosome's rows copied inside the step from ONE resident chromosome's "
                                    "worth of generated rows (every chromosome of the share holds the same genotypes)" % (wl.image_bytes / 1e9)),
                       "family": "every --gpus N of this default reports the SAME workload: BASELINE.json configs[2]'s samples / density / window / threshold at %s; at N = 8 "
                                 "the weak line is configs[2] itself (10M variants)" % ("the metric's 10,000,000 variants in total" if strong else "1,250,000 variants per GPU"),
                       "samples": cfg["samples"], "variants_total": cfg["variants"], "variants_rank0": wl.local_ct, "window_kb": cfg["window_kb"], "r2": cfg["r2"],
                       "subcontigs": len(wl.subs), "candidate_pairs_total": total_pairs, "candidate_pairs_per_rank": per_rank_pairs,
                       "shard_imbalance_max_over_mean": (max(per_rank_pairs) / (total_pairs / world)) if total_pairs else 1.0,
                       "pairs_above_threshold_rank0": ctr["pred_true"], "above_threshold_pairs_consumed_by_replay_rank0": ctr["replay_pairs"],
                       "variants_removed": int(removed.sum()), "resident": bool(wl.resident), "image_gb_rank0": wl.image_bytes / 1e9,
                       "engine_options": main_options,
                       "multiallelic_variants_rank0": (wl.multi or {}).get("variants", 0)},
            "roofline": roofline,
            "stage_ms": stage_ms(cmean, wl.image_bytes),
            "power_and_clock": smi.summary("main") if smi else None,
        }
        if world == 1 and not args.no_legs:   # (profiling runs -- tools/profile.sh, the in-run PMC passes -- carry the timed step's kernels only)
            try:
                out["headline_bits_check"] = headline_bits_check(pkg, torch, wl, removed)
            except Exception as ex:  # pragma: no cover
                out["headline_bits_check"] = {"checked": False, "why": str(ex)[:300]}
        if selfcheck is not None:
            out["multi_rank_selfcheck"] = selfcheck
    subset = None
    if rank == 0 and world > 1:
        subset = (wl.founder_ct, wl.window_bp, cfg["r2"], wl.chr_idx, wl.bps, wl.m_total)
    wl.close()
    if subset is not None:
        # ... and the combined bitmap of the N shards equals ONE engine's result on the last three chromosomes (chr20-22: a contiguous
        # range of the genome, 5 % of it; subcontigs are independent), computed here on rank 0 after its own share has left the device
        try:
            n_f, w_bp, r2v, chr_all, bps_all, m_tot = subset
            first = int(np.searchsorted(chr_all, 19, side="left"))
            ln = m_tot - first
            t0 = time.perf_counter()
            e1 = pkg.LdPruneEngine(n_f, w_bp, 1, True, r2v, device=local_rank)
            e1.set_variants(chr_all[first:], bps_all[first:])
            ptr, stride = e1.map_rows(0, ln)
            pkg.synth_genotypes_device(SEED, first, ln, n_f, args.missing_rate, ptr, stride)
            torch.cuda.synchronize()
            e1.load_genotypes_device(0, ln, ptr, stride, pkg.LDP_GENO_REF)
            one = np.asarray(e1.run(), dtype=bool)
            e1.close()
            torch.cuda.empty_cache()
            out["multi_rank_selfcheck"]["three_chromosome_subset"] = {
                "first_variant": first, "variants": int(ln), "identical_to_single_engine": bool(np.array_equal(one, np.asarray(removed[first:], dtype=bool))),
                "removed": int(one.sum()), "seconds": time.perf_counter() - t0,
                "what": "chr20-22 of the genome run alone through ONE engine on rank 0 against the stitched bitmap of the %d ranks" % world}
        except Exception as ex:  # pragma: no cover
            out["multi_rank_selfcheck"]["three_chromosome_subset"] = {"error": str(ex)[:300]}
    if rank == 0 and world == 1 and not (args.no_pmc or args.no_legs or args.no_cpu_baseline):
        # roofline.traffic of the headline workload, measured here and now (the device is free: the share has just been released)
        wargs = ["--workload", name] + (["--strong"] if strong else [])
        for flag, v in (("--samples", args.samples), ("--variants", args.variants), ("--window-kb", args.window_kb), ("--r2", args.r2), ("--spacing", args.spacing)):
            if v is not None:
                wargs += [flag, str(v)]
        if args.missing_rate:
            wargs += ["--missing-rate", repr(args.missing_rate)]
        for kv in args.option:
            wargs += ["--option", kv]
        t_pmc = time.perf_counter()
        tb, tk, tnote = support.pmc_traffic_in_run(__file__, wargs)
        r = out["roofline"]
        if tb:
            comp = r["hbm"]["compulsory_bytes_per_step"]
            r["traffic_replayed_from_profiles"] = {"traffic": r["traffic"], "traffic_source": r["traffic_source"]}
            r["traffic"], r["traffic_source"], r["traffic_over_compulsory"] = tb, tnote, (tb / comp) if comp else None
            r["traffic_pair_kernels"] = tk
            r["traffic_measurement_s"] = time.perf_counter() - t_pmc
        else:
            r["traffic_in_run_error"] = tnote
    e2e_box = [None]

    def start_e2e():
        # the metric's wall-clock leg: materialise the chr22-sized fileset and let the reference run (twice: fixed-width file, variable-width file) beside
        # whatever this script does next.  Called AFTER the legs whose numbers are walls per step: two reference processes and a 256-thread --make-pgen
        # on the host stretched those walls by 10-70 % when they ran beside them (profiles/r06_experiments.md); kernel times never moved.
        if e2e_box[0] is not None or not (rank == 0 and world == 1 and (not args.no_cpu_baseline) and (not args.no_e2e) and (not args.no_cli_compare) and cfg["samples"] > 100000):
            return
        try:
            e2e_box[0] = support.E2EChr22(pkg, torch, cfg, args.e2e_variants or int(round(CONFIGS[name]["variants"] * CHR22_FRACTION)), SEED, genome_layout)
            e2e_box[0].start(reference=True, variable_width=not args.no_e2e_variable)
        except Exception as ex:  # pragma: no cover
            e2e_box[0] = None
            out["e2e_error"] = str(ex)[:300]

    if rank == 0 and world == 1 and not args.no_legs:
        legs = {}
        half = max(2, min(args.steps, 10) // 2)

        def leg(cfg_l, missing, options, steps, multiallelic=None):
            w = Workload(pkg, torch, cfg_l, missing, 0, 1, local_rank, options, multiallelic)
            el, kk, rem = timed(w, steps, 1)
            c = {k: float(np.mean([q[k] for q in kk])) for k in kk[-1]}
            r = pair_roofline(c, cfg_l["samples"], w.local_ct, missing, cfg_l["variants"], cfg_l["window_kb"], options)
            res = {"ms_per_step": 1000.0 * el / steps, "pairs_per_s": c["candidate_pairs"] * steps / el, "count_pass_ms": c["ms_prepare"], "pair_kernels_ms": c["ms_pair_kernel"],
                   "kernel": r["kernel"], "routes": r["routes"], "pairs_counted_exactly": int(c["sparse_exact_pairs"]), "candidate_pairs": int(c["candidate_pairs"]),
                   "variants_removed": int(distmod.bitmap_to_mask(rem, cfg_l["variants"]).sum()), "roofline": r, "stage_ms": stage_ms(c, w.image_bytes)}
            if w.multi:
                eng = w.engines[0][0]
                ok = True
                for st, nm, sm, sq in w.multi["checks"]:
                    rec = eng.variant_recs(st, 1)[0]
                    ok = ok and (int(rec["nm_ct"]), int(rec["sum"]), int(rec["ssq"])) == (nm, sm, sq)
                res["multiallelic"] = {"variants": w.multi["variants"], "runs": len(w.multi["runs"]), "record_bytes": w.multi["nbytes"],
                                       "ms_per_step_in_ldp_load_pgen_records": float(np.mean(w.multi["ms_calls"][1:] or w.multi["ms_calls"])),
                                       "records_match_numpy_collapse": bool(ok),
                                       "what": "variants with two ALT alleles arriving as .pgen records resident in HBM (main track + aux track 1), decoded and "
                                               "collapsed major-vs-rest on the device inside every step (ldp_load_pgen_records, one call per run of consecutive variants); their "
                                               "records are checked against a numpy restatement of the collapse here -- the comparison of the collapse with the REFERENCE lives in "
                                               "tests/test_cli.py::test_cli_multiallelic_collapse_matches_reference and tests/test_pgen_device_decode.py"}
            w.close()
            return res

        # (a) BASELINE.json configs[1] (the narrow-band shape): 50,000 x 1,000,000, 200kb 0.5, with its own roofline, early termination
        # off, rows with 0.1 % / 1 % missing calls, and the reference on a 440,000-variant sample (both binaries end to end)
        try:
            c2 = dict(CONFIGS["config2"])
            steps2 = max(5, min(args.steps, 20))
            L = leg(c2, 0.0, {}, steps2)
            L["what"] = "BASELINE.json configs[1]: %d samples x %d variants at %d bp, --indep-pairwise %gkb %g; rows resident, count pass inside the step" % (
                c2["samples"], c2["variants"], c2["spacing"], c2["window_kb"], c2["r2"])
            L["exhaustive"] = {k: v for k, v in leg(c2, 0.0, {"early_exit": 0}, half).items() if k in ("ms_per_step", "pair_kernels_ms", "kernel")}
            L["exhaustive"]["what"] = "early termination off (ldp_debug_set_option early_exit 0)"
            for rate in (0.001, 0.01):
                M = leg(c2, rate, {}, half)
                L["missing_rate_%g" % rate] = {k: M[k] for k in ("ms_per_step", "pair_kernels_ms", "kernel", "routes", "pairs_counted_exactly", "variants_removed")}
                L["missing_rate_%g" % rate]["vs_complete_data_step"] = M["ms_per_step"] / L["ms_per_step"]
                L["missing_rate_%g" % rate]["mfma"] = M["roofline"]["mfma"]
                for key in ("traffic", "traffic_over_compulsory", "traffic_source"):   # (replayed like the main leg's: profiles/*_pmc_traffic.json of this workload)
                    L["missing_rate_%g" % rate][key] = M["roofline"][key]
            legs["config2"] = L
        except Exception as e:  # pragma: no cover
            legs["config2"] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()
        # (b) config 5's density on a slice: 5 % missing calls in every variant, 2 % of the variants multiallelic (runs of 400 consecutive
        # variants, one ldp_load_pgen_records call each: a loader hands over records in batches)
        if args.leg_variants > 0 and name == "config3":
            c5 = dict(CONFIGS["config3"], variants=args.leg_variants)
            try:
                L = leg(c5, 0.05, {}, 3, (max(1, c5["variants"] // 20000), 400))
                L["what"] = ("%d samples x %d variants at %d bp, --indep-pairwise %gkb %g, 5 %% missing calls in every variant, 2 %% of the variants multiallelic (see "
                             "`multiallelic`); rows resident, count pass inside the step" % (c5["samples"], c5["variants"], c5["spacing"], c5["window_kb"], c5["r2"]))
                C = leg(c5, 0.0, {}, 3)
                L["vs_complete_data_step_of_the_same_slice"] = L["ms_per_step"] / C["ms_per_step"]
                L["complete_data_step_of_the_same_slice"] = {k: C[k] for k in ("ms_per_step", "pair_kernels_ms", "kernel", "variants_removed")}
                L["complete_data_step_of_the_same_slice"]["roofline"] = {k: C["roofline"][k] for k in ("bound", "achieved", "frac", "traffic", "traffic_source", "traffic_over_compulsory")}
                legs["config5_density"] = L
                # (b') the same slice with a FEW missing calls -- what real call sets look like: 0.1 % stays on the 8 x 8 tiles (the tile
                # kernel's SPARSE instantiation), 1 % is beyond the interval path's limit (0.5 % on average) and takes the four-product quarter tiles
                M3 = {"what": "the config-3 density slice (%d x %d, %gkb %g) with missing calls in every variant; vs_complete_data_step is against the same slice's "
                              "complete-data step (%.2f ms)" % (c5["samples"], c5["variants"], c5["window_kb"], c5["r2"], C["ms_per_step"])}
                for rate in (0.001, 0.01):
                    Q = leg(c5, rate, {}, 3)
                    M3["missing_rate_%g" % rate] = {k: Q[k] for k in ("ms_per_step", "pair_kernels_ms", "kernel", "routes", "pairs_counted_exactly", "variants_removed")}
                    M3["missing_rate_%g" % rate].update({"vs_complete_data_step": Q["ms_per_step"] / C["ms_per_step"], "mfma": Q["roofline"]["mfma"],
                                                          "traffic": Q["roofline"]["traffic"], "traffic_over_compulsory": Q["roofline"]["traffic_over_compulsory"],
                                                          "traffic_source": Q["roofline"]["traffic_source"], "same_prune_set_as_complete_data_is_not_expected": True})
                legs["config3_density_missing"] = M3
            except Exception as e:  # pragma: no cover  (e.g. a smaller GPU)
                legs.setdefault("config5_density", {"error": str(e)[:300]})
            torch.cuda.empty_cache()
        # ---- from here on nothing is timed by the wall: the end-to-end references start now and run beside the rest ----
        start_e2e()
        if ("error" not in legs.get("config2", {"error": 1})) and not args.no_cpu_baseline:
            c2 = dict(CONFIGS["config2"])
            legs["config2"]["cpu_baseline"] = cpu_baseline(pkg, torch, c2["samples"], 440000, c2["spacing"], c2["window_kb"], c2["r2"], 0.0, cli_compare=not args.no_cli_compare)
        if ("error" not in legs.get("config5_density", {"error": 1})) and not args.no_cpu_baseline:
            c5 = dict(CONFIGS["config3"], variants=args.leg_variants)
            try:
                legs["config5_density"]["reference_slice"] = config5_reference_slice(pkg, torch, c5["samples"], args.cpu_sample_variants or 11000, c5["spacing"], c5["window_kb"], c5["r2"], 0.05)
            except Exception as ex:  # pragma: no cover
                legs["config5_density"]["reference_slice"] = {"error": str(ex)[:300]}
        # (c) config 4: the cross-chromosome tile set of --r2-unphased inter-chr
        if args.tile > 0:
            try:
                legs["config4_tiles"] = config4_tiles_leg(pkg, torch, local_rank, CONFIGS["config3"]["samples"], args.tile, 1024, args.no_cpu_baseline)
            except Exception as e:  # pragma: no cover
                legs["config4_tiles"] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
        out["legs"] = legs
        ceil = measured_ceilings()
        out["roofline"]["measured_ceilings"] = ceil
        if ceil.get("mfma_fp4_tflops_random_operands"):
            out["roofline"]["mfma"]["frac_of_measured_rate_random_operands"] = out["roofline"]["mfma"]["executed_tflops"] / ceil["mfma_fp4_tflops_random_operands"]
        if ceil.get("hbm_read_gbs"):
            out["roofline"]["hbm"]["frac_of_measured_read_rate"] = out["roofline"]["hbm"]["compulsory_gbs"] / ceil["hbm_read_gbs"]
        if ceil.get("stage_loop_probe_tflops") and ("wide" in str(out["roofline"].get("kernel"))) and not args.missing_rate:  # (the probe is the complete-data tile kernel's loop)
            # flat scalars (the driver's record keeps those): the stage loop's rate on THIS box in THIS run, and the kernel against it
            out["roofline"]["stage_loop_probe_frac_of_peak"] = ceil["stage_loop_probe_tflops"] / out["roofline"]["mfma"]["peak_tflops"]
            out["roofline"]["kernel_over_stage_loop_probe"] = out["roofline"]["mfma"]["executed_tflops"] / ceil["stage_loop_probe_tflops"]
            out["roofline"]["mfma"]["frac_of_stage_loop_probe"] = out["roofline"]["kernel_over_stage_loop_probe"]

    start_e2e()   # (no legs in this run: start it here)
    e2e = e2e_box[0]
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            m = args.cpu_sample_variants or (440000 if cfg["samples"] <= 100000 else 11000)  # ~2-20 s of reference time either way
            cb = cpu_baseline(pkg, torch, cfg["samples"], m, cfg["spacing"], cfg["window_kb"], cfg["r2"], args.missing_rate, cli_compare=not args.no_cli_compare)
            if e2e is not None:
                # the wall-clock half of the metric, measured on the chr22-sized fileset; the small slice above stays as the parity check of the
                # timed engine's generator (prune_set_identical_to_hip) and moves to `slice`
                try:
                    big = e2e.finish()
                except Exception as ex:  # pragma: no cover  (never let this leg take the line with it)
                    big = {"error": str(ex)[:300]}
                ref = big.get("reference_plink2") or {}
                hip = big.get("plink2_hip") or {}
                if ref.get("wall_s") and hip.get("wall_s") and ref.get("rc") == 0 and hip.get("rc") == 0:
                    cb["slice"] = {k: cb.pop(k) for k in ("e2e_wall_s", "sample", "wall_s", "value", "removed") if k in cb}
                    cb["e2e_wall_s"] = {"reference_plink2": ref["wall_s"], "plink2_hip": hip["wall_s"], "speedup": big.get("speedup"), "variants": big.get("variants"),
                                        "samples": big["samples"], "files_identical": big.get("files_identical"), "plink2_hip_phases": hip.get("phases"),
                                        "plink2_hip_wall_s_runs": hip.get("wall_s_runs"), "reference_compute_threads": ref.get("compute_threads"),
                                        "fileset": big.get("fileset"), "what": big.get("what"), "reference_note": ref.get("note"),
                                        "pgen_bytes": getattr(e2e, "file_bytes", None), "variable_width": big.get("variable_width")}
                    cb["value"] = big.get("reference_candidate_pairs_per_s")
                    cb["wall_s"] = ref["wall_s"]
                    if ref.get("compute_threads"):
                        cb["cores"] = ref["compute_threads"] + 1
                    cb["sample"] = ("%d variants x %d samples of the same generator as a fixed-width .pgen (a chr22-sized share of the metric's genome, %s): reference plink2 "
                                    "AVX2 end-to-end wall %.1f s, %d candidate pairs" % (big["variants"], big["samples"], big.get("fileset", ""), ref["wall_s"],
                                                                                         (hip.get("phases") or {}).get("candidate_pairs", 0)))
                else:
                    cb["e2e_chr22_measurement"] = big   # (incomplete: keep what there is, the slice's numbers stay in place)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = {"value": None, "unit": "variant-pairs/s", "cores": 0, "kind": "reference",
                                   "sample": "measured at N=1 only", **host_description()}
        if world > 1 and (not args.no_e2e) and cfg["samples"] > 100000:
            # the wall-clock half of the metric at N GPUs: `plink2-hip --gpus N` (one feeding thread per engine, each next to its device, the shards' rows
            # crossing N PCIe links at once) end to end on the chr22-sized fileset, and `--gpus 1` on the same files beside it.  The other ranks wait at the
            # final barrier with their shares still resident (a chr22-sized shard is 22 GB / N per device).
            try:
                e2n = support.E2EChr22(pkg, torch, cfg, args.e2e_variants or int(round(CONFIGS[name]["variants"] * CHR22_FRACTION)), SEED, genome_layout)
                e2n.start(reference=False, variable_width=False)
                tmpdir = e2n.tmp
                one = support.run_plink2_hip(tmpdir, "g", e2n.kb, cfg["r2"], "one", gpus=1) if tmpdir else None
                big = e2n.finish(gpus=world, compare_with="one")   # (removes the fileset)
                if one is not None:
                    big["plink2_hip_one_gpu"] = {k: one[k] for k in ("wall_s", "wall_s_runs", "rc", "phases")}
                    big["speedup_over_one_gpu"] = (one["wall_s"] / big["plink2_hip"]["wall_s"]) if (one.get("wall_s") and (big.get("plink2_hip") or {}).get("wall_s")) else None
                out["e2e_n_gpus"] = big
            except Exception as ex:  # pragma: no cover
                out["e2e_n_gpus"] = {"error": str(ex)[:300]}
        try:
            flatten_summary(out)
        except Exception as ex:  # pragma: no cover  (a summary must never take the line with it)
            out["roofline"]["summary_error"] = str(ex)[:200]
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
