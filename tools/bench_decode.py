#!/usr/bin/env python3
"""Device-side .pgen record decode (ldp_load_pgen_records) against the host decoder, on files the REFERENCE writes:
(1) a variable-width .pgen of the bench generator's genotypes (`plink2 --make-pgen`: LD-compressed, difflist, one-bit and plain
records), (2) a VCF-imported fileset whose variants have two or three ALT alleles (auxiliary track 1; major-vs-rest collapse).
Times per variant set: records resident in HBM -> rows counted in the engine (decode + count pass), the count pass alone on
already decoded rows, the host decoder's threads, and the same call with the file's bytes in host memory (PCIe inside).
One JSON line per file."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def best_of(f, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best


def engine(pkg, n, m):
    eng = pkg.LdPruneEngine(n, 200, 1, True, 0.5, order=2, device=0)
    chr_idx, bps = bench.genome_layout(m, 1, 2875)
    eng.set_variants(chr_idx, bps)
    return eng


def measure(pkg, torch, path, allele_cts, label, extra):
    f = pkg.PgenFile(path)
    m, n = f.variant_ct, f.sample_ct
    recs, _ = f.record_index()
    types = {}
    for q in range(m):
        t = int(recs[q].vrtype) & 15
        types[t] = types.get(t, 0) + 1
    ptr, nbytes = f.file_bytes()
    host_bytes = np.ctypeslib.as_array((pkg.ctypes.c_uint8 * nbytes).from_address(ptr))
    dev_bytes = torch.from_numpy(host_bytes.copy()).cuda()
    res = {"file": label, "samples": n, "variants": m, "file_bytes": nbytes, "row_bytes": m * ((n + 3) // 4), "record_types": types, **extra}
    eng = engine(pkg, n, m)

    def dev_call():
        eng.load_pgen_records(0, f, allele_cts=allele_cts, location=pkg.LDP_MEM_DEVICE, device_bytes=dev_bytes.data_ptr())
        torch.cuda.synchronize()   # (the engine has its own stream: wait for the device, the count pass included)

    dev_call()
    res["device_decode_plus_count_ms"] = 1e3 * best_of(dev_call)
    want = eng.variant_recs().copy()

    def host_bytes_call():
        eng.load_pgen_records(0, f, allele_cts=allele_cts)
        torch.cuda.synchronize()

    res["same_call_bytes_in_host_memory_ms"] = 1e3 * best_of(host_bytes_call)
    # the count pass alone: rows already decoded, resident in HBM
    rows_host = f.read(threads=0)
    res["host_decoder_all_threads_ms"] = 1e3 * best_of(lambda: f.read(threads=0), reps=2)
    res["host_decoder_one_thread_ms"] = 1e3 * best_of(lambda: f.read(threads=1), reps=1)
    rows_dev = torch.from_numpy(rows_host).cuda()
    biallelic = allele_cts is None or int(np.max(allele_cts)) <= 2

    def count_only():
        eng.load_genotypes_device(0, m, rows_dev.data_ptr(), rows_host.shape[1], pkg.LDP_GENO_REF)
        torch.cuda.synchronize()

    count_only()
    res["count_pass_on_decoded_rows_ms"] = 1e3 * best_of(count_only)
    if biallelic:
        got = eng.variant_recs()
        res["records_identical_to_host_decoded_rows"] = bool(all(np.array_equal(got[k], want[k]) for k in ("nm_ct", "sum", "ssq", "flags")))
    else:
        # host side of the collapse: the allele pairs of every multiallelic variant (the CLI's host path reads them one by one)
        multi = [v for v in range(m) if allele_cts[v] > 2]
        t0 = time.perf_counter()
        for v in multi[:200]:
            f.read_alleles(v, int(allele_cts[v]) - 1)
        res["host_allele_pairs_ms_per_variant"] = 1e3 * (time.perf_counter() - t0) / max(1, min(len(multi), 200))
        res["multiallelic_variants"] = len(multi)
    dec = res["device_decode_plus_count_ms"] - res["count_pass_on_decoded_rows_ms"]
    res["device_decode_ms"] = dec
    res["device_decode_rows_GBps"] = res["row_bytes"] / max(dec, 1e-6) / 1e6
    eng.close()
    f.close()
    return res


def measure_phased(pkg, torch, path, label, extra, allele_cts=None):
    """--indep-pairphase's load: main + hardcall-phase tracks -> haplotype rows (ldp_load_pgen_records_phased) against the host reader
    (ldp_pgen_read_phased) followed by the same engine's load of its rows.  allele_cts: records with several ALT alleles (collapse + phase bits in
    pgen_aux1_kernel); the host side of that comparison is the per-variant reader plink2-hip used until round 5 (ldp_pgen_read_alleles_phased)."""
    f = pkg.PgenFile(path)
    m, n = f.variant_ct, f.sample_ct
    ptr, nbytes = f.file_bytes()
    host_bytes = np.ctypeslib.as_array((pkg.ctypes.c_uint8 * nbytes).from_address(ptr))
    dev_bytes = torch.from_numpy(host_bytes.copy()).cuda()
    res = {"file": label, "samples": n, "variants": m, "file_bytes": nbytes, "haplotype_row_bytes": m * pkg.phased_row_bytes(2 * n), **extra}
    eng = pkg.LdPruneEngine(2 * n, 200, 1, True, 0.5, order=2, device=0)
    chr_idx, bps = bench.genome_layout(m, 1, 2875)
    eng.set_variants(chr_idx, bps)

    def dev_call():
        eng.load_pgen_records_phased(0, f, location=pkg.LDP_MEM_DEVICE, device_bytes=dev_bytes.data_ptr(), allele_cts=allele_cts)
        torch.cuda.synchronize()

    dev_call()
    res["device_decode_plus_count_ms"] = 1e3 * best_of(dev_call)
    want = eng.variant_recs().copy()

    def host_bytes_call():
        eng.load_pgen_records_phased(0, f, allele_cts=allele_cts)
        torch.cuda.synchronize()

    res["same_call_bytes_in_host_memory_ms"] = 1e3 * best_of(host_bytes_call)
    if allele_cts is not None:
        # (what the front-end did per variant before: allele pairs + phase bits on one host thread, then a row built from them)
        def host_alleles():
            for v in range(m):
                f.read_alleles_phased(v, int(allele_cts[v]) - 1)
        res["host_allele_pair_reader_one_thread_ms"] = 1e3 * best_of(host_alleles, reps=1)
        res["device_decode_plus_count_rows_GBps"] = res["haplotype_row_bytes"] / res["device_decode_plus_count_ms"] / 1e6
        eng.close()
        f.close()
        return res
    rows_host = f.read_phased(threads=0)
    res["host_decoder_all_threads_ms"] = 1e3 * best_of(lambda: f.read_phased(threads=0), reps=2)
    res["host_decoder_one_thread_ms"] = 1e3 * best_of(lambda: f.read_phased(threads=1), reps=1)
    rows_dev = torch.from_numpy(rows_host).cuda()

    def count_only():
        eng.load_genotypes_device(0, m, rows_dev.data_ptr(), rows_host.shape[1], pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
        torch.cuda.synchronize()

    count_only()
    res["count_pass_on_decoded_rows_ms"] = 1e3 * best_of(count_only)
    got = eng.variant_recs()
    res["records_identical_to_host_decoded_rows"] = bool(all(np.array_equal(got[k], want[k]) for k in ("nm_ct", "sum", "ssq", "flags")))
    dec = res["device_decode_plus_count_ms"] - res["count_pass_on_decoded_rows_ms"]
    res["device_decode_ms"] = dec
    res["device_decode_rows_GBps"] = res["haplotype_row_bytes"] / max(dec, 1e-6) / 1e6
    eng.close()
    f.close()
    return res


def write_phased_vcf(path, codes, rng):
    """codes: (m, n) genotype codes 0/1/2/3; every het gets a random phase, everything is written with '|' (a fully phased file)."""
    m, n = codes.shape
    table = np.array(["0|0", "0|1", "1|1", ".|.", "1|0"])
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##contig=<ID=1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GT\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            c = codes[v].astype(np.int64)
            c = np.where((c == 1) & (rng.random(n) < 0.5), 4, c)
            fh.write("1\t%d\tp%d\tA\tC\t.\t.\t.\tGT\t%s\n" % (1000 + 2875 * v, v, "\t".join(table[c])))


def write_phased_multiallelic_vcf(path, codes, rng, third_rate):
    """as write_multiallelic_vcf below, every call phased (a|b in either order)"""
    m, n = codes.shape
    fwd = np.array(["0|0", "0|1", "1|1", ".|.", "0|2", "1|2", "2|2"])
    rev = np.array(["0|0", "1|0", "1|1", ".|.", "2|0", "2|1", "2|2"])
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##contig=<ID=1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GT\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            c = codes[v].astype(np.int64)
            flip = rng.random(n) < (third_rate if v % 4 else 0.7)
            c = np.where(flip & (c == 1), 4, c)
            c = np.where(flip & (c == 2), np.where(rng.random(n) < 0.5, 5, 6), c)
            gt = np.where(rng.random(n) < 0.5, fwd[c], rev[c])
            fh.write("1\t%d\tq%d\tA\tC,G\t.\t.\t.\tGT\t%s\n" % (1000 + 2875 * v, v, "\t".join(gt)))


def write_multiallelic_vcf(path, codes, rng, third_rate):
    """codes: (m, n) int8 genotype codes 0/1/2/3 from the bench generator; some ALT copies become ALT2 (every variant has two ALT
    alleles in the header)."""
    m, n = codes.shape
    table = np.array(["0/0", "0/1", "1/1", "./.", "0/2", "1/2", "2/2"])
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##contig=<ID=1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GT\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            c = codes[v].astype(np.int64)
            flip = rng.random(n) < (third_rate if v % 4 else 0.7)   # (every fourth variant: ALT2 common enough to become the major allele sometimes)
            c = np.where(flip & (c == 1), 4, c)
            c = np.where(flip & (c == 2), np.where(rng.random(n) < 0.5, 5, 6), c)
            fh.write("1\t%d\tm%d\tA\tC,G\t.\t.\t.\tGT\t%s\n" % (1000 + 2875 * v, v, "\t".join(table[c])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=200000)
    ap.add_argument("--multi-variants", type=int, default=1500)
    ap.add_argument("--phased-variants", type=int, default=4000)
    ap.add_argument("--phased-samples", type=int, default=20000)
    ap.add_argument("--missing-rate", type=float, default=0.001)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    m, n = args.variants, args.samples
    stride = (n + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(bench.SEED, 0, m, n, args.missing_rate, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    del buf
    tmp = tempfile.mkdtemp(prefix="decbench_")
    ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
    lines = []
    try:
        chr_idx, bps = bench.genome_layout(m, 1, 2875)
        bench.write_plink1_fileset(os.path.join(tmp, "d"), host, n, chr_idx, bps)
        cp = subprocess.run([ref_bin, "--bfile", "d", "--make-pgen", "--out", "v"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert cp.returncode == 0, cp.stdout[-500:]
        lines.append(measure(pkg, torch, os.path.join(tmp, "v.pgen"), None, "reference --make-pgen of the bench generator's genotypes",
                             {"missing_rate": args.missing_rate, **bench.host_description()}))
        mm = args.multi_variants
        if mm:
            shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
            codes = ((host[:mm, :, None] >> shifts) & 3).reshape(mm, -1)[:, :n]
            write_multiallelic_vcf(os.path.join(tmp, "m.vcf"), codes, np.random.default_rng(1), 0.1)
            cp = subprocess.run([ref_bin, "--vcf", "m.vcf", "--make-pgen", "--out", "mv"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert cp.returncode == 0, cp.stdout[-500:]
            lines.append(measure(pkg, torch, os.path.join(tmp, "mv.pgen"), np.full(mm, 3), "reference --vcf import, two ALT alleles per variant (aux track 1)",
                                 {"missing_rate": args.missing_rate}))
        pm, pn = min(args.phased_variants, m), min(args.phased_samples, n)
        if pm:
            shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
            codes = ((host[:pm, :(pn + 3) // 4, None] >> shifts) & 3).reshape(pm, -1)[:, :pn]
            write_phased_vcf(os.path.join(tmp, "p.vcf"), codes, np.random.default_rng(2))
            cp = subprocess.run([ref_bin, "--vcf", "p.vcf", "--make-pgen", "--out", "pv"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert cp.returncode == 0, cp.stdout[-500:]
            lines.append(measure_phased(pkg, torch, os.path.join(tmp, "pv.pgen"), "reference --vcf import of a fully phased file (hardcall-phase track)",
                                        {"missing_rate": args.missing_rate}))
            qm = min(pm, args.multi_variants)
            if qm:
                write_phased_multiallelic_vcf(os.path.join(tmp, "q.vcf"), codes[:qm], np.random.default_rng(3), 0.1)
                cp = subprocess.run([ref_bin, "--vcf", "q.vcf", "--make-pgen", "--out", "qv"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                assert cp.returncode == 0, cp.stdout[-500:]
                lines.append(measure_phased(pkg, torch, os.path.join(tmp, "qv.pgen"), "reference --vcf import of a fully phased file, two ALT alleles per variant (aux track 1 + hardcall-phase track)",
                                            {"missing_rate": args.missing_rate}, allele_cts=np.full(qm, 3)))
    finally:
        subprocess.call(["rm", "-rf", tmp])
    for r in lines:
        line = json.dumps(r)
        print(line)
        if args.out:
            with open(args.out, "a") as fh:
                fh.write(line + "\n")


if __name__ == "__main__":
    main()
