#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host-side file reader (ldp_pgen.cpp: main track, phase track,
# multiallelic + phase, dosage tracks, sample subsetting) on every committed golden .pgen, and over plink2-hip's host-only paths
# (the .vcor number formatter, the zstd output stream), and over the engine's host logic.  No GPU involved.
#   bash tests/sanitize/run.sh
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
SAN="-std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
g++ $SAN -I"$R/include" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ "$R/tests/sanitize/reader_harness.cpp" "$R/tests/sanitize/geometry_stub.cpp" \
    "$R/plink-ng_amd/csrc/ldp_pgen.cpp" -o "$T/reader" -lpthread
G="$R/tests/golden/pgen"
"$T/reader" "$G/varwidth_small.pgen" 1 "$G/phased_small.pgen" 1 "$G/phased_partial.pgen" 1 "$G/phased_multi.pgen" 4 "$G/phased_multi_partial.pgen" 4 "$G/dosage_small.pgen" 1
LDTEST_PGEN_PORTABLE=1 "$T/reader" "$G/phased_partial.pgen" 1 "$G/phased_multi.pgen" 4 > /dev/null
# malformed input: byte flips / truncation of the golden files must end in an error code, never in a sanitizer report
python3 - "$T" "$G" <<'PY'
import sys, subprocess, numpy as np
T, G = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(1)
files = [("varwidth_small.pgen", 1), ("phased_small.pgen", 1), ("phased_multi.pgen", 4), ("phased_multi_partial.pgen", 4), ("dosage_small.pgen", 1)]
for it in range(150):
    name, alts = files[it % len(files)]
    data = bytearray(open(G + "/" + name, "rb").read())
    kind = int(rng.integers(0, 3))
    if kind == 1:
        data = data[:int(rng.integers(3, len(data)))]
    else:
        for _ in range(int(rng.integers(1, 6))):
            data[int(rng.integers(0, len(data) if kind == 0 else min(len(data), 2000)))] = int(rng.integers(0, 256))
    open(T + "/mut.pgen", "wb").write(data)
    r = subprocess.run([T + "/reader", T + "/mut.pgen", str(alts)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode in (0, 1) and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (it, name, kind, r.stderr[-800:])
print("reader on 150 corrupted files: errors only, no sanitizer report")
PY
if [ -f "$R/plink-ng_amd/lib/libldprune_hip.so" ]; then
  g++ $SAN -I/opt/rocm/include "$R"/plink-ng_amd/csrc/plink2_hip_cli.cpp "$R"/plink-ng_amd/csrc/p2h_*.cpp -o "$T/cli" -L"$R/plink-ng_amd/lib" -lldprune_hip -Wl,-rpath,"$R/plink-ng_amd/lib" -lpthread -ldl
  export ASAN_OPTIONS=detect_leaks=0
  python3 - "$T" "$R" <<'PY'
import sys, subprocess, numpy as np
T, R = sys.argv[1], sys.argv[2]
g = np.load(R + "/tests/golden/pgen/vcor_format_g6.npz")
open(T + "/bits.txt", "w").write("".join("%016x\n" % int(b) for b in g["bits"]))
out = subprocess.run([T + "/cli", "--debug-format-g6", T + "/bits.txt"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
assert out.returncode == 0 and out.stdout.split("\n")[:-1] == [str(t) for t in g["texts"]], out.stderr[-400:]
open(T + "/in.txt", "w").write("line\n" * 200000)
z = subprocess.run([T + "/cli", "--debug-zstd", T + "/in.txt", T + "/out.zst"], stderr=subprocess.PIPE, text=True)
assert z.returncode == 0, z.stderr[-400:]
# the host-only command paths: table parsing (alleles, centimorgans), filters, the --clump report handling with column
# sets / bins / ranges (a 1-bp radius pairs nothing, so no device is needed), and the command-line refusals
sys.path.insert(0, R + "/tests")
import os
import ldtools as L
import test_clump as TC
d = T + "/fs"
os.makedirs(d)
m = 400
raw = L.synth_raw_codes(m, 40, seed=3, missing_rate=0.02)
chroms = ["1"] * 150 + ["2"] * 150 + ["X"] * 100
bps = np.concatenate([np.arange(150), np.arange(150), np.arange(100)]) * 1000 + 1000
L.write_bed(d + "/d", raw, chroms, bps)
L.write_pgen_fixed(d + "/d", raw, chroms, bps)
TC.write_report(d + "/a.txt", m, 5)
TC.write_report(d + "/b.txt", m, 6, sig_rate=0.1)
TC.write_ranges(d + "/genes.txt", chroms, bps, 2)
open(d + "/drop.txt", "w").write("snp3 snp77\nsnp200\nnone_such\n")
TC.add_a1_column(d + "/a.txt", d + "/a1.txt", 3)
runs = [
    (["--bfile", "d", "--clump", "cols=+ref,+alt1,+alt,+bounds,+f,+a1", "a.txt", "b.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-bins", "1e-5,0.001", "0.2",
      "--clump-range", "genes.txt", "--clump-range-border", "3"], 0),
    (["--pfile", "d", "--clump", "cols=sp2", "a.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-log10", "output-only", "--clump-p1", "0.3", "--clump-p2", "1e-7"], 0),
    (["--bfile", "d", "--chr", "1,X", "--exclude", "drop.txt", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run"], 0),
    (["--bfile", "d", "--clump", "zs", "cols=+a1,+bounds", "a1.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-force-a1", "--clump-p1", "0.05", "--clump-p2", "1e-4"], 0),
    (["--bfile", "d", "--r-unphased", "cols=+nope"], 8),
    (["--bfile", "d", "--r2-unphased", "square", "--ld-window-cm", "1"], 8),
    (["--bfile", "d", "--clump", "a.txt", "cols=+f", "--clump-unphased"], 8),
    (["--pgen", "d.pgen", "--pgi", "nowhere.pgi", "--pvar", "d.pvar", "--psam", "d.psam", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run"], 0),
]
for args, want in runs:
    r = subprocess.run([T + "/cli"] + args + ["--out", "o"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == want and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (args, r.returncode, r.stdout[-300:], r.stderr[-800:])
# --clump over (variant, A1 allele) pairs (round 5): a fileset with multiallelic sites whose LAST variant is multiallelic, reports that name one allele per
# variant (nothing pairs: no device), --clump-force-a1 with two reports -- the reference's stale forced-A1 bit then points past its allele table
# (plink2_ld.cc:9357; it dies there), plink2-hip must print an empty name and stay inside its own
if L.have_ref():
    import pathlib
    d2 = T + "/fs2"
    os.makedirs(d2)
    alt_ct, _, _ = TC.multiallelic_clump_fileset(pathlib.Path(d2), 300, 40, 7, chrom_of=lambda v: "1", multi_rate=0.5)
    lines = open(d2 + "/d.pvar").read().split("\n")
    f = lines[-2].split("\t")
    f[4] = "C,G,T"
    lines[-2] = "\t".join(f)           # (the genotypes do not use the added alleles; the table does)
    open(d2 + "/d.pvar", "w").write("\n".join(lines))
    alt_ct[-1] = 3
    TC.write_allele_report(d2 + "/a.txt", alt_ct, 41, True, sig_rate=0.4)
    TC.write_allele_report(d2 + "/b.txt", alt_ct, 42, True, sig_rate=0.4)
    for extra in ([], ["--clump-force-a1"], ["--clump-force-a1", "--clump-p1", "0.5", "--clump-p2", "0.5"]):
        r = subprocess.run([T + "/cli", "--pfile", "d", "--clump", "cols=+a1,+f,+bounds", "a.txt", "b.txt", "--clump-unphased", "--clump-kb", "0.001"] + extra + ["--out", "o"], cwd=d2,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (extra, r.returncode, r.stdout[-300:], r.stderr[-800:])
print("plink2-hip host paths: clean")
PY
fi
if command -v hipcc > /dev/null; then
  # the engine's host logic (window planning in all three modes, sharding, greedy replay in both orders) on random
  # variant tables; the HIP runtime is linked but never finds a device here
  hipcc -std=c++17 -g -O1 --offload-arch=gfx950 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-gpu-sanitize -I"$R/include" \
      "$R/tests/sanitize/engine_host_harness.cpp" "$R/plink-ng_amd/csrc/ldp_engine.cpp" "$R/plink-ng_amd/csrc/ldp_engine_run.cpp" "$R/plink-ng_amd/csrc/ldp_engine_r2.cpp" "$R/plink-ng_amd/csrc/ldp_engine_load.cpp" "$R/plink-ng_amd/csrc/ldp_engine_shard.cpp" "$R/plink-ng_amd/csrc/ldp_kernels.hip" "$R/plink-ng_amd/csrc/ldp_pair_mfma.hip" "$R/plink-ng_amd/csrc/ldp_pair_wide.hip" "$R/plink-ng_amd/csrc/ldp_codes.hip" "$R/plink-ng_amd/csrc/ldp_pred_csr.hip" "$R/plink-ng_amd/csrc/ldp_pgen_decode.hip" \
      "$R/plink-ng_amd/csrc/ldp_synth.hip" "$R/plink-ng_amd/csrc/ldp_pgen.cpp" -o "$T/engine" -lpthread 2> "$T/engine_build.log" || { tail -5 "$T/engine_build.log"; exit 1; }
  ASAN_OPTIONS=detect_leaks=0 "$T/engine"
fi
echo "sanitizers: clean"
