#!/bin/bash
# round 6, batch s: the SPARSE instantiation of the tile kernel with the second half-stage's J fragments made during the first, against the plain form
# (lib/libldprune_hip_base.so): parity tests on the new library, then config 3's slice at 0 / 0.1 % / 0.3 % missing calls alternately on the two libraries
# (The libraries beside the tree's are built by hand before the call: `git stash` / a -D switch, build_library(), cp lib/libldprune_hip.so lib/libldprune_hip_<name>.so -- git-ignored,
# they travel with the snapshot.  profiles/r06_experiments.md section 4b says which sources each one was.)
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=plink-ng_amd/lib
cp $L/libldprune_hip.so $L/libldprune_hip_new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_properties.py -m gpu -x -q -k "wide or tile or config3 or band or three_kernel or missing" > gpurun_out/r06_pfs_tests.log 2>&1
tail -3 gpurun_out/r06_pfs_tests.log
timeout 300 python tests/fuzz_parity.py --seed 921 --cases 300 --wide-sparse 2>&1 | tail -1
for round in 1 2; do
  for which in new base; do
    cp $L/libldprune_hip_$which.so $L/libldprune_hip.so
    timeout 600 python tools/c3miss_leg.py --rates 0,0.001,0.003 --steps 4 > gpurun_out/r06_pfs_slice_${which}_$round.json 2>> gpurun_out/r06_pfs_slice.err
  done
done
cp $L/libldprune_hip_new.so $L/libldprune_hip.so
python - <<'PY'
import json
for rnd in (1, 2):
    for which in ("new", "base"):
        try:
            d = json.load(open("gpurun_out/r06_pfs_slice_%s_%d.json" % (which, rnd)))
            for rate, v in d["rates"].items():
                for name, x in v.items():
                    if isinstance(x, dict) and "pair_kernels_ms" in x:
                        print("slice", which, rnd, rate, x.get("kernel"), "step %.2f pair %.2f ms skipped %.3f removed %s exact %s" % (x["ms_per_step"], x["pair_kernels_ms"], x["early_termination_skipped_frac"], x["variants_removed"], x.get("pairs_counted_exactly")))
        except Exception as ex:
            print("slice", which, rnd, "??", ex)
PY
