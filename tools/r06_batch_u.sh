#!/bin/bash
# round 6, batch u: the quarter-tile kernel (config 5) with neighbouring waves on different V pairs, against the tree's map (lib/libldprune_hip_base.so)
# (The libraries beside the tree's are built by hand before the call: `git stash` / a -D switch, build_library(), cp lib/libldprune_hip.so lib/libldprune_hip_<name>.so -- git-ignored,
# they travel with the snapshot.  profiles/r06_experiments.md section 4b says which sources each one was.)
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=plink-ng_amd/lib
cp $L/libldprune_hip.so $L/libldprune_hip_new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "missing or four or tile4 or quarter or general or mixed" > gpurun_out/r06_t4map_tests.log 2>&1
tail -3 gpurun_out/r06_t4map_tests.log
timeout 300 python tests/fuzz_parity.py --seed 931 --cases 200 --wide-missing 2>&1 | tail -1
for round in 1 2; do
  for which in new base; do
    cp $L/libldprune_hip_$which.so $L/libldprune_hip.so
    timeout 600 python tools/c5_leg.py --forms tiles --steps 3 > gpurun_out/r06_t4map_c5_${which}_$round.json 2>> gpurun_out/r06_t4map.err
    timeout 600 python tools/c5_leg.py --forms tiles --steps 3 --missing-rate 0.01 > gpurun_out/r06_t4map_c1_${which}_$round.json 2>> gpurun_out/r06_t4map.err
    python - $which $round <<'PY'
import json, sys
w, r = sys.argv[1], sys.argv[2]
for tag in ("c5", "c1"):
    try:
        d = json.loads(open("gpurun_out/r06_t4map_%s_%s_%s.json" % (tag, w, r)).read().strip().splitlines()[-1])
        print(tag, w, r, json.dumps({k: v for k, v in d.items() if k not in ("samples", "variants", "window_kb", "r2")})[:400])
    except Exception as ex:
        print(tag, w, r, "??", ex)
PY
  done
done
cp $L/libldprune_hip_new.so $L/libldprune_hip.so
