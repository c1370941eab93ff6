#!/bin/bash
# round 6, batch t: wave -> rectangle maps of the 2 x 4 body, one library each (lib/libldprune_hip_{base,map2,map3,map4}.so), the share alternately
# (The libraries beside the tree's are built by hand before the call: `git stash` / a -D switch, build_library(), cp lib/libldprune_hip.so lib/libldprune_hip_<name>.so -- git-ignored,
# they travel with the snapshot.  profiles/r06_experiments.md section 4b says which sources each one was.)
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=plink-ng_amd/lib
B="python bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-pmc --no-e2e --no-cli-compare"
for round in 1 2; do
  for which in base map2 map3 map4; do
    cp $L/libldprune_hip_$which.so $L/libldprune_hip.so
    timeout 600 $B > gpurun_out/r06_maps_share_${which}_$round.json 2>> gpurun_out/r06_maps_share.err
    python - $which $round <<'PY'
import json, sys
w, r = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("gpurun_out/r06_maps_share_%s_%s.json" % (w, r)).read().strip().splitlines()[-1])
    print("share", w, r, "ms/step %.2f kernels %.2f frac %.4f removed %s power %s clock %s" % (d["ms_per_step"], d["stage_ms"]["pair_kernels"], d["roofline"]["frac"], d["config"]["variants_removed"],
          d.get("power_and_clock", {}).get("socket_power_w_median"), d.get("power_and_clock", {}).get("shader_clock_mhz_median")))
except Exception as ex:
    print("share", w, r, "??", ex)
PY
  done
done
cp $L/libldprune_hip_base.so $L/libldprune_hip.so
