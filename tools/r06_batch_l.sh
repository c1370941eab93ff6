#!/bin/bash
# round 6, batch l: the headline step with the count pass of later rows beside the pair kernels of earlier ones
# (LDP_EAGER_PAIRS=1: groups launched as their rows are counted; LDP_DEBUG_GROUPS: launch groups; LDP_DEBUG_COUNT_CUS: count pass confined to n CUs)
# (the CU-masked count stream -- LDP_DEBUG_COUNT_CUS -- was an experiment of this batch only and is not in the tree: profiles/r06_experiments.md section 4)
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export LDP_LIB_MEASURE=1
B="python bench.py --steps 6 --warmup 2 --no-legs --no-cpu-baseline --no-pmc --no-e2e --no-cli-compare"
out=gpurun_out/r06_overlap.jsonl
: > $out
run() {
  echo "{\"env\": \"$1\"}" >> $out
  env $1 timeout 600 $B 2>gpurun_out/r06_overlap.err | tail -1 >> $out
}
run "LDP_NONE=1"
run "LDP_DEBUG_GROUPS=8"
run "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=4"
run "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=8"
run "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=8 LDP_DEBUG_COUNT_CUS=16"
run "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=8 LDP_DEBUG_COUNT_CUS=32"
run "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=16 LDP_DEBUG_COUNT_CUS=24"
python - <<'PY'
import json
for ln in open("gpurun_out/r06_overlap.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    if "env" in d: print(d["env"]); continue
    print("   ms/step %.2f  kernels %.2f  count %.2f  launches %s  identical_bits %s" % (d["ms_per_step"], d["stage_ms"]["pair_kernels"], d["stage_ms"]["count_pass_codes_kernel"], d["roofline"].get("launches_per_step"), d.get("headline_bits_check",{}).get("identical")))
PY
