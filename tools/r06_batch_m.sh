#!/bin/bash
# round 6, batch m: SPARSE tiles' checkpoints counting stubborn pairs exactly from per-row lists of missing calls, against the box bound alone
set -u
# (the per-row lists and the counting checkpoint -- option sparse_lists -- were an experiment of this batch only and are not in the tree: profiles/r06_experiments.md section 1)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "few_missing or tile_plan or missing" > gpurun_out/r06_lists_tests.log 2>&1
tail -3 gpurun_out/r06_lists_tests.log
timeout 900 python tools/c3miss_leg.py --rates 0,0.0001,0.001,0.003 --steps 3 --sets "lists:;box:sparse_lists=0" > gpurun_out/r06_lists.json 2> gpurun_out/r06_lists.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_lists.json"))
for rate,v in d["rates"].items():
    for name in ("lists","box"):
        x=v[name]
        print(rate,name,"ms %.2f pair %.2f count %.2f skipped %.3f recount %s removed %s kernel %s"%(x["ms_per_step"],x["pair_kernels_ms"],x["count_pass_ms"],x["early_termination_skipped_frac"],x["pairs_counted_exactly"],x["variants_removed"],x["kernel"]))
    print(rate,"identical",v.get("prune_sets_identical"))
PY
