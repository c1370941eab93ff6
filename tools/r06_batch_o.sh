#!/bin/bash
# round 6, batch o: the diagonal tiles' 2 x 3 body in the SPARSE instantiation too -- launches with a few missing calls, with and without it
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "few_missing or tile_plan or wide" > gpurun_out/r06_diag_sparse_tests.log 2>&1
tail -2 gpurun_out/r06_diag_sparse_tests.log
timeout 300 python tests/fuzz_parity.py --wide-sparse --seed 801 --cases 300 2>&1 | tail -1
timeout 900 python tools/c3miss_leg.py --rates 0,0.001,0.003 --steps 4 --sets "split:;one:wide_diag_kernel=0" > gpurun_out/r06_diag_sparse_slice.json 2> gpurun_out/r06_diag_sparse_slice.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_diag_sparse_slice.json"))
for rate,v in d["rates"].items():
    for name in ("split","one"):
        x=v[name]; print(rate,name,"ms %.2f pair %.2f skipped %.3f recount %s removed %s %s"%(x["ms_per_step"],x["pair_kernels_ms"],x["early_termination_skipped_frac"],x["pairs_counted_exactly"],x["variants_removed"],x["kernel"]))
    print(rate,"identical",v.get("prune_sets_identical"))
PY
