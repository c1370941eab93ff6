#!/bin/bash
# (GPU box) round 6, batch h: the GPU suite on the CSR predicate rows, the share with and without them, the pinned ring's slot size
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -8) > $O/gpu_suite.txt
cat $O/gpu_suite.txt
python bench.py --no-legs --no-cpu-baseline --no-pmc --steps 6 --warmup 1 > $O/share_csr.json 2> $O/share_csr.err
python bench.py --no-legs --no-cpu-baseline --no-pmc --steps 6 --warmup 1 --option pred_csr=0 > $O/share_dense.json 2> $O/share_dense.err
python - <<'PY'
import json
for n in ("share_csr", "share_dense"):
    d = json.loads(open("gpurun_out/r06h/%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.2f" % d["ms_per_step"], "kernel %.2f" % d["roofline"]["kernel_ms_per_step"], "removed", d["config"]["variants_removed"], d["stage_ms"]["host_replay"])
PY
LDP_SWEEP=stage timeout 900 python tools/e2e_load_sweep.py > $O/load_sweep_stage.jsonl 2> $O/load_sweep_stage.err
cat $O/load_sweep_stage.jsonl | cut -c1-400
