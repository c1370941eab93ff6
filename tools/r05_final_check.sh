#!/bin/bash
# tools/r05_final_check.sh -- GPU box, the last check of round 5 on the final tree: the GPU suite, smoke(), the driver's bench command
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05final
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -6) > $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time
cat $O/gpu_suite.txt; tail -1 $O/smoke.txt; tail -3 $O/bench_n1.time; tail -c 400 $O/bench_n1.json
