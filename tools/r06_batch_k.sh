#!/bin/bash
# round 6, batch k: quarter-tile kernel with 4 stages (1024 samples) per barrier against 2 -- the same binary, option tile4_stages
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/c3miss_leg.py --rates 0.01,0.05 --steps 3 --sets "quad:tile4_stages=4;pair:tile4_stages=2" > gpurun_out/r06_t4_stages.json 2> gpurun_out/r06_t4_stages.err
timeout 600 python tools/c3miss_leg.py --config config5 --rates 0.01 --steps 3 --sets "quad:tile4_stages=4;pair:tile4_stages=2" > gpurun_out/r06_t4_stages_c5.json 2>> gpurun_out/r06_t4_stages.err
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "missing or tile4 or quarter or general or wide" > gpurun_out/r06_t4_stages_tests.log 2>&1
tail -3 gpurun_out/r06_t4_stages_tests.log
cat gpurun_out/r06_t4_stages.json gpurun_out/r06_t4_stages_c5.json | cut -c1-3000
