#!/bin/bash
# tools/r05_cp_sweep.sh -- GPU box, measurement build: where the tile kernel's checkpoints should sit at N = 500,000 (LDP_DEBUG_CP_FRACS)
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
export HSA_ENABLE_IPC_MODE_LEGACY=0 LDP_LIB_MEASURE=1
O=gpurun_out/r05i
mkdir -p $O
run() {  # name, fractions, extra bench args
  env ${2:+LDP_DEBUG_CP_FRACS=$2} timeout 300 python bench.py --no-legs --no-cpu-baseline --steps $4 --warmup 2 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 [$2]', round(d['ms_per_step'],2), round(r['kernel_ms_per_step'],2), round(r['mfma']['early_termination_skipped_frac'],4), d['config']['variants_removed'])"
}
for F in "" "0.561,0.575,0.600,0.650,0.800" "0.558,0.566,0.580,0.620,0.750" "0.563,0.570,0.590,0.630,0.750" "0.565,0.575,0.593,0.633,0.753" "0.565,0.580,0.600,0.640,0.760" "0.560,0.570,0.585,0.620,0.750" "0.565,0.593,0.633,0.713,0.913"; do
  run slice "$F" "--variants 120000" 12
done > $O/cp_sweep2.txt 2>&1
cat $O/cp_sweep2.txt
