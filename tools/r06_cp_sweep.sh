#!/bin/bash
# tools/r06_cp_sweep.sh -- (GPU box, measurement build) config 3's density slice at 0 / 0.1 % missing calls for several checkpoint lists:
# where the tile kernels' checkpoints should sit when a launch's rows miss a few calls.  Output: gpurun_out/r06c/cp_sweep.jsonl
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
export LDP_LIB_MEASURE=1
: > $OUT/cp_sweep.jsonl
for F in default 0.565,0.593,0.633,0.673,0.753 0.58,0.62,0.66,0.70,0.78 0.593,0.633,0.673,0.713,0.793 0.58,0.61,0.65,0.69,0.75 0.60,0.64,0.68,0.72,0.80; do
  if [ "$F" = default ]; then unset LDP_DEBUG_CP_FRACS; else export LDP_DEBUG_CP_FRACS=$F; fi
  echo "{\"cp_fracs\": \"$F\", \"result\": $(python $R/tools/c3miss_leg.py --rates 0,0.001,0.003 --steps 3 2>>$OUT/cp_sweep.err)}" >> $OUT/cp_sweep.jsonl
done
unset LDP_DEBUG_CP_FRACS
echo "{\"cp_fracs\": \"exhaustive\", \"result\": $(python $R/tools/c3miss_leg.py --rates 0,0.001 --steps 3 --sets 'x:early_exit=0' 2>>$OUT/cp_sweep.err)}" >> $OUT/cp_sweep.jsonl
