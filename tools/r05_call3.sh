#!/bin/bash
# tools/r05_call3.sh -- GPU box: the driver's bench command with the in-run PMC passes and the new config-5 reference slice; config 2's run timeline
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_full.time
tail -3 $O/bench_full.time; tail -c 400 $O/bench_full.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("ms_per_step", d["ms_per_step"], "frac", r["frac"], "traffic", r.get("traffic"), r.get("traffic_over_compulsory"), r.get("traffic_measurement_s"), r.get("traffic_in_run_error"))
    print("traffic_source", str(r.get("traffic_source"))[:200])
    print("kernels", r.get("traffic_pair_kernels"))
    print("c5 slice", json.dumps(d["legs"]["config5_density"].get("reference_slice"))[:900])
    print("e2e", json.dumps(d["cpu_baseline"].get("e2e_wall_s"))[:400])
except Exception as e:
    print("bench_full failed", e)
PY
env LDP_LIB_MEASURE=1 LDP_DEBUG_TIMELINE=1 timeout 300 python bench.py --workload config2 --steps 4 --warmup 2 --no-legs --no-cpu-baseline > $O/c2_timeline.json 2> $O/c2_timeline.err
grep -E "run timeline|recs copy" $O/c2_timeline.err | tail -8
