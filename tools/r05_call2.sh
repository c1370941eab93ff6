#!/bin/bash
# tools/r05_call2.sh -- GPU box, round 5, second batch: the driver's bench command on the new tree (power / bits check / chr22-sized
# end-to-end run), the whole GPU suite, the rest of the attribution (probe with power, single ablations, L2-resident DMA), config 2 with
# eager pair launches, the PMC profile of the config-4 shape.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
df -h /dev/shm /tmp > $O/space.txt 2>&1; nproc >> $O/space.txt; free -g >> $O/space.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_full.time
tail -3 $O/bench_full.time; tail -c 600 $O/bench_full.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "power", d.get("power_and_clock"))
    print("bits", d.get("headline_bits_check"))
    cb = d.get("cpu_baseline", {})
    print("e2e", json.dumps(cb.get("e2e_wall_s"))[:1500])
    print("cb keys", sorted(cb.keys()), "value", cb.get("value"), "e2e_err", d.get("e2e_error"), json.dumps(cb.get("e2e_chr22_measurement"))[:600])
    print("legs", {k: (v.get("ms_per_step"), v.get("error")) for k, v in d.get("legs", {}).items()})
except Exception as e:
    print("bench_full failed", e)
PY
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 400 python tools/attribution.py --steps 40 --ablations 0,2,4,64 --modes exhaustive > $O/attr_more.jsonl 2> $O/attr_more.err
timeout 200 python tools/attribution.py --probe > $O/attr_probe.jsonl 2> $O/attr_probe.err
cut -c1-420 $O/attr_more.jsonl; cat $O/attr_probe.jsonl
for SET in "" "LDP_EAGER_PAIRS=1" "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=4" "LDP_EAGER_PAIRS=1 LDP_DEBUG_GROUPS=8" "LDP_DEBUG_GROUPS=1"; do
  env LDP_LIB_MEASURE=1 $SET timeout 300 python bench.py --workload config2 --steps 20 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 [$SET]', round(d['ms_per_step'],3), d['stage_ms']['count_pass_codes_kernel'], d['stage_ms']['pair_kernels'], d['config']['variants_removed'])"
done > $O/config2_eager.txt 2>&1
cat $O/config2_eager.txt
LDP_PROF_ARGS="--only-config4" timeout 900 bash tools/profile.sh r05_c4 > $O/prof_c4.log 2>&1; tail -12 $O/prof_c4.log
