#!/bin/bash
# round 6, batch p: a J tile's far tile computed by its diagonal tile's workgroup (wide_merge) against two workgroups (2 x 3 diagonal body) and 2 x 4 everywhere
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_properties.py -m gpu -x -q -k "merged or wide or tile or config3 or band or three_kernel" > gpurun_out/r06_merge_tests.log 2>&1
tail -3 gpurun_out/r06_merge_tests.log
timeout 300 python tests/fuzz_parity.py --seed 901 --cases 400 2>&1 | tail -1
timeout 600 python tools/c3miss_leg.py --rates 0 --steps 4 --sets "merged:;two:wide_merge=0;one:wide_merge=0,wide_diag_kernel=0" > gpurun_out/r06_merge_slice.json 2> gpurun_out/r06_merge_slice.err
B="python bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-pmc --no-e2e --no-cli-compare"
timeout 600 $B > gpurun_out/r06_merge_share_merged.json 2> gpurun_out/r06_merge_share.err
timeout 600 $B --option wide_merge=0 > gpurun_out/r06_merge_share_two.json 2>> gpurun_out/r06_merge_share.err
timeout 600 $B --option wide_merge=0 --option wide_diag_kernel=0 > gpurun_out/r06_merge_share_one.json 2>> gpurun_out/r06_merge_share.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_merge_slice.json"))
for rate,v in d["rates"].items():
    for name in ("merged","two","one"):
        x=v[name]; print("slice",name,"ms %.2f pair %.2f skipped %.3f removed %s"%(x["ms_per_step"],x["pair_kernels_ms"],x["early_termination_skipped_frac"],x["variants_removed"]))
    print("identical",v.get("prune_sets_identical"))
for name in ("merged","two","one"):
    try:
        d=json.loads(open("gpurun_out/r06_merge_share_%s.json"%name).read().strip().splitlines()[-1])
        print("share",name,"ms/step %.2f kernels %.2f frac %.4f removed %s power %s clock %s beyond %.4f"%(d["ms_per_step"],d["stage_ms"]["pair_kernels"],d["roofline"]["frac"],d["config"]["variants_removed"],d.get("power_and_clock",{}).get("socket_power_w_median"),d.get("power_and_clock",{}).get("shader_clock_mhz_median"),d["roofline"]["mfma"]["computed_beyond_plan_frac"]))
    except Exception as ex: print(name,"??",ex)
PY
