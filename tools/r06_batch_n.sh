#!/bin/bash
# round 6, batch n: the diagonal tiles on a kernel of their own (2 x 3 rectangles, pair_mfma_wide_kernel<0, false, 3>) against the 2 x 4 kernel for every tile
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_properties.py -m gpu -x -q -k "wide or tile or config3 or band or three_kernel" > gpurun_out/r06_diag_tests.log 2>&1
tail -3 gpurun_out/r06_diag_tests.log
timeout 600 python tools/c3miss_leg.py --rates 0 --steps 4 --sets "split:;one:wide_diag_kernel=0" > gpurun_out/r06_diag_slice.json 2> gpurun_out/r06_diag_slice.err
B="python bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-pmc --no-e2e --no-cli-compare"
timeout 600 $B > gpurun_out/r06_diag_share_split.json 2> gpurun_out/r06_diag_share.err
timeout 600 $B --option wide_diag_kernel=0 > gpurun_out/r06_diag_share_one.json 2>> gpurun_out/r06_diag_share.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_diag_slice.json"))
for rate,v in d["rates"].items():
    for name in ("split","one"):
        x=v[name]; print("slice",name,"ms %.2f pair %.2f skipped %.3f removed %s"%(x["ms_per_step"],x["pair_kernels_ms"],x["early_termination_skipped_frac"],x["variants_removed"]))
    print("identical",v.get("prune_sets_identical"))
for name in ("split","one"):
    try:
        d=json.loads(open("gpurun_out/r06_diag_share_%s.json"%name).read().strip().splitlines()[-1])
        print("share",name,"ms/step %.2f kernels %.2f frac %.4f removed %s bits %s power %s clock %s"%(d["ms_per_step"],d["stage_ms"]["pair_kernels"],d["roofline"]["frac"],d["config"]["variants_removed"],d.get("headline_bits_check",{}).get("identical"),d.get("power_and_clock",{}).get("socket_power_w_median"),d.get("power_and_clock",{}).get("shader_clock_mhz_median")))
    except Exception as ex: print(name,"??",ex)
PY
