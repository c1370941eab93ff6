#!/bin/bash
# (GPU box) round 6, batch i: the quarter-tile kernel with shared V operands -- parity, then the config-3 density slice at 1 % / 5 % missing against the
# old form; the config-2 fused-count stage-loop measurement
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config3_parity.py tests/test_full_size_properties.py -x -q -m gpu -k "quarter or four_product or wide_band or config3 or three_kernel or missing" 2>&1 | tail -6) > $O/tests.log
cat $O/tests.log
timeout 300 python tests/fuzz_parity.py --wide-missing --cases 200 > $O/fuzz_wm.txt 2>&1; tail -2 $O/fuzz_wm.txt
timeout 300 python tests/fuzz_parity.py --wide-sparse --cases 150 > $O/fuzz_ws.txt 2>&1; tail -2 $O/fuzz_ws.txt
python tools/c3miss_leg.py --rates 0.01,0.05 --steps 3 --sets "share:;old:tile4_share=0" > $O/c3miss_tile4.json 2> $O/c3miss_tile4.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06i/c3miss_tile4.json"))
for r, v in d["rates"].items():
    for n in ("share", "old"):
        x = v[n]
        print(r, n, x["kernel"], "step %.1f ms kernels %.1f skip %.3f exact %d removed %d pred_true %d" % (x["ms_per_step"], x["pair_kernels_ms"], x["early_termination_skipped_frac"], x["pairs_counted_exactly"], x["variants_removed"], x["pred_true"]))
    print("  identical", v["prune_sets_identical"])
PY
bash tools/r06_c2_fuse.sh
