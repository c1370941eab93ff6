#!/bin/bash
# tools/r05_attribution.sh -- GPU box: where pair_mfma_wide_kernel's time goes (profiles/r05_experiments.md), both tile kernels.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. parity of the barrier-free kernel before anything is timed
timeout 600 python -m pytest tests/test_00_rows.py tests/test_cabi_symbols.py "tests/test_gpu_parity.py::test_wide_band_tiles_match_oracle" \
  "tests/test_gpu_parity.py::test_early_termination_wide_window" "tests/test_full_size_properties.py::test_three_kernel_families_agree_at_config3_density" \
  -x -q -m gpu > $O/parity.txt 2>&1
tail -3 $O/parity.txt
# 2. the slice (120,000 variants x 500,000 samples), every ablation of the barrier kernel, then the barrier-free one
timeout 900 python tools/attribution.py --steps 40 --ablations 0,32,1,7,8,9,15,16,25 > $O/attr_sync.jsonl 2> $O/attr_sync.err
timeout 600 python tools/attribution.py --steps 40 --ablations 0,32 --option wide_async=1 > $O/attr_async.jsonl 2> $O/attr_async.err
cut -c1-400 $O/attr_sync.jsonl; cut -c1-600 $O/attr_async.jsonl
# 3. the share (1.25M variants, 156 GB): both kernels through bench.py
timeout 600 python bench.py --steps 5 --warmup 1 --no-legs --no-cpu-baseline > $O/share_sync.json 2> $O/share_sync.err
timeout 600 python bench.py --steps 5 --warmup 1 --no-legs --no-cpu-baseline --option wide_async=1 > $O/share_async.json 2> $O/share_async.err
python - <<PY
import json
for n in ("share_sync", "share_async"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d["config"]["variants_removed"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -5 $O/*.err | cut -c1-300
