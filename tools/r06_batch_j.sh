#!/bin/bash
# (GPU box) round 6, batch j: major-allele-oriented image rows -- the GPU suite, fuzzers, the share with and without
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -12) > $O/gpu_suite.txt
cat $O/gpu_suite.txt
timeout 240 python tests/fuzz_parity.py --cases 300 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 240 python tests/fuzz_parity.py --wide-missing --cases 200 > $O/fuzz_wm.txt 2>&1; tail -1 $O/fuzz_wm.txt
timeout 240 python tests/fuzz_parity.py --wide-sparse --cases 200 > $O/fuzz_ws.txt 2>&1; tail -1 $O/fuzz_ws.txt
timeout 300 python tests/fuzz_cli.py --cases 120 > $O/fuzz_cli.txt 2>&1; tail -1 $O/fuzz_cli.txt
python bench.py --no-legs --no-cpu-baseline --no-pmc --steps 6 --warmup 2 > $O/share_orient.json 2> $O/share_orient.err
python bench.py --no-legs --no-cpu-baseline --no-pmc --steps 6 --warmup 2 --option orient_rows=0 > $O/share_plain.json 2> $O/share_plain.err
python - <<'PY'
import json
for n in ("share_orient", "share_plain"):
    d = json.loads(open("gpurun_out/r06j/%s.json" % n).read().strip().splitlines()[-1])
    print(n, "ms_per_step %.2f" % d["ms_per_step"], "kernel %.2f" % d["roofline"]["kernel_ms_per_step"], "frac %.4f" % d["roofline"]["frac"], "removed", d["config"]["variants_removed"], "count pass %.2f" % d["stage_ms"]["count_pass_codes_kernel"])
PY
python tools/c3miss_leg.py --config config2 --rates 0 --steps 10 --sets "orient:;plain:orient_rows=0" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for r,v in d['rates'].items():
    for n in ('orient','plain'):
        x=v[n]; print('config2',n,'step %.2f count %.2f kernels %.2f'%(x['ms_per_step'],x['count_pass_ms'],x['pair_kernels_ms']), x['variants_removed'])
    print(' identical', v['prune_sets_identical'])
"
