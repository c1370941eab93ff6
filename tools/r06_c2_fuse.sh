#!/bin/bash
# (GPU box, measurement build) config 2: what the stage loop of a count-fused narrow-band pair kernel costs -- the exhaustive DIAGFORM kernel as shipped,
# and with the fused-count work in its stage loop (LDP_DEBUG_DIAG_FUSE=1: one 16x16x128 selector MFMA per J block and k-step, the missing-call detector),
# beside today's step (count pass + pair kernel with its one checkpoint).  Output: gpurun_out/r06i/c2_fuse.jsonl
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06i
mkdir -p $O
export LDP_LIB_MEASURE=1
: > $O/c2_fuse.jsonl
echo "{\"what\": \"today: one checkpoint\", \"result\": $(python $R/tools/c3miss_leg.py --config config2 --rates 0 --steps 10 2>>$O/c2_fuse.err)}" >> $O/c2_fuse.jsonl
echo "{\"what\": \"exhaustive (early_exit 0)\", \"result\": $(python $R/tools/c3miss_leg.py --config config2 --rates 0 --steps 10 --sets 'x:early_exit=0' 2>>$O/c2_fuse.err)}" >> $O/c2_fuse.jsonl
export LDP_DEBUG_DIAG_FUSE=1
echo "{\"what\": \"exhaustive + fused-count work in the stage loop\", \"result\": $(python $R/tools/c3miss_leg.py --config config2 --rates 0 --steps 10 --sets 'x:early_exit=0' 2>>$O/c2_fuse.err)}" >> $O/c2_fuse.jsonl
python - <<'PY'
import json, os
for ln in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06i/c2_fuse.jsonl")):
    d = json.loads(ln)
    x = list(list(d["result"]["rates"].values())[0].values())[0]
    print("%-50s step %.2f ms | count pass %.2f | pair kernels %.2f | removed %d" % (d["what"], x["ms_per_step"], x["count_pass_ms"], x["pair_kernels_ms"], x["variants_removed"]))
PY
