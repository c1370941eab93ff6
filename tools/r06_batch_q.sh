#!/bin/bash
# round 6, batch q: tools/tile_shape_probe -- the tile kernel's stage loop in other wave / tile shapes (4 waves x 4 x 4 at 512 VGPRs, 8 x 12 tiles),
# each with the socket's power and the shader clock sampled beside it, and the kernel itself (exhaustive / default) on the config-3 slice for scale
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-c}
i=0
for s in ${SHAPES:-2x4 4x4 4x4pf 2x4pf 2x6 4x6 2x4p2 2x4}; do
  i=$((i+1)); n=${TAG}$i
  timeout 300 bash tools/power_sample.sh gpurun_out/r06_tile_shape_${s}_$n.smi.jsonl -- ${PROBE_BIN:-tools/_bin/tile_shape_probe} $s ${SECONDS_EACH:-4} > gpurun_out/r06_tile_shape_${s}_$n.json 2> gpurun_out/r06_tile_shape_${s}_$n.err
  python tools/tile_shape_summary.py gpurun_out/r06_tile_shape_${s}_$n.json
done
if [ -z "${NO_KERNEL:-}" ]; then
timeout 600 python tools/c3miss_leg.py --rates 0 --steps 4 --sets "default:;exhaustive:early_exit=0;exhaustive_2x4_diag:early_exit=0,wide_diag_kernel=0" > gpurun_out/r06_tile_shape_kernel_slice.json 2> gpurun_out/r06_tile_shape_kernel_slice.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_tile_shape_kernel_slice.json"))
for name, x in d["rates"]["0"].items():
    if isinstance(x, dict) and "pair_kernels_ms" in x:
        print("kernel", name, "pair %.2f ms  executed %.1f TFLOP/s  skipped %.3f" % (x["pair_kernels_ms"], x["mfma_executed_tflops"], x["early_termination_skipped_frac"]))
PY
fi
