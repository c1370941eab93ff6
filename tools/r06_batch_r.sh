#!/bin/bash
# round 6, batch r: the complete-data tile kernel with the second half-stage's J fragments made during the first (wide_stage_pair) against the tree's kernel:
# parity tests on the new library, then the share and the slice alternately on the two libraries (lib/libldprune_hip_base.so = the kernel before)
# (The libraries beside the tree's are built by hand before the call: `git stash` / a -D switch, build_library(), cp lib/libldprune_hip.so lib/libldprune_hip_<name>.so -- git-ignored,
# they travel with the snapshot.  profiles/r06_experiments.md section 4b says which sources each one was.)
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=plink-ng_amd/lib
cp $L/libldprune_hip.so $L/libldprune_hip_new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_properties.py -m gpu -x -q -k "wide or tile or config3 or band or three_kernel" > gpurun_out/r06_${TAGR:-pf}_tests.log 2>&1
tail -3 gpurun_out/r06_${TAGR:-pf}_tests.log
timeout 300 python tests/fuzz_parity.py --seed 911 --cases 300 2>&1 | tail -1
B="python bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-pmc --no-e2e --no-cli-compare"
for round in 1 2; do
  for which in new base; do
    cp $L/libldprune_hip_$which.so $L/libldprune_hip.so
    timeout 600 $B > gpurun_out/r06_${TAGR:-pf}_share_${which}_$round.json 2>> gpurun_out/r06_${TAGR:-pf}_share.err
    timeout 600 python tools/c3miss_leg.py --rates 0 --steps 4 --sets "default:;exhaustive:early_exit=0" > gpurun_out/r06_${TAGR:-pf}_slice_${which}_$round.json 2>> gpurun_out/r06_${TAGR:-pf}_slice.err
  done
done
cp $L/libldprune_hip_new.so $L/libldprune_hip.so
python tools/ab_summary.py ${TAGR:-pf}
