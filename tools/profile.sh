#!/bin/bash
# tools/profile.sh [round-tag] -- run on the GPU box (gpurun).  Collects, for `bench.py` at its default
# workload: (1) rocprofv3 --kernel-trace --stats, (2) separate --pmc passes (kernel-trace only, as
# MI355X_MICROARCH.md prescribes), and writes the raw CSVs under gpurun_out/prof_<tag>/ plus the summaries
# tools/summarize_prof.py derives under gpurun_out/profiles_<tag>/ (copy those into profiles/).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-legs ${LDP_PROF_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps ${LDP_PROF_TRACE_STEPS:-5} --warmup 1 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" \
         "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "${LDP_PROF_TRAFFIC_ONLY:-}" ] && [ $i -lt 4 ]; then continue; fi   # (only the two passes roofline.traffic comes from)
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- $BENCH --steps 1 --warmup 0 > $OUT/pmc$i.log 2>&1
done
python $R/tools/summarize_prof.py $OUT $R/gpurun_out/profiles_$TAG $TAG
