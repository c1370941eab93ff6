# tools/r06_final_evidence.sh -- GPU box, round 6: PMC profiles of every shape the bench line replays (valid only for the kernel sources they were
# collected on: bench.py compares the hashes), the driver's bench command, config 5's per-GPU share as the main line, the GPU suite, fuzzers,
# end-to-end CLI timings.  Results under gpurun_out/final6; the summaries are copied into profiles/ by hand afterwards.
set -u
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=${O:-gpurun_out/final6}
mkdir -p $O
(timeout 500 bash tools/profile.sh r06) > $O/prof_share.log 2>&1
(LDP_PROF_ARGS="--workload config2" timeout 400 bash tools/profile.sh r06_config2) > $O/prof_config2.log 2>&1
(LDP_PROF_ARGS="--variants 120000" timeout 400 bash tools/profile.sh r06_c3shape) > $O/prof_c3shape.log 2>&1
(LDP_PROF_ARGS="--variants 120000 --missing-rate 0.001" LDP_PROF_TRACE_STEPS=3 timeout 400 bash tools/profile.sh r06_c3miss001) > $O/prof_c3miss001.log 2>&1
(LDP_PROF_ARGS="--variants 120000 --missing-rate 0.01" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=2 timeout 400 bash tools/profile.sh r06_c3miss01) > $O/prof_c3miss01.log 2>&1
(LDP_PROF_ARGS="--variants 120000 --missing-rate 0.05" LDP_PROF_TRACE_STEPS=2 timeout 500 bash tools/profile.sh r06_c5shape) > $O/prof_c5shape.log 2>&1
(LDP_PROF_ARGS="--workload config2 --missing-rate 0.01" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=3 timeout 300 bash tools/profile.sh r06_miss01) > $O/prof_miss01.log 2>&1
(LDP_PROF_ARGS="--workload config2 --missing-rate 0.001" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=3 timeout 300 bash tools/profile.sh r06_miss001) > $O/prof_miss001.log 2>&1
(LDP_PROF_ARGS="--only-config4" timeout 600 bash tools/profile.sh r06_c4) > $O/prof_c4.log 2>&1
cd $GRAFT_REPO_ROOT
for d in gpurun_out/profiles_r06 gpurun_out/profiles_r06_config2 gpurun_out/profiles_r06_c3shape gpurun_out/profiles_r06_c3miss001 gpurun_out/profiles_r06_c3miss01 gpurun_out/profiles_r06_c5shape gpurun_out/profiles_r06_miss01 gpurun_out/profiles_r06_miss001 gpurun_out/profiles_r06_c4; do cp $d/* profiles/ 2>/dev/null; done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time
( time timeout 1500 python bench.py --workload config5 --steps 3 --warmup 1 --no-legs --no-e2e > $O/config5_share.json 2> $O/config5_share.err ) 2> $O/config5_share.time
( LDP_BENCH_ALT_MINOR=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-legs --no-cpu-baseline --no-pmc > $O/share_alt_minor.json 2> $O/share_alt_minor.err )
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -6) > $O/gpu_suite.txt
timeout 400 python tests/cli_e2e.py > $O/cli_e2e.txt 2>&1
timeout 300 python tests/cli_e2e.py --pgen > $O/cli_e2e_pgen.txt 2>&1
timeout 240 python tests/fuzz_parity.py --cases 400 > $O/fuzz.txt 2>&1
timeout 240 python tests/fuzz_parity.py --wide-missing --cases 300 > $O/fuzz_wm.txt 2>&1
timeout 300 python tests/fuzz_parity.py --wide-sparse --cases 300 > $O/fuzz_ws.txt 2>&1
timeout 240 python tests/fuzz_cli.py --cases 80 > $O/fuzz_cli.txt 2>&1
python tools/nengine_load.py > $O/nengine_load.txt 2> $O/nengine_load.err
cat $O/gpu_suite.txt; tail -3 $O/bench_n1.time; tail -c 300 $O/bench_n1.json; tail -3 $O/config5_share.time; tail -n 2 $O/fuzz.txt $O/fuzz_wm.txt $O/fuzz_ws.txt $O/fuzz_cli.txt 2>/dev/null | tail -n 12; tail -4 $O/cli_e2e.txt
