# tools/final_evidence.sh -- GPU box, end of round 5: PMC profiles of every shape the bench line replays (they are valid only for the
# kernel sources they were collected on: bench.py compares the hashes), the driver's bench command, the GPU suite, fuzzers, end-to-end
# CLI timings, decode and chrX rates.  Results under gpurun_out/final; the summaries are copied into profiles/ by hand afterwards.
set -u
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/final
mkdir -p $O
(timeout 500 bash tools/profile.sh r05) > $O/prof_share.log 2>&1
(LDP_PROF_ARGS="--workload config2" timeout 400 bash tools/profile.sh r05_config2) > $O/prof_config2.log 2>&1
(LDP_PROF_ARGS="--variants 120000" timeout 400 bash tools/profile.sh r05_c3shape) > $O/prof_c3shape.log 2>&1
(LDP_PROF_ARGS="--variants 120000 --missing-rate 0.05" LDP_PROF_TRACE_STEPS=2 timeout 500 bash tools/profile.sh r05_c5shape) > $O/prof_c5shape.log 2>&1
(LDP_PROF_ARGS="--workload config2 --missing-rate 0.01" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=3 timeout 300 bash tools/profile.sh r05_miss01) > $O/prof_miss01.log 2>&1
(LDP_PROF_ARGS="--workload config2 --missing-rate 0.001" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=3 timeout 300 bash tools/profile.sh r05_miss001) > $O/prof_miss001.log 2>&1
(LDP_PROF_ARGS="--only-config4" timeout 600 bash tools/profile.sh r05_c4) > $O/prof_c4.log 2>&1
cd $GRAFT_REPO_ROOT
for d in gpurun_out/profiles_r05 gpurun_out/profiles_r05_config2 gpurun_out/profiles_r05_c3shape gpurun_out/profiles_r05_c5shape gpurun_out/profiles_r05_miss01 gpurun_out/profiles_r05_miss001 gpurun_out/profiles_r05_c4; do cp $d/* profiles/ 2>/dev/null; done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_n1.time
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -6) > $O/gpu_suite.txt
timeout 400 python tests/cli_e2e.py > $O/cli_e2e.txt 2>&1
timeout 300 python tests/cli_e2e.py --pgen > $O/cli_e2e_pgen.txt 2>&1
timeout 240 python tests/fuzz_parity.py --cases 400 > $O/fuzz.txt 2>&1
timeout 240 python tests/fuzz_parity.py --wide-missing --cases 300 > $O/fuzz_wm.txt 2>&1
timeout 200 python tests/fuzz_parity.py --wide-async --cases 250 > $O/fuzz_async.txt 2>&1
timeout 240 python tests/fuzz_cli.py --cases 80 > $O/fuzz_cli.txt 2>&1
timeout 300 python tools/bench_decode.py --out $O/decode.jsonl > $O/decode.log 2>&1
cat $O/gpu_suite.txt; tail -3 $O/bench_n1.time; tail -c 300 $O/bench_n1.json; tail -n 3 $O/fuzz.txt $O/fuzz_wm.txt $O/fuzz_async.txt $O/fuzz_cli.txt 2>/dev/null | tail -n 12; tail -4 $O/cli_e2e.txt
