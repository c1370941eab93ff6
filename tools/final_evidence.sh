set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -6) > $O/gpu_suite.txt
(timeout 500 bash tools/profile.sh r04) > $O/prof_share.log 2>&1
(LDP_PROF_ARGS="--workload config2" timeout 400 bash tools/profile.sh r04_config2) > $O/prof_config2.log 2>&1
(LDP_PROF_ARGS="--variants 120000" timeout 400 bash tools/profile.sh r04_c3shape) > $O/prof_c3shape.log 2>&1
(LDP_PROF_ARGS="--variants 120000 --missing-rate 0.05" LDP_PROF_TRACE_STEPS=2 timeout 500 bash tools/profile.sh r04_c5shape) > $O/prof_c5shape.log 2>&1
(LDP_PROF_ARGS="--workload config2 --missing-rate 0.01" LDP_PROF_TRAFFIC_ONLY=1 LDP_PROF_TRACE_STEPS=3 timeout 300 bash tools/profile.sh r04_miss01) > $O/prof_miss01.log 2>&1
cd $GRAFT_REPO_ROOT
for d in gpurun_out/profiles_r04 gpurun_out/profiles_r04_config2 gpurun_out/profiles_r04_c3shape gpurun_out/profiles_r04_c5shape gpurun_out/profiles_r04_miss01; do cp $d/* profiles/ 2>/dev/null; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python tests/cli_e2e.py > $O/cli_e2e.txt 2>&1
timeout 300 python tests/fuzz_parity.py --cases 600 > $O/fuzz.txt 2>&1
timeout 300 python tests/fuzz_parity.py --wide-missing --cases 450 > $O/fuzz_wm.txt 2>&1
timeout 300 python tests/fuzz_cli.py --cases 100 > $O/fuzz_cli.txt 2>&1
timeout 400 python tools/bench_decode.py --out $O/decode.jsonl > $O/decode.log 2>&1
timeout 200 python tools/bench_r2.py --samples 50000 --variants 16384 --chrx 1024 > $O/chrx_50k.json 2> $O/chrx.err
timeout 200 python tools/bench_r2.py --samples 500000 --variants 8192 --chrx 512 > $O/chrx_500k.json 2>> $O/chrx.err
cat $O/gpu_suite.txt; tail -c 400 $O/bench_n1.json; tail -n 3 $O/fuzz.txt $O/fuzz_wm.txt $O/fuzz_cli.txt 2>/dev/null | tail -n 12
