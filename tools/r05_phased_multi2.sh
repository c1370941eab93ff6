#!/bin/bash
mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_pairphase.py tests/test_pgen_device_decode.py tests/test_cli.py -q -m gpu -x > gpurun_out/r05q/tests.txt 2>&1
tail -5 gpurun_out/r05q/tests.txt
timeout 600 python tools/bench_decode.py --variants 4000 --out gpurun_out/r05q/decode.jsonl > gpurun_out/r05q/decode.txt 2>&1
tail -c 1500 gpurun_out/r05q/decode.txt
