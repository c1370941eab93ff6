#!/bin/bash
mkdir -p gpurun_out/r05s
timeout 600 python -m pytest tests/test_pairphase.py tests/test_cli.py -q -m gpu -x > gpurun_out/r05s/tests.txt 2>&1
tail -6 gpurun_out/r05s/tests.txt
timeout 500 python tests/fuzz_cli.py --mode sexmultiphase --cases ${1:-150} --seed 41 > gpurun_out/r05s/fuzz_sexmultiphase.txt 2>&1
tail -3 gpurun_out/r05s/fuzz_sexmultiphase.txt
timeout 400 python tests/fuzz_cli.py --mode clumpmulti --cases ${1:-150} --seed 32 > gpurun_out/r05s/fuzz_clumpmulti.txt 2>&1
tail -3 gpurun_out/r05s/fuzz_clumpmulti.txt
