#!/bin/bash
mkdir -p gpurun_out/r05x
timeout 1200 python -m pytest tests/test_r2_unphased.py tests/test_clump.py tests/test_r2_flags.py -q -m gpu -x > gpurun_out/r05x/tests.txt 2>&1
tail -8 gpurun_out/r05x/tests.txt
timeout 400 python tests/fuzz_cli.py --mode clump --cases ${1:-80} --seed 31 > gpurun_out/r05x/fuzz_clump.txt 2>&1
tail -3 gpurun_out/r05x/fuzz_clump.txt
timeout 400 python tests/fuzz_cli.py --mode clumpmulti --cases ${1:-80} --seed 32 > gpurun_out/r05x/fuzz_clumpmulti.txt 2>&1
tail -3 gpurun_out/r05x/fuzz_clumpmulti.txt
timeout 500 python tests/fuzz_cli.py --mode all --cases ${2:-150} --seed 33 > gpurun_out/r05x/fuzz_all.txt 2>&1
tail -3 gpurun_out/r05x/fuzz_all.txt
