#!/bin/bash
# round 5: --clump over (variant, A1 allele) pairs -- the new GPU tests and a fuzz campaign against the reference
mkdir -p gpurun_out/r05m
timeout 600 python -m pytest tests/test_clump.py -q -m gpu -k "multiallelic" -x > gpurun_out/r05m/tests.txt 2>&1
tail -5 gpurun_out/r05m/tests.txt
timeout 500 python tests/fuzz_cli.py --mode clumpmulti --cases ${1:-120} --seed 11 --keep /tmp/fz > gpurun_out/r05m/fuzz.txt 2>&1
tail -4 gpurun_out/r05m/fuzz.txt
