#!/bin/bash
# round 5: phased records with several ALT alleles on the device -- the new tests, the pairphase suite, a fuzz campaign against the reference
mkdir -p gpurun_out/r05p
timeout 900 python -m pytest tests/test_pgen_device_decode.py tests/test_pairphase.py -q -m gpu -x > gpurun_out/r05p/tests.txt 2>&1
tail -15 gpurun_out/r05p/tests.txt
timeout 600 python tests/fuzz_cli.py --mode pairphase --cases ${1:-150} --seed 23 > gpurun_out/r05p/fuzz.txt 2>&1
tail -4 gpurun_out/r05p/fuzz.txt
