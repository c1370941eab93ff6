#!/bin/bash
# round 6, end: fresh seeds through every fuzzer on the final tree (the product library against the oracle / the reference binary)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_fuzz_campaign
mkdir -p $O
timeout 1100 python tests/fuzz_parity.py --seed 606 --cases 2000 > $O/fuzz_parity_seed606.txt 2>&1
timeout 900 python tests/fuzz_parity.py --wide-sparse --seed 607 --cases 1000 > $O/fuzz_wide_sparse_seed607.txt 2>&1
timeout 800 python tests/fuzz_parity.py --wide-missing --seed 608 --cases 1000 > $O/fuzz_wide_missing_seed608.txt 2>&1
timeout 1000 python tests/fuzz_cli.py --seed 609 --cases 300 > $O/fuzz_cli_seed609.txt 2>&1
for f in $O/*.txt; do echo "== $f"; tail -n 2 $f; done
