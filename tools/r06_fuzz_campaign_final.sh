#!/bin/bash
# round 6, after the last kernel change (the stage form and wave map of pair_mfma_wide_kernel): fresh seeds through every fuzzer once more
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_fuzz_campaign_final
mkdir -p $O
timeout 1100 python tests/fuzz_parity.py --seed 706 --cases 2000 > $O/fuzz_parity_seed706.txt 2>&1
timeout 900 python tests/fuzz_parity.py --wide-sparse --seed 707 --cases 1000 > $O/fuzz_wide_sparse_seed707.txt 2>&1
timeout 800 python tests/fuzz_parity.py --wide-missing --seed 708 --cases 1000 > $O/fuzz_wide_missing_seed708.txt 2>&1
timeout 1000 python tests/fuzz_cli.py --seed 709 --cases 300 > $O/fuzz_cli_seed709.txt 2>&1
for f in $O/*.txt; do echo "== $f"; tail -n 2 $f; done
