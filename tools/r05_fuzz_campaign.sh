#!/bin/bash
# tools/r05_fuzz_campaign.sh -- GPU box: the randomised differential drivers with other seeds and many more cases than the suite runs
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05h
mkdir -p $O
timeout 400 python tests/fuzz_parity.py --cases 3000 --seed 5 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout 300 python tests/fuzz_parity.py --wide-missing --cases 2000 --seed 6 > $O/fuzz_wm.txt 2>&1; tail -1 $O/fuzz_wm.txt
timeout 300 python tests/fuzz_parity.py --wide-async --cases 2000 --seed 7 > $O/fuzz_async.txt 2>&1; tail -1 $O/fuzz_async.txt
timeout 900 python tests/fuzz_cli.py --cases 500 --seed 3 > $O/fuzz_cli.txt 2>&1; tail -1 $O/fuzz_cli.txt
timeout 500 python tests/fuzz_cli.py --mode sexmulti --cases 400 --seed 9 > $O/fuzz_sexmulti.txt 2>&1; tail -1 $O/fuzz_sexmulti.txt
timeout 500 python tests/fuzz_cli.py --mode pairphase --cases 200 --seed 4 > $O/fuzz_pairphase.txt 2>&1; tail -1 $O/fuzz_pairphase.txt
timeout 500 python tests/fuzz_cli.py --mode clump --cases 150 --seed 8 > $O/fuzz_clump.txt 2>&1; tail -1 $O/fuzz_clump.txt
