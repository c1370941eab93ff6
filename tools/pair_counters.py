#!/usr/bin/env python3
"""Pair-kernel counters of one synthetic shape on the GPU box: tools/pair_counters.py SAMPLES VARIANTS MISSING_RATE WINDOW_KB R2 [SPACING_BP]
(candidate pairs, kernel milliseconds, block products, product k-steps run and skipped).  Used for profiles/r02_experiments.md."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]); m = int(sys.argv[2]); mr = float(sys.argv[3]); kb = float(sys.argv[4]); r2 = float(sys.argv[5]); spacing = int(sys.argv[6]) if len(sys.argv) > 6 else 2875
chr_idx, bps = bench.genome_layout(m, 1, spacing)
stride = (n + 3) // 4
buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
pkg.synth_genotypes_device(bench.SEED, 0, m, n, mr, buf.data_ptr(), stride)
torch.cuda.synchronize()
eng = pkg.LdPruneEngine(n, int(kb * 1000), 1, True, r2, device=0)
eng.set_variants(chr_idx, bps)
for it in range(2):
    eng.load_genotypes_device(0, m, buf.data_ptr(), stride, pkg.LDP_GENO_REF)
    rem = eng.run()
c = eng.counters()
keys = ["candidate_pairs", "ms_pair_mfma", "ms_pair_mfma_general", "mfma_block_products", "mfma_product_stages", "mfma_skipped_product_stages", "ms_prepare", "sparse_exact_pairs", "pred_true"]
print(json.dumps({k: c.get(k) for k in keys if k in c}), int(rem.sum()))
