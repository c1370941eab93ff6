// plink2_hip_cli.cpp -- `plink2-hip`: process-level drop-in for the --indep-pairwise path of plink2.
//
// Same flag spellings (2.0/plink2.cc:7238-7337, 2.0/plink2_help.cc:948-969), same inputs
// (.bed/.bim/.fam, fixed- and variable-width .pgen/.pvar/.psam) and byte-identical <out>.prune.in / <out>.prune.out
// (LdPruneWrite, 2.0/plink2_ld.cc:2464-2528).  Everything between "files are open" and "bitmap of removed
// variants" goes through the C ABI in include/ldprune_hip.h, i.e. through the HIP kernels; there is no CPU
// compute path here, so without a usable GPU the program exits with an error.
//
// What this front-end does itself (host C++): argument parsing incl. the reference's decimal scanner
// (ScanadvDouble, 2.0/include/plink2_string.cc:1264-1528), .fam/.psam founder detection
// (2.0/plink2_psam.cc:804-813), .bim/.pvar parsing, chromosome-0 stripping (StripUnplacedK,
// plink2_ld.cc:113-164), the sorted-positions and unique-ID checks (plink2.cc:2926, plink2_ld.cc:2573-2592),
// the <50-founders guard (plink2.cc:2063-2071) and the output writer.
// Multiallelic variants are collapsed major-vs-rest on the host (Get1Multiallelic semantics); chrX / chrY / MT get
// their sample-mapped rows (males het->missing, non-males x2, ...) built on the host as well.
// --r2-unphased / --r-unphased: the matrix shapes (square/square0/triangle as bin, bin4 or text, zs), the windowed and the
// inter-chr .vcor table with cols=, --ld-window, --ld-window-kb, --ld-window-cm, --ld-window-r2, --ld-snp / --ld-snps / --ld-snp-list,
// --parallel; number formatting restated from dtoa_g.  --clump (several reports, --clump-allow-overlap, cols=, bins, -log10,
// ranges, sex chromosomes, (variant, A1 allele) pairs of multiallelic sites).
// Not yet supported (reported as such with exit 63, never silently mis-handled): dosage data outside --indep-pairwise on the autosomes,
// more than 254 ALT alleles (multiallelic sites on chrX/Y/MT are taken by every command since round 5), major-allele-oriented
// r^2 outputs on chrY/MT.
// The front-end is split into translation units of one concern each (p2h_cli.h: what they share): p2h_util.cpp (logging, number
// scanning / formatting), p2h_args.cpp (command line), p2h_tables.cpp (.psam / .pvar tables, host-built rows), p2h_inputs.cpp (filters,
// load_inputs), p2h_clump.cpp (--clump), p2h_r2.cpp (--r2-unphased outputs), p2h_prune.cpp (--indep-pairwise / --indep-pairphase); this file
// keeps main().
#include "p2h_cli.h"

using namespace p2h;


// test hook (no GPU needed): lines `both_x unsquared flip1 flip2 mflip1 mflip2  nm sum1 ssq1 sum2 ssq2 dot  (the same six for the male
// founders)` in, the chrX-weighted r^2 (or r) out as the hex bits of the double -- XWeighted's arithmetic against values the
// reference computed (tests/test_r2_flags.py)
int debug_xweighted(const char* path) {
  FILE* df = fopen(path, "r");
  if (!df) {
    die(3, "Error: Failed to open %s.\n", path);
  }
  double nan_ref;
  {
    const uint64_t bits = 0xfff8000000000000ull;
    memcpy(&nan_ref, &bits, 8);
  }
  int bx, us, f1, f2, g1, g2;
  long long v[12];
  while (fscanf(df, "%d %d %d %d %d %d %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld", &bx, &us, &f1, &f2, &g1, &g2, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5],
                &v[6], &v[7], &v[8], &v[9], &v[10], &v[11]) == 18) {
    ldp_pair_stats_t ta, tm;
    ta.nm = static_cast<uint32_t>(v[0]); ta.sum1 = static_cast<int32_t>(v[1]); ta.ssq1 = static_cast<uint32_t>(v[2]);
    ta.sum2 = static_cast<int32_t>(v[3]); ta.ssq2 = static_cast<uint32_t>(v[4]); ta.dot = static_cast<int32_t>(v[5]);
    tm.nm = static_cast<uint32_t>(v[6]); tm.sum1 = static_cast<int32_t>(v[7]); tm.ssq1 = static_cast<uint32_t>(v[8]);
    tm.sum2 = static_cast<int32_t>(v[9]); tm.ssq2 = static_cast<uint32_t>(v[10]); tm.dot = static_cast<int32_t>(v[11]);
    const double r = XWeighted::weighted(XWeighted::counts(ta, f1 != 0, f2 != 0), XWeighted::counts(tm, g1 != 0, g2 != 0), bx != 0, us != 0, nan_ref);
    uint64_t bits;
    memcpy(&bits, &r, 8);
    printf("%016llx\n", static_cast<unsigned long long>(bits));
  }
  fclose(df);
  return 0;
}

int main(int argc, char** argv) {
  if ((argc == 3) && (strcmp(argv[1], "--debug-xweighted") == 0)) {
    return debug_xweighted(argv[2]);
  }
  Session S;
  load_inputs(S, argc, argv);
  return S.A.have_r2 ? run_r2(S) : run_prune(S);
}
