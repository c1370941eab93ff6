"""Collected first (the driver runs `pytest -x`): ONE small oracle / reference comparison per row of SURVEY.md section 8, each a
representative case of the full test of that row elsewhere in tests/.  A late failure in a long module can then never leave a
row without evidence on the driver's GPU box.  Everything here runs in well under a minute; the exhaustive grids stay where
they are."""
import os

import numpy as np
import pytest

import ldtools as T
import test_cli as CLI
import test_clump as CLUMP
import test_golden as GOLD
import test_gpu_parity as P
import test_integration_binding as BIND
import test_pairphase as PH
import test_pgen_device_decode as DEC
import test_r2_unphased as R2
import test_sample_map as SM
from test_cli import cli  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_rows_a6_a7_a10_a11_a19_conversion_aggregates_allele_counts(gpu_pkg):
    """PgrGetInv1 & co. / VariantAggs / SplitHomRef2het / FillVaggs / allele-frequency pass: planes, aggregates, flags and the
    exact maj_freq doubles against the oracle, one sample count per encoding."""
    P.test_prepare_planes_and_aggregates(gpu_pkg, 1025, "ref")
    P.test_prepare_planes_and_aggregates(gpu_pkg, 65, "bed")
    P.test_prepare_planes_and_aggregates(gpu_pkg, 100, "inverse")


def test_rows_a8_a9_a12_a16_a17_scan_window_dot_decision(gpu_pkg):
    """IndepPairwiseThread / window iterator / DotprodWords / decision + tie-break / stitch: every candidate pair's 6-tuple and
    the prune set against the oracle (complete data, both orders, bp and count windows)."""
    P.test_tile_kernel_and_prune_set(gpu_pkg, P.RUN_CASES[0])
    P.test_tile_kernel_and_prune_set(gpu_pkg, P.RUN_CASES[4])


def test_rows_a13_a14_a15_missing_calls_and_dispatch(gpu_pkg):
    """SumSsqWords / SumSsqNmWords / R2Components dispatch: the hand-checked KAT integers, mixed tiles, and the device-side route
    (read from the route words, not from timings)."""
    GOLD.test_hip_kat_integers(gpu_pkg)
    P.test_mixed_missingness_tiles(gpu_pkg)
    P.test_rows_with_a_few_missing_calls(gpu_pkg, 9000, 0.001, 0.8, 0.1, None)
    P.test_partial_reload_keeps_the_route_of_the_resident_rows(gpu_pkg)


def test_rows_a3_a4_a5_split_balance_orchestration(gpu_pkg):
    """LdPruneSubcontigSplitAll / LoadBalance / IndepPairwise orchestration: shards over 3 ranks give the unsharded set;
    device-pointer input; the reference's golden prune sets."""
    P.test_sharded_union_equals_unsharded(gpu_pkg, 3)
    P.test_device_pointer_input(gpu_pkg)
    GOLD.test_hip_reproduces_reference(gpu_pkg, GOLD.GOLDEN[0], "ref")


def test_rows_a1_a2_a18_cli_files_byte_identical(gpu_pkg, cli, tmp_path):
    """LdPrune setup / StripUnplaced / LdPruneWrite: plink2-hip's .prune.in/.prune.out against the reference binary's (chr0
    variants, non-founders, .bed input)."""
    CLI.test_cli_byte_identical_to_reference(gpu_pkg, cli, tmp_path, CLI.CLI_CASES[4])


def test_row_f1_variable_width_pgen(gpu_pkg, cli, tmp_path):
    CLI.test_cli_byte_identical_to_reference(gpu_pkg, cli, tmp_path, CLI.CLI_CASES[7])


def test_row_f1_records_decoded_on_the_device(gpu_pkg, tmp_path):
    """ldp_load_pgen_records: the committed reference-written file (every main-track record type) and a VCF-imported multiallelic
    fileset (aux track 1, major-vs-rest collapse) against the host reader."""
    DEC.test_committed_variable_width_file(gpu_pkg)
    if T.have_ref():
        DEC.test_multiallelic_records_are_collapsed_on_the_device(gpu_pkg, tmp_path, 150, 300, 2, 5)


def test_row_a20_r2_matrix(gpu_pkg, tmp_path):
    """--r2-unphased matrices: rows against the oracle's doubles, one reference-written binary file."""
    R2.test_matrix_rows_match_oracle(gpu_pkg, 150, 130, 0.05)
    R2.test_cli_matrix_files_byte_identical(gpu_pkg, tmp_path, "triangle", "bin")


def test_row_f3_r2_table(gpu_pkg, tmp_path):
    """--r2-unphased windowed table (filters) and the inter-chr table, byte-identical to the reference's .vcor."""
    R2.test_cli_vcor_table_byte_identical(gpu_pkg, tmp_path, R2.VCOR_CASES[1])


def test_row_f3_r2_inter_chr(gpu_pkg, tmp_path):
    R2.test_cli_inter_chr_table_byte_identical(gpu_pkg, tmp_path, ["--ld-window-r2", "0.02"])


def test_row_f4_pairphase():
    PH.test_hip_pairphase_matches_reference_golden()
    PH.test_hip_pairphase_matches_oracle(100, 0.03)


def test_row_f4_clump(gpu_pkg, cli, tmp_path):
    CLUMP.test_clump_matches_reference(gpu_pkg, cli, tmp_path, CLUMP.CLUMP_CASES[3])


def test_row_b_reference_side_binding(gpu_pkg, tmp_path):
    """the reference's own LdPrune() calling the C-ABI library (oracle/_ref/plink2_hipld) writes the stock
    reference's files."""
    BIND.test_reference_ldprune_through_the_c_abi_matches_stock_reference(gpu_pkg, tmp_path, 1)


def test_row_b_binding_pairphase(gpu_pkg, tmp_path):
    BIND.test_reference_pairphase_through_the_c_abi_matches_stock_reference(gpu_pkg, tmp_path, ["30kb", "0.5"], 0)


def test_row_a5_sex_chromosome_sample_map(gpu_pkg):
    SM.test_mapped_rows_match_host_built_rows(gpu_pkg, 90, 31, 29, "ref")


def test_matrix_pipe_exactness_limit(gpu_pkg):
    """Either side of kMfMaxFounders the tile kernels' 6-tuples equal the one-wave-per-pair reference kernel's."""
    P.test_sample_counts_around_the_matrix_pipe_limit(gpu_pkg, 0, 0.0)
    P.test_sample_counts_around_the_matrix_pipe_limit(gpu_pkg, 257, 0.01)


def test_row_e_rccl_allgather_from_the_c_abi(gpu_pkg):
    P.test_rccl_allgather_from_the_c_abi_with_one_rank(gpu_pkg)


def test_row_a12_wide_band_tiles(gpu_pkg):
    """config 3's band shape (a window of many row-blocks): the 8 x 8 tile kernel against the oracle"""
    P.test_wide_band_tiles_match_oracle(gpu_pkg, P.WIDE_CASES[0])


def test_row_e_bench_step_under_torchrun(gpu_pkg, tmp_path):
    """One rank under torch.distributed.run: RCCL is initialised and the bitmap exchange runs (an identity at one rank)."""
    import test_config3_parity as C3
    C3.test_bench_step_under_torchrun_initialises_rccl(gpu_pkg, tmp_path)
