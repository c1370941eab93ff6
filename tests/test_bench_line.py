"""bench.py's line, the parts that need no GPU: the flat scalar summary the driver's record can keep (it drops nested objects of `roofline` /
`cpu_baseline` and every other top-level key), and the helpers moved to tools/bench_support.py."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_flat_summary_repeats_the_legs_as_scalars():
    import bench
    out = {
        "roofline": {"bound": "mfma", "frac": 0.48, "mfma": {"nested": 1}},
        "power_and_clock": {"socket_power_w_median": 1345.0, "socket_power_cap_w": 1400.0, "shader_clock_mhz_median": 2190.0},
        "headline_bits_check": {"checked": True, "identical": True},
        "legs": {
            "config2": {"kernel": "pair_mfma_kernel", "ms_per_step": 7.4, "pair_kernels_ms": 3.6,
                        "roofline": {"bound": "hbm", "frac": 0.44, "traffic_over_compulsory": 1.2},
                        "missing_rate_0.001": {"kernel": "pair_mfma_kernel", "ms_per_step": 7.9, "vs_complete_data_step": 1.07},
                        "missing_rate_0.01": {"kernel": "pair_mfma_general_kernel", "ms_per_step": 15.0, "vs_complete_data_step": 2.0},
                        "cpu_baseline": {"value": 1.1e7, "prune_set_identical_to_hip": True}},
            "config5_density": {"kernel": "pair_mfma_tile4_kernel", "ms_per_step": 119.0, "pair_kernels_ms": 111.0, "roofline": {"bound": "mfma", "frac": 0.47},
                                "vs_complete_data_step_of_the_same_slice": 3.9, "complete_data_step_of_the_same_slice": {"ms_per_step": 30.5},
                                "reference_slice": {"prune_set_identical_to_hip": True}},
            "config3_density_missing": {"missing_rate_0.001": {"kernel": "pair_mfma_wide_kernel<SPARSE>", "ms_per_step": 39.0, "vs_complete_data_step": 1.3,
                                                                 "pairs_counted_exactly": 705, "traffic_over_compulsory": 3.6},
                                        "missing_rate_0.01": {"kernel": "pair_mfma_tile4_kernel", "ms_per_step": 93.0, "vs_complete_data_step": 3.1, "pairs_counted_exactly": 48}},
            "config4_tiles": {"error": "out of memory"},
        },
        "cpu_baseline": {"value": 2e6, "e2e_wall_s": {"reference_plink2": 133.0, "plink2_hip": 0.8, "speedup": 166.0, "files_identical": True, "pgen_bytes": 22095625012,
                                                       "plink2_hip_phases": {"file_to_hbm_gbs": 40.0},
                                                       "variable_width": {"reference_plink2_wall_s": 161.0, "speedup": 240.0, "pgen_bytes": 12646071129, "bytes_vs_fixed_width": 0.57,
                                                                          "make_pgen_s": 16.0, "files_identical_to_reference_on_the_same_file": True,
                                                                          "plink2_hip": {"wall_s": 0.67, "phases": {"file_to_hbm_s": 0.51}}}}},
        "e2e_n_gpus": {"gpus": 8, "identical_to_one_gpu": True, "plink2_hip": {"wall_s": 0.5, "rc": 0, "phases": {"file_to_hbm_s": 0.2}}},
    }
    bench.flatten_summary(out)
    r, cb = out["roofline"], out["cpu_baseline"]
    assert r["leg_config2_ms_per_step"] == 7.4 and r["leg_config2_frac"] == 0.44 and r["leg_config2_bound"] == "hbm" and r["leg_config2_traffic_x_compulsory"] == 1.2
    assert r["leg_config2_miss0.01_x_complete"] == 2.0 and r["leg_config2_reference_identical"] is True
    assert r["leg_config3_miss0.001_kernel"] == "pair_mfma_wide_kernel<SPARSE>" and r["leg_config3_miss0.001_x_complete"] == 1.3 and r["leg_config3_miss0.001_recounted_pairs"] == 705
    assert r["leg_config5_density_x_complete"] == 3.9 and r["leg_config3_density_complete_ms_per_step"] == 30.5 and r["leg_config5_density_reference_identical"] is True
    assert "leg_config4_tiles_error" in r
    assert r["timed_steps_socket_power_w_median"] == 1345.0 and r["headline_bits_check_identical"] is True
    assert r["e2e_plink2_hip_gpus"] == 8 and r["e2e_plink2_hip_gpus_wall_s"] == 0.5 and r["e2e_plink2_hip_gpus_identical_to_one_gpu"] is True
    assert cb["e2e_fixed_width_reference_s"] == 133.0 and cb["e2e_fixed_width_plink2_hip_s"] == 0.8 and cb["e2e_fixed_width_file_to_hbm_gbs"] == 40.0
    assert cb["e2e_variable_width_reference_s"] == 161.0 and cb["e2e_variable_width_plink2_hip_s"] == 0.67 and cb["e2e_variable_width_bytes_vs_fixed"] == 0.57
    # every key added is a scalar: that is the point
    for obj in (r, cb):
        for k, v in obj.items():
            if k.startswith(("leg_", "e2e_fixed", "e2e_variable", "e2e_plink2", "timed_steps_", "headline_bits")):
                assert not isinstance(v, (dict, list)), k
    json.dumps(out)


def test_config5_is_a_named_workload():
    import bench
    c5 = bench.CONFIGS["config5"]
    assert c5["missing_rate"] == 0.05 and c5["multiallelic_frac"] == 0.02 and c5["samples"] == 500000 and bench.PER_GPU_VARIANTS["config5"] == 1250000


def test_bench_support_is_importable_without_a_gpu():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_support as support
    assert support.REF_BIN.endswith("oracle/_ref/plink2") and support.CLI_BIN.endswith("plink-ng_amd/bin/plink2-hip")
    assert support._PHASES.search("[timing] setup+parse 0.081 s | genotype load (file -> HBM bit-planes) 0.516 s | run 0.064 s (pair kernel 41.6 ms, replay 0.6 ms; 272029960 candidate pairs) | x")
