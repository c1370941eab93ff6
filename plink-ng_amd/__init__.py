"""plink-ng_amd: MI355X-native --indep-pairwise (pairwise-LD pruning) engine.

The product is the C-ABI shared library built from csrc/ (include/ldprune_hip.h) plus the C++
`plink2-hip` front-end.  This module is only the thin ctypes plumbing that tests/ and bench.py use to
drive the C ABI from Python; it holds no algorithmic code and has NO CPU fallback: if the HIP library
is missing or no GPU is usable, compute calls raise.

The directory name contains a hyphen, so import it through `load_package()` in __graft_entry__.py
(registered as module `plink_ng_amd`).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
BIN_DIR = os.path.join(_HERE, "bin")
LIB_PATH = os.path.join(LIB_DIR, "libldprune_hip.so")
# the measurement build (-DLDP_MEASURE, csrc/ldp_env.h): environment presets, timing prints and the ablation kernels of tools/ --
# never loaded by tests/, bench.py's default run or plink2-hip; LDP_LIB_MEASURE=1 makes lib() load it (tools/attribution.py)
MEASURE_LIB_PATH = os.path.join(LIB_DIR, "libldprune_hip_measure.so")
CLI_PATH = os.path.join(BIN_DIR, "plink2-hip")

LDP_OK, LDP_ERR_INVALID, LDP_ERR_NOMEM, LDP_ERR_GPU, LDP_ERR_STATE, LDP_ERR_UNSUPPORTED = range(6)
LDP_GENO_INVERSE, LDP_GENO_REF, LDP_GENO_BED = 0, 1, 2
LDP_GENO_PHASED = 4  # OR into INVERSE / REF: --indep-pairphase rows (include/ldprune_hip.h)
LDP_GENO_MAPPED = 8  # OR into REF / BED: file-wide rows, columns gathered on the device (ldp_set_sample_map)
LDP_ERR_UNPHASED = 6
LDP_MEM_HOST, LDP_MEM_DEVICE = 0, 1

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


class LdpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ldp error %d: %s" % (code, msg))
        self.code = code


class ldp_params(ctypes.Structure):
    _fields_ = [("founder_ct", ctypes.c_uint32), ("prune_window_size", ctypes.c_uint32),
                ("prune_window_incr", ctypes.c_uint32), ("window_is_bp", ctypes.c_uint32),
                ("plink1_order", ctypes.c_uint32), ("prune_last_param", ctypes.c_double),
                ("device", ctypes.c_int32), ("stream", ctypes.c_void_p)]


class ldp_pair_stats_t(ctypes.Structure):
    _fields_ = [("nm", ctypes.c_uint32), ("sum1", ctypes.c_int32), ("ssq1", ctypes.c_uint32),
                ("sum2", ctypes.c_int32), ("ssq2", ctypes.c_uint32), ("dot", ctypes.c_int32)]


class ldp_variant_rec(ctypes.Structure):
    _fields_ = [("nm_ct", ctypes.c_uint32), ("sum", ctypes.c_int32), ("ssq", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("n_homref", ctypes.c_uint32), ("n_het", ctypes.c_uint32),
                ("n_homalt", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class ldp_pgen_rec(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("length", ctypes.c_uint32), ("vrtype", ctypes.c_uint8), ("allele_ct", ctypes.c_uint8),
                ("reserved", ctypes.c_uint16)]


class ldp_counters(ctypes.Structure):
    _fields_ = [("candidate_pairs", ctypes.c_uint64), ("computed_pairs", ctypes.c_uint64),
                ("replay_pairs", ctypes.c_uint64), ("pred_true", ctypes.c_uint64),
                ("ms_prepare", ctypes.c_double), ("ms_pair_kernel", ctypes.c_double),
                ("ms_pair_fast", ctypes.c_double), ("ms_pair_general", ctypes.c_double),
                ("ms_replay", ctypes.c_double), ("ms_run_total", ctypes.c_double),
                ("pair_kernel_launches", ctypes.c_uint32), ("subcontig_ct", ctypes.c_uint32),
                ("owned_subcontig_ct", ctypes.c_uint32), ("window_max", ctypes.c_uint32),
                ("tile_unit_chunks", ctypes.c_uint64), ("early_exit_unit_chunks", ctypes.c_uint64),
                ("ms_pair_mfma", ctypes.c_double), ("mfma_block_products", ctypes.c_uint64),
                ("mfma_product_stages", ctypes.c_uint64), ("mfma_skipped_product_stages", ctypes.c_uint64),
                ("ms_pair_mfma_general", ctypes.c_double), ("sparse_exact_pairs", ctypes.c_uint64),
                ("route_complete_launches", ctypes.c_uint32), ("route_sparse_launches", ctypes.c_uint32),
                ("route_general_launches", ctypes.c_uint32), ("wide_tiles", ctypes.c_uint32),
                ("mfma_extra_product_stages", ctypes.c_uint64), ("four_tile_launches", ctypes.c_uint32),
                ("decoded_in_place_rows", ctypes.c_uint32), ("sparse_tile_launches", ctypes.c_uint32), ("reserved0", ctypes.c_uint32)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


PAIR_STATS_DTYPE = np.dtype([("nm", "<u4"), ("sum1", "<i4"), ("ssq1", "<u4"), ("sum2", "<i4"), ("ssq2", "<u4"), ("dot", "<i4")])
R2_HIT_DTYPE = np.dtype([("first", "<u4"), ("second", "<u4"), ("r2", "<f8")])
VARIANT_REC_DTYPE = np.dtype([("nm_ct", "<u4"), ("sum", "<i4"), ("ssq", "<u4"), ("flags", "<u4"), ("n_homref", "<u4"),
                              ("n_het", "<u4"), ("n_homalt", "<u4"), ("reserved", "<u4")])

# Every symbol include/ldprune_hip.h declares (checked by tests/test_cabi_symbols.py).
CABI_SYMBOLS = [
    "ldp_create", "ldp_destroy", "ldp_last_error", "ldp_device_count", "ldp_prewarm", "ldp_set_variants", "ldp_get_subcontigs",
    "ldp_set_shard", "ldp_get_band", "ldp_load_genotypes", "ldp_load_genotypes_fd", "ldp_set_sample_map", "ldp_set_maj_freqs", "ldp_set_preferred", "ldp_run",
    "ldp_run_with_stats", "ldp_pair_stats", "ldp_debug_set_variant_recs", "ldp_debug_replay_pairs", "ldp_debug_mfma_plan",
    "ldp_get_variant_recs", "ldp_get_maj_freqs", "ldp_get_planes", "ldp_get_counters", "ldp_synth_genotypes",
    "ldp_set_variants_matrix", "ldp_r2_unphased_rows", "ldp_r2_unphased_hits", "ldp_r2_unphased_block", "ldp_r2_unphased_block_hits", "ldp_r2_unphased_block_x", "ldp_r2_unphased_block_x_hits", "ldp_pair_stats_block", "ldp_set_variants_vcor", "ldp_r2_unphased_band_rows",
    "ldp_pgen_open", "ldp_pgen_info", "ldp_pgen_has_dosage", "ldp_pgen_variant_has_dosage", "ldp_pgen_dosage_sums", "ldp_pgen_direct_rows", "ldp_pgen_direct_fd", "ldp_pgen_read", "ldp_pgen_last_error", "ldp_pgen_close",
    "ldp_pgen_variant_is_multiallelic", "ldp_pgen_provisional_ref", "ldp_pgen_open_indexed", "ldp_set_r_signed", "ldp_set_variants_vcor_cm", "ldp_pgen_read_alleles", "ldp_pgen_read_phased", "ldp_pgen_read_alleles_phased", "ldp_subset_samples", "ldp_phased_row_bytes", "ldp_phased_phase_offset",
    "ldp_debug_set_option", "ldp_pgen_debug_force_portable", "ldp_matrix_pipe_max_founders", "ldp_map_rows", "ldp_release_device", "ldp_debug_wide_plan",
    "ldp_allgather_removed", "ldp_comm_init_all", "ldp_comm_destroy", "ldp_shard_segment_words", "ldp_pack_removed_segment", "ldp_stitch_removed_segments", "ldp_load_pgen_records", "ldp_load_pgen_records_phased", "ldp_pgen_file_bytes", "ldp_pgen_record_index", "ldp_device_numa_node",
    "ldp_use_private_copy_threads",
]


def _sources():
    return [os.path.join(CSRC, f) for f in ("ldp_kernels.hip", "ldp_codes.hip", "ldp_pair_mfma.hip", "ldp_pair_wide.hip", "ldp_pred_csr.hip", "ldp_pgen_decode.hip", "ldp_synth.hip", "ldp_engine.cpp", "ldp_engine_run.cpp", "ldp_engine_r2.cpp",
                                          "ldp_engine_load.cpp", "ldp_engine_shard.cpp", "ldp_pgen.cpp", "ldp_topology.cpp")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, measure=False):
    """Compile the HIP kernels + host runtime into lib/libldprune_hip.so for gfx950 (hipcc cross-compiles
    without a GPU).  In-tree so the .so travels with the repo snapshot.  One object per source under lib/_obj/, stale ones
    recompiled in parallel, then one link.  measure=True: the measurement build, lib/libldprune_hip_measure.so (-DLDP_MEASURE)."""
    headers = [os.path.join(CSRC, "ldp_device.h"), os.path.join(CSRC, "ldp_pair_device.h"), os.path.join(CSRC, "ldp_mfma_device.h"),
               os.path.join(CSRC, "ldp_env.h"), os.path.join(CSRC, "ldp_engine_internal.h"),
               os.path.join(REPO, "include", "ldprune_hip.h"), os.path.join(REPO, "include", "ldprune_hip_debug.h")]
    headers = [h for h in headers if os.path.exists(h)]
    lib_path = MEASURE_LIB_PATH if measure else LIB_PATH
    flags = HIPCC_FLAGS + (["-DLDP_MEASURE"] if measure else [])
    if not force and not _stale(lib_path, _sources() + headers):
        return lib_path  # (a snapshot on the GPU box carries the library but not the objects: nothing to do there)
    obj_dir = os.path.join(LIB_DIR, "_obj_measure" if measure else "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs, todo = [], []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            todo.append((src, obj))

    def compile_one(job):
        cmd = ["hipcc"] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, todo))
    if todo or force or _stale(lib_path, objs):
        cmd = ["hipcc"] + flags + ["-shared", "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib_path


def cli_sources():
    """the front-end's translation units: main() + one unit per concern (csrc/p2h_cli.h is what they share)"""
    return [os.path.join(CSRC, f) for f in ("plink2_hip_cli.cpp", "p2h_util.cpp", "p2h_args.cpp", "p2h_tables.cpp", "p2h_inputs.cpp", "p2h_clump.cpp", "p2h_r2.cpp",
                                            "p2h_prune.cpp")]


def build_cli(force=False, verbose=False):
    srcs = cli_sources()
    if not os.path.exists(srcs[0]):
        return None
    deps = srcs + [LIB_PATH, os.path.join(CSRC, "p2h_cli.h"), os.path.join(REPO, "include", "ldprune_hip.h"), os.path.join(REPO, "include", "ldprune_hip_debug.h")]
    if force or _stale(CLI_PATH, deps):
        os.makedirs(BIN_DIR, exist_ok=True)
        cmd = ["hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "-o", CLI_PATH] + srcs + ["-L" + LIB_DIR, "-lldprune_hip",
               "-Wl,-rpath,$ORIGIN/../lib", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI_PATH


_lib = None


def lib():
    """Load the C-ABI library.  torch (when importable) is imported first so that the process ends up
    with ONE libamdhip64 (torch bundles its own copy under the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    path = MEASURE_LIB_PATH if os.environ.get("LDP_LIB_MEASURE") == "1" else LIB_PATH
    if not os.path.exists(path):
        raise LdpError(LDP_ERR_GPU, "HIP extension %s is missing; run __graft_entry__.build()" % path)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    L = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    u32p = ctypes.POINTER(ctypes.c_uint32)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    f64p = ctypes.POINTER(ctypes.c_double)
    L.ldp_create.argtypes = [ctypes.POINTER(ldp_params), ctypes.POINTER(vp)]
    L.ldp_destroy.argtypes = [vp]
    L.ldp_destroy.restype = None
    L.ldp_last_error.argtypes = [vp]
    L.ldp_last_error.restype = ctypes.c_char_p
    L.ldp_device_count.argtypes = []
    L.ldp_set_variants.argtypes = [vp, ctypes.c_uint32, u32p, u32p]
    L.ldp_get_subcontigs.argtypes = [vp, u32p, u32p, ctypes.c_uint32]
    L.ldp_set_shard.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, u32p]
    L.ldp_get_band.argtypes = [vp, u32p, u64p]
    L.ldp_load_genotypes.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    L.ldp_set_maj_freqs.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, f64p]
    L.ldp_set_sample_map.argtypes = [vp, ctypes.c_uint32, u32p, ctypes.POINTER(ctypes.c_uint8)]
    L.ldp_set_preferred.argtypes = [vp, u64p]
    L.ldp_run.argtypes = [vp, u64p]
    L.ldp_run_with_stats.argtypes = [vp, u64p, vp, ctypes.c_uint64]
    L.ldp_pair_stats.argtypes = [vp, ctypes.c_uint32, u32p, u32p, vp]
    L.ldp_debug_set_variant_recs.argtypes = [vp, vp]
    L.ldp_debug_replay_pairs.argtypes = [vp, ctypes.c_uint64, u32p, u32p, u64p]
    L.ldp_map_rows.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), u64p]
    L.ldp_release_device.argtypes = [vp]
    L.ldp_load_pgen_records.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ldp_pgen_rec),
                                        ctypes.POINTER(ldp_pgen_rec), ctypes.c_uint32, u32p]
    L.ldp_load_pgen_records_phased.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ldp_pgen_rec),
                                               ctypes.POINTER(ldp_pgen_rec), ctypes.c_uint32, u32p]
    L.ldp_pgen_file_bytes.argtypes = [vp, u64p]
    L.ldp_pgen_file_bytes.restype = ctypes.c_void_p
    L.ldp_pgen_record_index.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ldp_pgen_rec), u32p]
    L.ldp_allgather_removed.argtypes = [vp, vp, u64p, u64p]
    L.ldp_comm_init_all.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]
    L.ldp_load_genotypes_fd.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
    L.ldp_shard_segment_words.argtypes = [vp, u64p]
    L.ldp_pack_removed_segment.argtypes = [vp, u64p, u64p]
    L.ldp_stitch_removed_segments.argtypes = [vp, u64p, u64p]
    L.ldp_comm_destroy.argtypes = [vp]
    L.ldp_comm_destroy.restype = None
    L.ldp_debug_wide_plan.argtypes = [vp, u32p, u32p, ctypes.c_uint64]
    L.ldp_matrix_pipe_max_founders.argtypes = []
    L.ldp_matrix_pipe_max_founders.restype = ctypes.c_uint32
    L.ldp_debug_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
    L.ldp_pgen_debug_force_portable.argtypes = [ctypes.c_int]
    L.ldp_debug_mfma_plan.argtypes = [vp, u32p, u32p, ctypes.c_uint64, u32p, u32p]
    L.ldp_get_variant_recs.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp]
    L.ldp_get_maj_freqs.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, f64p]
    L.ldp_get_planes.argtypes = [vp, ctypes.c_uint32, u32p, u32p]
    L.ldp_get_counters.argtypes = [vp, ctypes.POINTER(ldp_counters)]
    L.ldp_synth_genotypes.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, vp,
                                      ctypes.c_uint64, ctypes.c_int, vp]
    L.ldp_set_variants_matrix.argtypes = [vp, ctypes.c_uint32]
    L.ldp_r2_unphased_rows.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, vp, ctypes.c_uint64]
    L.ldp_r2_unphased_hits.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.ldp_r2_unphased_block.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, vp, ctypes.c_uint64]
    L.ldp_r2_unphased_block_hits.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, vp, ctypes.c_uint64,
                                             ctypes.POINTER(ctypes.c_uint64)]
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.ldp_r2_unphased_block_x.argtypes = [vp, vp, u8p, u8p, u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, vp,
                                          ctypes.c_uint64]
    L.ldp_r2_unphased_block_x_hits.argtypes = [vp, vp, u8p, u8p, u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_double,
                                               vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.ldp_pair_stats_block.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64]
    L.ldp_set_variants_vcor.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.c_uint32, ctypes.c_uint32]
    L.ldp_r2_unphased_band_rows.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, vp, ctypes.c_uint64]
    L.ldp_pgen_open.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp)]
    L.ldp_pgen_info.argtypes = [vp, u32p, u32p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.ldp_pgen_direct_rows.argtypes = [vp, u64p]
    L.ldp_pgen_direct_rows.restype = ctypes.c_void_p
    L.ldp_pgen_read.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_uint32]
    L.ldp_pgen_variant_is_multiallelic.argtypes = [vp, ctypes.c_uint32]
    L.ldp_pgen_variant_has_dosage.argtypes = [vp, ctypes.c_uint32]
    L.ldp_pgen_dosage_sums.argtypes = [vp, ctypes.c_uint32, vp, u64p, u64p]
    L.ldp_phased_row_bytes.argtypes = [ctypes.c_uint32]
    L.ldp_phased_row_bytes.restype = ctypes.c_uint64
    L.ldp_phased_phase_offset.argtypes = [ctypes.c_uint32]
    L.ldp_phased_phase_offset.restype = ctypes.c_uint64
    L.ldp_pgen_read_phased.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint32,
                                       ctypes.POINTER(ctypes.c_uint32)]
    L.ldp_subset_samples.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8), vp, ctypes.c_uint64,
                                     ctypes.c_int, ctypes.c_uint32]
    L.ldp_pgen_read_alleles_phased.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32] + [ctypes.POINTER(ctypes.c_uint8)] * 4
    L.ldp_pgen_provisional_ref.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint64]
    L.ldp_pgen_read_alleles.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8)]
    L.ldp_pgen_last_error.argtypes = [vp]
    L.ldp_pgen_last_error.restype = ctypes.c_char_p
    L.ldp_pgen_close.argtypes = [vp]
    L.ldp_pgen_close.restype = None
    _lib = L
    return L


def hip_memcpy_dtod(dst_ptr, src_ptr, nbytes):
    """hipMemcpy device -> device through the process's HIP runtime (test / benchmark plumbing: fills mapped image rows from a
    torch tensor)."""
    lib()
    rt = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return int(rt.hipMemcpy(ctypes.c_void_p(dst_ptr), ctypes.c_void_p(src_ptr), ctypes.c_size_t(nbytes), 3))  # 3 = hipMemcpyDeviceToDevice


def comm_init_all(devices):
    """ncclCommInitAll over `devices` through the C ABI (ldp_comm_init_all): a list of ncclComm_t handles (integers)."""
    n = len(devices)
    devs = (ctypes.c_int * n)(*devices)
    comms = (ctypes.c_void_p * n)()
    rc = lib().ldp_comm_init_all(n, devs, comms)
    if rc != LDP_OK:
        raise LdpError(rc, "ldp_comm_init_all failed")
    return [int(c) for c in comms]


def comm_destroy(comm):
    lib().ldp_comm_destroy(ctypes.c_void_p(comm))


def matrix_pipe_max_founders():
    """Largest founder_ct the matrix-pipe kernels take (f32 accumulators integer-exact); beyond it the popcount kernels run."""
    return int(lib().ldp_matrix_pipe_max_founders())


def device_count():
    return int(lib().ldp_device_count())


def synth_genotypes_host(seed, first_variant, n_variants, founder_ct, missing_rate=0.0):
    """Host-side run of the synthetic generator (small sizes): (n_variants, ceil(founder_ct/4)) uint8 REF codes."""
    out = np.zeros((n_variants, (founder_ct + 3) // 4), dtype=np.uint8)
    rc = lib().ldp_synth_genotypes(seed, first_variant, n_variants, founder_ct, missing_rate, out.ctypes.data_as(ctypes.c_void_p),
                                   out.strides[0] if n_variants else (founder_ct + 3) // 4, LDP_MEM_HOST, None)
    if rc != LDP_OK:
        raise LdpError(rc, "ldp_synth_genotypes failed")
    return out


def synth_genotypes_device(seed, first_variant, n_variants, founder_ct, missing_rate, device_ptr, stride_bytes, stream=None):
    rc = lib().ldp_synth_genotypes(seed, first_variant, n_variants, founder_ct, missing_rate, ctypes.c_void_p(device_ptr), stride_bytes,
                                   LDP_MEM_DEVICE, ctypes.c_void_p(stream) if stream else None)
    if rc != LDP_OK:
        raise LdpError(rc, "ldp_synth_genotypes failed")


def phased_row_bytes(hap_ct):
    """ldp_phased_row_bytes (include/ldprune_hip.h)"""
    s = hap_ct // 2
    return (((s + 3) // 4 + 3) & ~3) + (s + 7) // 8


def pack_phased_rows(codes_packed, phaseinfo_bits, sample_ct):
    """(M, >= ceil(S/4)) uint8 packed 2-bit codes + (M, S) 0/1 phaseinfo -> rows in the LDP_GENO_PHASED layout"""
    m = codes_packed.shape[0]
    cb = (sample_ct + 3) // 4
    off = (cb + 3) & ~3
    out = np.zeros((m, phased_row_bytes(2 * sample_ct)), dtype=np.uint8)
    out[:, :cb] = np.ascontiguousarray(codes_packed).view(np.uint8).reshape(m, -1)[:, :cb]
    bits = np.packbits(np.asarray(phaseinfo_bits, dtype=np.uint8) & 1, axis=1, bitorder="little")
    out[:, off:off + bits.shape[1]] = bits
    return out


def subset_samples(rows, raw_sample_ct, keep_mask, phased=False, threads=0):
    """ldp_subset_samples: rows (M, stride) uint8 -> rows of the kept samples (packed 2-bit codes [+ phase bits])"""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    keep = np.asarray(keep_mask, dtype=bool)
    kept = int(keep.sum())
    out_stride = phased_row_bytes(2 * kept) if phased else (kept + 3) // 4
    out = np.zeros((rows.shape[0], max(out_stride, 1)), dtype=np.uint8)
    mask = np.packbits(keep, bitorder="little")
    rc = lib().ldp_subset_samples(rows.ctypes.data_as(ctypes.c_void_p), rows.strides[0] if rows.shape[0] else 1, rows.shape[0], raw_sample_ct,
                                  mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), out.ctypes.data_as(ctypes.c_void_p), out.strides[0], 1 if phased else 0, threads)
    if rc != LDP_OK:
        raise LdpError(rc, "ldp_subset_samples failed")
    return out


class PgenFile:
    """ctypes mirror of the ldp_pgen_* reader (main track of .bed / .pgen files)."""

    def __init__(self, path, sample_ct_hint=0, variant_ct_hint=0):
        self._L = lib()
        self._h = ctypes.c_void_p()
        rc = self._L.ldp_pgen_open(path.encode(), sample_ct_hint, variant_ct_hint, ctypes.byref(self._h))
        if rc != LDP_OK:
            msg = self._L.ldp_pgen_last_error(self._h).decode() if self._h else "open failed"
            self.close()
            raise LdpError(rc, msg)
        m, n = ctypes.c_uint32(), ctypes.c_uint32()
        mode, enc, multi = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._L.ldp_pgen_info(self._h, ctypes.byref(m), ctypes.byref(n), ctypes.byref(mode), ctypes.byref(enc), ctypes.byref(multi))
        self.variant_ct, self.sample_ct, self.mode, self.encoding, self.has_multiallelic = m.value, n.value, mode.value, enc.value, bool(multi.value)

    def read(self, first=0, n=None, threads=0):
        n = self.variant_ct - first if n is None else n
        out = np.zeros((n, (self.sample_ct + 3) // 4), dtype=np.uint8)
        rc = self._L.ldp_pgen_read(self._h, first, n, out.ctypes.data_as(ctypes.c_void_p), out.strides[0] if n else 1, threads)
        if rc != LDP_OK:
            raise LdpError(rc, self._L.ldp_pgen_last_error(self._h).decode())
        return out

    def file_bytes(self):
        """(host pointer, byte count) of the reader's mapping of the file (ldp_pgen_file_bytes)."""
        nb = ctypes.c_uint64()
        p = self._L.ldp_pgen_file_bytes(self._h, ctypes.byref(nb))
        return int(p), int(nb.value)

    def record_index(self, first=0, n=None, allele_cts=None):
        """(ctypes array of ldp_pgen_rec for variants [first, first + n), the variant the first record's LD chain builds on or None);
        allele_cts (optional, per variant) overrides the default of 2."""
        n = self.variant_ct - first if n is None else n
        recs = (ldp_pgen_rec * max(n, 1))()
        base = ctypes.c_uint32()
        rc = self._L.ldp_pgen_record_index(self._h, first, n, recs, ctypes.byref(base))
        if rc != LDP_OK:
            raise LdpError(rc, "ldp_pgen_record_index failed")
        if allele_cts is not None:
            for q in range(n):
                recs[q].allele_ct = int(allele_cts[q])
        return recs, (None if base.value == 0xffffffff else int(base.value))

    def read_phased(self, first=0, n=None, sample_mask=None, threads=0):
        """Rows in the LDP_GENO_REF | LDP_GENO_PHASED layout (2-bit codes, padding to a dword, phaseinfo bits).
        Raises LdpError(LDP_ERR_UNPHASED) -- .unphased_variant set -- when a het call of a masked sample has no phase."""
        n = self.variant_ct - first if n is None else n
        out = np.zeros((n, phased_row_bytes(2 * self.sample_ct)), dtype=np.uint8)
        mask = None
        if sample_mask is not None:
            mask = np.packbits(np.asarray(sample_mask, dtype=bool), bitorder="little")
        bad = ctypes.c_uint32(0xffffffff)
        rc = self._L.ldp_pgen_read_phased(self._h, first, n, out.ctypes.data_as(ctypes.c_void_p), out.strides[0] if n else 1,
                                          mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) if mask is not None else None, threads, ctypes.byref(bad))
        if rc != LDP_OK:
            err = LdpError(rc, self._L.ldp_pgen_last_error(self._h).decode())
            err.unphased_variant = bad.value
            raise err
        return out

    def read_alleles(self, variant, alt_ct):
        """(allele_lo, allele_hi) uint8 arrays over samples: 0 REF, k ALTk, 255 missing."""
        lo = np.zeros(self.sample_ct, dtype=np.uint8)
        hi = np.zeros(self.sample_ct, dtype=np.uint8)
        rc = self._L.ldp_pgen_read_alleles(self._h, variant, alt_ct, lo.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                           hi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        if rc != LDP_OK:
            raise LdpError(rc, self._L.ldp_pgen_last_error(self._h).decode())
        return lo, hi

    def read_alleles_phased(self, variant, alt_ct):
        """(allele_lo, allele_hi, phasepresent, phaseinfo) over samples; phaseinfo 1 = the higher allele on the first haplotype"""
        lo = np.zeros(self.sample_ct, dtype=np.uint8)
        hi = np.zeros(self.sample_ct, dtype=np.uint8)
        nb = (self.sample_ct + 7) // 8
        pp = np.zeros(nb, dtype=np.uint8)
        pi = np.zeros(nb, dtype=np.uint8)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        rc = self._L.ldp_pgen_read_alleles_phased(self._h, variant, alt_ct, lo.ctypes.data_as(u8p), hi.ctypes.data_as(u8p),
                                                  pp.ctypes.data_as(u8p), pi.ctypes.data_as(u8p))
        if rc != LDP_OK:
            raise LdpError(rc, self._L.ldp_pgen_last_error(self._h).decode())
        return (lo, hi, np.unpackbits(pp, bitorder="little")[:self.sample_ct], np.unpackbits(pi, bitorder="little")[:self.sample_ct])

    def is_multiallelic(self, variant):
        return bool(self._L.ldp_pgen_variant_is_multiallelic(self._h, variant))

    def has_dosage(self, variant=None):
        """some record of the file (variant None) / this variant's record carries a dosage track"""
        if variant is None:
            return bool(self._L.ldp_pgen_has_dosage(self._h))
        return bool(self._L.ldp_pgen_variant_has_dosage(self._h, variant))

    def dosage_sums(self, variant, sample_mask=None):
        """(ref, alt) allele dosage sums of a biallelic variant in the reference's units (16384 per ALT allele copy) over the
        samples whose bit is set in sample_mask (bool array over the file's samples; None = all)"""
        ref, alt = ctypes.c_uint64(0), ctypes.c_uint64(0)
        mask = None
        if sample_mask is not None:
            mask = np.packbits(np.asarray(sample_mask, dtype=bool), bitorder="little")
        rc = self._L.ldp_pgen_dosage_sums(self._h, variant, mask.ctypes.data_as(ctypes.c_void_p) if mask is not None else None,
                                          ctypes.byref(ref), ctypes.byref(alt))
        if rc != LDP_OK:
            raise LdpError(rc, self._L.ldp_pgen_last_error(self._h).decode())
        return int(ref.value), int(alt.value)

    def close(self):
        if self._h:
            self._L.ldp_pgen_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def kb_window(kb):
    """plink2.cc:7266: window in bp = int32(kb * 1000 * (1 + 2^-44))."""
    return int(float(kb) * 1000 * (1 + 2.0 ** -44))


class LdPruneEngine:
    """Mirror of the C ABI, one engine per GPU (one per process under torch.distributed)."""

    def __init__(self, founder_ct, window, step=1, window_is_bp=False, r2=0.2, order=2, device=-1, stream=None):
        self._L = lib()
        p = ldp_params()
        p.founder_ct = int(founder_ct)
        p.prune_window_size = int(window)
        p.prune_window_incr = int(step)
        p.window_is_bp = 1 if window_is_bp else 0
        p.plink1_order = 1 if order == 1 else 0
        p.prune_last_param = float(r2)
        p.device = int(device)
        p.stream = ctypes.c_void_p(stream) if stream else None
        self._h = ctypes.c_void_p()
        rc = self._L.ldp_create(ctypes.byref(p), ctypes.byref(self._h))
        if rc != LDP_OK:
            raise LdpError(rc, "ldp_create failed")
        self.founder_ct = int(founder_ct)
        self.variant_ct = 0

    def close(self):
        if self._h:
            self._L.ldp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LDP_OK:
            raise LdpError(rc, self._L.ldp_last_error(self._h).decode())

    def set_option(self, name, value):
        """Kernel-selection switch of this engine (ldp_debug_set_option): 'early_exit', 'pair_mfma' (before set_variants),
        'pair_sparse', 'sparse_frac', ... and the test hooks 'replay_steps', 'decode_rows', 'decode_no_lds', 'x_rows' (include/ldprune_hip_debug.h).
        Results never depend on them."""
        self._ck(self._L.ldp_debug_set_option(self._h, name.encode(), float(value)))
        return self

    # ---- planning
    def set_variants(self, chr_idx, bps=None):
        chr_idx = _u32(chr_idx)
        self.variant_ct = len(chr_idx)
        bp_ptr = None
        if bps is not None:
            bps = _u32(bps)
            assert len(bps) == len(chr_idx)
            bp_ptr = _ptr(bps, ctypes.c_uint32)
        self._ck(self._L.ldp_set_variants(self._h, self.variant_ct, _ptr(chr_idx, ctypes.c_uint32), bp_ptr))

    def set_variants_matrix(self, variant_ct):
        self.variant_ct = int(variant_ct)
        self._ck(self._L.ldp_set_variants_matrix(self._h, self.variant_ct))

    def set_r_signed(self, mode):
        """0: r^2 (default); 1: the r2_unphased_* calls return r, major-allele orientation; 2: r, REF orientation (--r-unphased)."""
        self._ck(self._L.ldp_set_r_signed(self._h, int(mode)))

    def r2_unphased_rows(self, row_first=0, row_ct=None, as_float=False):
        """Lower-triangle rows (incl. diagonal) of the --r2-unphased matrix: (row_ct, row_first+row_ct) array."""
        row_ct = self.variant_ct - row_first if row_ct is None else row_ct
        ld = row_first + row_ct
        out = np.zeros((row_ct, ld), dtype=np.float32 if as_float else np.float64)
        self._ck(self._L.ldp_r2_unphased_rows(self._h, row_first, row_ct, 1 if as_float else 0, out.ctypes.data_as(ctypes.c_void_p), ld))
        return out

    def r2_unphased_block(self, row_first, row_ct, col_first, col_ct, as_float=False):
        """Column block of the same rows: (row_ct, col_ct) array, element [j - row_first, i - col_first] for i < j (and the
        diagonal where it falls inside the block); everything else 0."""
        out = np.zeros((row_ct, col_ct), dtype=np.float32 if as_float else np.float64)
        self._ck(self._L.ldp_r2_unphased_block(self._h, row_first, row_ct, col_first, col_ct, 1 if as_float else 0, out.ctypes.data_as(ctypes.c_void_p), col_ct))
        return out

    def r2_unphased_block_hits(self, min_r2, row_first, row_ct, col_first, col_ct, capacity=1 << 20):
        out = np.zeros(max(capacity, 1), dtype=R2_HIT_DTYPE)
        found = ctypes.c_uint64()
        self._ck(self._L.ldp_r2_unphased_block_hits(self._h, row_first, row_ct, col_first, col_ct, float(min_r2), out.ctypes.data_as(ctypes.c_void_p), capacity,
                                                    ctypes.byref(found)))
        got = out[:min(found.value, capacity)]
        return np.sort(got, order=["first", "second"]), found.value

    def pair_stats_block(self, row_first, row_ct, col_first, col_ct):
        """ldp_pair_stats_block: the six integers of the pairs i < j of a dense block from the pair kernels, (row_ct, col_ct) structured array."""
        out = np.zeros((row_ct, col_ct), dtype=PAIR_STATS_DTYPE)
        self._ck(self._L.ldp_pair_stats_block(self._h, row_first, row_ct, col_first, col_ct, out.ctypes.data_as(ctypes.c_void_p), col_ct))
        return out

    @staticmethod
    def _flags(a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))

    def r2_unphased_block_x(self, block, male, is_x, row_first, col_first, flip_all=None, flip_male=None, unsquared=False):
        """ldp_r2_unphased_block_x: overwrites, in `block` (what r2_unphased_block returned for the same rows / columns), the elements of
        the pairs with a chrX variant by their male-weighted r^2 (or r); `male` = the male founders' engine or None."""
        k1, p1 = self._flags(is_x)
        k2, p2 = self._flags(flip_all)
        k3, p3 = self._flags(flip_male)
        self._ck(self._L.ldp_r2_unphased_block_x(self._h, male._h if male is not None else None, p1, p2, p3, row_first, block.shape[0], col_first, block.shape[1],
                                                 1 if block.dtype == np.float32 else 0, 1 if unsquared else 0, block.ctypes.data_as(ctypes.c_void_p), block.shape[1]))
        return block

    def r2_unphased_block_x_hits(self, male, is_x, min_r2, row_first, row_ct, col_first, col_ct, flip_all=None, flip_male=None, unsquared=False, capacity=1 << 20):
        k1, p1 = self._flags(is_x)
        k2, p2 = self._flags(flip_all)
        k3, p3 = self._flags(flip_male)
        out = np.zeros(max(capacity, 1), dtype=R2_HIT_DTYPE)
        found = ctypes.c_uint64()
        self._ck(self._L.ldp_r2_unphased_block_x_hits(self._h, male._h if male is not None else None, p1, p2, p3, row_first, row_ct, col_first, col_ct,
                                                      1 if unsquared else 0, float(min_r2), out.ctypes.data_as(ctypes.c_void_p), capacity, ctypes.byref(found)))
        got = out[:min(found.value, capacity)]
        return np.sort(got, order=["first", "second"]), found.value

    def r2_unphased_hits(self, min_r2, row_first=0, row_ct=None, capacity=1 << 20):
        """Pairs first < second (second among the rows) with |r^2| >= min_r2, filtered on the device; sorted here by
        (first, second).  Returns (structured array, total found) -- found > len(array) means the buffer was too small."""
        row_ct = self.variant_ct - row_first if row_ct is None else row_ct
        out = np.zeros(max(capacity, 1), dtype=R2_HIT_DTYPE)
        found = ctypes.c_uint64()
        self._ck(self._L.ldp_r2_unphased_hits(self._h, row_first, row_ct, float(min_r2), out.ctypes.data_as(ctypes.c_void_p), capacity, ctypes.byref(found)))
        out = out[:min(found.value, capacity)]
        return out[np.lexsort((out["second"], out["first"]))], found.value

    def set_variants_vcor(self, chr_idx, bps, bp_radius, var_ct_radius=0x7fffffff):
        """Windowed plan of the --r2-unphased table (--ld-window-kb / --ld-window)."""
        chr_idx = _u32(chr_idx)
        bps = _u32(bps)
        self.variant_ct = len(chr_idx)
        self._ck(self._L.ldp_set_variants_vcor(self._h, self.variant_ct, _ptr(chr_idx, ctypes.c_uint32), _ptr(bps, ctypes.c_uint32),
                                               int(bp_radius), int(var_ct_radius)))

    def r2_unphased_band_rows(self, row_first=0, row_ct=None, as_float=False):
        """r^2 of the candidate pairs of second variants [row_first, row_first+row_ct), band order (see band())."""
        row_ct = self.variant_ct - row_first if row_ct is None else row_ct
        lo, _ = self.band()
        j = np.arange(row_first, row_first + row_ct, dtype=np.int64)
        n = int((j - lo[row_first:row_first + row_ct]).sum())
        out = np.zeros(max(n, 1), dtype=np.float32 if as_float else np.float64)
        self._ck(self._L.ldp_r2_unphased_band_rows(self._h, row_first, row_ct, 1 if as_float else 0, out.ctypes.data_as(ctypes.c_void_p), n))
        return out[:n]

    def subcontigs(self):
        ct = ctypes.c_uint32()
        self._ck(self._L.ldp_get_subcontigs(self._h, ctypes.byref(ct), None, 0))
        info = np.zeros(2 * max(ct.value, 1), dtype=np.uint32)
        self._ck(self._L.ldp_get_subcontigs(self._h, ctypes.byref(ct), _ptr(info, ctypes.c_uint32), ct.value))
        return [(int(info[2 * k]), int(info[2 * k + 1])) for k in range(ct.value)]

    def set_shard(self, rank, world):
        n = len(self.subcontigs())
        owner = np.zeros(max(n, 1), dtype=np.uint32)
        self._ck(self._L.ldp_set_shard(self._h, rank, world, _ptr(owner, ctypes.c_uint32)))
        return owner[:n].copy()

    def band(self):
        lo = np.zeros(max(self.variant_ct, 1), dtype=np.uint32)
        tot = ctypes.c_uint64()
        self._ck(self._L.ldp_get_band(self._h, _ptr(lo, ctypes.c_uint32), ctypes.byref(tot)))
        return lo[:self.variant_ct], tot.value

    # ---- data
    def load_genotypes_host(self, first_variant, rows, encoding=LDP_GENO_INVERSE):
        """rows: C-contiguous 2-D uint8/uint64 array, one packed 2-bit row per variant."""
        rows = np.ascontiguousarray(rows)
        stride = rows.strides[0]
        self._ck(self._L.ldp_load_genotypes(self._h, first_variant, rows.shape[0], rows.ctypes.data_as(ctypes.c_void_p),
                                            stride, LDP_MEM_HOST, encoding))

    def load_genotypes_fd(self, first_variant, n, fd, file_offset, stride_bytes, encoding=LDP_GENO_REF):
        """Fixed-width rows straight from an open file descriptor (ldp_load_genotypes_fd): pread into the pinned ring."""
        self._ck(self._L.ldp_load_genotypes_fd(self._h, int(first_variant), int(n), int(fd), int(file_offset), int(stride_bytes), int(encoding)))

    def load_genotypes_device(self, first_variant, n, device_ptr, stride_bytes, encoding=LDP_GENO_INVERSE):
        self._ck(self._L.ldp_load_genotypes(self._h, first_variant, n, ctypes.c_void_p(device_ptr), stride_bytes,
                                            LDP_MEM_DEVICE, encoding))

    def load_pgen_records(self, first_variant, pgen, raw_first=None, n=None, allele_cts=None, location=LDP_MEM_HOST, device_bytes=None):
        """ldp_load_pgen_records: variants [first_variant, +n) of the engine <- records [raw_first, +n) of the PgenFile `pgen`, decoded
        on the device.  device_bytes: device pointer of a copy of the whole file (location = LDP_MEM_DEVICE).  Returns the major
        alleles (uint32 array; 0xffffffff for variants with one ALT allele)."""
        raw_first = first_variant if raw_first is None else raw_first
        n = pgen.variant_ct - raw_first if n is None else n
        recs, base = pgen.record_index(raw_first, n, allele_cts)
        base_rec = None
        if base is not None:
            base_rec = pgen.record_index(base, 1)[0]
        ptr, nbytes = pgen.file_bytes()
        if location == LDP_MEM_DEVICE:
            ptr = int(device_bytes)
        maj = np.zeros(max(n, 1), dtype=np.uint32)
        self._ck(self._L.ldp_load_pgen_records(self._h, int(first_variant), int(n), ctypes.c_void_p(ptr), nbytes, location, recs,
                                               base_rec if base_rec is not None else None, pgen.sample_ct, _ptr(maj, ctypes.c_uint32)))
        return maj[:n]

    def load_pgen_records_phased(self, first_variant, pgen, raw_first=None, n=None, location=LDP_MEM_HOST, device_bytes=None, allele_cts=None):
        """ldp_load_pgen_records_phased (--indep-pairphase): main + hardcall-phase tracks of records [raw_first, +n) decoded on the device
        into the engine's haplotype rows (founder_ct = 2 x the file's samples).  Raises LdpError(LDP_ERR_UNPHASED) with `.variant` = the
        lowest variant that has a het call without phase.  allele_cts (optional, per record; the .pvar's ALT count + 1): records with more
        than two alleles are collapsed on their major allele, their phase bits following the collapse."""
        raw_first = first_variant if raw_first is None else raw_first
        n = pgen.variant_ct - raw_first if n is None else n
        recs, base = pgen.record_index(raw_first, n, allele_cts)
        base_rec = pgen.record_index(base, 1)[0] if base is not None else None
        ptr, nbytes = pgen.file_bytes()
        if location == LDP_MEM_DEVICE:
            ptr = int(device_bytes)   # (the whole file's bytes, resident in HBM)
        bad = ctypes.c_uint32(0xffffffff)
        rc = self._L.ldp_load_pgen_records_phased(self._h, int(first_variant), int(n), ctypes.c_void_p(ptr), nbytes, location, recs,
                                                  base_rec if base_rec is not None else None, pgen.sample_ct, ctypes.byref(bad))
        if rc != LDP_OK:
            err = LdpError(rc, self._L.ldp_last_error(self._h).decode())
            err.variant = int(bad.value)
            raise err

    def map_rows(self, first_variant, n):
        """(device pointer, stride in bytes) of the engine's own image rows of variants [first_variant, first_variant + n): write
        REF- or INVERSE-coded 2-bit rows there and pass the same pointer / stride to load_genotypes_device() -- they are counted in
        place, nothing is copied (ldp_map_rows)."""
        ptr = ctypes.c_void_p()
        stride = ctypes.c_uint64()
        self._ck(self._L.ldp_map_rows(self._h, int(first_variant), int(n), ctypes.byref(ptr), ctypes.byref(stride)))
        return int(ptr.value), int(stride.value)

    def allgather_removed(self, comm, removed_bitmap_words):
        """ldp_allgather_removed: this rank's bitmap (uint64 words over all variants) -> the global one, through ONE ncclAllGather
        of shard-order segments on the communicator `comm` (an ncclComm_t as an integer / c_void_p)."""
        words = np.ascontiguousarray(removed_bitmap_words, dtype=np.uint64)
        out = np.zeros((self.variant_ct + 63) // 64 + 1, dtype=np.uint64)
        self._ck(self._L.ldp_allgather_removed(self._h, ctypes.c_void_p(comm), _ptr(words, ctypes.c_uint64), _ptr(out, ctypes.c_uint64)))
        return out

    def segment_words(self):
        """64-bit words of one padded shard segment (ldp_shard_segment_words)."""
        w = ctypes.c_uint64()
        self._ck(self._L.ldp_shard_segment_words(self._h, ctypes.byref(w)))
        return int(w.value)

    def pack_removed_segment(self, removed_bitmap_words):
        """This rank's removed bits in shard order (ldp_pack_removed_segment): segment_words() uint64 words."""
        words = np.ascontiguousarray(removed_bitmap_words, dtype=np.uint64)
        seg = np.zeros(self.segment_words(), dtype=np.uint64)
        self._ck(self._L.ldp_pack_removed_segment(self._h, _ptr(words, ctypes.c_uint64), _ptr(seg, ctypes.c_uint64)))
        return seg

    def stitch_removed_segments(self, segments):
        """world x segment_words() words, rank-major -> the global removed bitmap (ldp_stitch_removed_segments)."""
        segs = np.ascontiguousarray(segments, dtype=np.uint64).reshape(-1)
        out = np.zeros((self.variant_ct + 63) // 64 + 1, dtype=np.uint64)
        self._ck(self._L.ldp_stitch_removed_segments(self._h, _ptr(segs, ctypes.c_uint64), _ptr(out, ctypes.c_uint64)))
        return out

    def release_device(self):
        """Free the engine's device memory, keep its plan (ldp_release_device)."""
        self._ck(self._L.ldp_release_device(self._h))

    def set_sample_map(self, raw_sample_ct, src_sample, het_to_missing=None):
        """Column f of the engine's rows = sample src_sample[f] of rows loaded with LDP_GENO_MAPPED; het_to_missing[f] != 0
        turns a het call there into a missing one (chrX males, chrY, MT)."""
        src = np.ascontiguousarray(src_sample, dtype=np.uint32)
        het = None if het_to_missing is None else np.ascontiguousarray(het_to_missing, dtype=np.uint8)
        self._ck(self._L.ldp_set_sample_map(self._h, raw_sample_ct, _ptr(src, ctypes.c_uint32), None if het is None else _ptr(het, ctypes.c_uint8)))

    def set_maj_freqs(self, first_variant, freqs):
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        self._ck(self._L.ldp_set_maj_freqs(self._h, first_variant, len(freqs), _ptr(freqs, ctypes.c_double)))

    def set_preferred(self, mask_bool):
        if mask_bool is None:
            self._ck(self._L.ldp_set_preferred(self._h, None))
            return
        bits = np.packbits(np.asarray(mask_bool, dtype=bool), bitorder="little")
        buf = np.zeros(((self.variant_ct + 63) // 64) * 8, dtype=np.uint8)
        buf[:len(bits)] = bits
        self._ck(self._L.ldp_set_preferred(self._h, _ptr(buf.view(np.uint64), ctypes.c_uint64)))

    # ---- compute
    def _removed_buf(self):
        return np.zeros((self.variant_ct + 63) // 64 + 1, dtype=np.uint64)

    def _to_bool(self, bm):
        return np.unpackbits(bm.view(np.uint8), bitorder="little")[:self.variant_ct].astype(bool)

    def run(self):
        bm = self._removed_buf()
        self._ck(self._L.ldp_run(self._h, _ptr(bm, ctypes.c_uint64)))
        return self._to_bool(bm)

    def run_bitmap(self):
        bm = self._removed_buf()
        self._ck(self._L.ldp_run(self._h, _ptr(bm, ctypes.c_uint64)))
        return bm

    def run_with_stats(self):
        _, cand = self.band()
        stats = np.zeros(max(cand, 1), dtype=PAIR_STATS_DTYPE)
        bm = self._removed_buf()
        self._ck(self._L.ldp_run_with_stats(self._h, _ptr(bm, ctypes.c_uint64), stats.ctypes.data_as(ctypes.c_void_p), len(stats)))
        return self._to_bool(bm), stats[:cand]

    def pair_stats(self, first, second):
        first, second = _u32(first), _u32(second)
        out = np.zeros(max(len(first), 1), dtype=PAIR_STATS_DTYPE)
        self._ck(self._L.ldp_pair_stats(self._h, len(first), _ptr(first, ctypes.c_uint32), _ptr(second, ctypes.c_uint32),
                                        out.ctypes.data_as(ctypes.c_void_p)))
        return out[:len(first)]

    def debug_set_variant_recs(self, recs):
        recs = np.ascontiguousarray(recs, dtype=VARIANT_REC_DTYPE)
        assert len(recs) == self.variant_ct
        self._ck(self._L.ldp_debug_set_variant_recs(self._h, recs.ctypes.data_as(ctypes.c_void_p)))

    def debug_replay_pairs(self, first, second):
        first, second = _u32(first), _u32(second)
        bm = self._removed_buf()
        self._ck(self._L.ldp_debug_replay_pairs(self._h, len(first), _ptr(first, ctypes.c_uint32), _ptr(second, ctypes.c_uint32),
                                                _ptr(bm, ctypes.c_uint64)))
        return self._to_bool(bm)

    def debug_mfma_plan(self):
        """(workgroups as an (n, 63) uint32 array, lo in shard-local indices) -- see ldp_debug_mfma_plan."""
        n = ctypes.c_uint32(0)
        lc = ctypes.c_uint32(0)
        self._ck(self._L.ldp_debug_mfma_plan(self._h, ctypes.byref(n), None, 0, None, ctypes.byref(lc)))
        words = np.zeros((max(n.value, 1), 63), dtype=np.uint32)
        lo = np.zeros(max(lc.value, 1), dtype=np.uint32)
        self._ck(self._L.ldp_debug_mfma_plan(self._h, ctypes.byref(n), _ptr(words, ctypes.c_uint32), words.size, _ptr(lo, ctypes.c_uint32),
                                             ctypes.byref(lc)))
        return words[:n.value], lo[:lc.value]

    def debug_wide_plan(self):
        """tiles of the wide-band plan as an (n, 5) uint32 array: jv, vv, jend, mask low / high word (ldp_debug_wide_plan)."""
        n = ctypes.c_uint32(0)
        self._ck(self._L.ldp_debug_wide_plan(self._h, ctypes.byref(n), None, 0))
        words = np.zeros((max(n.value, 1), 5), dtype=np.uint32)
        self._ck(self._L.ldp_debug_wide_plan(self._h, ctypes.byref(n), _ptr(words, ctypes.c_uint32), words.size))
        return words[:n.value]

    # ---- inspection
    def variant_recs(self, first=0, n=None):
        n = self.variant_ct - first if n is None else n
        out = np.zeros(max(n, 1), dtype=VARIANT_REC_DTYPE)
        self._ck(self._L.ldp_get_variant_recs(self._h, first, n, out.ctypes.data_as(ctypes.c_void_p)))
        return out[:n]

    def maj_freqs(self, first=0, n=None):
        n = self.variant_ct - first if n is None else n
        out = np.zeros(max(n, 1), dtype=np.float64)
        self._ck(self._L.ldp_get_maj_freqs(self._h, first, n, _ptr(out, ctypes.c_double)))
        return out[:n]

    def planes(self, variant):
        w = (self.founder_ct + 31) // 32
        hom = np.zeros(w, dtype=np.uint32)
        r2h = np.zeros(w, dtype=np.uint32)
        self._ck(self._L.ldp_get_planes(self._h, variant, _ptr(hom, ctypes.c_uint32), _ptr(r2h, ctypes.c_uint32)))
        return hom, r2h

    def counters(self):
        c = ldp_counters()
        self._ck(self._L.ldp_get_counters(self._h, ctypes.byref(c)))
        return c.asdict()
