"""The C-ABI shared library must load (no GPU needed) and export every entry point that
include/ldprune_hip.h declares -- no compute calls here."""
import ctypes
import os
import re


def declared_symbols():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for hdr in ("ldprune_hip.h", "ldprune_hip_debug.h"):   # the boundary, and the test / measurement hooks kept out of it
        text = open(os.path.join(repo, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ldp_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_the_boundary_header_carries_no_test_hooks():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(repo, "include", "ldprune_hip.h")).read(), flags=re.S)
    assert "ldp_debug_" not in text and "ldp_synth_" not in text


def test_header_and_binding_agree(pkg):
    assert declared_symbols() == sorted(pkg.CABI_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    L = ctypes.CDLL(pkg.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), "missing export: " + name


def test_the_shipped_library_reads_no_environment(pkg):
    """No LDP_* variable can change which kernel runs or what it computes: the names do not even reach the binary (csrc/ldp_env.h; the
    measurement build -DLDP_MEASURE, lib/libldprune_hip_measure.so, is the one that knows them), and neither does an ablation kernel."""
    blob = open(pkg.LIB_PATH, "rb").read()
    for marker in (b"LDP_DEBUG", b"LDP_PAIR_", b"LDP_EARLY", b"LDP_EAGER", b"LDP_STREAM", b"LDP_PGEN_", b"ldp_measure_"):
        assert marker not in blob, marker
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(repo, "plink-ng_amd", "csrc")
    for f in os.listdir(csrc):
        front_end = f.startswith("p2h_") or (f == "plink2_hip_cli.cpp")
        if f.endswith((".cpp", ".hip", ".h")) and (f != "ldp_env.h") and not front_end:
            assert "getenv(" not in open(os.path.join(csrc, f)).read(), f + ": the library's environment goes through LDP_ENV (ldp_env.h)"
        if front_end:   # ... and the front-end takes its test hooks as --debug-* flags
            assert "getenv(\"LDP_" not in open(os.path.join(csrc, f)).read(), f


def test_struct_layouts_match_the_header(pkg):
    # sizes the C side was compiled with (plain C structs, natural alignment)
    assert ctypes.sizeof(pkg.ldp_pair_stats_t) == 24 == pkg.PAIR_STATS_DTYPE.itemsize
    assert ctypes.sizeof(pkg.ldp_variant_rec) == 32 == pkg.VARIANT_REC_DTYPE.itemsize
    assert ctypes.sizeof(pkg.ldp_params) == 48
    assert ctypes.sizeof(pkg.ldp_counters) == 200


def test_device_count_never_fails(pkg):
    assert pkg.device_count() >= 0


def test_device_numa_node_is_minus_one_where_there_is_no_such_device(pkg):
    """ldp_device_numa_node (round 5: plink2-hip binds its loading threads to the device's NUMA node): -1 for a device that does not exist, an
    integer >= -1 otherwise -- never an error, never a side effect on the caller's threads."""
    import ctypes
    import os
    L = pkg.lib()
    L.ldp_device_numa_node.argtypes = [ctypes.c_int]
    L.ldp_device_numa_node.restype = ctypes.c_int
    before = os.sched_getaffinity(0)
    n = pkg.device_count()
    assert L.ldp_device_numa_node(-1) == -1 and L.ldp_device_numa_node(n) == -1 and L.ldp_device_numa_node(1 << 20) == -1
    for d in range(n):
        assert L.ldp_device_numa_node(d) >= -1
    assert os.sched_getaffinity(0) == before


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under plink-ng_amd/ or include/ may mention it."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for root in (os.path.join(repo, "plink-ng_amd"), os.path.join(repo, "include")):
        for dirpath, _, files in os.walk(root):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "ldoracle" not in src and "oracle/" not in src and "ldtools" not in src, os.path.join(dirpath, f)


def test_private_copy_threads_come_and_go_with_their_engine(pkg):
    """ldp_use_private_copy_threads (round 6: plink2-hip --gpus N feeds every engine from a thread of its own): no device needed -- the pool is
    host threads --, idempotent, NULL refused, and ldp_destroy joins the threads it started (the process-wide pool is never joined)."""
    import threading
    L = pkg.lib()
    L.ldp_use_private_copy_threads.argtypes = [ctypes.c_void_p]
    L.ldp_use_private_copy_threads.restype = ctypes.c_int
    assert L.ldp_use_private_copy_threads(None) != 0
    before = threading.active_count()
    def native_threads():
        return len(os.listdir("/proc/self/task"))
    n0 = native_threads()
    eng = pkg.LdPruneEngine(64, 10, 1, False, 0.5, device=0)
    assert L.ldp_use_private_copy_threads(eng._h) == 0
    n1 = native_threads()
    assert L.ldp_use_private_copy_threads(eng._h) == 0 and native_threads() == n1   # idempotent
    assert n1 > n0 or (os.cpu_count() or 1) == 1
    eng.close()
    assert native_threads() == n0 and threading.active_count() == before
