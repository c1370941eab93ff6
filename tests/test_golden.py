"""Golden vectors recorded from the reference binary (tests/golden/make_golden.py).

  * not gpu: the CPU oracle must reproduce every golden prune set and the reference's exact r^2 doubles
    (this is what pins the oracle);
  * gpu: the HIP path, driven through the C ABI, must reproduce them too.
"""
import glob
import os

import numpy as np
import pytest

import ldtools as T

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    m, n = int(z["m"]), int(z["n"])
    raw = T.unpack_2bit(z["raw_packed"].reshape(m, -1).view(np.uint64), n)
    wa = [str(a) for a in z["window_args"]]
    if wa[0].lower().endswith("kb"):
        window, step, is_bp = int(float(wa[0][:-2]) * 1000 * (1 + 2.0 ** -44)), 1, True
    else:
        window, step, is_bp = int(wa[0]), (int(wa[1]) if len(wa) > 1 else 1), False
    return dict(raw=raw, m=m, n=n, chr_idx=z["chroms"].astype(np.uint32), bps=z["bps"].astype(np.uint32), window=window, step=step,
                is_bp=is_bp, r2=float(z["r2"]), order=int(z["order"]), removed=z["removed"].astype(bool),
                r2_square=z["r2_square"] if "r2_square" in z.files else None, summary=str(z["ref_summary"]))


def test_golden_files_present():
    assert len(GOLDEN) >= 13


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_reference(path):
    g = load(path)
    inv, mf, _ = T.oracle_prepare(g["raw"])
    got, _ = T.oracle_indep_pairwise(inv, g["n"], g["chr_idx"], g["bps"], mf, g["window"], g["step"], g["is_bp"], g["r2"], g["order"])
    assert np.array_equal(got, g["removed"])
    assert ("%d/%d variants removed" % (got.sum(), g["m"])) in g["summary"]


def test_oracle_kat_integers_and_r2_doubles():
    """hand-computed 6-tuples and the reference's --r2-unphased doubles, to the last bit."""
    g = load([p for p in GOLDEN if p.endswith("kat_missing.npz")][0])
    inv, mf, altmaj = T.oracle_prepare(g["raw"])
    assert not altmaj.any()
    hom, r2h, vaggs = T.oracle_split(inv, g["n"])
    want = {(0, 1): (90, 40, 80, 15, 75, 50), (0, 2): (95, 45, 85, 5, 85, 35), (1, 2): (92, 17, 77, 2, 82, 47)}
    for (i, j), tup in want.items():
        st = T.oracle_pair_stats(hom, r2h, vaggs, g["n"], i, j)
        assert st.astuple() == tup
        cov, v1, v2 = T.oracle_r2(st)
        assert cov * cov / (v1 * v2) == g["r2_square"][i, j]  # plink2_ld.cc:6673-6681, exact double
    # ALT frequencies the reference's --freq printed for this data: 0.263158 / 0.407609 / 0.45
    assert np.allclose(1.0 - mf, [50 / 190.0, 75 / 184.0, 90 / 200.0], rtol=0, atol=1e-15)
    gq = load([p for p in GOLDEN if p.endswith("kat_quirk_o2.npz")][0])
    inv, mf, _ = T.oracle_prepare(gq["raw"])
    hom, r2h, vaggs = T.oracle_split(inv, gq["n"])
    st = T.oracle_pair_stats(hom, r2h, vaggs, gq["n"], 0, 1)
    assert st.astuple() == (100, 60, 100, 40, 100, 80)
    # the step-7 quirk: B is removed by C, then still removes A
    assert list(gq["removed"]) == [True, True, False]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("encoding", ["ref", "bed", "inverse"])
def test_hip_reproduces_reference(gpu_pkg, path, encoding):
    pkg = gpu_pkg
    g = load(path)
    eng = pkg.LdPruneEngine(g["n"], g["window"], g["step"], g["is_bp"], g["r2"], order=g["order"], device=0)
    eng.set_variants(g["chr_idx"], g["bps"])
    if encoding == "ref":
        eng.load_genotypes_host(0, T.pack_2bit(g["raw"]), pkg.LDP_GENO_REF)
    elif encoding == "bed":
        lut = np.array([3, 2, 0, 1], dtype=np.uint8)
        rec = (g["n"] + 3) // 4  # .bed rows are ceil(n/4) bytes: exercises the unaligned-row path
        rows = np.ascontiguousarray(T.pack_2bit(lut[g["raw"]]).view(np.uint8).reshape(g["m"], -1)[:, :rec])
        eng.load_genotypes_host(0, rows, pkg.LDP_GENO_BED)
    else:
        inv, mf, _ = T.oracle_prepare(g["raw"])
        eng.load_genotypes_host(0, inv, pkg.LDP_GENO_INVERSE)
        eng.set_maj_freqs(0, mf)
    got = eng.run()
    eng.close()
    assert np.array_equal(got, g["removed"]), "%d vs %d removed" % (got.sum(), g["removed"].sum())


@pytest.mark.gpu
def test_hip_kat_integers(gpu_pkg):
    pkg = gpu_pkg
    g = load([p for p in GOLDEN if p.endswith("kat_missing.npz")][0])
    eng = pkg.LdPruneEngine(g["n"], 50, 1, False, 0.5, device=0)
    eng.set_variants(g["chr_idx"], g["bps"])
    eng.load_genotypes_host(0, T.pack_2bit(g["raw"]), pkg.LDP_GENO_REF)
    st = eng.pair_stats([0, 0, 1], [1, 2, 2])
    assert [tuple(int(x) for x in r) for r in st] == [(90, 40, 80, 15, 75, 50), (95, 45, 85, 5, 85, 35), (92, 17, 77, 2, 82, 47)]
    removed, tile_stats = eng.run_with_stats()
    # band in count mode with window 50: pairs (0,1), (0,2), (1,2) in pair_off order j=1:(0,1); j=2:(0,2),(1,2)
    assert [tuple(int(x) for x in r) for r in tile_stats] == [(90, 40, 80, 15, 75, 50), (95, 45, 85, 5, 85, 35), (92, 17, 77, 2, 82, 47)]
    assert not removed.any()
    eng.close()
