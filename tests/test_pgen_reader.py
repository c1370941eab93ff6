"""The .pgen/.bed main-track reader (ldp_pgen_*).  Round trips through the REFERENCE's writer: data is
written as fixed-width .pgen by the test, re-encoded by oracle/_ref/plink2 `--make-pgen` into the standard
variable-width format (LD-compressed, one-bit and difflist records), and must decode back to the same codes.
A small reference-written variable-width file is also committed under tests/golden/ so the decoder is pinned
where the reference binary is absent."""
import os
import subprocess

import numpy as np
import pytest

import ldtools as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pgen")


def codes(rows, n):
    """(m, ceil(n/4)) uint8 packed rows -> (m, n) codes"""
    pad = (-rows.shape[1]) % 8
    return T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n)


def structured_codes(m, n, seed):
    """Rows that make the reference's writer pick every main-track record type."""
    rng = np.random.default_rng(seed)
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.01, ld_copy_prob=0.6, redraw=0.01)
    for v in range(0, m, 7):       # sparse vs all-0: difflist type 4
        raw[v] = 0
        raw[v, rng.integers(0, n, size=max(1, n // 50))] = rng.integers(1, 4)
    for v in range(3, m, 11):      # sparse vs all-2: type 6
        raw[v] = 2
        raw[v, rng.integers(0, n, size=max(1, n // 60))] = rng.integers(0, 2)
    for v in range(5, m, 13):      # mostly two categories: one-bit type 1
        raw[v] = np.where(rng.random(n) < 0.4, 1, 0)
        raw[v, rng.integers(0, n, size=3)] = 3
    for v in range(6, m, 17):      # inverted copy of the previous row: LD-compressed inverted (type 3)
        inv = raw[v - 1].copy()
        inv[raw[v - 1] == 0] = 2
        inv[raw[v - 1] == 2] = 0
        raw[v] = inv
    raw[1] = 3                     # all missing: type 7
    return raw


def test_fixed_width_and_bed_rows(pkg, tmp_path):
    m, n = 50, 77
    raw = T.synth_raw_codes(m, n, 3, missing_rate=0.05)
    prefix = str(tmp_path / "f")
    T.write_pgen_fixed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    T.write_bed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    f = pkg.PgenFile(prefix + ".pgen")
    assert (f.variant_ct, f.sample_ct, f.mode, f.encoding) == (m, n, 2, pkg.LDP_GENO_REF)
    assert np.array_equal(codes(f.read(), n), raw)
    f.close()
    b = pkg.PgenFile(prefix + ".bed", n, m)
    assert (b.mode, b.encoding) == (1, pkg.LDP_GENO_BED)
    lut = np.array([2, 3, 1, 0], dtype=np.uint8)  # bed -> pgen
    assert np.array_equal(lut[codes(b.read(5, 20), n)], raw[5:25])
    b.close()
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(prefix + ".bed", n + 4, m)   # wrong dimensions -> size check (pgenlib_read.cc:767-789)
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(prefix + ".pgen", n, m + 1)


def test_committed_variable_width_file(pkg):
    path = os.path.join(GOLD, "varwidth_small.pgen")
    z = np.load(os.path.join(GOLD, "varwidth_small_codes.npz"))
    f = pkg.PgenFile(path)
    assert f.mode == 0x10 and (f.variant_ct, f.sample_ct) == (int(z["m"]), int(z["n"]))
    got = codes(f.read(), f.sample_ct)
    assert np.array_equal(got, T.unpack_2bit(z["raw_packed"].reshape(int(z["m"]), -1).view(np.uint64), int(z["n"])))
    # every main-track record type the writer produced is exercised
    assert set(int(t) for t in z["vrtypes_present"]) >= {0, 1, 2, 4, 6}
    # random access into the middle of LD-compressed runs
    for first, cnt in [(17, 5), (100, 1), (299, 40)]:
        part = codes(f.read(first, cnt), f.sample_ct)
        assert np.array_equal(part, got[first:first + cnt])
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed", [(400, 50, 1), (900, 300, 2), (70000, 40, 3), (300, 70000, 4)])
def test_roundtrip_through_reference_writer(pkg, tmp_path, m, n, seed):
    raw = structured_codes(m, n, seed)
    prefix = str(tmp_path / "f")
    T.write_pgen_fixed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "v.pgen"))
    assert f.mode == 0x10 and (f.variant_ct, f.sample_ct) == (m, n)
    got = codes(f.read(threads=4), n)
    assert np.array_equal(got, raw)
    part = codes(f.read(m // 3, min(50, m - m // 3)), n)
    assert np.array_equal(part, raw[m // 3:m // 3 + min(50, m - m // 3)])
    f.close()


def test_malformed_files_are_rejected(pkg, tmp_path):
    src = open(os.path.join(GOLD, "varwidth_small.pgen"), "rb").read()
    bad = str(tmp_path / "t.pgen")
    open(bad, "wb").write(src[:len(src) // 2])            # truncated records
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(bad).read()
    open(bad, "wb").write(src[:2] + bytes([0x21]) + src[3:])  # external-index mode: unsupported
    with pytest.raises(pkg.LdpError) as ei:
        pkg.PgenFile(bad)
    assert ei.value.code == pkg.LDP_ERR_UNSUPPORTED
    open(bad, "wb").write(b"\x00\x01\x02")
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(bad)
