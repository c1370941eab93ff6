"""Dosage tracks (pgen_spec.tex:566-606) as far as the prune needs them: the reference takes every allele frequency -- hence the
major allele and the tie-break of --indep-pairwise -- from dosages where a file has them (LoadAlleleAndGenoCountsThread,
plink2_data.cc:2421-2443; GetBasicGenotypeCountsAndDosage16s, pgenlib_read.cc:7917), the hardcalls only feed r^2.
ldp_pgen_dosage_sums() must reproduce the two sums exactly.  The files are written by the test (records: plain main track,
optionally the hardcall-phase track, then the dosage list / bit array / one value per sample; the fixed-width modes 0x03 and
0x04), so the expected sums are numpy's; the reference reads the same files and its --freq agrees to its six digits."""
import os

import numpy as np
import pytest

import ldtools as T


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def deltalist(ids, n):
    """pgen_spec.tex:367-391 without the 2-bit values"""
    L = len(ids)
    out = varint(L)
    if not L:
        return out
    idw = 1 if n <= 256 else (2 if n <= 65536 else (3 if n <= (1 << 24) else 4))
    G = (L + 63) // 64
    firsts = b"".join(int(ids[64 * g]).to_bytes(idw, "little") for g in range(G))
    groups = []
    for g in range(G):
        chunk = ids[64 * g:64 * g + 64]
        groups.append(b"".join(varint(int(chunk[k] - chunk[k - 1])) for k in range(1, len(chunk))))
    sizes = bytes(len(groups[g]) - 63 for g in range(G - 1))
    return out + firsts + sizes + b"".join(groups)


def synth_dosages(raw, rng, rate):
    """per call: a dosage (0..32768 per diploid sample, within 0.3 of its hardcall) or -1 = none"""
    m, n = raw.shape
    base = np.where(raw == 3, 0, raw).astype(np.int64) * 16384
    jitter = rng.integers(-4900, 4900, size=raw.shape)
    d = np.clip(base + jitter, 0, 32768)
    has = (rng.random(raw.shape) < rate) & (raw != 3)
    extra = (raw == 3) & (rng.random(raw.shape) < 0.3 * rate)      # a dosage without a hardcall (too uncertain to call)
    d = np.where(extra, rng.integers(5000, 11000, size=raw.shape), d)
    return np.where(has | extra, d, -1)


def dosage_records(raw, dos, kinds, phaseinfo=None):
    """kinds[v]: 0 no dosage track, 1 list, 2 one value per sample, 3 bit array"""
    m, n = raw.shape
    rec = (n + 3) // 4
    packed = T.pack_2bit(raw).view(np.uint8).reshape(m, -1)[:, :rec]
    records, vrtypes = [], np.zeros(m, dtype=np.uint8)
    for v in range(m):
        body = packed[v].tobytes()
        vt = 0
        het = np.flatnonzero(raw[v] == 1)
        if phaseinfo is not None and len(het) and v % 3 != 2:
            # hardcall-phase track in front of the dosages; every third of these leaves some het calls unphased (explicit phasepresent)
            if v % 3 == 0:
                bits = np.zeros(1 + len(het), dtype=np.uint8)
                bits[1:] = phaseinfo[v, het] & 1
                body += np.packbits(bits, bitorder="little").tobytes()
            else:
                present = (np.arange(len(het)) % 4 != 1).astype(np.uint8)
                bits = np.concatenate([[1], present]).astype(np.uint8)
                body += np.packbits(bits, bitorder="little").tobytes()
                body += np.packbits(phaseinfo[v, het][present == 1] & 1, bitorder="little").tobytes()
            vt |= 0x10
        ids = np.flatnonzero(dos[v] >= 0)
        if kinds[v] == 1:
            body += deltalist(ids, n) + dos[v, ids].astype("<u2").tobytes()
            vt |= 0x20
        elif kinds[v] == 2:
            body += np.where(dos[v] >= 0, dos[v], 65535).astype("<u2").tobytes()
            vt |= 0x40
        elif kinds[v] == 3:
            body += np.packbits(dos[v] >= 0, bitorder="little").tobytes() + dos[v, ids].astype("<u2").tobytes()
            vt |= 0x60
        records.append(body)
        vrtypes[v] = vt
    return records, vrtypes


def expected_sums(raw_v, dos_v, mask):
    """numpy restatement: a sample's dosage where it has one, its hardcall otherwise"""
    has = (dos_v >= 0) & mask
    alt = int(dos_v[has].sum()) + int((np.where(raw_v == 3, 0, raw_v)[(~(dos_v >= 0)) & mask]).sum()) * 16384
    nm = int(has.sum()) + int(((raw_v != 3) & (dos_v < 0) & mask).sum())
    return nm * 32768 - alt, alt


def freq_agrees(text, x):
    """the reference prints six significant digits"""
    if x == 0.0:
        return float(text) == 0.0
    return abs(float(text) - x) <= 0.6 * 10.0 ** (np.floor(np.log10(x)) - 5)


def make_case(m, n, seed, fixed_kind_two=False):
    rng = np.random.default_rng(seed)
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.06)
    kinds = np.full(m, 2) if fixed_kind_two else rng.integers(0, 4, size=m)
    rate = np.where(kinds == 1, 0.05, 0.6)[:, None]
    dos = synth_dosages(raw, rng, rate)
    dos[kinds == 0] = -1
    two = (kinds == 2)
    # one value per sample: every called sample has one (pgen_spec.tex:601-604), a sample without a call may
    dos[two] = np.where((raw[two] != 3) & (dos[two] < 0), np.clip(raw[two].astype(np.int64) * 16384, 0, 32768), dos[two])
    return raw, dos, kinds, rng


@pytest.mark.parametrize("n", [37, 300, 70001])
def test_dosage_sums_of_handwritten_records(pkg, tmp_path, n):
    m = 60 if n < 1000 else 12
    raw, dos, kinds, rng = make_case(m, n, n)
    _, info = T.synth_phase(raw, n + 1)
    records, vrtypes = dosage_records(raw, dos, kinds, info)
    prefix = str(tmp_path / "d")
    T.write_pgen_records(prefix, records, vrtypes, n, ["1"] * m, np.arange(m) + 1)
    f = pkg.PgenFile(prefix + ".pgen")
    assert f.has_dosage() == bool((kinds > 0).any())
    rows = f.read()
    pad = (-rows.shape[1]) % 8
    assert np.array_equal(T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n), raw)  # (hardcalls unaffected)
    sub = rng.random(n) < 0.7
    for v in range(m):
        assert f.has_dosage(v) == bool(kinds[v] > 0)
        assert f.dosage_sums(v) == expected_sums(raw[v], dos[v], np.ones(n, dtype=bool)), (v, kinds[v])
        assert f.dosage_sums(v, sub) == expected_sums(raw[v], dos[v], sub), (v, kinds[v])
    f.close()
    if T.have_ref() and n < 1000:
        cp = T.run_ref(["--pfile", "d", "--freq", "--nonfounders", "--out", "r"], str(tmp_path))
        assert cp.returncode == 0, cp.stdout
        got = [ln.split() for ln in open(str(tmp_path / "r.afreq")) if not ln.startswith("#")]
        assert len(got) == m
        for v in range(m):
            ref, alt = expected_sums(raw[v], dos[v], np.ones(n, dtype=bool))
            if ref + alt:
                assert freq_agrees(got[v][4], alt / (ref + alt)) and int(got[v][5]) == (ref + alt) // 16384, (v, got[v])


@pytest.mark.parametrize("mode", [3, 4])
def test_fixed_width_dosage_modes(pkg, tmp_path, mode):
    """Storage modes 0x03 / 0x04 (pgen_spec.tex:142-145; PgfiInitPhase1, pgenlib_read.cc:885-913): hardcalls, one dosage per sample
    and -- mode 4 -- one phased-dosage difference per sample, fixed record width."""
    m, n = 45, 83
    raw, dos, kinds, rng = make_case(m, n, 100 + mode, fixed_kind_two=True)
    rec = (n + 3) // 4
    packed = T.pack_2bit(raw).view(np.uint8).reshape(m, -1)[:, :rec]
    prefix = str(tmp_path / "f")
    with open(prefix + ".pgen", "wb") as fh:
        fh.write(bytes([0x6C, 0x1B, mode]) + np.uint32(m).tobytes() + np.uint32(n).tobytes() + bytes([0x40]))
        for v in range(m):
            fh.write(packed[v].tobytes() + np.where(dos[v] >= 0, dos[v], 65535).astype("<u2").tobytes())
            if mode == 4:
                fh.write(np.zeros(n, dtype="<i2").tobytes())
    with open(prefix + ".pvar", "w") as fh:
        fh.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join("1\t%d\tsnp%d\tA\tC\n" % (v + 1, v) for v in range(m)))
    with open(prefix + ".psam", "w") as fh:
        fh.write("#IID\tSEX\n" + "".join("s%d\t2\n" % s for s in range(n)))
    f = pkg.PgenFile(prefix + ".pgen")
    assert (f.variant_ct, f.sample_ct, f.encoding) == (m, n, pkg.LDP_GENO_REF) and f.has_dosage()
    rows = f.read()
    pad = (-rows.shape[1]) % 8
    assert np.array_equal(T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n), raw)
    recs, base = f.record_index()
    assert all(int(recs[v].length) == rec + n * (2 if mode == 3 else 4) for v in range(m))
    sub = rng.random(n) < 0.5
    for v in range(m):
        assert f.dosage_sums(v) == expected_sums(raw[v], dos[v], np.ones(n, dtype=bool))
        assert f.dosage_sums(v, sub) == expected_sums(raw[v], dos[v], sub)
    f.close()
    if T.have_ref():
        # (one thread: the reference's multithreaded allele-count pass crashes or reports a malformed file on these modes at most
        # sample counts -- nobody writes them --, single-threaded it reads them)
        cp = T.run_ref(["--pfile", "f", "--freq", "--threads", "1", "--out", "r"], str(tmp_path))
        assert cp.returncode == 0, cp.stdout
        got = [ln.split() for ln in open(str(tmp_path / "r.afreq")) if not ln.startswith("#")]
        for v in range(m):
            ref, alt = expected_sums(raw[v], dos[v], np.ones(n, dtype=bool))
            if ref + alt:
                assert freq_agrees(got[v][4], alt / (ref + alt)) and int(got[v][5]) == (ref + alt) // 16384, (v, got[v])
    # a truncated file is refused with the reference's message
    data = open(prefix + ".pgen", "rb").read()
    open(prefix + "_cut.pgen", "wb").write(data[:-3])
    with pytest.raises(pkg.LdpError) as ei:
        pkg.PgenFile(prefix + "_cut.pgen")
    assert "Unexpected .pgen file size" in str(ei.value)


def test_dosage_sums_without_a_dosage_track_are_the_hardcall_counts(pkg, tmp_path):
    m, n = 20, 61
    raw = T.synth_raw_codes(m, n, 9, missing_rate=0.1)
    prefix = str(tmp_path / "h")
    T.write_pgen_fixed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    T.write_bed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    for path, args in ((prefix + ".pgen", ()), (prefix + ".bed", (n, m))):
        f = pkg.PgenFile(path, *args)
        assert not f.has_dosage()
        for v in range(m):
            assert f.dosage_sums(v) == expected_sums(raw[v], np.full(n, -1), np.ones(n, dtype=bool))
        f.close()


@pytest.mark.parametrize("freq,order,window,step,is_bp,seed", [(0.3, 2, 60, 4, False, 31), (0.95, 1, 200000, 1, True, 96), (0.05, 2, 100, 1, False, 7)])
def test_oracle_prune_with_dosage_frequencies_reproduces_the_reference(pkg, tmp_path, freq, order, window, step, is_bp, seed):
    """What plink2-hip does with such files, end to end on the CPU: hardcalls -> the oracle's greedy scan, with the major allele's
    frequency of every variant taken from ldp_pgen_dosage_sums in ComputeAlleleFreqs' arithmetic.  Same list as the reference's
    --indep-pairwise on a file its own --dummy wrote (dosage lists and bit arrays); hardcall frequencies give another one."""
    if not T.have_ref():
        pytest.skip("oracle/_ref/plink2 not built")
    n, m = 130, 1500
    cp = T.run_ref(["--dummy", str(n), str(m), "dosage-freq=%g" % freq, "--seed", str(seed), "--threads", "2", "--make-pgen", "--out", "dos"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    wargs = ["%dkb" % (window // 1000)] if is_bp else [str(window), str(step)]
    kept, removed, _ = T.ref_indep_pairwise(str(tmp_path / "dos"), wargs, 0.03, order=order, threads=4)
    f = pkg.PgenFile(str(tmp_path / "dos.pgen"))
    rows = f.read()
    pad = (-rows.shape[1]) % 8
    raw = T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n)
    inv, mf_hard, _ = T.oracle_prepare(raw)
    mf = np.zeros(m)
    for v in range(m):
        ref_dd, alt_dd = f.dosage_sums(v)
        tot = ref_dd + alt_dd
        ref_freq = float(ref_dd) * (1.0 / float(tot)) if tot else 0.5
        mf[v] = (1.0 - ref_freq) if ref_freq < 0.5 else ref_freq
    f.close()
    chr_idx = np.zeros(m, dtype=np.uint32)
    bps = np.arange(m, dtype=np.uint32)      # (--dummy: POS 0, 1, 2, ...)
    got, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, 0.03, order)
    want = np.ones(m, dtype=bool)
    want[[int(x[3:]) for x in kept]] = False
    assert len(kept) + len(removed) == m and np.array_equal(got, want)
    if freq >= 0.3:
        hard, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf_hard, window, step, is_bp, 0.03, order)
        assert not np.array_equal(hard, want)


def test_golden_dosage_file(pkg):
    """tests/golden/pgen/dosage_small.*: a file the reference's --dummy wrote, its --freq and two of its --indep-pairwise lists
    (tests/golden/make_golden_dosage.py) -- pins the dosage reader and the frequency arithmetic where the reference binary is absent."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pgen")
    g = np.load(os.path.join(gold, "dosage_small.npz"))
    n, m = int(g["n"]), int(g["m"])
    f = pkg.PgenFile(os.path.join(gold, "dosage_small.pgen"))
    assert (f.variant_ct, f.sample_ct) == (m, n) and f.has_dosage()
    rows = f.read()
    pad = (-rows.shape[1]) % 8
    raw = T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n)
    mf = np.zeros(m)
    kinds = set()
    for v in range(m):
        ref_dd, alt_dd = f.dosage_sums(v)
        tot = ref_dd + alt_dd
        assert tot // 16384 == int(g["obs_ct"][v]) and freq_agrees(str(g["alt_freq_text"][v]), alt_dd / tot)
        ref_freq = float(ref_dd) * (1.0 / float(tot))
        mf[v] = (1.0 - ref_freq) if ref_freq < 0.5 else ref_freq
        kinds.add(int(f.record_index(v, 1)[0][0].vrtype) & 0x60)
    assert {0x20, 0x60} <= kinds   # dosage lists and bit arrays both occur
    f.close()
    inv, _, _ = T.oracle_prepare(raw)
    chr_idx, bps = np.zeros(m, dtype=np.uint32), np.arange(m, dtype=np.uint32)
    got, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 50, 5, False, 0.3, 2)
    assert np.array_equal(got, g["removed_count_o2"])
    got, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 30000, 1, True, 0.45, 1)
    assert np.array_equal(got, g["removed_kb_o1"])
