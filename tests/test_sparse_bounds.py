"""The interval test of DESIGN 4.1d (classify_sparse in csrc/ldp_pair_mfma.hip), restated in numpy: per-variant counts confine
the five pairwise-complete statistics of a pair; whenever interval arithmetic settles the prune predicate it settles it the way
the exact integers do.  A property of the mathematics, checked without a GPU (the kernel itself is compared with the six-product
kernel and the oracle in tests/test_gpu_parity.py::test_rows_with_a_few_missing_calls)."""
import numpy as np
import pytest


def classify(N, thresh, dot, Mi, Si, Qi, Mj, Sj, Qj):
    nm_lo, nm_hi = max(N - Mi - Mj, 0.0), N - max(Mi, Mj)
    k1, k2 = min(Mj, Qi), min(Mi, Qj)
    s1_lo, s1_hi, s2_lo, s2_hi = Si - k1, Si + k1, Sj - k2, Sj + k2
    a0, a1 = dot * nm_lo, dot * nm_hi
    a_lo, a_hi = min(a0, a1), max(a0, a1)
    p = [s1_lo * s2_lo, s1_lo * s2_hi, s1_hi * s2_lo, s1_hi * s2_hi]
    c_lo, c_hi = a_lo - max(p), a_hi - min(p)
    c2_hi = max(c_lo * c_lo, c_hi * c_hi)
    c2_lo = 0.0 if (c_lo <= 0.0 <= c_hi) else min(c_lo * c_lo, c_hi * c_hi)

    def sq(lo, hi):
        return (0.0 if (lo <= 0.0 <= hi) else min(lo * lo, hi * hi)), max(lo * lo, hi * hi)
    q1_lo, q1_hi = sq(s1_lo, s1_hi)
    q2_lo, q2_hi = sq(s2_lo, s2_hi)
    v1_lo, v1_hi = max((Qi - k1) * nm_lo - q1_hi, 0.0), max(Qi * nm_hi - q1_lo, 0.0)
    v2_lo, v2_hi = max((Qj - k2) * nm_lo - q2_hi, 0.0), max(Qj * nm_hi - q2_lo, 0.0)
    rhs_lo, rhs_hi = thresh * v1_lo * v2_lo, thresh * v1_hi * v2_hi
    if c2_lo > rhs_hi * (1.0 + 1e-9) + 1.0:
        return 1
    if c2_hi * (1.0 + 1e-9) + 1.0 < rhs_lo:
        return 0
    return 2


@pytest.mark.parametrize("n,miss,r2", [(2000, 0.002, 0.5), (20000, 0.001, 0.2), (5000, 0.01, 0.8), (800, 0.05, 0.1), (50000, 0.0005, 0.5)])
def test_interval_test_never_contradicts_the_exact_predicate(n, miss, r2):
    rng = np.random.default_rng(n)
    thresh = r2 * (1 + 2.0 ** -44)
    settled = total = 0
    for trial in range(400):
        maf = rng.uniform(0.005, 0.5, 2)
        u0 = rng.random(n)
        # anything from independent to identical, with a crowd near the threshold
        share = rng.choice([0.0, 1.0, np.sqrt(r2), rng.random()])
        x = np.zeros((2, n), dtype=np.int64)
        for k in range(2):
            u = u0 if (k == 0) else np.where(rng.random(n) < share, u0, rng.random(n))
            p = maf[k]
            x[k] = np.where(u < (1 - p) ** 2, 1, np.where(u < (1 - p) ** 2 + 2 * p * (1 - p), 0, -1))
        gone = rng.random((2, n)) < miss
        if trial % 7 == 0:      # the adversarial placement: j's gaps on i's minor-allele carriers
            carriers = np.flatnonzero(x[0] != 1)
            gone[1] = False
            gone[1, carriers[:max(1, int(miss * n))]] = True
        x = np.where(gone, 0, x)
        present = ~gone
        Mi, Mj = int(gone[0].sum()), int(gone[1].sum())
        Si, Sj = int(x[0].sum()), int(x[1].sum())
        Qi, Qj = int((x[0] != 0).sum()), int((x[1] != 0).sum())
        dot = int((x[0] * x[1]).sum())
        both = present[0] & present[1]
        nm = int(both.sum())
        s1, s2 = int(x[0][both].sum()), int(x[1][both].sum())
        q1, q2 = int((x[0][both] != 0).sum()), int((x[1][both] != 0).sum())
        cov = float(dot * nm - s1 * s2)
        exact = cov * cov > thresh * float(q1 * nm - s1 * s1) * float(q2 * nm - s2 * s2)
        cls = classify(float(n), thresh, float(dot), Mi, Si, Qi, Mj, Sj, Qj)
        total += 1
        if cls != 2:
            settled += 1
            assert cls == int(exact), (trial, cls, exact, Mi, Mj)
    if miss <= 0.002:
        assert settled > 0.5 * total     # ... and at the rates the path is routed to it does settle most pairs
