"""CPU tests: oracle pieces against the reference's OWN objects (fet.c, multtest.c, utils.c compiled
unmodified into oracle/_ref/libref_parts.so by `make -C oracle ref`).  Skipped where the reference tree
is not mounted (the GPU box)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref_parts()
    if r is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return r


def test_fisher_exact_bitwise(oracle, ref):
    rng = np.random.default_rng(0)
    L = oracle.lib()
    for i in range(4000):
        hi = 30 if i % 3 == 0 else 6000
        a, b, c, d = [int(x) for x in rng.integers(0, hi, 4)]
        o = [C.c_double() for _ in range(3)]
        r = [C.c_double() for _ in range(3)]
        q1 = L.orc_fisher_exact(a, b, c, d, *o)
        q2 = ref.kt_fisher_exact(a, b, c, d, *r)
        assert q1 == q2 and all(x.value == y.value for x, y in zip(o, r)), (a, b, c, d)


def test_multtest_bitwise(oracle, ref):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    for n in (1, 2, 10, 50, 333):
        for alpha in (0.001, 0.05, 0.25, 1.0):
            p = np.round(rng.random(n) ** 3, 4)          # ties on purpose
            a, b = p.copy(), p.copy()
            L.orc_bonf_corr(a.ctypes.data_as(dp), n, 0)
            ref.bonf_corr(b.ctypes.data_as(dp), n, 0)
            assert np.array_equal(a, b)
            a, b = p.copy(), p.copy()
            L.orc_holm_bonf_corr(a.ctypes.data_as(dp), n, alpha, 0)
            ref.holm_bonf_corr(b.ctypes.data_as(dp), n, alpha, 0)
            assert np.array_equal(a, b)
            for ntests in (0, 2 * n):
                rej = (C.c_long * n)()
                n1 = L.orc_fdr(p.ctypes.data_as(dp), n, alpha, ntests, rej)
                ptr = C.POINTER(C.c_long)()
                n2 = ref.fdr(p.ctypes.data_as(dp), n, alpha, ntests, C.byref(ptr))
                assert n1 == n2
                assert sorted(rej[i] for i in range(n1)) == sorted(ptr[i] for i in range(n2))


def test_int_median_and_dbl_cmp(oracle, ref):
    rng = np.random.default_rng(2)
    L = oracle.lib()
    for n in (1, 2, 3, 10, 11, 100):
        v = rng.integers(0, 42, n).astype(np.int32)
        p = v.ctypes.data_as(C.POINTER(C.c_int))
        assert L.orc_int_median(p, n) == ref.int_median(p, n)
    L.orc_dbl_cmp.argtypes = [C.c_void_p, C.c_void_p]
    for x, y in [(1.0, 1.0 + 1e-16), (1.0, 1.0 + 3e-16), (0.5, 0.25), (1e-20, 2e-20)]:
        a, b = C.c_double(x), C.c_double(y)
        assert L.orc_dbl_cmp(C.byref(a), C.byref(b)) == ref.dbl_cmp(C.byref(a), C.byref(b))
