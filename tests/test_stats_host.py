"""CPU: host-side logic of orphics_amd.stats / mpi (Statistics, Stats, get_stats,
mpi_distribute) against the reference fixtures and the closed forms of the
reference's own orphics/tests/test_stats.py (P = 1 here; P = 2 in
test_distributed_cpu.py)."""
import os

import numpy as np
import pytest

from orphics_amd import mpi, stats


@pytest.fixture(scope="module")
def gs(golden_dir):
    return np.load(os.path.join(golden_dir, "stats_reference.npz"))


def test_statistics_matches_reference_fixture(gs, tmp_path):
    X = gs["X"]
    acc = stats.Statistics(comm=None)
    acc.extend("A", X[:20])
    for row in X[20:]:
        acc.add("A", row)
    for s in gs["stack_in"]:
        acc.add_stack("S", s)
    with pytest.raises(RuntimeError):
        acc.mean("A")
    acc.allreduce()
    assert acc.count("A") == int(gs["st_count"])
    np.testing.assert_allclose(acc.mean("A"), gs["st_mean"], rtol=0, atol=0)
    np.testing.assert_allclose(acc.cov("A"), gs["st_cov"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(acc.cov("A", ddof=0), gs["st_cov0"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(acc.var("A"), gs["st_var"], rtol=1e-14)
    np.testing.assert_allclose(acc.stack_sum("S"), gs["st_stack_sum"], rtol=0, atol=0)
    assert acc.stack_count("S") == int(gs["st_stack_count"])
    path = tmp_path / "red.npz"
    acc.save_reduced(path)
    keys = set(np.load(path).files)
    assert set(gs["npz_keys"]) <= keys            # the reference's schema is a subset ...
    assert "stack/S/K" in keys                     # ... plus the stack count the reference forgets
    back = stats.Statistics.load_reduced(path)
    np.testing.assert_allclose(back.cov("A"), gs["st_cov"], rtol=1e-14, atol=1e-15)
    assert back.stack_count("S") == 5
    with pytest.raises(KeyError):
        acc.mean("nope")
    with pytest.raises(ValueError):
        acc.add_stack("A", np.zeros(3))             # label already in stats mode
    with pytest.raises(ValueError):
        acc.add("A", np.zeros(7))                   # dim mismatch


def test_get_stats_and_legacy_stats(gs):
    X = gs["X"]
    g = stats.get_stats(X)
    for k in ("mean", "cov", "covmean", "err", "errmean", "corr"):
        np.testing.assert_allclose(g[k], gs["gs_" + k], rtol=1e-13, atol=1e-15)
    st = stats.Stats()
    for row in X:
        st.add_to_stats("v", row)
    for s in gs["stack_in"]:
        st.add_to_stack("k", s)
    st.get_stats(verbose=False)
    st.get_stacks(verbose=False)
    np.testing.assert_allclose(st.stats["v"]["mean"], gs["legacy_mean"], rtol=1e-14)
    np.testing.assert_allclose(st.stats["v"]["errmean"], gs["legacy_errmean"], rtol=1e-13)
    np.testing.assert_allclose(st.stacks["k"], gs["legacy_stack"], rtol=1e-14)
    with pytest.raises(TypeError):
        st.add_to_stats("c", np.zeros(3, complex))
    with pytest.raises(AssertionError):
        st.add_to_stats("stats", np.zeros(3))


def closed_forms(P):
    N = P * (P + 1) // 2
    SUM = sum((r + 1) * (r + 2) // 2 for r in range(P))
    return N, SUM / N


def test_reference_closed_forms_single_rank():
    """orphics/tests/test_stats.py:12-183 with P = 1 (rtol = atol = 0 like the reference)."""
    acc = stats.Statistics(comm=None)
    acc.extend("A", np.arange(1, 2, dtype=np.float64).reshape(1, 1))
    acc.extend("C", np.tile(np.array([0., 0.]), (1, 1)))
    acc.add_stack("M", np.arange(6, dtype=np.float64).reshape(2, 3))
    acc.allreduce()
    N, MEAN = closed_forms(1)
    np.testing.assert_allclose(acc.mean("A")[0], MEAN, rtol=0, atol=0)
    assert np.isnan(acc.cov("C", ddof=1)).all()          # N = 1 -> NaN, as the reference expects
    assert np.allclose(acc.stack_sum("M"), np.arange(6.).reshape(2, 3)) and acc.stack_count("M") == 1


def test_mpi_distribute_and_fake_comm(golden_dir):
    m = np.load(os.path.join(golden_dir, "mpi_reference.npz"))
    for nt, nc in m["pairs"]:
        num_each, dist = mpi.mpi_distribute(int(nt), int(nc))
        assert np.array_equal(num_each, m[f"num_each_{nt}_{nc}"])
        assert np.array_equal([d[0] for d in dist], m[f"first_{nt}_{nc}"])
    with pytest.raises(AssertionError):
        mpi.mpi_distribute(3, 5)
    num_each, dist = mpi.mpi_distribute(3, 5, allow_empty=True)
    assert list(num_each) == [0, 0, 1, 1, 1]
    c = mpi.fakeMpiComm()
    assert c.Get_rank() == 0 and c.Get_size() == 1
    comm, rank, mine = mpi.distribute(7, verbose=False, comm=c)
    assert rank == 0 and mine == list(range(7))


def test_lensforecast_knoxcov_formula():
    """cosmology.py:1054-1082: per-bin var = 2 (C+N)^2 / ((2l+1) dl fsky) for an auto spectrum."""
    from orphics_amd import cosmology
    ells = np.arange(2, 3000)
    clkk = 1e-7 * (ells / 100.) ** -1.2
    nl = 2e-8 * np.ones_like(clkk) * (1 + (ells / 1500.) ** 2)
    LF = cosmology.LensForecast()
    LF.loadKK(ells, clkk, ells, nl)
    edges = np.arange(100, 2000, 100)
    var, s1, s2 = LF.KnoxCov("kk", "kk", edges, 0.4)
    for i, (a, b) in enumerate(zip(edges[:-1], edges[1:])):
        e = np.arange(a, b + 1)
        tot = np.sum(e * (np.interp(e, ells, clkk) + np.interp(e, ells, nl))) / np.sum(e)
        sig = np.sum(e * np.interp(e, ells, clkk)) / np.sum(e)
        v = 2 * tot ** 2 / (2 * (a + b) / 2. + 1) / (b - a) / 0.4
        assert abs(var[i] / v - 1) < 1e-12 and abs(s1[i] / (sig ** 2 / v) - 1) < 1e-12
    sn, errs = LF.sn(edges, 0.4, "kk")
    assert abs(sn - np.sqrt(s1.sum())) < 1e-12 and np.allclose(errs, np.sqrt(var))
    assert np.allclose(cosmology.knox_cov(3.0, 50), 2 * 9. / 50)


def test_lensforecast_matches_the_reference_methods():
    """LensForecast.KnoxCov / sn / sigmaClSquared vs outputs of the REFERENCE's own method definitions
    (cosmology.py:976-1094, executed by tests/golden/make_golden_forecast.py): autos, crosses, mixed pairs, ntot."""
    import os
    from orphics_amd import cosmology
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forecast_reference.npz"))
    ells = g["ells"]
    LF = cosmology.LensForecast()
    LF.loadKK(ells, g["kk"], ells, g["n_kk"])
    LF.loadGenericCls("gg", ells, g["gg"], ells, g["n_gg"])
    LF.loadGenericCls("kg", ells, g["kg"])
    edges, fsky = g["edges"], float(g["fsky"])
    for xy, wz in (("kk", "kk"), ("kg", "kg"), ("kk", "kg"), ("gg", "kk")):
        for ntot in (False, True):
            tag = "%s_%s_%d" % (xy, wz, int(ntot))
            cov, s1, s2 = LF.KnoxCov(xy, wz, edges, fsky, ntot=ntot)
            for got, key in ((cov, "cov_"), (s1, "s1_"), (s2, "s2_")):
                assert np.allclose(got, g[key + tag], rtol=1e-13, atol=0), (tag, key)
    sn, errs = LF.sn(edges, fsky, "kg")
    assert abs(sn / float(g["sn_kg"]) - 1) < 1e-13 and np.allclose(errs, g["errs_kg"], rtol=1e-13, atol=0)
    assert np.allclose(LF.sigmaClSquared("kk", edges, fsky), g["sigma2_kk"], rtol=1e-13, atol=0)
