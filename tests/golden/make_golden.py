#!/usr/bin/env python3
"""Generate golden input/output vectors from the REAL reference.

Run in the build container only (it needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/orphics/stats.py and mpi.py (the only hot-path
modules importable without pixell), feeds them seeded inputs and
stores inputs + outputs as small .npz fixtures next to this script.  The
fixtures are data; no reference source travels.
"""
import os
import sys
import io
import contextlib

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
os.environ.setdefault("DISABLE_MPI", "true")

with contextlib.redirect_stdout(io.StringIO()):
    from orphics import stats as rstats  # noqa: E402
    from orphics import mpi as rmpi  # noqa: E402


def modlmap(Ny, Nx, res):
    ly = np.fft.fftfreq(Ny, res) * 2 * np.pi
    lx = np.fft.fftfreq(Nx, -res) * 2 * np.pi
    return np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)


def bin2d_cases():
    out = {}
    rng = np.random.default_rng(0)
    res = 2.0 * np.pi / 180. / 60.
    Ny, Nx = 48, 64
    ml = modlmap(Ny, Nx, res)
    edges = np.arange(100., 3000., 200.)
    data = rng.standard_normal((Ny, Nx)) * (1 + ml / 1000.)
    w = rng.uniform(0.5, 2.0, (Ny, Nx))
    b = rstats.bin2D(ml, edges)
    out["a_modlmap"], out["a_edges"], out["a_data"], out["a_weights"] = ml, edges, data, w
    out["a_digitized"] = b.digitized
    c, r = b.bin(data)
    out["a_cents"], out["a_res"] = c, r
    _, r, cnt = b.bin(data, get_count=True)
    out["a_count"] = cnt
    _, r = b.bin(data, weights=w)
    out["a_res_w"] = r
    _, _, cw = b.bin(data, weights=w, get_count=True)
    out["a_count_w"] = cw
    dn = data.copy()
    dn[rng.uniform(size=dn.shape) < 0.1] = np.nan
    out["a_data_nan"] = dn
    _, r, cnt = b.bin(dn, mask_nan=True, get_count=True)
    out["a_res_nan"], out["a_count_nan"] = r, cnt
    with np.errstate(all="ignore"):
        _, r, s = b.bin(data, err=True)
    out["a_err_ref_shifted"] = s  # the reference's (buggy, shifted) std

    # H2: exact ties on integer edges (Delta ell = 21600/4096*2 = 10.546875)
    res05 = 0.5 * np.pi / 180. / 60.
    Nt = 64
    # scale so that the fundamental equals 10.546875 like 4096^2 @0.5'
    step = res05 * 4096 / Nt
    ly = np.fft.fftfreq(Nt, step) * 2 * np.pi
    lx = np.fft.fftfreq(Nt, -step) * 2 * np.pi
    mlt = np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)
    fund = abs(ly[1])
    edges_t = np.array([0., fund * 1, fund * 5, fund * 10, fund * 13, fund * 20, fund * 25])
    bt = rstats.bin2D(mlt, edges_t)
    dt = rng.standard_normal((Nt, Nt))
    out["t_modlmap"], out["t_edges"], out["t_data"] = mlt, edges_t, dt
    out["t_digitized"] = bt.digitized
    _, r, cnt = bt.bin(dt, get_count=True)
    out["t_res"], out["t_count"] = r, cnt

    # H3: nothing exceeds the last edge -> [1:-1] drops a real bin
    ml3 = np.array([[1.5, 2.5, 3.5, 2.2], [3.9, 1.1, 2.9, 3.1]])
    e3 = np.array([1., 2., 3., 4.])
    d3 = np.arange(8.).reshape(2, 4)
    b3 = rstats.bin2D(ml3, e3)
    c3, r3 = b3.bin(d3)
    out["h3_modrmap"], out["h3_edges"], out["h3_data"] = ml3, e3, d3
    out["h3_cents"], out["h3_res"], out["h3_digitized"] = c3, r3, b3.digitized

    # appendix-A tie semantics
    v = np.array([[2., 4., 4.0000001, 8., 8.1]])
    e = np.array([2., 4., 6., 8.])
    out["tie_vals"], out["tie_edges"] = v, e
    out["tie_digitized"] = rstats.bin2D(v, e).digitized
    return out


def stats_cases():
    out = {}
    rng = np.random.default_rng(1)
    X = rng.standard_normal((37, 6)) * np.arange(1, 7)
    out["X"] = X
    g = rstats.get_stats(X)
    for k in ("mean", "cov", "covmean", "err", "errmean", "corr"):
        out["gs_" + k] = np.asarray(g[k])

    acc = rstats.Statistics(comm=None)
    acc.extend("A", X[:20])
    for row in X[20:]:
        acc.add("A", row)
    stack = rng.standard_normal((5, 4, 3))
    for s in stack:
        acc.add_stack("S", s)
    acc.allreduce()
    out["stack_in"] = stack
    out["st_count"] = np.array(acc.count("A"))
    out["st_mean"] = acc.mean("A")
    out["st_cov"] = acc.cov("A")
    out["st_cov0"] = acc.cov("A", ddof=0)
    out["st_var"] = acc.var("A")
    out["st_stack_sum"] = acc.stack_sum("S")
    out["st_stack_count"] = np.array(acc.stack_count("S"))
    path = os.path.join(HERE, "_tmp_reduced.npz")
    acc.save_reduced(path)
    d = np.load(path)
    out["npz_keys"] = np.array(sorted(d.files))
    os.remove(path)

    # legacy Stats container (fake comm)
    st = rstats.Stats()
    for row in X:
        st.add_to_stats("v", row)
    for s in stack:
        st.add_to_stack("k", s)
    st.get_stats(verbose=False)
    st.get_stacks(verbose=False)
    out["legacy_mean"] = st.stats["v"]["mean"]
    out["legacy_errmean"] = st.stats["v"]["errmean"]
    out["legacy_stack"] = st.stacks["k"]
    return out


def mpi_cases():
    out = {}
    pairs = [(10, 1), (10, 2), (10, 3), (10, 4), (1000, 8), (7, 7), (1003, 8), (17, 5)]
    out["pairs"] = np.array(pairs)
    for nt, nc in pairs:
        num_each, dist = rmpi.mpi_distribute(nt, nc)
        out[f"num_each_{nt}_{nc}"] = np.asarray(num_each)
        out[f"first_{nt}_{nc}"] = np.array([d[0] for d in dist])
    return out


if __name__ == "__main__":
    np.savez(os.path.join(HERE, "bin2d_reference.npz"), **bin2d_cases())
    np.savez(os.path.join(HERE, "stats_reference.npz"), **stats_cases())
    np.savez(os.path.join(HERE, "mpi_reference.npz"), **mpi_cases())
    print("golden fixtures written to", HERE)
