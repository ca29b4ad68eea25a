"""CPU, world_size = 2 over gloo: the N > 1 path -- TorchComm, Statistics.allreduce,
legacy Stats gather, mpi.distribute and the MC driver's tensor all-reduce.
The closed forms are the reference's own (orphics/tests/test_stats.py:12-183),
which is meant to run under ``mpirun -n P``; here P = 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from orphics_amd import mc, mpi, stats
        comm = mpi.TorchComm()
        P = world
        out = {}
        # --- test_scalar_mean_with_closed_form
        m_r = rank + 1
        acc = stats.Statistics(comm=comm)
        acc.extend("A", np.arange(1, m_r + 1, dtype=np.float64).reshape(m_r, 1))
        # --- test_covariance_from_outer_sums_closed_form
        acc.extend("C", np.tile(np.array([float(rank), 2.0 * rank]), (m_r, 1)))
        # --- test_label_present_on_subset_of_ranks
        acc.extend("train", np.ones((m_r, 1)))
        if rank % 2 == 0:
            acc.extend("valid", 2.0 * np.ones((m_r, 1)))
        # --- test_stack_2d_array
        base = np.arange(6, dtype=np.float64).reshape(2, 3)
        acc.add_stack("S", (rank + 1) * base)
        # --- test_var_equals_diag_cov
        acc.extend("profiles", np.tile(np.array([rank, 2. * rank, 3. * rank]), (m_r, 1)))
        acc.allreduce()
        out["meanA"] = acc.mean("A")[0]
        out["covC"] = acc.cov("C", ddof=1)
        out["train"] = acc.mean("train")[0]
        out["valid"] = acc.mean("valid")[0]
        out["valid_n"] = acc.count("valid")
        out["stack"] = acc.stack_sum("S")
        out["stack_n"] = acc.stack_count("S")
        out["var"] = acc.var("profiles")
        out["covdiag"] = np.diag(acc.cov("profiles"))
        # --- mode mismatch across ranks must raise everywhere
        bad = stats.Statistics(comm=comm)
        if rank == 0:
            bad.add("x", np.zeros(3))
        else:
            bad.add_stack("x", np.zeros(3))
        try:
            bad.allreduce()
            out["mismatch"] = False
        except ValueError:
            out["mismatch"] = True
        # --- legacy Stats gather-to-root
        st = stats.Stats(comm=comm)
        for i in range(rank + 2):
            st.add_to_stats("v", np.array([rank + i, 2.0 * i]))
        st.add_to_stack("k", np.full((2, 2), float(rank + 1)))
        st.get_stats(verbose=False)
        st.get_stacks(verbose=False)
        if rank == 0:
            out["legacy_n"] = st.vectors["v"].shape[0]
            out["legacy_mean"] = st.stats["v"]["mean"]
            out["legacy_stack"] = st.stacks["k"]
        # --- task split + the MC driver's single all-reduce
        _, _, mine = mpi.distribute(7, verbose=False, comm=comm)
        out["tasks"] = mine
        n = torch.tensor([len(mine)], dtype=torch.int64)
        S = torch.tensor([float(sum(mine)), 1.0], dtype=torch.float64)
        mc.allreduce_tensors([n, S], comm)
        out["mc_n"], out["mc_S"] = int(n.item()), S.numpy().copy()
        # --- mean-field stack with a declared support: only the active region goes through the all-reduce; a rank that
        #     stacked nothing still takes part with the same buffer; the local accumulator is left untouched
        reg = stats.Statistics(comm=comm, device=torch.device("cpu"))
        reg.PACK_LIMIT = 100                                  # (so that this small plane takes the large-stack path)
        ny, kp, rb, w = 16, 12, 3, 5
        if rank == 1:
            plane = reg.device_stack("mf", (ny, kp, 2), support=(rb, w))
            plane[:rb, :w] += 2.0
            plane[ny - rb + 1:, :w] += 3.0
            reg.note_stacked("mf", 4)
            keep = plane.clone()
        reg.add("b", np.array([1.0 + rank, 2.0]))
        reg.allreduce()
        full = reg.stack_sum("mf")
        want = np.zeros((ny, kp, 2)); want[:rb, :w] = 2.0; want[ny - rb + 1:, :w] = 3.0
        out["region_ok"] = bool(np.array_equal(full, want)) and reg.stack_count("mf") == 4 and reg.count("b") == 2
        if rank == 1:
            out["region_ok"] = out["region_ok"] and bool(torch.equal(plane, keep))
        comm.Barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


def test_world_size_two_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, o = q.get(timeout=240)
        res[r] = o
    for p in procs:
        p.join(60)
    for r in (0, 1):
        assert not isinstance(res[r], str), res[r]
    P = 2
    N = P * (P + 1) // 2
    SUM = sum((r + 1) * (r + 2) // 2 for r in range(P))
    for r in (0, 1):
        o = res[r]
        np.testing.assert_allclose(o["meanA"], SUM / N, rtol=0, atol=0)
        S1 = sum(rr * (rr + 1) for rr in range(P))
        T = sum(rr * rr * (rr + 1) for rr in range(P))
        M = float(T - S1 * S1 / N) * np.array([[1., 2.], [2., 4.]])
        np.testing.assert_allclose(o["covC"], M / (N - 1), rtol=0, atol=0)
        assert o["train"] == 1.0 and o["valid"] == 2.0 and o["valid_n"] == 1
        assert np.allclose(o["stack"], 3.0 * np.arange(6.).reshape(2, 3)) and o["stack_n"] == 2
        np.testing.assert_allclose(o["var"], o["covdiag"], rtol=0, atol=1e-12)
        assert o["mismatch"] is True
        assert o["mc_n"] == 7 and np.allclose(o["mc_S"], [21.0, 2.0])
        assert o["region_ok"] is True
    assert res[0]["tasks"] == [0, 1, 2] and res[1]["tasks"] == [3, 4, 5, 6]   # remainder on the LAST rank
    assert res[0]["legacy_n"] == 5
    np.testing.assert_allclose(res[0]["legacy_mean"], np.mean([[0, 0], [1, 2], [1, 0], [2, 2], [3, 4]], axis=0))
    np.testing.assert_allclose(res[0]["legacy_stack"], np.full((2, 2), 1.5))


def test_abort_on_exception_exits_nonzero_and_launcher_reaps_peers(tmp_path):
    """mpi.mpi_abort_on_exception (role of mpi.py:31-39): a rank whose user loop raises leaves with a non-zero
    exit code at once; the peer stuck in its next collective is reaped by the launcher (bench.spawn_ranks logic)."""
    import subprocess
    import time
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
dist.init_process_group("gloo")
from orphics_amd import mpi
comm = mpi.TorchComm()
with mpi.mpi_abort_on_exception(comm):
    if comm.Get_rank() == 1:
        raise ValueError("boom on rank 1")
    comm.Barrier()          # rank 0 waits for a peer that is gone
print("unreachable on rank 1")
''' % ROOT)
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    while procs[1].poll() is None and time.time() - t0 < 120:
        time.sleep(0.1)
    assert procs[1].returncode == 1
    err = procs[1].stderr.read()
    assert "boom on rank 1" in err and "unreachable" not in procs[1].stdout.read()
    procs[0].terminate()
    procs[0].wait(timeout=60)
    assert procs[0].returncode != 0
