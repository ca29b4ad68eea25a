"""bench.py launch plumbing: --gpus N must be honoured (children started by the parent before any GPU call) or
refused loudly -- never a silent 1-GPU run labelled n_gpus = 1 (round-1 advice)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_resolve_world_rules():
    import bench
    assert bench.resolve_world(1, {}) == (1, 0, 0, False)
    assert bench.resolve_world(4, {}) == (4, 0, 0, True)                      # parent must spawn the ranks
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}
    assert bench.resolve_world(4, env) == (4, 2, 2, False)                    # under torchrun
    with pytest.raises(SystemExit):
        bench.resolve_world(8, env)                                           # torchrun world contradicts --gpus
    with pytest.raises(SystemExit):
        bench.resolve_world(1, {"WORLD_SIZE": "2", "RANK": "0"})


def test_gpus_more_than_visible_is_refused():
    """On a box with fewer GPUs than --gpus the parent exits non-zero before starting anything."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OA_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "only" in (r.stderr + r.stdout) and "GPU" in (r.stderr + r.stdout)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gloo_rehearsal():
    """The N > 1 code path end to end on whatever GPUs exist: 2 ranks started by bench.py itself (child torchrun),
    gloo for the moment all-reduce when only one GPU is visible, RCCL when there are two."""
    import json
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["OA_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--n", "1024",
                        "--res", "2.0", "--no-cpu", "--no-extras", "--preroll", "0.1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["steps"] == 6
    assert d["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("windowed", [False, True])
def test_bench_config_mc_two_ranks_rehearsal(windowed):
    """``bench.py --gpus 2 --config mc`` (BASELINE config 4 as the sharded job: mpi_distribute split, one oa_mc_run call per
    rank, packed moment all-reduce + region-only mean-field reduce, per-rank compute and reduce times reported): 2 ranks
    started by bench.py itself on whatever GPUs exist (gloo on a 1-GPU box, RCCL on two)."""
    import json
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["OA_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "mc", "--mc-n", "1024", "--mc-sims", "61", "--res", "2.0",
           "--prec", "f64"] + (["--mc-windowed"] if windowed else [])
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["nsims"] == 61 and d["check"]["sims_counted"] == 61 and d["check"]["stacked"] == 61
    assert [x["sims"] for x in sorted(d["per_rank"], key=lambda x: x["rank"])] == [30, 31]      # remainder on the last rank (mpi.py:81-83)
    assert d["value"] > 0
    if not windowed:      # (a tapered 1024^2 patch couples modes: its debiased N0 is not the analytic full-plane N_L)
        assert d["check"]["max_abs_pull"] < 6.0 and d["check"]["max_rel_dev_vs_analytic_N0"] < 0.5


@pytest.mark.gpu
def test_mc_two_ranks_rccl():
    """mc.GaussianN0MonteCarlo.run sharded over 2 ranks with the RCCL all-reduce (needs >= 2 visible GPUs)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the 1-GPU boxes cover this path with gloo: test_distributed_cpu.py)")
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(os.environ["LOCAL_RANK"]); torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
from orphics_amd import cosmology, lensing, maps, mc, mpi, stats
from orphics_amd.geometry import FlatGeometry
N = 512; shape = (N, N); g = FlatGeometry.from_res(shape, 2.0); th = cosmology.default_theory(); ml = g.modlmap()
beam = maps.gauss_beam(ml, 1.5); noise = np.full(shape, cosmology.white_noise_power(1.0))
tm = maps.mask_kspace(shape, g, lmin=300, lmax=2000); km = maps.mask_kspace(shape, g, lmin=20, lmax=3500)
q = lensing.qest(shape, g, th, noise2d=noise, beam2d=beam, kmask=tm, kmask_K=km, unlensed_equals_lensed=True, dtype="f32")
tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
edges = np.linspace(100, 3000, 12)
st = mc.GaussianN0MonteCarlo(q, tot, edges, comm=mpi.TorchComm(), base_seed=5).run(33)
assert st.count("n0") == 33
if rank == 0:
    ref = mc.GaussianN0MonteCarlo(q, tot, edges, comm=mpi.fakeMpiComm(), base_seed=5).run(33)
    np.testing.assert_allclose(st.mean("n0"), ref.mean("n0"), rtol=1e-12)
    print("MC2 OK")
dist.barrier(); dist.destroy_process_group()
''' % ROOT
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29612", f.name], capture_output=True, text=True, timeout=900)
    os.unlink(f.name)
    assert r.returncode == 0 and "MC2 OK" in r.stdout, r.stderr[-2000:]
