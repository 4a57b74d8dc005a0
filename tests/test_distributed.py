"""The N > 1 path on CPU: two gloo ranks each own a shard; one all-reduce of the per-step sums per flush."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

torch.set_default_dtype(torch.float64)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, nt, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import sc_oracle as orc
    from semiclassical_amd import distributed as D
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(1)
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = cases.load("hk_as5_chi002")
    n_total = g["zi"].shape[1]
    counts = [D.shard_count(n_total, i, world) for i in range(world)]
    lo = sum(counts[:rank])
    sl = slice(lo, lo + counts[rank])
    # this rank's shard of the SAME initial conditions, normalised with the global N
    pot = cases.oracle_potential(g)
    prop = orc.HKOracle(cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"]))
    prop.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]),
                                cases.T(g["zi"][:, sl]), cases.T(g["probi"][sl]))
    prop.ntraj = n_total                      # Monte-Carlo weight 1/(N_total P)
    slots = torch.zeros((nt, 5))
    for t in range(nt):
        c = torch.sum(prop.autocorrelation_qp() / prop._mc_weight())
        slots[t, 0], slots[t, 1] = c.real, c.imag
        slots[t, 4] = rank + 1.0              # rank-local column must survive the flush
        prop.step(pot, float(g["dt"]))
    D.flush_correlations(slots)
    assert slots[0, 4] == rank + 1.0
    if rank == 0:
        q.put(slots[:, :2].numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flush_equals_single_process():
    nt, world = 12, 2
    g = cases.load("hk_as5_chi002")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, nt, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from semiclassical_amd import hostmath
    phase = np.exp(1j * hostmath.time_grid(nt, float(g["dt"])) * float(g["E0"]))
    cauto = (out[:, 0] + 1j * out[:, 1]) * phase
    assert cases.rel_err(cauto, g["cauto"][:nt]) < 1e-12


def test_shard_counts():
    from semiclassical_amd import distributed as D
    assert [D.shard_count(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert sum(D.shard_count(10 ** 6, r, 8) for r in range(8)) == 10 ** 6
    assert D.shard_count(5, 0, 1) == 5


def test_flush_is_identity_without_process_group():
    from semiclassical_amd import distributed as D
    s = torch.arange(10.0).reshape(2, 5)
    assert torch.equal(D.flush_correlations(s.clone()), s)


def test_shard_slices_partition_the_batch():
    from semiclassical_amd import distributed as D
    idx = np.arange(1003)
    parts = [idx[D.shard_slice(1003, r, 8)] for r in range(8)]
    assert np.array_equal(np.concatenate(parts), idx)
    assert [len(p) for p in parts] == [D.shard_count(1003, r, 8) for r in range(8)]


@pytest.mark.parametrize("nproc", [2, 3])
def test_bench_self_launcher_forms_a_group(nproc):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N rank processes itself (the parent
    only waits); here the ranks form a gloo group on CPU and report their coordinates, the engine is not imported."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [(d["rank"], d["world"], d["local_rank"]) for d in rows] == [(i, nproc, i) for i in range(nproc)]
    assert all(d["sum"] == nproc * (nproc + 1) / 2 for d in rows)


def test_launcher_reports_a_failing_rank():
    from semiclassical_amd import distributed as D
    script = os.path.join(ROOT, "tests", "_rank_fail.py")
    assert D.launch_local_ranks([script], 2, timeout=120) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("case,tol", [("hk_as5_chi002", 1e-9), ("hk_as60", 1e-9), ("wm_methylium", 1e-8)])
def test_two_engine_ranks_on_one_gpu(case, tol, tmp_path):
    """Two rank processes share cuda:0; each runs the HIP engine on half of the golden initial conditions with the
    global N as weight, the raw sums are flushed through distributed.flush_correlations (gloo) and must reproduce the
    single-process golden C_auto / k_ic (cli.py:453-458 normalisation)."""
    from semiclassical_amd import distributed as D
    g = cases.load(case)
    nt = min(12, len(g["cauto"]))
    out = str(tmp_path / "flush.npz")
    with_norm = case != "hk_as60"                # the O(n^2) norm of the small cases only
    rc = D.launch_local_ranks([os.path.join(ROOT, "tests", "_rank_engine.py"), case, str(nt), out], 2, timeout=600,
                              extra_env={"SC_DIST_BACKEND": "gloo", "SC_TEST_DEVICE": "0", "SC_TEST_NORM": "1" if with_norm else ""})
    assert rc == 0, f"a rank process failed (largest exit code {rc})"        # a rank that dies AFTER the flush fails too
    r = np.load(out)
    assert int(r["world"]) == 2 and str(r["backend"]) == "gloo"
    assert cases.rel_err(r["cauto"], g["cauto"][:nt]) < tol
    assert cases.rel_err(r["kic"], g["kic"][:nt]) < tol
    if with_norm:
        # norm() across ranks: each rank sums its bras against the all-gathered kets; both ranks must return
        # the norm the single-process engine computes for the whole ensemble after the same nt steps
        from tests.engine_cases import engine_potential, engine_propagator
        whole = engine_propagator(g)
        whole.run(engine_potential(g), float(g["dt"]), nt, float(g["E0"]))
        want = whole.norm()
        assert r["norms"].shape == (2,) and np.all(np.abs(r["norms"] - want) < 1e-10 * want), (r["norms"], want)
        # ... while the DEFAULT norm() stays rank-local although a process group exists (only rank 0 called it: a collective
        # would have hung): the norm of rank 0's shard with the weights it carries
        sl = D.shard_slice(g["zi"].shape[1], 0, 2)
        part = engine_propagator(g, select=sl, ntraj_total=g["zi"].shape[1])
        part.run(engine_potential(g), float(g["dt"]), nt, float(g["E0"]))
        assert abs(float(r["local_norm_rank0"]) - part.norm()) < 1e-10 * part.norm()


@pytest.mark.gpu
@pytest.mark.parametrize("prop,tol", [("HK", 1e-12), ("WM", 1e-12)])
def test_two_ranks_through_the_driver(prop, tol, tmp_path):
    """The product driver under a process group (north_star: trajectory batches shard across the GPUs, one all-reduce per
    flush): two rank processes on cuda:0 run `run_semiclassical_dynamics` on the reference's methylium example task, each
    on ITS half of every batch of the reference's sampled initial conditions; rank 0's correlations.npz must be the
    reference driver's (tests/golden/driver_methylium.npz, cli.py:321-324, 374-476)."""
    from semiclassical_amd import distributed as D
    g = cases.load("driver_methylium")
    out = str(tmp_path / "correlations.npz")
    rc = D.launch_local_ranks([os.path.join(ROOT, "tests", "_rank_driver.py"), prop, out], 2, timeout=600,
                              extra_env={"SC_DIST_BACKEND": "gloo", "SC_TEST_DEVICE": "0",
                                         "SC_TEST_NORM_EVERY": "11" if prop == "HK" else ""})
    assert rc == 0, f"a rank process failed (largest exit code {rc})"
    got = dict(np.load(out))
    assert int(got["trajectories"]) == 96 and str(got["propagator"]) == prop
    assert np.array_equal(got["times"], g[f"{prop}_times"])
    assert cases.rel_err(got["autocorrelation"], g[f"{prop}_autocorrelation"]) < tol
    assert cases.rel_err(got["ic_correlation"], g[f"{prop}_ic_correlation"]) < tol


@pytest.mark.gpu
def test_rccl_flush_through_the_c_abi():
    """sc_comm_* / sc_flush_allreduce bind librccl by themselves (no torch.distributed).  One GPU per box: a communicator
    of ONE rank -- ncclGetUniqueId, ncclCommInitRank and ncclAllReduce(ncclDouble, ncclSum) really run on the device
    and leave the sums of the only rank unchanged; the rank-local column survives."""
    from semiclassical_amd import distributed as D
    from semiclassical_amd._lib import lib
    assert lib.sc_comm_available() > 0
    dev = torch.device("cuda", 0)
    comm = D.RcclCommunicator(0, 1, dev)
    try:
        assert comm.rank_count() == (0, 1)
        slots = torch.rand((64, 5), dtype=torch.float64, device=dev)
        want = slots.clone()
        D.flush_correlations(slots, comm=comm)
        torch.cuda.synchronize(dev)
        assert torch.equal(slots, want)
        big = torch.rand(1 << 20, dtype=torch.float64, device=dev)
        keep = big.clone()
        comm.all_reduce_sum(big)
        torch.cuda.synchronize(dev)
        assert torch.equal(big, keep)
    finally:
        comm.destroy()


def test_comm_entry_points_reject_bad_arguments():
    """no GPU needed: argument checks of the flush entry points (librccl itself is present in the image)"""
    import ctypes
    from semiclassical_amd._lib import lib
    assert lib.sc_comm_available() > 0
    assert lib.sc_flush_allreduce(None, 8, None, None) != 0
    assert b"communicator" in lib.sc_last_error() or b"buffer" in lib.sc_last_error()
    handle = ctypes.c_void_p()
    assert lib.sc_comm_init(None, 2, 0, ctypes.byref(handle)) != 0
    buf = ctypes.create_string_buffer(128)
    assert lib.sc_comm_init(buf, 2, 5, ctypes.byref(handle)) != 0
    assert lib.sc_comm_destroy(None) == 0


@pytest.mark.gpu
def test_driver_flushes_through_the_c_abi_communicator(tmp_path):
    """run_semiclassical_dynamics(task, comm=RcclCommunicator): the per-batch flush goes through sc_flush_allreduce (a
    one-rank RCCL communicator on this box) and must leave the single-process result bit for bit."""
    from semiclassical_amd import distributed as D, driver
    g = cases.load("hk_as5_chi002")
    model = tmp_path / "AS_model.dat"
    rows = np.vstack((g["omega"] * 219474.63, 0.5 * g["omega"] * g["q0"] ** 2 * np.sign(g["q0"]), g["nac"],
                      np.full(5, 0.02))).T
    np.savetxt(model, rows)
    res = []
    for tag in ("plain", "rccl"):
        out = tmp_path / f"{tag}.npz"
        task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": str(model)},
                "propagator": "HK", "batch_size": 600, "num_trajectories": 1200, "num_steps": 16, "time_step_fs": 0.04,
                "results": {"correlations": str(out)}, "manual_seed": 5}
        comm = D.RcclCommunicator(0, 1, "cuda:0") if tag == "rccl" else None
        try:
            driver.run_semiclassical_dynamics(task, device="cuda:0", comm=comm)
        finally:
            if comm is not None:
                comm.destroy()
        res.append(dict(np.load(out)))
    assert int(res[1]["trajectories"]) == 1200
    assert np.array_equal(res[0]["autocorrelation"], res[1]["autocorrelation"])
    assert np.array_equal(res[0]["ic_correlation"], res[1]["ic_correlation"])


@pytest.mark.gpu
def test_torch_nccl_backend_on_the_visible_gpus(tmp_path):
    """the transport bench.py / the driver use for N > 1 -- torch.distributed's `nccl` backend, which is RCCL on ROCm -- forms
    a process group over every GPU this box shows (one per process; a single rank on the one-GPU test boxes) and
    all-reduces a slot buffer on the device"""
    from semiclassical_amd import distributed as D
    n = min(torch.cuda.device_count(), 6)
    out = tmp_path / "nccl.json"
    rc = D.launch_local_ranks([os.path.join(ROOT, "tests", "_rank_nccl.py"), str(out)], n, timeout=300)
    assert rc == 0
    r = json.loads(out.read_text())
    assert r["backend"] == "nccl" and r["world"] == n and r["ok"]


@pytest.mark.gpu
def test_driver_command_line_shares_batches_among_ranks(tmp_path):
    """`python -m semiclassical_amd.driver dynamics input.json --gpus 2`: the command starts two rank processes itself; with
    device sampling every rank draws ITS slice of each batch (same seed, first_index) -- the ensemble, and therefore the
    result file, is that of the single-process run."""
    g = cases.load("hk_as5_chi002")
    model = tmp_path / "AS_model.dat"
    rows = np.vstack((g["omega"] * 219474.63, 0.5 * g["omega"] * g["q0"] ** 2 * np.sign(g["q0"]), g["nac"],
                      np.full(5, 0.02))).T
    np.savetxt(model, rows)
    res = {}
    for gpus in (1, 2):
        out = tmp_path / f"c{gpus}.npz"
        task = {"task": "dynamics", "potential": {"type": "anharmonic AS", "model_file": str(model)},
                "propagator": "HK", "batch_size": 700, "num_trajectories": 1400, "num_steps": 14, "time_step_fs": 0.04,
                "results": {"correlations": str(out)}, "manual_seed": 11, "sampling": "device"}
        inp = tmp_path / f"in{gpus}.json"
        inp.write_text(json.dumps({"semi": [task]}))
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        env["SC_DIST_BACKEND"] = "gloo"                      # two ranks on the one GPU of the test box
        r = subprocess.run([sys.executable, "-m", "semiclassical_amd.driver", "dynamics", str(inp), "--gpus", str(gpus), "--cuda", "0"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[gpus] = dict(np.load(out))
    assert int(res[2]["trajectories"]) == 1400
    assert cases.rel_err(res[2]["autocorrelation"], res[1]["autocorrelation"]) < 1e-12
    assert cases.rel_err(res[2]["ic_correlation"], res[1]["ic_correlation"]) < 1e-12


@pytest.mark.gpu
def test_bench_flow_with_two_ranks_sharing_the_gpu():
    """bench.py's multi-rank flow end to end (barriers, max over ranks, gathered rank table, one flush) as torch.distributed.run starts
    it, rehearsed on ONE GPU (--share-gpu: gloo, both ranks on cuda:0): the two shards of one global ensemble reproduce the
    single-process correlation function of the same ensemble"""
    import json
    import subprocess
    import sys
    from semiclassical_amd import distributed as D
    common = ["--steps", "4", "--warmup", "1", "--no-configs", "--no-cpu-baseline", "--ntraj-total", "3000"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(D.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu"] + common,
                         capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and [r["trajectories"] for r in b["ranks"]] == [1500, 1500] and "REHEARSAL" in b["data"]
    assert a["config"]["trajectories_total"] == b["config"]["trajectories_total"] == 3000
    assert np.allclose(a["C_auto_last"], b["C_auto_last"], rtol=1e-12, atol=1e-14)
