"""Multi-rank tests of the distributed path.  CPU (gloo, 2 / 3 / 8 ranks): decomposition algorithm + communicator callbacks.
GPU: both ranks on cuda:0 through the host-staged test communicator vs the 1-rank HIP result."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, world=2, timeout=int(os.environ.get("BK_DIST_TIMEOUT", "300"))):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            # a rank that hangs must leave its last words in the report: kill every rank, collect what each of them printed
            for q in procs:
                q.kill()
            tails = []
            for r, q in enumerate(procs):
                try:
                    t, _ = q.communicate(timeout=20)
                except Exception:  # noqa: BLE001
                    t = ""
                tails.append(f"--- rank {r} (last output) ---\n{(t or '')[-2500:]}")
            pytest.fail(f"{mode} x{world}: no result within {timeout} s\n" + "\n".join(tails))
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert "checks OK" in o, o[-2000:]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_decomposition_gloo_cpu(world):
    """2 and 3 ranks (ragged split), and the 8 ranks of the node the scaling runs use (slabs of 3-4 planes)."""
    _run("cpu", world)


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_host_communicator():
    _run("gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 4])
def test_ragged_and_thin_slabs_host_communicator(world):
    """3 and 4 ranks on one GPU: z extents that do not divide evenly, slabs of 2-3 planes (every z-chunk is a face chunk).  The
    host-staged communicator enqueues its collectives in the stream (proxy thread) since round 4: the ranks run the RCCL ranks'
    code path -- block Arnoldi / device-resident chunks with in-stream all-reduces, halo exchange on the second stream -- with the
    defaults and with every variant switched (tests/dist_worker.py: VARIANTS; two lanes on ranks left the automatic list in round 6:
    it used to hang intermittently, see there)."""
    _run("gpu_many", world)


@pytest.mark.gpu
def test_ragged_multi_million_unknown_slabs_host_communicator():
    """128 x 128 x 129 on 2 ranks (65 + 64 planes), default options = the code path RCCL ranks run, vs 1 rank."""
    _run("gpu_ragged", 2)


@pytest.mark.gpu
def test_real_rccl_ranks_when_two_gpus_are_visible():
    """The RCCL calls themselves (ncclSend / ncclRecv groups, ncclAllReduce) with one rank per GPU.  Skipped on a
    single-GPU box -- there the same code paths run through the host-staged communicator above."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    _run("rccl", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_real_rccl_ranks_on_a_full_node(world):
    """The driver's scaling node: 4 and 8 RCCL ranks (one per GPU) -- ragged and thin slabs, the slab z-solve with 16-plane
    slabs, device-resident Arnoldi chunks with in-stream all-reduces, halo overlap on the second stream, one PALC corrector
    iteration, each against the 1-rank result; then the host-driven variants of the same paths.  Fires the moment
    >= `world` GPUs are visible; skipped on the 1-GPU box (the same checks run there over the host-staged communicator)."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs >= {world} visible GPUs")
    _run("rccl", world, timeout=1200)


@pytest.mark.gpu
def test_branch_workload_on_two_rccl_ranks_matches_one_rank():
    """BASELINE config 5 in miniature over RCCL: `bench.py --workload branch` (corrector + eigensolve + Bordered tangent per
    step, every step ONE bk_cont_step call per rank) on 64 x 64 x 64 with 2 ranks reproduces the 1-rank branch."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    args = ["--workload", "branch", "--size", "64", "--steps", "3", "--nev", "4", "--eig-tol", "1e-7"]
    one = _bench(["--gpus", "1"] + args, timeout=900)
    two = _bench(["--gpus", "2"] + args, timeout=900)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr[-1500:], two.stderr[-1500:])
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and len(a["param"]) == len(b["param"])
    assert all(abs(x - y) <= 1e-8 for x, y in zip(a["param"], b["param"])), (a["param"], b["param"])
    assert a["n_unstable"] == b["n_unstable"]


def _bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, OMP_NUM_THREADS="2", **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd="/tmp")


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launches_its_ranks_from_a_bare_shell(world):
    """`python bench.py --gpus N` without WORLD_SIZE in the environment (the form the driver uses) must start the N ranks
    itself; --dry-launch stops after every rank has reported in (no GPU here).  Exactly one JSON line on stdout."""
    import json
    r = _bench(["--gpus", str(world), "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["dry_launch"] and rec["world_size"] == world and rec["ranks_reported"] and rec["launcher"] == "self"


def test_bench_under_an_external_launcher_does_not_relaunch():
    import json
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["launcher"] == "external"


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r = _bench(["--gpus", "2", "--dry-launch"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
