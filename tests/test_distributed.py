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


def _run(mode, world=2, timeout=600):
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
            for q in procs:
                q.kill()
            raise
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
    """3 and 4 ranks on one GPU: z extents that do not divide evenly, slabs of 2-3 planes (every z-chunk is a face chunk)."""
    _run("gpu_many", world)


@pytest.mark.gpu
def test_real_rccl_ranks_when_two_gpus_are_visible():
    """The RCCL calls themselves (ncclSend / ncclRecv groups, ncclAllReduce) with one rank per GPU.  Skipped on a
    single-GPU box -- there the same code paths run through the host-staged communicator above."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    _run("rccl", 2)
