"""CPU test of the launch planning the HIP launchers share (bifurcationkit.jl_amd/csrc/launch_plan.h): the z-chunk length of
the streaming Swift-Hohenberg kernel, replayed on the host against the measured optimum and against its own contract."""
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plan(tmp_path_factory):
    exe = tmp_path_factory.mktemp("plan") / "launch_plan_check"
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "launch_plan_check.cpp"), "-o", str(exe)],
                   check=True)

    def run(cases):
        inp = "\n".join(f"{nz} {tiles} {res} {int(split)}" for nz, tiles, res, split in cases) + "\n"
        out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.split()
        return [int(x) for x in out]
    return run


def test_measured_optima_at_512_cubed_and_its_slabs(plan):
    """profiles/r2_jvp_zchunk_sweep_512.jsonl: three chunks per tile column (768 workgroups = one round of the 3 resident
    workgroups on each of the 256 CUs) are the fastest at the full grid and at the slabs of 2 / 4 / 8 ranks."""
    res = 3 * 256
    got = plan([(512, 256, res, False), (256, 256, res, False), (128, 256, res, False), (64, 256, res, False)])
    assert got == [171, 86, 43, 22]


def test_contract_over_many_shapes(plan):
    cases = [(nz, tiles, res, split) for nz in (1, 2, 5, 8, 12, 16, 22, 31, 32, 64, 100, 151, 256, 512, 1000)
             for tiles in (1, 4, 64, 256, 1024) for res in (96, 768) for split in (False, True)]
    for (nz, tiles, res, split), zc in zip(cases, plan(cases)):
        assert 1 <= zc <= nz
        nzc = math.ceil(nz / zc)
        assert nzc <= 64
        assert zc >= min(8, nz) or nzc == 1, (nz, tiles, res, split, zc)
        # never worse than one chunk per column or the 16-plane chunks of rounds 1-2 under the plan's own cost
        def cost(z):
            c = math.ceil(nz / z)
            if split and c >= 3:
                return (math.ceil((c - 2) * tiles / res) + math.ceil(2 * tiles / res)) * (z + 4)
            return math.ceil(c * tiles / res) * (z + 4)
        assert cost(zc) <= cost(nz)
        if nz >= 16 and math.ceil(nz / 16) <= 64:
            assert cost(zc) <= cost(16)


def test_overlapped_halo_path_keeps_interior_chunks(plan):
    """With the halo exchange overlapped the two face chunks wait for the exchange: on an 8-rank slab of 512^3 the plan leaves
    interior chunks to compute meanwhile."""
    zc, = plan([(64, 256, 768, True)])
    assert math.ceil(64 / zc) >= 3
