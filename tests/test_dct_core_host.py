"""CPU test: host replay of the fast-DCT phases (bifurcationkit.jl_amd/csrc/dct_core.h -- the exact index math and
butterflies the LDS kernel dct_fast.hip executes) against scipy.fft.dct / idct."""
import os
import subprocess

import numpy as np
import pytest
import scipy.fft as sfft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# "product": the layout the library is built with (dct_core.h: BK_DCT_LAYOUT 1 -- the XOR swizzle and padded twiddle tables
# the bank-conflict model prefers); "round3": BK_DCT_LAYOUT 0, the previous layout kept as the A/B reference
@pytest.fixture(scope="module", params=["product", "round3"])
def exe(request, tmp_path_factory):
    out = tmp_path_factory.mktemp("dct") / ("dct_core_check_" + request.param)
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "dct_core_check.cpp"), "-o", str(out),
           "-DBK_DCT_TWI(j)=bk::dctc::twi(j)"]
    if request.param == "round3":
        cmd += ["-DBK_DCT_LAYOUT=0"]
    subprocess.run(cmd, check=True)
    return str(out)


@pytest.mark.parametrize("N", [4, 8, 16, 32, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])        # forward / inverse, radix-2 / grouped radix-8
def test_dct_core_matches_scipy(exe, N, mode):
    rng = np.random.default_rng(N + mode)
    a, b = rng.standard_normal(N), rng.standard_normal(N)
    inp = f"{mode} {N}\n" + " ".join(map(repr, a.tolist())) + "\n" + " ".join(map(repr, b.tolist()))
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    o = np.array(list(map(float, out))).reshape(N, 2)
    f = (lambda x: sfft.idct(x, type=2, norm="ortho")) if mode & 1 else (lambda x: sfft.dct(x, type=2, norm="ortho"))
    assert np.abs(o[:, 0] - f(a)).max() < 1e-13 and np.abs(o[:, 1] - f(b)).max() < 1e-13


@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024])
@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8, 9])  # fused schedule: forward / inverse / forward-symbol-inverse; 7-9: contiguous axis
def test_dct_fused_schedule_matches_scipy(exe, N, mode):
    rng = np.random.default_rng(N + mode)
    a, b = rng.standard_normal(N), rng.standard_normal(N)
    inp = f"{mode} {N}\n" + " ".join(map(repr, a.tolist())) + "\n" + " ".join(map(repr, b.tolist()))
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    o = np.array(list(map(float, out))).reshape(N, 2)
    dct = lambda x: sfft.dct(x, type=2, norm="ortho")
    idct = lambda x: sfft.idct(x, type=2, norm="ortho")
    k = np.arange(N)
    sa, sb = 1.0 / (1.0 + 0.01 * k), 1.0 / (2.0 + 0.02 * k * k)          # the harness' test symbol
    mode = mode - 3 if mode >= 7 else mode
    if mode == 4:
        ra, rb = dct(a), dct(b)
    elif mode == 5:
        ra, rb = idct(a), idct(b)
    else:
        ra, rb = idct(sa * dct(a)), idct(sb * dct(b))
    assert np.abs(o[:, 0] - ra).max() < 1e-13 and np.abs(o[:, 1] - rb).max() < 1e-13


@pytest.mark.parametrize("N", [64, 128, 256, 512])
def test_fused_roundtrip_spectral_dot_is_the_physical_dot(exe, N):
    """The merged middle pass can return sum_k sym(k) X_k^2 per line; the transforms are orthonormal, so that is
    x . idct(sym * dct(x)) -- the r . M^-1 r that MINRES / CG need after every preconditioner application."""
    rng = np.random.default_rng(N)
    a, b = rng.standard_normal(N), rng.standard_normal(N)
    inp = f"10 {N}\n" + " ".join(map(repr, a.tolist())) + "\n" + " ".join(map(repr, b.tolist()))
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    o = np.array(list(map(float, out))).reshape(N + 1, 2)
    k = np.arange(N)
    sa, sb = 1.0 / (1.0 + 0.01 * k), 1.0 / (2.0 + 0.02 * k * k)
    dct = lambda x: sfft.dct(x, type=2, norm="ortho")
    idct = lambda x: sfft.idct(x, type=2, norm="ortho")
    za, zb = idct(sa * dct(a)), idct(sb * dct(b))
    assert np.abs(o[:N, 0] - za).max() < 1e-13 and np.abs(o[:N, 1] - zb).max() < 1e-13
    assert abs(o[N, 0] - a @ za) < 1e-13 * abs(a @ za) * N and abs(o[N, 1] - b @ zb) < 1e-13 * abs(b @ zb) * N


@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024])
def test_lane_pair_split_of_the_merged_middle_is_bitwise_the_unsplit_round_trip(exe, N):
    """Round 6 (dct_core.h: mid_half_fwd / mid_half_pairs / mid_half_inv): the merged middle of the z round trip shared by two
    lanes -- each owns one top group, the upper halves are exchanged around the (k, N - k) pairing -- performs the SAME arithmetic
    per value as fused_mid<2>: the replay is bitwise equal (modes 11 vs 6), and the spectral dot of the split (mode 12) agrees with
    the unsplit one (mode 10) to the rounding of a differently ordered sum."""
    rng = np.random.default_rng(3 * N)
    a, b = rng.standard_normal(N), rng.standard_normal(N)
    body = " ".join(map(repr, a.tolist())) + "\n" + " ".join(map(repr, b.tolist()))
    run = lambda mode: np.array(list(map(float, subprocess.run([exe], input=f"{mode} {N}\n" + body, capture_output=True, text=True,
                                                                check=True).stdout.split())))
    o6, o11, o10, o12 = run(6), run(11), run(10), run(12)
    assert np.array_equal(o6, o11)
    assert np.array_equal(o10[:2 * N], o12[:2 * N])
    assert np.abs(o10[2 * N:] - o12[2 * N:]).max() <= 1e-14 * np.abs(o10[2 * N:]).max()
