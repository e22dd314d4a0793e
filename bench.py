#!/usr/bin/env python
"""bench.py -- Newton-Krylov PALC corrector steps/s on the 3-D Swift-Hohenberg grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or -- when WORLD_SIZE is not
set -- bench.py re-executes itself through torch.distributed.run on 127.0.0.1 with a free port, so the bare
`python bench.py --gpus 8` works as well.  Rank 0 prints the ONE JSON line either way.  `--dry-launch` only proves the
launch path (every rank joins a gloo group, rank 0 prints a JSON record; no GPU needed).

Workload (BASELINE.json configs[4]/[3]): SH3d on an n^3 grid (default 512^3, fp64, 1 GiB per
vector), l = 0.1, nu = 1.2 (examples/SH3d.jl:86), GMRES(30) rtol 1e-9 (SH3d.jl:93) with the exact spectral
preconditioner Pl = (L1 + I)^-1 (`lu(L1 + I)`, examples/SH2d-fronts.jl:121; the `+ 1` of the reference's own
512^2 GPU example, examples/SH2d-fronts-cuda.jl:63), PALC with theta = 0.5, ds = -0.001,
BorderingBLS(check_precision = false) (SH3d.jl:160-163).

Synthetic state with a checkable answer.  The branch point is the hexagon pattern of the reference's examples
(`sol_hexa`, examples/SH3d.jl:127; guess cos x + cos(x/2) cos(sqrt(3) y/2), examples/SH2d-fronts.jl:47,
SH2d-fronts-cuda.jl:71) as z-invariant hexagonal prisms: Newton-converge it on ONE periodic cell
[-2pi, 2pi) x [-2pi/sqrt3, 2pi/sqrt3) x [-pi, pi) with 64 x 32 x 32 points (h = 0.196 in x -- the grid spacing of
examples/SH2d-fronts-cuda.jl:66-69), then tile it by even reflections to 8 x 16 x 16 cells = 512^3.  Because the
Neumann-ghost boundary rule IS an even reflection, the tiled field is an exact discrete solution on the big grid, and
every quantity of the big corrector (residual history, p) must reproduce the single-cell run -- a size-independent
parity check against the CPU oracle at full size (tests/test_gpu_fullsize.py).  The state is linearly stable (largest
cell eigenvalue -0.17), so the preconditioned operator is well conditioned and the GMRES iteration count does not
depend on rounding noise (the unstable square pattern cos x cos y of SH3d.jl:77 tiled the same way needs 47-146
operator applications per step depending on the build: its Krylov spaces are polluted by growing asymmetric modes).
The kernels do not know about the symmetry: every byte of every 1 GiB vector is streamed.

One "step" = one pass through the corrector loop src/continuation/Palc.jl:237-295 from the PALC predictor:
finite-difference dF/dp (1 residual), Jacobian handle, bordered solve (2 preconditioned GMRES solves to
rtol 1e-9), update, residual, norms.  Every timed step repeats that first corrector iteration from the same
predictor, so the work per step is identical and nothing is cached between steps.  Inputs are resident in HBM
before the timed region; synthetic data (the reference's own initial guess, Newton-converged on the device).

Prints ONE JSON line (rank 0).  `roofline` is for the kernel with the largest share of the timed region,
measured with HIP events on the library's stream (bk_prof_*); `cpu_baseline` times the NumPy/SciPy oracle
(reference formulation: assembled sparse L1, MGS2 GMRES) on a bounded sample on this host.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / hipIpcGetMemHandle fail with the legacy mode); the
# launcher normally exports it -- keep it for a bare `torchrun bench.py` as well (must be set before the HIP runtime loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="grid points per axis (BASELINE metric: 512)")
    ap.add_argument("--size-z", type=int, default=0, help="z extent if different from --size (cost-model runs: the z-slab one of "
                                                          "8 ranks owns, e.g. --size 512 --size-z 64 --opt dct_slab_emulate=8)")
    ap.add_argument("--shift", type=float, default=1.0, help="preconditioner (L1 + shift I)^-1")
    ap.add_argument("--cpu-sample", type=int, default=3, help="CPU-baseline sample: tiles per axis of the cell (0: skip)")
    ap.add_argument("--no-precond", action="store_true")
    ap.add_argument("--sh-kernel", type=int, default=1)
    ap.add_argument("--dgks-eta", type=float, default=None)
    ap.add_argument("--opt", action="append", default=[], help="library tuning option key=value (experiments)")
    ap.add_argument("--trace", action="store_true", help="after the timed region, repeat the step once with the solver trace on and "
                                                         "report the residual history of its linear solves (config.trace)")
    ap.add_argument("--workload", default="corrector", choices=["corrector", "branch"],
                    help="corrector: the BASELINE metric (PALC corrector steps/s); branch: BASELINE config 5 -- --steps native "
                         "continuation steps (corrector + 15 eigenvalues + Bordered tangent + predictor per step)")
    ap.add_argument("--ls-maxiter", type=int, default=150, help="restart cycles of GMRES(30) (SH3d.jl:93: 150)")
    ap.add_argument("--no-full", action="store_true", help="skip the run-to-convergence correctors of the setup (big grid and cell): the "
                                                           "shift-0 pairing on the tiled domain does not converge within any sensible budget")
    ap.add_argument("--block-log", action="store_true", help="after the timed region, repeat the step once with the block log on and "
                                                             "report every Arnoldi block (config.block_log)")
    ap.add_argument("--no-fixed", action="store_true", help="skip the fixed_input record (the step from the committed cell states)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state record (corrector of a running branch)")
    ap.add_argument("--nev", type=int, default=15)
    ap.add_argument("--eig-tol", type=float, default=1e-8)
    ap.add_argument("--eig-sigma", type=float, default=0.1, help="branch workload: shift of the shift-invert eigensolver (SH3d.jl: 0.1)")
    ap.add_argument("--eig-dim", type=int, default=0, help="Krylov dimension of the eigensolver (0: max(30, nev + 30), examples/SH3d.jl:109)")
    ap.add_argument("--eig-inner-rtol", type=float, default=1e-9, help="branch workload: rtol of the eigensolver's inner solves (SH3d.jl:115: 1e-9)")
    ap.add_argument("--eig-thick", type=int, default=1, help="branch workload: eigensolve starts from the previous step's Ritz vectors")
    ap.add_argument("--eig-inner-dim", type=int, default=30, help="branch workload, --eig-inner gmres: Krylov dimension of the inner GMRES")
    ap.add_argument("--eig-inner", default="minres", choices=["gmres", "minres"], help="branch workload: inner solver of the shift-invert eigensolver")
    ap.add_argument("--dry-launch", action="store_true", help="launch-path check only: every rank joins a gloo group and reports "
                                                              "in, rank 0 prints a JSON record (no GPU work)")
    ap.add_argument("--linsolver", default="gmres", choices=["gmres", "minres"],
                    help="gmres: GMRESKrylovKit(30), the reference example's solver (the headline workload); minres: "
                         "KrylovLS(KrylovAlg = :minres), valid because the SH Jacobian is symmetric (experiment)")
    return ap.parse_args()


CELL = (64, 32, 32)                                            # points per cell (x, y, z)
CELL_L = (2.0 * math.pi, 2.0 * math.pi / math.sqrt(3.0), math.pi)     # half-widths of the cell


def tiles_for(n, nz=0):
    dims = (n, n, nz or n)
    if any(d % c for d, c in zip(dims, CELL)):
        raise SystemExit(f"--size / --size-z must be multiples of {CELL}")
    return tuple(d // c for d, c in zip(dims, CELL))


def tile_cell(cell_vec, tiles, slab, device):
    """Even-reflection tiling of a cell field (flat, x fastest) to this rank's z-slab of the big grid."""
    import torch
    cx, cy, cz = CELL
    def idx(nc, T):
        return torch.cat([torch.arange(nc) if c % 2 == 0 else torch.arange(nc - 1, -1, -1) for c in range(T)]).to(device)
    c = cell_vec.reshape(cz, cy, cx)                      # [z, y, x]
    lo, hi = slab
    out = c[idx(cz, tiles[2])[lo:hi]][:, idx(cy, tiles[1])][:, :, idx(cx, tiles[0])]
    return out.contiguous().reshape(-1)


def hex_guess_np():
    """0.5 (cos x + cos(x/2) cos(sqrt(3) y/2)), constant in z, on the cell grid (x fastest)."""
    import numpy as np
    ax = [-l + 2.0 * l / n * np.arange(n) for n, l in zip(CELL, CELL_L)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    s = 0.5 * (np.cos(X) + np.cos(X / 2.0) * np.cos(math.sqrt(3.0) * Y / 2.0)) + 0.0 * Z
    return np.ascontiguousarray(s.reshape(-1, order="F"))


def cell_branch_points(ctx_cell, hip, shift, ds, eta=150.0):
    """Two Newton-converged points of the hexagon branch on the cell (device)."""
    prob = hip.SwiftHohenberg(ctx_cell, CELL, CELL_L, l=0.1, nu=1.2)
    P = hip.DCTPreconditioner(prob, shift)
    ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=150, Pl=P)
    u0 = prob.vec(hex_guess_np())
    s0 = hip.newton_native(prob, u0, 0.1, ls, tol=1e-10, max_iterations=40, norm_inf=True)
    s1 = hip.newton_native(prob, s0["u"], 0.1 + ds / eta, ls, tol=1e-10, max_iterations=20, norm_inf=True)
    return prob, ls, s0, s1


def cpu_baseline(tiles, shift, steps=1):
    """Time the oracle's corrector step (reference formulation: assembled sparse L1, MGS2 GMRES) on a tiles[0] x
    tiles[1] x tiles[2] tiling of the cell, single thread."""
    import numpy as np
    from oracle import bordered, krylov, operators, palc
    ds = -0.001
    shc = operators.SwiftHohenberg(CELL, CELL_L)
    pc = palc.Problem(lambda x, p: shc.F(x, p, 1.2), lambda x, p: (lambda dx: shc.dF(x, p, 1.2, dx)))
    Plc = operators.dct_preconditioner(CELL, CELL_L, shift)
    cls_s = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9,
                                                                 atol=1e-12, Pl=Plc)[:3]
    c0 = palc.newton(pc, hex_guess_np(), 0.1, cls_s, tol=1e-10, max_iterations=40, normN=palc.norminf)
    c1 = palc.newton(pc, c0["u"], 0.1 + ds / 150.0, cls_s, tol=1e-10, max_iterations=20, normN=palc.norminf)
    idx = [np.concatenate([np.arange(nc) if c % 2 == 0 else np.arange(nc)[::-1] for c in range(T)])
           for nc, T in zip(CELL, tiles)]
    cx, cy, cz = CELL
    tile = lambda v: np.ascontiguousarray(v.reshape(cz, cy, cx)[np.ix_(idx[2], idx[1], idx[0])]).reshape(-1)
    dims = tuple(c * t for c, t in zip(CELL, tiles))
    ls = tuple(l * t for l, t in zip(CELL_L, tiles))
    sh = operators.SwiftHohenberg(dims, ls)
    Pl = operators.dct_preconditioner(dims, ls, shift)
    ols = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9,
                                                               atol=1e-12, Pl=Pl)[:3]
    prob = palc.Problem(lambda x, p: sh.F(x, p, 1.2), lambda x, p: (lambda dx: sh.dF(x, p, 1.2, dx)))
    z0, z1 = (tile(c0["u"]), 0.1), (tile(c1["u"]), 0.1 + ds / 150.0)
    tau = palc.secant_tangent(z1, z0, ds, 0.5)
    zp = palc.add_tangent(z0, tau, ds)
    bls = lambda *a, **k: bordered.bordering_bls(ols, *a, check_precision=False, **k)
    t0 = time.perf_counter()
    itl = 0
    for _ in range(steps):
        so = palc.newton_palc(prob, z0, tau, zp, ds, 0.5, bls, tol=0.0, max_iterations=1, normN=palc.norminf)
        itl = so["itlineartot"]
    dt = (time.perf_counter() - t0) / steps
    return dict(seconds_per_step=dt, n=sh.N, dims=dims, itlinear=itl, residuals=so["residuals"],
                cell_newton=(c0["converged"], c0["itnewton"]), umax=float(np.abs(c0["u"]).max()))


def cpu_baseline_cpp(tiles, shift):
    """Time oracle/cpu_ref.cpp -- the C++/OpenMP restatement of the reference's CSR formulation (assembled L1 = A*A,
    SpMV, MGS2 GMRES(30), Bordering) -- on all host cores, on a tiling of the cell.  The cell solutions come from the
    NumPy oracle; the binary is rebuilt with -march=native for the machine it runs on (falls back to the portable
    build of oracle/Makefile)."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    from oracle import krylov, operators, palc
    odir = os.path.join(ROOT, "oracle")
    exe = os.path.join(odir, "_build", "cpu_ref_native")
    try:
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", os.path.join(odir, "cpu_ref.cpp"), "-o", exe],
                       check=True, capture_output=True, timeout=300)
    except Exception:
        exe = os.path.join(odir, "_build", "cpu_ref")
        if not os.path.exists(exe):
            raise RuntimeError("oracle/_build/cpu_ref missing (run __graft_entry__.build())")
    ds = -0.001
    shc = operators.SwiftHohenberg(CELL, CELL_L)
    pc = palc.Problem(lambda x, p: shc.F(x, p, 1.2), lambda x, p: (lambda dx: shc.dF(x, p, 1.2, dx)))
    Plc = operators.dct_preconditioner(CELL, CELL_L, shift)
    cls_s = lambda J, r, a0=0.0, a1=1.0: krylov.gmres_krylovkit(J, r, a0, a1, krylovdim=30, maxiter=150, rtol=1e-9,
                                                                 atol=1e-12, Pl=Plc)[:3]
    c0 = palc.newton(pc, hex_guess_np(), 0.1, cls_s, tol=1e-10, max_iterations=40, normN=palc.norminf)
    c1 = palc.newton(pc, c0["u"], 0.1 + ds / 150.0, cls_s, tol=1e-10, max_iterations=20, normN=palc.norminf)
    idx = [np.concatenate([np.arange(nc) if c % 2 == 0 else np.arange(nc)[::-1] for c in range(T)])
           for nc, T in zip(CELL, tiles)]
    cx, cy, cz = CELL
    tile = lambda v: np.ascontiguousarray(v.reshape(cz, cy, cx)[np.ix_(idx[2], idx[1], idx[0])]).reshape(-1)
    dims = tuple(c * t for c, t in zip(CELL, tiles))
    ls = tuple(l * t for l, t in zip(CELL_L, tiles))
    tmp = tempfile.mkdtemp(prefix="bk_cpu_ref_")
    try:
        f0, f1 = os.path.join(tmp, "u0.bin"), os.path.join(tmp, "u1.bin")
        tile(c0["u"]).tofile(f0)
        tile(c1["u"]).tofile(f1)
        cmd = [exe, *map(str, dims), *map(repr, ls), "0.1", "1.2", repr(float(shift)), repr(ds), "0.5", f0, "0.1", f1,
               repr(0.1 + ds / 150.0)]
        best = None
        for nt in thread_counts():                # keep the fastest: more threads than memory channels can lose
            env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PROC_BIND="spread", OMP_PLACES="cores")
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, check=True)
            o = json.loads(r.stdout.strip().splitlines()[-1])
            if best is None or o["seconds_per_step"] < best["seconds_per_step"]:
                best = o
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    best["dims"] = dims
    return best


def available_cpus():
    """CPUs this process may use: affinity mask, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def thread_counts():
    n = available_cpus()
    cand = sorted({min(n, c) for c in (8, 16, 32, 64, n)})
    return cand


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment): start the N ranks through
    torch.distributed.run on the loopback address with a free port and hand their output through.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, available_cpus() // max(args.gpus, 1))))
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args, rank, world):
    """Launch-path check: all ranks meet in a gloo group; rank 0 prints what it saw."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        seen = int(round(float(t.item())))
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": args.gpus, "world_size": world,
                          "rank_sum": seen, "ranks_reported": seen == world * (world + 1) // 2,
                          "launcher": "external" if os.environ.get("BK_BENCH_SELF_LAUNCHED") != "1" else "self"}))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.environ["BK_BENCH_SELF_LAUNCHED"] = "1"
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}), or unset WORLD_SIZE to let bench.py start them")
    if args.dry_launch:
        dry_launch(args, rank, world)
        return
    import torch
    import torch.distributed as dist

    if world > 1 and os.environ.get("BK_BENCH_HOSTCOMM") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         f"(BK_BENCH_HOSTCOMM=1 shares GPU 0 between the ranks over a host-staged test communicator)")
    if os.environ.get("BK_BENCH_HOSTCOMM") == "1":
        local = 0                                   # test mode: all ranks share GPU 0
    if world > 1:
        # a collective that never completes (a rank died, a link is down) must end the run with a traceback instead of
        # hanging the node: every rank exits after BK_BENCH_WATCHDOG_S seconds (default 15 min; 0 disables)
        import faulthandler
        wd = int(os.environ.get("BK_BENCH_WATCHDOG_S", "900"))
        if wd > 0:
            faulthandler.dump_traceback_later(wd, exit=True)
    torch.cuda.set_device(local)
    from bk_amd import hip

    comm = None
    hostcomm_mode = os.environ.get("BK_BENCH_HOSTCOMM") == "1"     # test mode: ranks share GPU 0 over gloo + host staging
    if hostcomm_mode:
        from bk_amd import hostcomm
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        comm = hostcomm.comm_tuple()
    elif world > 1 or os.environ.get("BK_FORCE_DIST") == "1":       # BK_FORCE_DIST: exercise the RCCL bootstrap with 1 rank
        if world == 1 and "MASTER_ADDR" not in os.environ:           # ... also from a bare shell
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]))
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt = torch.tensor(list(hip.Context.unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(idt, 0)
        comm = ("rccl", rank, world, bytes(idt.cpu().tolist()))
    ctx = hip.Context(local, comm)
    ctx.set_option("sh_kernel", args.sh_kernel)
    if args.dgks_eta is not None:
        ctx.set_option("dgks_eta", args.dgks_eta)
    n = args.size
    nzz = args.size_z or n
    tiles = tiles_for(n, nzz)
    big_l = tuple(l * t for l, t in zip(CELL_L, tiles))
    ds, theta = -0.001, 0.5
    # ---- setup (untimed)
    t_setup = time.perf_counter()
    ctx_cell = ctx if world == 1 else hip.Context(local)
    cprob, cls_, c0, c1 = cell_branch_points(ctx_cell, hip, args.shift, ds)
    # experiment options apply from here on: the two cell solutions above -- the INPUT of the timed step, whose secant amplifies
    # their last digits to ~1e-6 of the tangent -- are the same for every option set, so that an A/B compares the same step
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, float(v_))
    prob = hip.SwiftHohenberg(ctx, (n, n, nzz), big_l, l=0.1, nu=1.2)
    P = None if args.no_precond else hip.DCTPreconditioner(prob, args.shift)
    ls = hip.GMRESKrylovKit(dim=30, rtol=1e-9, atol=1e-12, maxiter=args.ls_maxiter, Pl=P)       # SH3d.jl:93
    if args.linsolver == "minres":
        ls = hip.KrylovLSSymmetric("minres", rtol=1e-9, atol=1e-12, itmax=4000, Pl=P)
    bls = hip.BorderingBLS(ls, check_precision=False)                               # SH3d.jl:163
    B = hip.BorderedArray

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    p0, p1 = 0.1, 0.1 + ds / 150.0
    u0 = hip.HipVec(ctx, tile_cell(c0["u"].t, tiles, prob.slab, ctx.torch_device), prob.nglobal)
    u1 = hip.HipVec(ctx, tile_cell(c1["u"].t, tiles, prob.slab, ctx.torch_device), prob.nglobal)
    res0 = prob.residual(u0, p0).norminf()           # the tiled field is an exact discrete solution
    res1 = prob.residual(u1, p1).norminf()
    z0, z1 = B(u0, p0), B(u1, p1)
    tau = z1.copy().add_(z0, -1.0)
    nrm = math.sqrt(tau.u.inner(tau.u) / prob.nglobal * theta + tau.p * tau.p * (1 - theta))
    tau.scale_(math.copysign(1.0, ds) / nrm)                                        # Secant tangent, Tangents.jl:28-42
    z_pred = z0.copy().add_(tau, ds)
    none_ = {"converged": None, "itnewton": None, "itlineartot": None, "residuals": [], "u": B(u0, float("nan"))}
    full = none_ if args.no_full else hip.newton_palc_native(prob, z0, tau, z_pred, ds, theta, bls, tol=1e-9, max_iterations=15,
                                                            p_min=-0.1, p_max=0.15, norm_inf=True)
    # the same corrector on the single cell: must give the same trajectory
    cb_ = hip.BorderedArray
    cz0, cz1 = cb_(c0["u"], p0), cb_(c1["u"], p1)
    ctau = cz1.copy().add_(cz0, -1.0)
    cn = math.sqrt(ctau.u.inner(ctau.u) / cprob.nglobal * theta + ctau.p * ctau.p * (1 - theta))
    ctau.scale_(math.copysign(1.0, ds) / cn)
    cfull = none_ if args.no_full else hip.newton_palc_native(cprob, cz0, ctau, cz0.copy().add_(ctau, ds), ds, theta,
                                                             hip.BorderingBLS(cls_, check_precision=False), tol=1e-9, max_iterations=15,
                                                             p_min=-0.1, p_max=0.15, norm_inf=True)
    barrier()
    t_setup = time.perf_counter() - t_setup

    from bk_amd import continuation as Cn

    def branch_setup(eig):
        nopt = Cn.NewtonPar(tol=1e-9, max_iterations=15, linsolver=ls, eigsolver=eig)                # SH3d.jl:160
        cp = Cn.ContinuationPar(ds=ds, dsmin=1e-4, dsmax=0.005, p_min=-0.1, p_max=0.15, max_steps=1, nev=args.nev,
                                detect_bifurcation=3 if eig is not None else 0, newton_options=nopt)
        alg = Cn.PALC(tangent="bordered", theta=theta, bls=hip.BorderingBLS(None, check_precision=False))   # SH3d.jl:163
        return cp, alg

    if args.workload == "branch":
        run_branch_workload(args, ctx, hip, Cn, prob, P, ls, u0, branch_setup, barrier, rank, world, n, tiles, t_setup)
        if dist.is_initialized():
            dist.destroy_process_group()
        ctx.close()
        return

    def one_step():
        return hip.newton_palc_native(prob, z0, tau, z_pred, ds, theta, bls, tol=0.0, max_iterations=1,
                                      p_min=-0.1, p_max=0.15, norm_inf=True)

    for _ in range(args.warmup):
        last = one_step()
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one_step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if hostcomm_mode else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- fixed input (VERDICT r5 Next 7): the SAME step from the committed cell solutions tests/golden/bench_cell_states.npz
    # (CPU oracle, generated once: scripts/gen_bench_cell_states.py) -- its operator-application count does not depend on the last
    # digits of this build's own cell Newton solves, so two builds can be compared by it
    fixed = None
    fx = os.path.join(ROOT, "tests", "golden", "bench_cell_states.npz")
    if os.path.exists(fx) and not args.no_fixed and args.shift == 1.0:
        import numpy as np
        d = np.load(fx)
        if tuple(int(v) for v in d["cell"]) == CELL:
            dev = ctx.torch_device
            tl_ = lambda a: hip.HipVec(ctx, tile_cell(torch.from_numpy(np.ascontiguousarray(a)).to(dev), tiles, prob.slab, dev), prob.nglobal)
            fz0, fz1 = B(tl_(d["u0"]), float(d["p0"])), B(tl_(d["u1"]), float(d["p1"]))
            ftau = fz1.copy().add_(fz0, -1.0)
            fn_ = math.sqrt(ftau.u.inner(ftau.u) / prob.nglobal * theta + ftau.p * ftau.p * (1 - theta))
            ftau.scale_(math.copysign(1.0, ds) / fn_)
            fpred = fz0.copy().add_(ftau, ds)
            fstep = lambda: hip.newton_palc_native(prob, fz0, ftau, fpred, ds, theta, bls, tol=0.0, max_iterations=1,
                                                   p_min=-0.1, p_max=0.15, norm_inf=True)
            for _ in range(min(args.warmup, 2)):
                fl = fstep()
            barrier()
            tf0 = time.perf_counter()
            nf = max(1, min(args.steps, 10))
            for _ in range(nf):
                fl = fstep()
            barrier()
            fms = (time.perf_counter() - tf0) / nf * 1e3
            fixed = {"what": "the same corrector pass started from the committed cell solutions tests/golden/bench_cell_states.npz "
                             "(CPU oracle, scripts/gen_bench_cell_states.py) instead of this build's own cell Newton solves",
                     "steps": nf, "ms_per_step": fms, "steps_per_s": 1e3 / fms, "itlinear": fl["itlineartot"],
                     "ms_per_operator_application": fms / max(fl["itlineartot"], 1), "residuals": fl["residuals"], "p": fl["u"].p}
            del fz0, fz1, ftau, fpred

    # ---- block log: the timed step once more with the library's block log on (bk_solver_block_log): per Arnoldi block the steps
    # issued / accepted, the last pivot ratio (conditioning margin against the truncation threshold 1e-8) and the Newton shifts
    block_log = None
    if args.block_log:
        ctx.set_option("gmres_block_log", 1)
        ctx.solver_block_log()
        bl = one_step()
        recs = ctx.solver_block_log()
        ctx.set_option("gmres_block_log", 0)
        clean = lambda r: {k_: (None if isinstance(v_, float) and v_ != v_ else v_) for k_, v_ in r.items()}
        first = [r for r in recs if r["j"] == 0]
        block_log = {"itlinear": bl["itlineartot"], "blocks": [clean(r) for r in recs],
                     "first_block_last_pivot_ratio": [r["last_pivot_ratio"] for r in first],
                     "min_last_pivot_ratio": min((r["last_pivot_ratio"] for r in recs if r["got"] > 0), default=None),
                     "truncated_blocks": sum(1 for r in recs if r["got"] < r["steps"])}

    trace_rec = None
    if args.trace:
        ctx.set_option("solver_trace", 1)
        ctx.solver_history()
        tl = one_step()
        trace_rec = {"itlinear": tl["itlineartot"], "histories": ctx.solver_history()}
        ctx.set_option("solver_trace", 0)

    # ---- steady state: the corrector of a RUNNING branch (state, Bordered tangent and step size after two native
    # continuation steps), run to convergence -- next to the first-corrector headline (Weak 2)
    steady = None
    if not args.no_steady:
        cp, alg = branch_setup(None)
        cp.max_steps = 2
        grab = {}
        Cn.continuation_native(prob, u0, p0, alg, cp, normC=Cn.norminf,
                               finalise_solution=lambda get, r: grab.update(get(), step=r.step, itlinear=r.itlinear) or True)
        zs, ts, dss = grab["z"], grab["tau"], grab["ds"]
        zps = zs.copy().add_(ts, dss)
        run_s = lambda: hip.newton_palc_native(prob, zs, ts, zps, dss, theta, bls, tol=1e-9, max_iterations=15,
                                               p_min=-0.1, p_max=0.15, norm_inf=True)
        run_s()
        barrier()
        ts0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ss = run_s()
        barrier()
        dts = (time.perf_counter() - ts0) / reps
        steady = {"what": "corrector of continuation step 3 (state / Bordered tangent / ds after two native bk_cont_step calls), "
                          "run to convergence (tol 1e-9)", "ms_per_corrector": dts * 1e3, "steps_per_s": 1.0 / dts,
                  "converged": ss["converged"], "itnewton": ss["itnewton"], "itlinear": ss["itlineartot"],
                  "ms_per_operator_application": dts * 1e3 / max(ss["itlineartot"], 1), "ds": dss, "p": zs.p,
                  "itlinear_of_native_step_2": grab["itlinear"], "residuals": ss["residuals"]}

    kernels = {}
    for name in ("jvp", "residual", "multidot", "multiaxpy", "dct_pass", "blas1", "combine", "transpose", "alltoall",
                 "halo"):
        e = ctx.prof_get(name)
        if e["calls"]:
            kernels[name] = dict(ms_total=e["ms"], calls=e["calls"], avg_ms=e["ms"] / e["calls"],
                                 alg_gb_per_call=e["bytes"] / e["calls"] / 1e9,
                                 gbs=e["bytes"] / max(e["ms"], 1e-9) / 1e6)
    compute = {k: v for k, v in kernels.items() if k not in ("alltoall", "halo")}
    # the JVP + GMRES inner loop as a whole (north-star target: >= 60 % of peak HBM): algorithmic bytes of every profiled
    # kernel of the timed steps over the sum of their durations
    inner = None
    if compute:
        gb = sum(v["alg_gb_per_call"] * v["calls"] for v in compute.values())
        ms = sum(v["ms_total"] for v in compute.values())
        inner = dict(alg_gb=gb, kernel_ms=ms, gbs=gb / max(ms, 1e-9) * 1e3, frac_of_peak=gb / max(ms, 1e-9) * 1e3 / HBM_PEAK_GBS,
                     kernel_time_share_of_wall=ms / max(dt * 1e3, 1e-9))
    dom = max(compute, key=lambda k: compute[k]["ms_total"]) if compute else None
    # what a reader of a multi-GPU number needs to believe it: the ranks the communicator itself reports (ncclCommCount), the
    # measured cost of the two collectives of the hot path on THIS node, and their share of the timed region
    comm_rec = None
    if world > 1:
        kind, crank, cranks = ctx.comm_info()
        ar_us = ctx.comm_probe("allreduce", 32, 50)                 # the projections of one Arnoldi step
        halo_cnt = 2 * n * n                                        # 2 planes per side
        halo_us = ctx.comm_probe("halo", halo_cnt, 20)
        nb = 2 if 0 < rank < world - 1 else 1
        wall_ms = dt * 1e3
        comm_rec = {"backend": kind, "ranks_in_communicator": cranks, "allreduce_32_doubles_us": ar_us,
                    "halo_exchange_us": halo_us, "halo_bytes_per_neighbour": 8 * halo_cnt,
                    "halo_gbs_per_direction": 8.0 * halo_cnt / max(halo_us, 1e-9) / 1e3, "halo_neighbours_rank0": nb,
                    "share_of_wall": {k_: kernels[k_]["ms_total"] / max(wall_ms, 1e-9) for k_ in ("halo", "alltoall", "transpose")
                                      if k_ in kernels},
                    "note": "halo / alltoall shares are event-timed spans on rank 0's streams (the halo exchange runs on its own "
                            "stream under the interior z-chunks of the JVP)",
                    # the first real multi-GPU line proves itself: the distributed corrector on the tiled grid against the SAME corrector
                    # on the one cell, solved by this rank alone on a single-rank context (no communicator) -- the tiling property makes
                    # them the same trajectory (tests/test_gpu_fullsize.py), so p must agree to the solver tolerance and the
                    # operator-application counts to the rounding-noise wobble of the last Arnoldi steps (+-2 per solve)
                    "parity_vs_1rank": None if args.no_full else {
                        "p_distributed": full["u"].p, "p_cell_on_one_rank": cfull["u"].p, "abs_diff_p": abs(full["u"].p - cfull["u"].p),
                        "itlinear_distributed": full["itlineartot"], "itlinear_cell_on_one_rank": cfull["itlineartot"],
                        "itnewton": [full["itnewton"], cfull["itnewton"]],
                        "residuals_distributed": full["residuals"], "residuals_cell_on_one_rank": cfull["residuals"],
                        "ok": bool(full["converged"] and cfull["converged"] and abs(full["u"].p - cfull["u"].p) <= 1e-9 and
                                   abs(full["itlineartot"] - cfull["itlineartot"]) <= 4 * max(full["itnewton"], 1))}}
    roofline = None
    if dom:
        k = kernels[dom]
        # PMC traffic comes from a separate rocprofv3 pass (scripts/gpu_profile.sh -> profiles/rN_pmc_hbm_traffic.json); it is
        # only reported when that pass was taken on EXACTLY the kernel sources running now (fingerprint), else null
        traffic, tsrc = None, None
        import glob
        import hashlib
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
        pmc_file = cands[-1] if cands else None
        pmc_name = {"dct_pass": "dct_pass (all dct_f* kernels)", "jvp": "sh_stream_kernel<true, true>", "blas1": "axpbyz_kernel<2, true>"}.get(dom)
        if world == 1 and n == 512 and nzz == 512 and pmc_file:
            pmc = json.load(open(pmc_file))
            hsh = hashlib.sha256()
            cdir = os.path.join(ROOT, "bifurcationkit.jl_amd", "csrc")
            for f_ in sorted(os.listdir(cdir)):
                if f_.endswith((".hip", ".h")):
                    hsh.update(open(os.path.join(cdir, f_), "rb").read())
            rel = os.path.relpath(pmc_file, ROOT)
            if pmc.get("_meta", {}).get("sources_sha") != hsh.hexdigest()[:16]:
                tsrc = f"{rel} is stale (taken on other kernel sources): re-run scripts/gpu_profile.sh"
            elif pmc_name in pmc:
                traffic = pmc[pmc_name]["read_bytes"] + pmc[pmc_name]["write_bytes"]
                tsrc = f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 on gfx950; same kernel sources)"
            elif dom in ("multiaxpy", "multidot"):
                # a family of <KB> instantiations with k + 2 (k + 1) streams each: launch-weighted mean over the PMC pass's own
                # launches (the same step, hence the same distribution of k as the timed steps)
                fam = [v for k_, v in pmc.items() if k_.startswith((dom + "_kernel<", dom + "_c_kernel<")) and v.get("n")]
                if fam:
                    traffic = sum(v["n"] * (v["read_bytes"] + v["write_bytes"]) for v in fam) / sum(v["n"] for v in fam)
                    tsrc = (f"{rel} (launch-weighted mean over the {dom}[_c]_kernel<KB> instantiations; rocprofv3 --pmc FETCH_SIZE / "
                            f"WRITE_SIZE in separate passes, FETCH x2 on gfx950; same kernel sources)")
        roofline = dict(kernel=dom, bound="hbm", achieved=k["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=k["gbs"] / HBM_PEAK_GBS, traffic=traffic, traffic_source=tsrc, avg_ms=k["avg_ms"],
                        calls=k["calls"], alg_bytes_per_launch=k["alg_gb_per_call"] * 1e9)

    # two lanes (docs/history.md 8c): the two solves of the bordered system run concurrently where a single solve does not saturate
    # HBM (default: one rank, vectors <= 128 MiB -- not the 512^3 headline); the per-kernel event timings then overlap
    tl_opt = [float(kv.split("=")[1]) for kv in args.opt if kv.startswith("two_lanes=")]
    two_lanes = bool(tl_opt[-1] != 0.0) if tl_opt else bool(world == 1 and prob.nlocal <= (1 << 24))
    if two_lanes and roofline is not None:
        roofline["note"] = ("two lanes: kernels of the two concurrent solves overlap, per-kernel durations (and this fraction) "
                            "are those of kernels sharing the device; compare ms_per_step")
    # block Arnoldi: operator applications issued inside blocks over the whole run, and those void (tails of
    # truncated blocks) -- itlinear_per_step counts consumed applications only
    try:
        gmres_blocks = {"operator_applications_in_blocks": ctx.get_option("gmres_block_steps"),
                        "truncated_tails": ctx.get_option("gmres_block_truncated"),
                        "speculated_past_convergence": ctx.get_option("gmres_block_unconsumed")}
    except Exception:  # noqa: BLE001
        gmres_blocks = None
    fd_opt = [float(kv.split("=")[1]) for kv in args.opt if kv.startswith("fd_dparam=")]
    dfdp = "literal" if (fd_opt and fd_opt[-1] == 0.0) else "routed"
    sf_opt = [float(kv.split("=")[1]) for kv in args.opt if kv.startswith("gmres_stencil_free=")]
    stencil_free = bool((sf_opt[-1] if sf_opt else 1.0) != 0.0) and P is not None and args.linsolver == "gmres"
    if rank == 0:
        ms = dt / max(args.steps, 1) * 1e3
        out = {
            "metric": "newton_krylov_corrector_steps_per_s", "value": args.steps / dt, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"SH3d {n}^3 PALC corrector (Palc.jl:237-295 pass), "
                                   f"{'GMRES(30)' if args.linsolver == 'gmres' else 'KrylovLS(:minres)'} rtol 1e-9, "
                                   f"Pl = (L1+shift)^-1 (DCT), BorderingBLS",
                       "grid": [n, n, nzz], "unknowns": prob.nglobal, "parallelism": f"z-slabs x{world}",
                       "itlinear_per_step": last["itlineartot"], "residual_after_step": last["residuals"][-1],
                       # (the count moves between 24 and 26 with the last digits of the cell solutions: from the 8th Arnoldi step on the
                       # solve fits the rounding noise of its own right-hand side -- compare rounds by this figure)
                       "ms_per_operator_application": ms / max(last["itlineartot"], 1),
                       "cell": list(CELL), "tiles": list(tiles), "h": [2 * l / c for l, c in zip(CELL_L, CELL)],
                       "precond_shift": args.shift, "gmres_restart_cycles_max": args.ls_maxiter,
                       # (the engine does not look at the linear solver's flag, src/Newton.jl:93 -- a solve that stops at the restart limit
                       # shows up here: both solves of the bordered system used their whole budget of 30 x maxiter applications)
                       "linear_solves_hit_the_restart_limit": bool(last["itlineartot"] >= 2 * 30 * args.ls_maxiter),
                       "precond_pairing": ("Pl = cholesky(L1), examples/SH3d.jl:88-93" if args.shift == 0.0 else
                                           "Pl = lu(L1 + I), examples/SH2d-fronts.jl:121" if args.shift == 1.0 else "Pl = L1 + shift I"),
                       # which dF/dp the corrector's right-hand side uses (src/continuation/Palc.jl:239-240): "routed" = the
                       # cancellation-free one-pass quotient bk_residual_dparam (what the Julia binding's dispatch installs),
                       # "literal" = (F(x, p + eps) - F(x, p)) / eps from two residual evaluations (--opt fd_dparam=0)
                       "dfdp": dfdp,
                       "state": "z-invariant hexagons (stable), l = 0.1, nu = 1.2",
                       "cell_umax": c0["u"].norminf(),
                       "cell_newton": {"converged": c0["converged"], "itnewton": c0["itnewton"],
                                       "residual": c0["residuals"][-1]},
                       "tiled_state_residual_inf": [res0, res1],
                       "full_corrector": {"converged": full["converged"], "itnewton": full["itnewton"],
                                          "itlinear": full["itlineartot"], "residuals": full["residuals"],
                                          "p": full["u"].p},
                       "cell_corrector": {"converged": cfull["converged"], "itnewton": cfull["itnewton"],
                                          "itlinear": cfull["itlineartot"], "residuals": cfull["residuals"],
                                          "p": cfull["u"].p},
                       "two_lanes": two_lanes, "gmres_blocks": gmres_blocks, "block_log": block_log, "trace": trace_rec,
                       "arnoldi_operator": ("stencil-free: Pl^-1 J = -I + Pl^-1 diag(g(u) + shift) for Pl = L1 + shift I (exact; "
                                            "src/LinearSolver.jl:270-277 `_linmap` rearranged) -- the pointwise factor rides in the "
                                            "x-forward transform pass, -I is a Hessenberg shift, the stencil kernel runs only in the "
                                            "residuals and in each solve's explicit residual check (--opt gmres_stencil_free=0: the "
                                            "literal chain stencil -> Pl^-1)") if stencil_free else "literal chain: stencil kernel, then Pl^-1",
                       "jvp_calls_per_step": (kernels["jvp"]["calls"] / max(args.steps, 1)) if "jvp" in kernels else 0,
                       "setup_seconds": t_setup, "sh_kernel": args.sh_kernel, "linsolver": args.linsolver,
                       "preconditioner": "none" if P is None else "dct"},
            "roofline": roofline, "inner_loop": inner, "fixed_input": fixed, "steady_state": steady, "kernels": kernels, "comm": comm_rec,
        }
        cb = None
        if world == 1 and args.cpu_sample > 0:
            try:
                c = cpu_baseline((args.cpu_sample,) * 3, args.shift)
                scaled = (1.0 / c["seconds_per_step"]) * (c["n"] / prob.nglobal)
                note_np = (f"NumPy/SciPy oracle, 1 thread: 1 corrector step on SH3d {c['dims']} ({c['n']} unknowns, same "
                           f"cell tiling, h and solver settings, {c['itlinear']} GMRES operator applications) took "
                           f"{c['seconds_per_step']:.2f} s -> {scaled:.3e} steps/s scaled by the unknowns ratio to {n}^3")
                cb = {"value": scaled, "unit": "steps/s", "cores": 1, "kind": "port", "sample": note_np +
                      "; CPU restatement of the reference path (assembled sparse L1, MGS2 GMRES, DCT preconditioner), not Julia"}
                try:                                  # second baseline, SURVEY 8(d)(ii): C++/OpenMP on all host cores
                    # (sample: BASELINE config 4's own size, 256^3, for the default --cpu-sample 3: 4 x 8 x 8 cells; ~10 s per step on 16 threads)
                    cs1 = args.cpu_sample + 1
                    cc = cpu_baseline_cpp((cs1, 2 * cs1, 2 * cs1), args.shift)
                    scaled_c = (1.0 / cc["seconds_per_step"]) * (cc["n"] / prob.nglobal)
                    cb = {"value": scaled_c, "unit": "steps/s", "cores": cc["threads"], "kind": "port",
                          "numpy_1thread_value": scaled,
                          "sample": f"C++/OpenMP restatement of the reference's CSR formulation (oracle/cpu_ref.cpp: assembled "
                                    f"L1 = A*A with {cc['nnz_L1']} nonzeros, SpMV, MGS2 GMRES(30), Bordering, FFT-based DCT "
                                    f"preconditioner) on {cc['threads']} threads: 1 corrector step on SH3d {cc['dims']} "
                                    f"({cc['n']} unknowns, {cc['itlinear']} operator applications) took "
                                    f"{cc['seconds_per_step']:.2f} s, scaled by the unknowns ratio to {n}^3; " + note_np +
                                    "; CPU restatements of the reference path, not Julia"}
                except Exception as e:
                    cb["sample"] += f"; (C++/OpenMP baseline unavailable: {e!r})"
            except Exception as e:  # the baseline must not take the bench line down
                cb = {"value": None, "unit": "steps/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        out["cpu_baseline"] = cb
        FINAL_LINE.append(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
    ctx.close()


def run_branch_workload(args, ctx, hip, Cn, prob, P, ls, u0, branch_setup, barrier, rank, world, n, tiles, t_setup):
    """BASELINE config 5: a PALC branch with the reference example's settings (examples/SH3d.jl:160-166: Bordered tangent,
    BorderingBLS(check_precision = false), ds = -0.001, dsmax = 0.005, Newton tol 1e-9, normC = norminf, detect_bifurcation 3,
    nev = 15: shift-invert eigensolve sigma = 0.1, Krylov dimension 45 after every step), every step ONE bk_cont_step call."""
    import torch
    els = (hip.KrylovLSSymmetric("minres", rtol=args.eig_inner_rtol, atol=1e-12, itmax=4000, Pl=P) if args.eig_inner == "minres"
           else hip.GMRESKrylovKit(dim=args.eig_inner_dim, rtol=args.eig_inner_rtol, atol=1e-12, maxiter=150, Pl=P))
    eig = hip.ShiftInvert(args.eig_sigma, els, tol=args.eig_tol, maxiter=20, hermitian=True, save_vectors=False,
                          krylovdim=args.eig_dim if args.eig_dim > 0 else None)
    cp, alg = branch_setup(eig)
    cp.max_steps = args.steps
    ctx.set_option("eig_thick_start", args.eig_thick)
    per = []
    last_t = [time.perf_counter()]
    init = {}

    def on_init(r):
        torch.cuda.synchronize()
        t = time.perf_counter()
        init.update(seconds=t - last_t[0], eig_solves=r.eig_numops, eig_inner_iterations=int(ctx.get_option("eig_last_inner_ops")),
                    eig_converged=bool(r.eig_converged))
        last_t[0] = t

    def fin(get, r):
        torch.cuda.synchronize()
        t = time.perf_counter()
        per.append(dict(step=r.step, seconds=t - last_t[0], p=r.p, ds=r.ds_used, itnewton=r.itnewton, itlinear=r.itlinear,
                        eig_solves=r.eig_numops, eig_inner_iterations=int(ctx.get_option("eig_last_inner_ops")),
                        eig_converged=bool(r.eig_converged), n_unstable=r.n_unstable,
                        rightmost=[r.vals_re[i] for i in range(min(4, r.nvals))]))
        last_t[0] = t
        return True

    barrier()
    t0 = time.perf_counter()
    last_t[0] = t0
    br = Cn.continuation_native(prob, u0, 0.1, alg, cp, normC=Cn.norminf, finalise_solution=fin, on_init=on_init)
    barrier()
    nst = len(br.param) - 1
    t_init = init.get("seconds", 0.0)                         # two Newton solves + the eigensolve at the first point
    t_steps = sum(p_["seconds"] for p_ in per)
    if rank == 0:
        FINAL_LINE.append(json.dumps({
            "metric": "palc_continuation_steps_per_s", "value": nst / max(t_steps, 1e-9), "unit": "steps/s", "n_gpus": world,
            "steps": nst, "warmup": 0, "ms_per_step": t_steps / max(nst, 1) * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SH3d {n}^3 PALC branch (BASELINE config 5): corrector + {args.nev} eigenvalues "
                                   f"(ShiftInvert sigma {args.eig_sigma:g}, Krylov-Schur dim {args.eig_dim if args.eig_dim > 0 else max(30, args.nev + 30)}, tol {args.eig_tol:g}, inner "
                                   f"{args.eig_inner}{'(%d)' % args.eig_inner_dim if args.eig_inner == 'gmres' else ''} rtol {args.eig_inner_rtol:g}, thick start {args.eig_thick}) + Bordered tangent + "
                                   f"predictor per step",
                       "grid": [n, n, n], "tiles": list(tiles), "parallelism": f"z-slabs x{world}",
                       "eig_settings_note": ("the reference example asks for tol 1e-12, maxiter 20, krylovdim 45 (examples/SH3d.jl:109); on the tiled "
                                             "domain the wanted eigenvalues sit in a dense band (relative gaps 1e-3 in 1/(lambda - sigma)) and "
                                             "Krylov-Schur stops at the restart limit with eig_converged = false there (run with --eig-tol 1e-12 to "
                                             "see it: per_step[].eig_converged); 1e-8 is the tolerance it reaches within 20 restarts"),
                       "all_eigensolves_converged": all(p_["eig_converged"] for p_ in per) and bool(init.get("eig_converged", True)),
                       "initialisation": init, "setup_seconds": t_setup},
            "per_step": per, "param": br.param, "n_unstable": br.n_unstable}))


# The ONE JSON line is the last thing this process writes: communicator / context teardown first (RCCL and the runtime
# print to stdout on their way out -- "Librccl path : ..." came AFTER the line when it was printed before the teardown),
# then the line, flushed; a rank that has nothing to print exits quietly.
FINAL_LINE = []

if __name__ == "__main__":
    main()
    sys.stdout.flush()
    try:                                       # RCCL's version banner sits in the C stdio buffer when stdout is a pipe or a file
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if FINAL_LINE:
        print(FINAL_LINE[-1], flush=True)
