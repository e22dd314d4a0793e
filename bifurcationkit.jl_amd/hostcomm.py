"""Test communicator: the bk_allreduce_fn / bk_sendrecv_fn callbacks of include/bkhip.h implemented over
``torch.distributed`` (gloo, CPU tensors).  It lets two ranks share one GPU -- RCCL refuses that -- so the whole
distributed path (halo exchange, batched dot all-reduce, DCT transposes) can be exercised on a single-GPU box,
and it runs on a CPU-only machine for the callback plumbing itself.  Production multi-GPU runs use RCCL
(``Context(device, ("rccl", rank, nranks, unique_id))``)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# `user` of the callbacks: NULL for the context's own communicator, 1 for the communicator registered for its second lane
# (bk_ctx_set_lane_comm; comm_tuple below passes lane_user = 1).  Each has its own gloo group: the two are driven by two
# library-owned proxy threads concurrently and their collectives must never be matched crosswise.
_groups = {}


def _group(user):
    lane = int(user) if user else 0
    return _groups.get(lane)                     # None = the default group


def allreduce(user, buf, n, op):
    try:
        arr = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(arr.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=_group(user))
        arr[:] = t.numpy()
        return 0
    except Exception as e:  # pragma: no cover - surfaced as a library error
        print("hostcomm.allreduce failed:", e)
        return 1


def sendrecv(user, sbuf, ns, dst, rbuf, nr, src):
    try:
        reqs = []
        rt = None
        if ns and dst >= 0:
            st = torch.from_numpy(np.ctypeslib.as_array(sbuf, shape=(ns,)).copy())
            reqs.append(dist.isend(st, dst, group=_group(user)))
        if nr and src >= 0:
            rt = torch.empty(nr, dtype=torch.float64)
            reqs.append(dist.irecv(rt, src, group=_group(user)))
        for r in reqs:
            r.wait()
        if rt is not None:
            np.ctypeslib.as_array(rbuf, shape=(nr,))[:] = rt.numpy()
        return 0
    except Exception as e:  # pragma: no cover
        print("hostcomm.sendrecv failed:", e)
        return 1


def comm_tuple():
    """Argument for ``hip.Context(device, comm=...)`` once ``torch.distributed`` (gloo) is initialised.  Collective: also
    creates the gloo group of the second lane."""
    if 1 not in _groups and dist.get_world_size() > 1:
        _groups[1] = dist.new_group(backend="gloo")
    return ("host", dist.get_rank(), dist.get_world_size(), allreduce, sendrecv, 1 if 1 in _groups else None)


def slab(n, rank, nranks):
    """Balanced contiguous decomposition [lo, hi) of n planes -- the rule bk_problem_create uses."""
    base, rem = divmod(n, nranks)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
