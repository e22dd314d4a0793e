"""Debug: per-phase timing of the fused DCT kernel (option dct_trace) on an n^3 grid."""
import ctypes as C
import math
import sys

import torch

import bk_amd
from bk_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = hip.Context(0)
prob = hip.SwiftHohenberg(ctx, (n, n, n), (math.pi * n / 32,) * 3)
v = hip.HipVec(ctx, torch.rand(n ** 3, dtype=torch.float64, device="cuda"))
out = v.similar()
P = hip.DCTPreconditioner(prob, 1.0)
f = lambda: ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
f(); f()
ctx.set_option("dct_trace", 1)
f()
