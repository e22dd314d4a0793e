"""20 applications of the spectral preconditioner at n^3 (default 512): the workload of a per-pass rocprofv3 A/B
(scripts/micro/dct_pass_ab.sh: the two LDS layouts of dct_core.h, BKHIP_LIB selects the library)."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bk_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = hip.Context(0)
for kv in sys.argv[2:]:                      # library options key=value (A/B runs)
    k_, v_ = kv.split("=")
    ctx.set_option(k_, float(v_))
prob = hip.SwiftHohenberg(ctx, (n, n, n), (math.pi * n / 32,) * 3)
g = torch.Generator(device="cuda").manual_seed(0)
v = hip.HipVec(ctx, torch.rand(n ** 3, dtype=torch.float64, device="cuda", generator=g))
out = v.similar()
P = hip.DCTPreconditioner(prob, 1.0)
for _ in range(20):
    ctx.check(ctx.lib.bk_precond_apply(P.h, C.c_void_p(v.t.data_ptr()), C.c_void_p(out.t.data_ptr())))
torch.cuda.synchronize()
