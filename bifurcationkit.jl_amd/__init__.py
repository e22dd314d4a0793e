"""bifurcationkit.jl_amd -- MI355X-native Newton-Krylov corrector for BifurcationKit.jl's PALC continuation.

Only what the hot path needs:
  csrc/      hand-written HIP (gfx950) kernels + the C ABI (include/bkhip.h) -> lib/libbkhip.so
  _lib.py    ctypes binding of the C ABI (the Python twin of julia/BifurcationKitHIP.jl)
  hip.py     mirror of the reference's plugin surface (linear / bordered / eigen solvers, problems)
  continuation.py   minimal restatement of the caller (newton, newton_palc, PALC loop) for parity tests

Import name: ``bk_amd`` (the directory name is not a valid Python identifier; ``bk_amd.py`` at the repo
root registers this package under that name).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "hip", "continuation"]
