"""CPU test: the C-ABI shared library loads and exports every symbol include/bkhip.h declares (no compute
calls -- there is no GPU here), and the ctypes binding covers the same set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bkhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(bk_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_the_boundary():
    names = _declared()
    for must in ["bk_gmres", "bk_gmres2", "bk_bls_bordering", "bk_bls_matrixfree", "bk_bls_block_bordering", "bk_gmres_cshift", "bk_jacobian_adjoint", "bk_bls_bordering_cshift", "bk_bls_block_matrixfree", "bk_newton_deflated", "bk_cont_create", "bk_cont_step", "bk_eig_shiftinvert", "bk_eig_krylovkit",
                 "bk_newton_palc", "bk_newton", "bk_residual", "bk_jacobian", "bk_op_apply", "bk_vec_dot",
                 "bk_precond_sh_create", "bk_ctx_create_dist"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from bk_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"symbols declared in include/bkhip.h but not exported: {missing}"


def test_ctypes_binding_matches_header():
    from bk_amd import _lib
    declared = set(_declared())
    bound = set(_lib.SIGNATURES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    lib = _lib.load()
    assert lib.bk_version() >= 100


def test_no_cpu_fallback_in_product():
    """The product path must not import the oracle."""
    pkg = os.path.join(ROOT, "bifurcationkit.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_library_does_not_link_a_vendor_blas():
    """rocBLAS only serves the cross-check option dct_gemm = 2 and is dlopen()ed there (csrc/dct.hip: rocblas_api); the product
    library's own dependencies are the HIP runtime and RCCL."""
    import subprocess
    lib = os.path.join(ROOT, "bifurcationkit.jl_amd", "lib", "libbkhip.so")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    needed = [l.split("[")[1].split("]")[0] for l in out.splitlines() if "NEEDED" in l]
    assert not [n for n in needed if "blas" in n.lower()], needed
    assert any("amdhip64" in n for n in needed) and any("rccl" in n for n in needed), needed
