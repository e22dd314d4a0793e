"""Keeps the reference-fixture consumer (tests/test_reference_fixtures.py) executable: generates a file of the fixture schema
from the oracle (tests/golden/make_schema_fixture.py -- NOT reference data, it pins nothing) and runs the CPU consumers
against it in a subprocess.  Guards against rot of the plumbing that will pin the oracle to the real BifurcationKit.jl the
day julia/gen_fixtures.jl is run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_consumers_run_end_to_end_on_a_schema_file(tmp_path):
    fx = str(tmp_path / "schema_fixture.json")
    env = dict(os.environ, BK_FIXTURES=fx, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", MKL_NUM_THREADS="2")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_schema_fixture.py"), fx], check=True, timeout=900, env=env)
    d = json.load(open(fx))
    assert d["generator"] == "oracle" and {"sh3d_22", "sh2d_151x100", "cgl_41x21"} <= set(d)
    # the keys julia/gen_fixtures.jl writes per Swift-Hohenberg case
    want = {"dims", "ls", "l", "nu", "F_u0", "dF_u0_probe1", "newton", "gmres", "gmres_shift", "bordering", "matrixfree",
            "shift_invert", "branch", "minres", "cg"}
    assert want <= set(d["sh3d_22"]) and want <= set(d["sh2d_151x100"]) and want <= set(d["sh2d_128x64"])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_fixtures.py"), "-q", "-m",
                        "not gpu", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # (sh3d_64 is the one case the schema file does not hold: its two consumers skip)
    assert "4 passed, 1 skipped" in r.stdout, r.stdout[-500:]
    # and the real file is still absent / untouched: nothing oracle-generated may sit in tests/golden/
    real = os.path.join(ROOT, "tests", "golden", "julia_fixtures.json")
    if os.path.exists(real):
        assert json.load(open(real)).get("generator") != "oracle"
