"""Extract the one literal golden vector the reference's tests hold for this path: the 5x5 matrix and its
eigenvalues / eigenvectors at test/linear_solvers/test_linear.jl:595-614.  Run in the build container
(where /root/reference exists); the JSON it writes is committed because the reference does not travel to
the GPU box.

    python tests/golden/make_golden.py
"""
import json
import os
import re

REF = "/root/reference/test/linear_solvers/test_linear.jl"
HERE = os.path.dirname(os.path.abspath(__file__))


def _cplx(tok):
    tok = tok.replace("im", "j").replace(" ", "")
    return complex(tok)


def main():
    lines = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "J0 = [0.688714" in l)
    mat = []
    for l in lines[start:start + 5]:
        l = l.replace("J0 = [", "").replace("]", "")
        mat.append([float(x) for x in l.split()])
    i_vals = next(i for i in range(start, start + 20) if "_vals .≈ [" in lines[i])
    vals = []
    for l in lines[i_vals:i_vals + 5]:
        l = l.split("[")[-1].replace("])", "").strip()
        vals.append(_cplx(l))
    i_vecs = next(i for i in range(i_vals, i_vals + 20) if "norminf(_vecs - [" in lines[i])
    vecs = []
    for l in lines[i_vecs:i_vecs + 5]:
        l = l.split("[")[-1].split("]")[0]
        toks = re.findall(r"[-+]?\d*\.\d+[-+]\d*\.\d+im", l.replace(" ", ""))
        assert len(toks) == 5, (l, toks)
        vecs.append([_cplx(t) for t in toks])
    out = dict(
        source="BifurcationKit.jl test/linear_solvers/test_linear.jl:595-614",
        J0=mat,
        vals=[[v.real, v.imag] for v in vals],
        vecs=[[[v.real, v.imag] for v in row] for row in vecs],
        vals_rtol="isapprox default (sqrt(eps))", vecs_atol=1e-6)
    with open(os.path.join(HERE, "eig5x5.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote eig5x5.json")


if __name__ == "__main__":
    main()
