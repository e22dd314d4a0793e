"""CPU oracle for the Newton-Krylov PALC corrector path of BifurcationKit.jl.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product (``bifurcationkit.jl_amd/``) never
imports, links or executes anything in here; it fails loudly when the HIP
extension is missing instead of falling back to this code.

What it is: a NumPy/SciPy restatement of the reference's own formulation
(assembled sparse operators + host Krylov loops), every function citing the
reference ``file:line`` it follows (paths relative to ``/root/reference``).

Parity status ("pinned" = checked against something the reference itself
holds):
  * Newton / PALC corrector / MatrixBLS / Secant and Bordered tangents /
    step-size control / eigenvalue-count detection / bisection / both-sides
    merge (palc.py, bordered.py, bifurcations.py): pinned by the known answers
    the reference's own tests assert at rtol sqrt(eps) -- special points of
    test/hopf_codim_2/COModel.jl:31-34, bisection intervals of lorenz84.jl:63-66,
    the fold of test/fold_codim_2/codim2.jl:54-55, the bifurcation points and
    kernel dimensions of test/continuation/test_bif_detection.jl:62-98
    (tests/golden/reference_known_answers.json, tests/test_reference_known_answers.py).
  * dense/eigen plumbing: pinned by the literal 5x5 golden eigen-decomposition
    of ``test/linear_solvers/test_linear.jl:595-614`` (tests/golden/eig5x5.json)
    and by the reference tests' own identities (solver == dense ``\\``).
  * Krylov arithmetic (KrylovKit.linsolve / eigsolve, IterativeSolvers.gmres):
    the packages are un-vendored third-party dependencies (Project.toml:53,56,
    no Manifest) and Julia is not installed, so these are restatements of the
    published algorithms -> "parity unpinned" against real package output;
    anchored on the reference's call sites and test identities instead.
  * PDE-level outputs (SH2d/SH3d/cGL2d residuals, branches, eigenvalues):
    "parity unpinned" -- no reference test pins them.
"""
