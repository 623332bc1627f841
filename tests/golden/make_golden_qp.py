"""Golden ``dq`` from the reference's own QP path -- RECIPE, to be run where quadprog exists.

The reference solves its QP with ``qpsolvers.solve_problem(problem, solver="quadprog")``
(``pink/solve_ik.py:270``).  Neither ``qpsolvers`` nor ``quadprog`` can be installed in the
build container (no network, not in the wheelhouse), so the QP half of the oracle is *parity
unpinned* (``oracle/pink_oracle.py`` header).  This script closes that gap the moment it is run in an
environment that has both packages:

    pip install qpsolvers quadprog
    python tests/golden/make_golden_qp.py        # writes tests/golden/quadprog_dq.npz

It needs nothing else (no Pinocchio, no /root/reference): the problems are

* the ``(P, q, G, h[, A, b])`` that the reference's ``pink.build_ik`` produced for the fixtures of
  ``pink_build_ik.npz`` (``make_golden.py``), and
* a seeded sample of the three BASELINE configurations (``pink_amd.synthetic``), stacked into
  Pink's dense form by the NumPy restatement whose stacking half *is* pinned by those fixtures.

For every problem the file stores quadprog's ``x`` and whether a solution was found;
``tests/test_quadprog_golden.py`` then holds the C oracle, the CPU wave emulator and the MI355X
kernel to those vectors (1e-9 absolute) and is skipped, with this explanation, while the file is
absent.
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SAMPLE = 32  # instances per synthetic configuration / regime


def fixture_problems():
    """(name, P, q, G, h, A, b) of every fixture the reference's build_ik produced."""
    g = np.load(os.path.join(HERE, "pink_build_ik.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    for n in names:
        A = g[f"{n}/A"] if f"{n}/A" in g else None
        b = g[f"{n}/b"] if f"{n}/b" in g else None
        yield f"fixture/{n}", g[f"{n}/P"], g[f"{n}/qvec"], g[f"{n}/G"], g[f"{n}/h"], A, b


def synthetic_problems():
    """(name, P, q, G, h, None, None) for SAMPLE instances of each BASELINE configuration and
    bounds regime, in Pink's dense form (box rows as +-e_i rows, then the barrier rows)."""
    from oracle import c_oracle
    from pink_amd import synthetic

    for cfg in ("ur5", "draco3", "jvrc"):
        for bounds, jac in (("tight", "dense"), ("kinematic", "kinematic")):
            terms = synthetic.make_terms(cfg, SAMPLE, bounds=bounds, jacobians=jac)
            pf = synthetic.pink_form(terms)
            Hc = c_oracle.solve_ik_batch(**pf, want_Hc=True, solve=False)
            for i in range(SAMPLE):
                yield f"synthetic/{cfg}/{bounds}/{i}", Hc["H"][i], Hc["c"][i], pf["G"][i], pf["h"][i], None, None


def main(path=None):
    try:
        import qpsolvers
        import quadprog  # noqa: F401  (the backend qpsolvers dispatches to)
    except ImportError as exc:
        raise SystemExit(f"{exc}: this recipe needs `pip install qpsolvers quadprog` (absent in the build container)")
    out = {}
    n = 0
    for name, P, q, G, h, A, b in list(fixture_problems()) + list(synthetic_problems()):
        problem = qpsolvers.Problem(P, q, G, h, A, b)
        sol = qpsolvers.solve_problem(problem, solver="quadprog")  # pink/solve_ik.py:270
        found = bool(sol.found and sol.x is not None)
        out[f"{name}/found"] = found
        out[f"{name}/x"] = np.asarray(sol.x, dtype=np.float64) if found else np.zeros_like(q)
        n += 1
    out["meta/qpsolvers_version"] = np.array(getattr(qpsolvers, "__version__", "?"))
    out["meta/sample"] = SAMPLE
    path = path or os.path.join(HERE, "quadprog_dq.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {n} problems solved by quadprog through qpsolvers {out['meta/qpsolvers_version']}")


if __name__ == "__main__":
    main()
