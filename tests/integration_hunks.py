"""The integration patch of INTEGRATION.md section 1 in machine-applicable form: (reference file, anchor text, replacement, expected
number of occurrences).  tests/test_ref_glue.py applies them to the reference's source text IN MEMORY (nothing is copied into
this repository) and runs the patched modules; INTEGRATION.md prints the same hunks as a diff (a test keeps the two in step)."""
from __future__ import annotations

import os
import sys
import types
from contextlib import contextmanager
from types import SimpleNamespace

HUNKS = [
    # 1. registry: the solver context (CSC structure, like DIFFCP -- interfaces/__init__.py:26-33)
    ("interfaces/__init__.py",
     '''    options = _merge_verbose(kwargs, verbose)

    if solver == "DIFFCP":''',
     '''    options = _merge_verbose(kwargs, verbose)

    if solver == "MI355":                       # AMD Instinct MI355X engine: CSC structure, like DIFFCP
        from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx

        return MI355_ctx(
            param_prob.reduced_P.problem_data_index,
            param_prob.reduced_A.problem_data_index,
            cone_dims,
            data.get("lower_bound"),
            data.get("upper_bound"),
            options,
            reduced_A_mat=param_prob.reduced_A.reduced_mat,      # constant-A decided once from the parameter map (as MOREAU_ctx does)
        )

    if solver == "DIFFCP":''', 1),
    # 2. registry: the torch plugin class (interfaces/__init__.py:76-101)
    ("interfaces/__init__.py",
     '''    match solver:
        case "MPAX":
            from cvxpylayers.interfaces.mpax_if import _CvxpyLayer
''',
     '''    match solver:
        case "MI355":
            from cvxpylayers_amd.interfaces.mi355_if import _CvxpyLayer

            return _CvxpyLayer
        case "MPAX":
            from cvxpylayers.interfaces.mpax_if import _CvxpyLayer
''', 1),
    # 3. canonicalise as DIFFCP: CVXPY must be given a solver name it knows, and the engine wants DIFFCP's (A, b, c, cones)
    #    (utils/parse_args.py:447-462, both the gp and the standard branch)
    ("utils/parse_args.py",
     '''                solver=solver,
                gp=False,''',
     '''                solver="DIFFCP" if solver == "MI355" else solver,
                gp=False,''', 2),
    # 4. warm start through the reference frontend (torch/cvxpylayer.py:425-429, 464-473): the reference only lets MOREAU through
    #    and only fills `ws` from its MOREAU cache; the MI355 plugin keeps the previous solution itself (warm_start=True)
    ("torch/cvxpylayer.py",
     '''        if warm_start and self.ctx.solver != "MOREAU":
            raise ValueError(
                "warm_start=True is only supported with solver='MOREAU'. "''',
     '''        if warm_start and self.ctx.solver not in ("MOREAU", "MI355"):
            raise ValueError(
                "warm_start=True is only supported with solver='MOREAU'. "''', 1),
    ("torch/cvxpylayer.py",
     '''        ws = None
        if warm_start and self._warm_start_cache is not None:''',
     '''        ws = True if (warm_start and self.ctx.solver == "MI355") else None     # MI355: the plugin reuses its previous solution
        if warm_start and self._warm_start_cache is not None:''', 1),
]


def patched_source(pkg_dir: str, rel: str) -> str:
    src = open(os.path.join(pkg_dir, rel)).read()
    for f, anchor, repl, count in HUNKS:
        if f != rel:
            continue
        assert src.count(anchor) == count, f"hunk anchor for {rel} found {src.count(anchor)} times, expected {count}"
        src = src.replace(anchor, repl)
    return src


def as_diff() -> str:
    """The hunks as the unified-diff-style text INTEGRATION.md shows."""
    out = []
    for f, anchor, repl, _ in HUNKS:
        a, b = anchor.split("\n"), repl.split("\n")
        out.append(f"--- a/src/cvxpylayers/{f}\n+++ b/src/cvxpylayers/{f}\n@@")
        import difflib
        for line in difflib.ndiff(a, b):
            if line.startswith("?"):
                continue
            out.append(line[0] + line[2:] if line[0] in "+-" else " " + line[2:])
    return "\n".join(out)


@contextmanager
def patched_reference_modules(ns):
    """Context manager.  Executes the patched source text of interfaces/__init__.py and torch/cvxpylayer.py as modules (the un-patched modules of `ns`
    stay importable, so `from cvxpylayers.interfaces.diffcp_if import ...` inside them still resolves) and byte-compiles the patched
    utils/parse_args.py.  Returns a namespace shaped like ref_glue.reference_modules()'s."""
    import ref_glue
    pkg = ref_glue._PKG
    compile(patched_source(pkg, "utils/parse_args.py"), "parse_args.py (patched)", "exec")
    inter = types.ModuleType("cvxpylayers.interfaces"); inter.__package__ = "cvxpylayers.interfaces"; inter.__path__ = [os.path.join(pkg, "interfaces")]
    exec(compile(patched_source(pkg, "interfaces/__init__.py"), "interfaces/__init__.py (patched)", "exec"), inter.__dict__)
    saved = sys.modules.get("cvxpylayers.interfaces")
    sys.modules["cvxpylayers.interfaces"] = inter           # the frontend does `from cvxpylayers.interfaces import get_torch_cvxpylayer` per call
    cl = types.ModuleType("cvxpylayers.torch.cvxpylayer"); cl.__package__ = "cvxpylayers.torch"
    try:
        exec(compile(patched_source(pkg, "torch/cvxpylayer.py"), "torch/cvxpylayer.py (patched)", "exec"), cl.__dict__)
        yield SimpleNamespace(interfaces=inter, cvxpylayer=cl, parse_args=ns.parse_args, diffcp_if=ns.diffcp_if, diffcp=ns.diffcp)
    finally:
        sys.modules["cvxpylayers.interfaces"] = saved
