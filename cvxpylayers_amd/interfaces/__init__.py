"""Solver-plugin registry, mirroring cvxpylayers/interfaces/__init__.py:13-101 for the one key this
repository provides.  "MI355" canonicalises exactly like "DIFFCP" (CSC structure of [A_cvx | b_cvx],
interfaces/__init__.py:26-33) so both see identical (A, b, c, cones)."""


def get_solver_ctx(solver, param_prob, cone_dims, data, kwargs, verbose=False):
    options = dict(kwargs) if kwargs else {}
    if verbose:
        options["verbose"] = True
    if solver == "MI355":
        from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx

        return MI355_ctx(
            param_prob.reduced_P.problem_data_index if getattr(param_prob, "reduced_P", None) is not None else None,
            param_prob.reduced_A.problem_data_index,
            cone_dims,
            data.get("lower_bound") if data else None,
            data.get("upper_bound") if data else None,
            options,
            reduced_A_mat=getattr(param_prob.reduced_A, "reduced_mat", None),          # the parameter map: constant-A is decided structurally, once (moreau_if.py:234-256)
        )
    raise RuntimeError("Unknown solver. Check if your solver is supported by CVXPYlayers")


def get_torch_cvxpylayer(solver):
    if solver == "MI355":
        from cvxpylayers_amd.interfaces.mi355_if import _CvxpyLayer

        return _CvxpyLayer
    raise RuntimeError("Unknown solver. Check if your solver is supported by CVXPYlayers")
