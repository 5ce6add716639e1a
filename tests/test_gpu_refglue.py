"""GPU replay of the reference-glue fixtures (tests/golden/refglue_*.npz: outputs of the REFERENCE's own frontend + DIFFCP plugin code,
executed by tests/golden/make_refglue.py with diffcp's arithmetic served by the CPU oracle; tests/test_ref_glue.py re-checks them
wherever /root/reference exists).  Nothing here reads /root/reference.

  * plugin boundary: the tensors the reference plugin received (q_eval, A_eval) go into MI355's _CvxpyLayer.apply; primal / dual
    and, for the same incoming (dprimal, ddual), dq_eval / dA_eval must agree -- sign of A, position of b, gradient packing, batch
    axis handling are the reference's, not a restatement;
  * whole layer: the same parameters through cvxpylayers_amd.torch.CvxpyLayer; variables and parameter gradients must agree
    (flattening order, canonical column order, broadcast-gradient sums, svec unpacking, Fortran reshape).
Tolerances (fp64, both sides at eps 1e-10): values 1e-6 (1 + |x|_inf), gradients 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

import ref_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(got, want, tol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() if want.size else 0.0
    assert err <= tol * (1.0 + np.abs(want).max()), (what, err)


@pytest.mark.parametrize("name", list(ref_cases.CASES))
def test_plugin_boundary_matches_the_reference_plugin(name):
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    f = np.load(os.path.join(GOLD, f"refglue_{name}.npz"))
    tpl = ref_cases.CASES[name]()["template"]
    ctx = MI355_ctx(None, tpl.A_structure, tpl.cone_dims, options=dict(ref_cases.SOLVER_ARGS))
    q = torch.tensor(f["q_eval"], device="cuda", requires_grad=True)
    A = torch.tensor(f["A_eval"], device="cuda", requires_grad=True)
    primal, dual, _, _ = _CvxpyLayer.apply(None, q, A, ctx, {"mode": "dense"}, True, None)
    _close(primal, f["primal"], 1e-6, "primal"); _close(dual, f["dual"], 1e-6, "dual")
    torch.autograd.backward([primal, dual], [torch.tensor(f["dprimal"], device="cuda"), torch.tensor(f["ddual"], device="cuda")])
    _close(q.grad, f["dq_eval"], 1e-5, "dq_eval"); _close(A.grad, f["dA_eval"], 1e-5, "dA_eval")


@pytest.mark.parametrize("name", list(ref_cases.CASES))
def test_whole_layer_matches_the_reference_frontend(name):
    from cvxpylayers_amd.torch import CvxpyLayer
    f = np.load(os.path.join(GOLD, f"refglue_{name}.npz"))
    case = ref_cases.CASES[name]()
    layer = CvxpyLayer(template=case["template"], solver_args=dict(ref_cases.SOLVER_ARGS))
    params = [torch.tensor(f[f"param{k}"], device="cuda", requires_grad=True) for k in range(len(case["params"]))]
    outs = layer(*params)
    assert len(outs) == len(case["weights"])
    for k, o in enumerate(outs):
        _close(o, f[f"out{k}"], 1e-6, f"out{k}")
    sum((o * torch.tensor(f[f"weight{k}"], device="cuda")).sum() for k, o in enumerate(outs)).backward()
    for k, p in enumerate(params):
        _close(p.grad, f[f"grad{k}"], 1e-5, f"grad{k}")
