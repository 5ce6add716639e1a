"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep reproducing them
(CPU), and the HIP engine must match them through the C ABI (GPU).  Tolerances as in BASELINE.md: solutions
1e-6 (1 + |x|_inf) at tight eps, gradients 1e-5 relative."""
import glob
import os

import numpy as np
import pytest

from cvxpylayers_amd import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(f).startswith(("refglue_", "ref_notebook_")))   # refglue_*: tests/test_ref_glue.py, tests/test_gpu_refglue.py; ref_notebook_*: tests/test_notebook_golden.py


def load(path):
    d = np.load(path)
    cones = {"z": int(d["z"]), "l": int(d["l"]), "q": [int(v) for v in d["q"]], "s": [int(v) for v in d["s"]]}
    if "ep" in d.files and int(d["ep"]):
        cones["ep"] = int(d["ep"])
    if "p" in d.files and len(d["p"]):
        cones["p"] = [float(v) for v in d["p"]]
    n, B, seed = int(d["n"]), int(d["B"]), int(d["seed"])
    A, b, c = P.generate(n, cones, B, seed=seed)
    return d, n, cones, A, b, c


def relerr(got, want):
    return np.abs(got - want).max() / (1 + np.abs(want).max())


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(path):
    from oracle import oracle
    d, n, cones, A, b, c = load(path)
    Pm = d["P"] if "P" in d.files else None
    r = oracle.solve_batch(A, b, c, cones, P=Pm, eps=1e-10, max_iters=200000)
    assert (r["status"] == 1).all()
    assert relerr(r["x"], d["x"]) < 1e-9 and relerr(r["y"], d["y"]) < 1e-9 and relerr(r["s"], d["sl"]) < 1e-9
    g = oracle.adjoint_batch(A, b, c, cones, d["x"], d["y"], d["sl"], d["dx"], d["dy"], P=Pm, mode="dense")
    assert relerr(g["dA"], d["dA"]) < 1e-8 and relerr(g["db"], d["db"]) < 1e-8 and relerr(g["dc"], d["dc"]) < 1e-8
    if Pm is not None:
        assert relerr(g["dP"], d["dP"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_engine_matches_golden(path):
    import torch
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    d, n, cones, A, b, c = load(path)
    tpl = P.dense_template(n, cones)
    pst = P_bm = None
    if "P" in d.files:         # quadratic objective: upper-triangle structure, values batch-major
        rows, ptr = [], [0]
        for j in range(n):
            rows.extend(range(j + 1)); ptr.append(len(rows))
        pst = (np.asarray(rows, dtype=np.int32), np.asarray(ptr, dtype=np.int32))
        pcols = np.repeat(np.arange(n), np.diff(pst[1]))
        P_bm = torch.from_numpy(np.ascontiguousarray(d["P"][:, pst[0], pcols])).cuda()
    try:
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, torch.device("cuda", 0), p_structure=pst)
    except NotImplementedError as e:       # cone type the device path rejects explicitly (CE_E_UNSUPPORTED)
        pytest.skip(str(e))
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda())
    x, y, s, iters, status, resid = eng.solve(A_bm, torch.from_numpy(q_eval).cuda(), make_settings(dict(acceleration_lookback=0, eps=1e-10, max_iters=200000)), P_bm=P_bm)
    assert (status.cpu().numpy() == 1).all()
    for got, want in ((x, d["x"]), (y, d["y"]), (s, d["sl"])):
        err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
        assert err.max() < 1e-6, err.max()
    xr, yr, sr, dx, dy = (torch.from_numpy(d[k]).cuda() for k in ("x", "y", "sl", "dx", "dy"))
    if P_bm is not None:
        dA, dq, adj, dP = eng.vjp(A_bm, xr, yr, sr, dx, dy, P_bm=P_bm)
        wantP = d["dP"][:, pst[0], pcols] + np.where(pst[0] != pcols, d["dP"][:, pcols, pst[0]], 0.0)
        assert relerr(dP.cpu().numpy(), wantP) < 1e-5
    else:
        dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, dx, dy)
    assert (adj.cpu().numpy() == 0).all()
    dA = dA.cpu().numpy(); dq = dq.cpu().numpy()
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    want = np.empty_like(dA)
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -d["dA"][:, i, j] if j < n else d["db"][:, i]     # [-dA.data, db[b_idx]]  (diffcp_if.py:91)
    assert relerr(dA, want) < 1e-5
    assert relerr(dq[:n], d["dc"].T) < 1e-5 and np.abs(dq[n]).max() == 0
