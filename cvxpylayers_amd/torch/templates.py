"""Cvxpy-free construction of CanonTemplate objects.

CVXPY's DPP canonicalisation produces AFFINE maps from the stacked parameter vector (p, 1) to the data of the cone program
(utils/parse_args.py:447-462: ParamConeProg.reduced_A.reduced_mat, .q).  When CVXPY is not installed (this build image) the same
object can be obtained from any function that builds the solver-form data (A, b, c) from parameter values, provided that
function is affine in the parameters: evaluate it at 0 and at the unit vectors ("affine probing").  The templates in the
tests are built this way from hand-canonicalised problem families that restate the reference's own test problems.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import scipy.sparse as sp

from cvxpylayers_amd.problems import cone_rows
from cvxpylayers_amd.torch.cvxpylayer import CanonTemplate, VariableRecovery


def template_from_affine_builder(builder: Callable, param_shapes: Sequence[tuple], cones: dict,
                                 var_recover: Sequence[VariableRecovery], col_order: Sequence[int] | None = None) -> CanonTemplate:
    """builder(*param_arrays) -> (A (m,n), b (m,), c (n,)) in SOLVER form (A x + s = b), affine in the parameters; a fourth
    return value P (n,n) (symmetric, affine in the parameters) adds the quadratic objective 1/2 x^T P x: its upper triangle becomes
    the template's P structure / P_map (what plugins in SUPPORTS_QUAD_OBJ receive as P_eval).

    Parameters are flattened in Fortran order (torch/cvxpylayer.py:40-55).  `col_order[i]` = position of user parameter i in
    the canonical parameter vector (CVXPY orders columns by parameter id, not by user order; utils/parse_args.py:330-385);
    default: user order.  The boundary stores CVXPY's sign convention: the augmented matrix is [A_cvx | b_cvx] with
    A_cvx = -A (diffcp_if.py:59-67).
    """
    sizes = [int(np.prod(s)) if len(s) else 1 for s in param_shapes]
    npar = len(sizes)
    col_order = list(range(npar)) if col_order is None else list(col_order)
    by_col = sorted(range(npar), key=lambda i: col_order[i])
    offs = {}
    o = 0
    for i in by_col:
        offs[i] = o
        o += sizes[i]
    ptot = o

    def evaluate(pvec):
        args = [pvec[offs[i]:offs[i] + sizes[i]].reshape(param_shapes[i], order="F") for i in range(npar)]
        out = builder(*args)
        A, b, c = out[:3]
        Pm = np.asarray(out[3], float) if len(out) > 3 else None
        return np.asarray(A, float), np.asarray(b, float), np.asarray(c, float), Pm

    A0, b0, c0, P0 = evaluate(np.zeros(ptot))
    m, n = A0.shape
    assert m == cone_rows(cones), "cone dims do not add up to the number of rows"
    dA, db, dc, dP = [], [], [], []
    for k in range(ptot):
        e = np.zeros(ptot); e[k] = 1.0
        A1, b1, c1, P1 = evaluate(e)
        dA.append(A1 - A0); db.append(b1 - b0); dc.append(c1 - c0)
        if P0 is not None:
            dP.append(P1 - P0)
    # structural pattern: any entry that is non-zero for some parameter value
    patA = (A0 != 0)
    patb = (b0 != 0)
    for k in range(ptot):
        patA |= dA[k] != 0
        patb |= db[k] != 0
    aug_pat = np.concatenate([patA, patb[:, None]], axis=1)
    indices, indptr = [], [0]
    for j in range(n + 1):
        rows = np.nonzero(aug_pat[:, j])[0]
        indices.extend(rows.tolist()); indptr.append(len(indices))
    indices = np.asarray(indices, dtype=np.int32); indptr = np.asarray(indptr, dtype=np.int32)
    colidx = np.repeat(np.arange(n + 1), np.diff(indptr))
    nnz_aug = len(indices)

    def aug_values(A, b):          # CSC data of [A_cvx | b_cvx] = [-A | b]
        aug = np.concatenate([-A, b[:, None]], axis=1)
        return aug[indices, colidx]

    rows_l, cols_l, vals_l = [], [], []
    qr, qc, qv = [], [], []
    v0 = aug_values(A0, b0)
    nz = np.nonzero(v0)[0]
    rows_l.extend(nz.tolist()); cols_l.extend([ptot] * len(nz)); vals_l.extend(v0[nz].tolist())
    nzc = np.nonzero(c0)[0]
    qr.extend(nzc.tolist()); qc.extend([ptot] * len(nzc)); qv.extend(c0[nzc].tolist())
    for k in range(ptot):
        v = aug_values(dA[k], db[k])
        nz = np.nonzero(v)[0]
        rows_l.extend(nz.tolist()); cols_l.extend([k] * len(nz)); vals_l.extend(v[nz].tolist())
        nzc = np.nonzero(dc[k])[0]
        qr.extend(nzc.tolist()); qc.extend([k] * len(nzc)); qv.extend(dc[k][nzc].tolist())
    A_map = sp.csr_array(sp.coo_array((vals_l, (rows_l, cols_l)), shape=(nnz_aug, ptot + 1)))
    q_map = sp.csr_array(sp.coo_array((qv, (qr, qc)), shape=(n + 1, ptot + 1)))
    col_offsets = [offs[i] for i in range(npar)]
    P_map = P_structure = None
    if P0 is not None:
        patP = np.triu(P0 != 0)
        for k in range(ptot):
            patP |= np.triu(dP[k] != 0)
        p_idx, p_ptr = [], [0]
        for j in range(n):
            r_ = np.nonzero(patP[:, j])[0]
            p_idx.extend(r_.tolist()); p_ptr.append(len(p_idx))
        p_idx = np.asarray(p_idx, dtype=np.int32); p_ptr = np.asarray(p_ptr, dtype=np.int32)
        p_cols = np.repeat(np.arange(n), np.diff(p_ptr))
        pr, pc, pv = [], [], []
        v0 = P0[p_idx, p_cols]
        nzp = np.nonzero(v0)[0]
        pr.extend(nzp.tolist()); pc.extend([ptot] * len(nzp)); pv.extend(v0[nzp].tolist())
        for k in range(ptot):
            v = dP[k][p_idx, p_cols]
            nzp = np.nonzero(v)[0]
            pr.extend(nzp.tolist()); pc.extend([k] * len(nzp)); pv.extend(v[nzp].tolist())
        P_map = sp.csr_array(sp.coo_array((pv, (pr, pc)), shape=(len(p_idx), ptot + 1)))
        P_structure = (p_idx, p_ptr, (n, n))
    return CanonTemplate([tuple(s) for s in param_shapes], col_offsets, A_map, q_map,
                         (indices, indptr, (m, n + 1)), dict(cones), list(var_recover), P_map=P_map, P_structure=P_structure)
