"""The two solver backends every layer-level parity test runs on: the CPU oracle (tests/oracle_layer.py; `-m "not gpu"`) and the HIP engine
through cvxpylayers_amd.torch.CvxpyLayer (`-m gpu`).  Both return callables  layer(*params) -> tuple of variables  with the frontend's
recovery conventions (Fortran reshape, symmetric-primal unpacking, batch axis only when a parameter is batched)."""
import numpy as np
import pytest
import torch

TIGHT = dict(eps=1e-10, max_iters=200000)


def cpu_layer(template, **kw):
    """(params...) -> tuple of variables, solver = CPU oracle; same variable recovery conventions as the frontend for the cases used here."""
    from oracle_layer import OracleLayer
    L = OracleLayer(template, **{**TIGHT, **kw})

    def call(*params):
        primal, _ = L(*[p.double() for p in params])
        outs = []
        for v in template.var_recover:
            d = primal[:, v.primal]
            if v.unpack_fn == "svec_primal":
                k = v.shape[0]; iu = np.triu_indices(k)
                M = torch.zeros(d.shape[0], k, k, dtype=torch.float64)
                M[:, iu[0], iu[1]] = d; M = M + M.transpose(1, 2) - torch.diag_embed(torch.diagonal(M, dim1=1, dim2=2))
                outs.append(M)
            else:
                outs.append(d.reshape((d.shape[0],) + tuple(reversed(v.shape))).permute(0, *range(len(v.shape), 0, -1)) if len(v.shape) > 1
                            else d.reshape((d.shape[0],) + tuple(v.shape)))
        batched = any(p.dim() == len(s) + 1 for p, s in zip(params, template.param_shapes))
        return tuple(o if batched else o[0] for o in outs)
    return call


def gpu_layer(template, **kw):
    from cvxpylayers_amd.torch import CvxpyLayer
    layer = CvxpyLayer(template=template, solver_args={**TIGHT, **kw})
    return lambda *params: layer(*[p.cuda() if p.device.type != "cuda" else p for p in params])


BACKENDS = [pytest.param(cpu_layer, id="oracle"), pytest.param(gpu_layer, id="engine", marks=pytest.mark.gpu)]


