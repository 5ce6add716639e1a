"""GPU debug: the notebook cases through the engine with every intermediate printed (tests/test_notebook_golden.py says only pass / fail)."""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import notebook_cases as nc
from cvxpylayers_amd.torch import CvxpyLayer
GOLD = os.path.join(ROOT, "tests", "golden")
np.set_printoptions(precision=6, linewidth=200)
warnings.simplefilter("always")

def ot():
    f = np.load(os.path.join(GOLD, "ref_notebook_ot.npz"))
    for eps in (1e-6, 1e-10):
        layer = CvxpyLayer(template=nc.ot_template(3, 3), solver_args=dict(eps=eps, max_iters=200000))
        x = torch.tensor(f["x"], requires_grad=True); y = torch.tensor(f["y"], requires_grad=True)
        a = torch.tensor(f["a"]); b = torch.tensor(f["b"]); e = torch.tensor(f["eps"])
        C = (x[:, None] - y[None, :]).pow(2)
        P, = layer(C.cuda(), a.cuda(), b.cuda(), e.cuda())
        print("OT eps", eps, "iters", layer.info["iters"].cpu().numpy(), "status", layer.info["status"].cpu().numpy(), "resid", layer.info["resid"].cpu().numpy())
        Pc = P.cpu(); print("P - notebook", np.abs(Pc.detach().numpy() - f["P"]).max())
        xs = torch.tensor(f["x"], requires_grad=True); ys = torch.tensor(f["y"], requires_grad=True)
        Ps = nc.sinkhorn((xs[:, None] - ys[None, :]).pow(2), a, b, e); Ps[2, 2].backward()
        print("P - sinkhorn", np.abs(Pc.detach().numpy() - Ps.detach().numpy()).max())
        Pc[2, 2].backward()
        print("x.grad", x.grad.numpy(), "want", f["x_grad"], "sinkhorn", xs.grad.numpy())
        print("y.grad", y.grad.numpy(), "want", f["y_grad"], "sinkhorn", ys.grad.numpy())

def supply():
    f = np.load(os.path.join(GOLD, "ref_notebook_supply.npz"))
    layer = CvxpyLayer(template=nc.supply_chain_template(), solver_args=dict(eps=1e-9, max_iters=200000, acceleration_lookback=0))
    stats = []
    def policy(x, P, q):
        u, = layer(x, P, q)
        stats.append((int(layer.info["iters"].max()), int(layer.info["status"].min())))
        return u
    loss = nc.supply_chain_sim(policy, device="cuda")
    torch.manual_seed(0)
    P_sqrt = torch.eye(4, dtype=torch.float64, device="cuda", requires_grad=True)
    q = (-3.0 * torch.ones(4, 1, dtype=torch.float64, device="cuda")).requires_grad_(True)
    opt = torch.optim.SGD([P_sqrt, q], lr=0.05)
    for epoch, want in enumerate(f["valid"]):
        with torch.no_grad():
            v = loss([P_sqrt, q], 20, 1, seed=0).item()
        torch.manual_seed(epoch); opt.zero_grad()
        stats.clear()
        c = loss([P_sqrt, q], 20, 5, seed=epoch + 1); c.backward()
        print("epoch", epoch, "valid", v, "want", want, "diff", v - want, "train cost", c.item(), "max iters/min status", max(s[0] for s in stats), min(s[1] for s in stats),
              "|gP|", P_sqrt.grad.norm().item(), "|gq|", q.grad.norm().item())
        print("   gq", q.grad.cpu().numpy().ravel())
        opt.step()

if __name__ == "__main__":
    which = sys.argv[1:] or ["ot", "supply"]
    if "ot" in which: ot()
    if "supply" in which: supply()
