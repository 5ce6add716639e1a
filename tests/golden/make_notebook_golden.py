"""Generates tests/golden/ref_notebook_*.npz  --  run from the repo root (build container only: reads /root/reference):
    python tests/golden/make_notebook_golden.py

These fixtures ARE outputs of the reference: numbers that the real cvxpylayers -> diffcp -> SCS stack printed into the example
notebooks the reference ships (/root/reference/examples/torch/*.ipynb, stored cell outputs).  This script only PARSES them out of the
notebook JSON (nothing is solved here) and stores them next to the notebooks' inputs, which it regenerates at full precision with the
notebooks' own seeds (torch / numpy generators reproduce the printed 4-digit inputs; asserted below).  The layers themselves are
restated cvxpy-free in tests/notebook_cases.py; tests/test_notebook_golden.py replays the fixtures through the oracle (CPU) and the
HIP engine (GPU).

  ref_notebook_ot.npz        optimal_transport.ipynb cells 6-14: entropic optimal transport (9 exponential cones); stored: x, y (printed
                             inputs, cells 9-10), P (cell 11: forward), x.grad, y.grad of P[2,2] (cells 13-14: diffcp's adjoint)
  ref_notebook_lqr.npz       lqr.ipynb cells 2-4: the LQR value-function SDP (PSD cones of order 6 and 4) solved by SCS through CVXPY;
                             stored: optimal value (17 digits), P_lqr (8 decimals)
  ref_notebook_tutorial.npz  tutorial.ipynb cells 16-17: fit_lr(Xtrain, ytrain, 0, 0) -> (a, b) (SOC program)
  ref_notebook_supply.npz    supply_chain.ipynb cell 9: closed-loop cost of the baseline policy (17 digits: 20 sequential forward solves) and
                             the validation costs after each of the first SGD epochs (5 digits: each one forward + diffcp backward through
                             20 time steps x batch 5, so they pin the gradients as well)
  ref_notebook_denoise.npz   signal_denoising.ipynb cells 15-18: mean squared error of the smoothing layer (two SOCs of 102 and 101 rows, n = 102) over the
                             500 training signals for ten values of lambda, and over the 100 validation signals at the best lambda (4 decimals each)
"""
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NB = "/root/reference/examples/torch"
_NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def cells(name):
    return json.load(open(os.path.join(NB, name)))["cells"]


def out_text(cell):
    s = ""
    for o in cell.get("outputs", []):
        if "text" in o:
            s += "".join(o["text"])
        elif "data" in o and "text/plain" in o["data"]:
            s += "".join(o["data"]["text/plain"])
    return s


def numbers(s):
    s = re.sub(r"(grad_fn|dtype|requires_grad)=[^,)]*", "", s)
    return np.array([float(v) for v in re.findall(_NUM, s)])


def find_cell(cs, needle):
    hits = [c for c in cs if c["cell_type"] == "code" and needle in "".join(c["source"])]
    assert len(hits) == 1, (needle, len(hits))
    return hits[0]


def ot():
    cs = cells("optimal_transport.ipynb")
    x_p = numbers(out_text([c for c in cs if "".join(c["source"]).strip() == "x"][0]))
    y_p = numbers(out_text([c for c in cs if "".join(c["source"]).strip() == "y"][0]))
    P_p = numbers(out_text(find_cell(cs, "print(P)"))).reshape(3, 3)
    xg = numbers(out_text(find_cell(cs, "x.grad")))
    yg = numbers(out_text(find_cell(cs, "y.grad")))
    assert "torch.manual_seed(6)" in "".join(find_cell(cs, "torch.manual_seed").get("source"))
    torch.set_default_dtype(torch.double)
    torch.manual_seed(6)                                          # cell 6
    x = torch.randn(3); y = torch.randn(3)
    torch.set_default_dtype(torch.float32)
    assert np.abs(x.numpy() - x_p).max() < 5.1e-5 and np.abs(y.numpy() - y_p).max() < 5.1e-5, "seeded inputs != printed inputs"
    np.savez(os.path.join(HERE, "ref_notebook_ot.npz"), x=x.numpy(), y=y.numpy(), x_printed=x_p, y_printed=y_p, a=np.full(3, 1 / 3), b=np.full(3, 1 / 3),
             eps=np.array([1.0]), P=P_p, x_grad=xg, y_grad=yg)
    print("ot", P_p, xg, yg)


def lqr():
    cs = cells("lqr.ipynb")
    val = numbers(out_text(find_cell(cs, "print(result)")))
    P = numbers(out_text([c for c in cs if "".join(c["source"]).strip() == "P_lqr"][0])).reshape(4, 4)
    assert "np.random.seed(0)" in "".join(find_cell(cs, "np.random.seed").get("source"))
    np.random.seed(0)                                             # cell 2
    n, m = 4, 2
    A = np.random.randn(n, n)
    A /= np.max(np.abs(np.linalg.eig(A)[0]))
    B = np.random.randn(n, m)
    np.savez(os.path.join(HERE, "ref_notebook_lqr.npz"), A=A, B=B, Q0=np.eye(n), R0=np.eye(m), W=0.25 * np.eye(n), value=val, P_lqr=P)
    print("lqr", val, P)


def tutorial():
    from sklearn.model_selection import train_test_split
    cs = cells("tutorial.ipynb")
    ab = numbers(out_text(find_cell(cs, "fit_lr(Xtrain, ytrain, torch.zeros(1), torch.zeros(1))")))
    assert ab.shape == (2,)
    torch.manual_seed(0); np.random.seed(0)                       # cell 16, second half
    n, N = 1, 60
    X = np.random.randn(N, n)
    theta = np.random.randn(n)
    y = X @ theta + .5 * np.random.randn(N)
    Xtrain, Xtest, ytrain, ytest = train_test_split(X, y, test_size=.5)
    np.savez(os.path.join(HERE, "ref_notebook_tutorial.npz"), Xtrain=Xtrain, ytrain=ytrain, a=ab[:1], b=ab[1:])
    print("tutorial", ab)


def supply():
    cs = cells("supply_chain.ipynb")
    txt = out_text(find_cell(cs, "Baseline cost"))
    base = float(re.search(r"Baseline cost:\s*(" + _NUM + ")", txt).group(1))
    valid = [float(v) for v in re.findall(r"epoch \d+, valid (" + _NUM + ")", txt)]
    np.savez(os.path.join(HERE, "ref_notebook_supply.npz"), baseline=np.array(base), valid=np.array(valid[:8]),
             time_horizon=20, batch_size=5, lr=0.05)
    print("supply", base, valid[:8])


def denoise():
    """signal_denoising.ipynb cells 3, 14-18: the one-parameter smoothing layer  min ||x - y||^2 + lam ||diff(y)||^2  evaluated by the reference on the WHOLE
    training set (batch of 500 signals of length 100) for ten values of lam, and on the validation set (100 signals) at the best one: mean squared errors
    printed to 4 decimals.  The data are regenerated with the notebook's seeds; that they ARE the notebook's data is checked on the printed value for
    lam = 0 (there y = x whatever the solver does: the printed mse is a function of the data alone)."""
    import math
    cs = cells("signal_denoising.ipynb")
    txt = out_text(find_cell(cs, "for value in tqdm(lambda_values)"))
    lams = [float(v) for v in re.findall(r"lambda\s+tensor\((" + _NUM + r")\)", txt)]
    mses = [float(v) for v in re.findall(r"mse tensor\((" + _NUM + r")\)", txt)]
    assert len(lams) == 10 and len(mses) == 10, (lams, mses)
    best = numbers(out_text(find_cell(cs, "print(best_lambda)")))[0]
    lowest = numbers(out_text(find_cell(cs, "print(lowest_loss)")))[0]
    val_mse = numbers(out_text(find_cell(cs, "print(one_param_mse)")))[0]
    src = "".join(find_cell(cs, "torch.random.manual_seed(0)")["source"])
    assert "np.random.seed(0)" in src and "N_train = 500" in src and "n = 100" in src
    torch.set_default_dtype(torch.double)
    torch.random.manual_seed(0); np.random.seed(0)                 # cell 3
    N_train, N_val, n = 500, 100, 100
    Sigma_sqrt = 0.1 * np.random.randn(n, n)
    Sigma = Sigma_sqrt.T @ Sigma_sqrt
    normal = torch.distributions.MultivariateNormal(loc=torch.zeros(n), covariance_matrix=torch.tensor(Sigma))
    eval_pts = torch.linspace(0, 2 * math.pi, n)
    X, bs = [], []
    for i in range(N_train + N_val):
        b = np.random.uniform(low=1, high=3)
        x = torch.cos(b * eval_pts).clone(); x += normal.sample()
        X.append(x.numpy()); bs.append(b)
    torch.set_default_dtype(torch.float32)
    X = np.stack(X); bs = np.asarray(bs)
    Y = np.cos(bs[:, None] * eval_pts.double().numpy()[None, :])
    assert abs(((X[:N_train] - Y[:N_train]) ** 2).mean() - mses[0]) < 5.1e-5, "seeded data != the notebook's data (lam = 0 is y = x)"
    lam_exact = np.linspace(0, 20, 10)                             # torch.linspace(0, 20, 10): the printed values are these, rounded
    assert np.abs(lam_exact - np.asarray(lams)).max() < 5.1e-5 and abs(best - lam_exact[7]) < 5.1e-5 and abs(lowest - min(mses)) < 1e-12
    np.savez_compressed(os.path.join(HERE, "ref_notebook_denoise.npz"), X=X, b=bs, eval_pts=eval_pts.double().numpy(), lams=lam_exact, mse=np.asarray(mses),
                        best=np.array(lam_exact[7]), val_mse=np.array(val_mse), N_train=N_train)
    print("denoise", lams, mses, best, val_mse)


def adp():
    """convex_approximate_dynamic_programming.ipynb cell 3: the 100 `(iter k) loss: v` lines of the training run (printed with %g: 6 significant digits).
    Every line after the first depends on the GRADIENTS of all steps before it (SGD with momentum through 200 chained policy solves per step)."""
    cs = cells("convex_approximate_dynamic_programming.ipynb")
    cell = find_cell(cs, "results = train(iters=100)")
    src = "".join(cell["source"])
    assert "lr=.02, momentum=.9" in src and "def eval_loss(N=8, T=25)" in src and "torch.manual_seed(1)" in src
    losses = [float(v) for v in re.findall(r"\(iter \d+\) loss: (" + _NUM + ")", out_text(cell))]
    assert len(losses) == 100, len(losses)
    np.savez(os.path.join(HERE, "ref_notebook_adp.npz"), losses=np.asarray(losses))
    print("adp", losses[:5], "...", losses[-1])


def monotone():
    """monotonic_output_regression.ipynb cells 5-6, 9-12: numbers the reference printed with full precision -- the validation loss of the least-squares fit
    (a function of the DATA, whose targets Y, Yval are outputs of the layer on batches of 100 and 50), the loss of the layer at the least-squares and at the
    true parameters, and the first validation loss of the training run (theta = 0).  The training trace itself is not replayable: its batches come from
    DataLoader(shuffle=True), whose use of the global generator changed between torch releases."""
    cs = cells("monotonic_output_regression.ipynb")
    one = lambda src: numbers(out_text([c for c in cs if c["cell_type"] == "code" and "".join(c["source"]).strip() == src][0]))[0]
    first_val = float(re.search(r"001 \| (" + _NUM + ")", out_text(find_cell(cs, "val_losses, train_losses = fit("))).group(1))
    src = "".join(find_cell(cs, "def get_data(N, n, m, theta)")["source"])
    assert "torch.manual_seed(0)" in src and "get_data(100, n, m, theta_true)" in src and "get_data(50, n, m, theta_true)" in src
    np.savez(os.path.join(HERE, "ref_notebook_monotone.npz"), lstsq_val_loss=np.array(one("lstsq_val_loss")), bayes_val_loss=np.array(one("bayes_val_loss")),
             train_loss_lstsq=np.array(one("print(loss(X, Y, theta_lstsq).item())")), first_val_loss=np.array(first_val), n=20, m=10)
    print("monotone", one("lstsq_val_loss"), one("bayes_val_loss"), one("print(loss(X, Y, theta_lstsq).item())"), first_val)


def constrained_lqr():
    """constrained_lqr.ipynb cells 13-16: the closed-loop cost of the LQR value function used as a control-Lyapunov policy (17 digits; 100 sequential batch-6 solves at eps 1e-8) and
    the `it: k, loss: v` lines of the training run (3 decimals).  Both start from sqrtm(P_lqr), P_lqr = the notebook's own CVXPY / SCS solve of the LQR SDP, which the notebook asserts
    against the Riccati solution only to atol 1e-3 (its printed optimal value is 6.4e-5 from tr(P_are W)); the test starts from the Riccati solution and carries that slack."""
    cs = cells("constrained_lqr.ipynb")
    clf = numbers(out_text([c for c in cs if c["cell_type"] == "code" and "".join(c["source"]).strip() == "clf_lqr, clf_lb"][0]))
    txt = out_text(find_cell(cs, "opt = torch.optim.SGD([P_sqrt, q], lr=.1)"))
    losses = [float(v) for v in re.findall(r"it: \d+, loss: (" + _NUM + "), dist", txt)]
    val = numbers(out_text(find_cell(cs, "P_lqr = P.value")))[0]
    src = "".join(find_cell(cs, "np.random.seed(1)")["source"])
    assert "n, m = 8, 2" in src and "u_max = .1" in src and len(losses) == 100
    np.savez(os.path.join(HERE, "ref_notebook_clqr.npz"), clf_lqr=np.array(clf[0]), losses=np.asarray(losses[:10]), sdp_value=np.array(val))
    print("constrained_lqr", clf[0], losses[:6], val)


if __name__ == "__main__":
    ot(); lqr(); tutorial(); supply(); denoise(); adp(); monotone(); constrained_lqr()
