"""Randomised sweep of the engine against the CPU oracle (test infrastructure; GPU box only): random template shapes (sizes across all kernel variants, random
cone mixes, random sparsity, optionally a duplicated equality row = a rank-deficient adjoint system on every instance), random strictly feasible data.  Per shape
  * forward at eps 1e-9, acceleration off on both sides: x / y / s on the instances both sides solve; status in {solved, solved-inaccurate} where the oracle solves;
  * forward with the one-pair acceleration on both sides: x;
  * the plugin's default adjoint (search-free elimination + device-side LSQR re-solve) at the oracle's point: against the oracle's dense elimination on unflagged
    instances (1e-5), against its LSQR mode on re-solved ones; finite gradients, no instance left without one.
Iteration counts are REPORTED when they differ by more than a check interval / 1 % (slowly converging LP-like instances follow rounding-level different trajectories
over thousands of iterations) but are not a failure.  sweep() returns (failures, notes, lines)."""
import os, sys, time
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
from oracle import oracle
from kit import TIGHT_LSQR


def boundary(tpl, g, n):
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    want = np.empty((tpl.nnz_aug, g["dA"].shape[0]))
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]
    return want


def random_shape(rng, ext=False):
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 14, 15, 22, 30, 31, 40, 50, 51, 62, 75, 98, 104, 110]))
    cones = {"z": 0, "l": 0, "q": []}
    if ext:      # PSD blocks / exponential / power triples too (smaller n: the eigen-solves of the oracle are slow)
        n = int(rng.choice([2, 3, 5, 8, 13, 14, 20, 30, 40, 55]))
        kind = rng.integers(0, 3)
        if kind != 1: cones["s"] = [int(k) for k in rng.integers(2, 7, size=int(rng.integers(1, 3)))]
        if kind != 0: cones["ep"] = int(rng.integers(1, 5))
        if kind == 2 and rng.random() < 0.5: cones["p"] = [float(rng.choice([0.3, 0.5, -0.6]))]
    budget = int(rng.integers(max(n, 2), 2 * n + 12))          # rows: m >= n mostly (bounded problems)
    if rng.random() < 0.4:
        cones["z"] = int(rng.integers(1, max(2, n // 3 + 1)))
    rows = cones["z"]
    nq = int(rng.integers(0, 5)) if n > 1 else 0
    for _ in range(nq):
        d = int(rng.integers(2, max(3, min(n + 1, 26))))
        if rows + d > budget: break
        cones["q"].append(d); rows += d
    rows += P.cone_rows({k: v for k, v in cones.items() if k in ("s", "ep", "p")})
    cones["l"] = max(budget - rows, 1 if not cones["q"] else 0)
    dens = float(rng.choice([1.0, 1.0, 0.6, 0.3]))
    return n, cones, dens


def sweep(n_shapes=40, seed0=1, B=16, verbose=True, ext=False):
    rng = np.random.default_rng(seed0)
    dev = torch.device("cuda", 0)
    fails, notes, lines = [], [], []
    for it in range(n_shapes):
        n, cones, dens = random_shape(rng, ext)
        m = P.cone_rows(cones)
        pat = rng.random((m, n)) < dens
        pat[np.arange(m), rng.integers(0, n, m)] = True
        pat[rng.integers(0, m, n), np.arange(n)] = True
        seed = int(rng.integers(1 << 30))
        tag = f"shape {it}: n={n} m={m} cones={cones} density={dens} seed={seed}"
        msgs, info_msgs = [], []
        tpl = P.dense_template(n, cones, pattern=pat)
        A, b, c = P.generate(n, cones, B, seed=seed)
        A = A * pat[None]
        # b, c re-derived so that the masked problem is still strictly feasible:  b = A x0 + s0, c = -A^T y0 with interior points
        r2 = np.random.default_rng(seed + 7)
        x0 = r2.standard_normal((B, n)); s0, y0 = P._interior_point(r2, cones, B)
        dup = cones["z"] >= 2 and m > n + 1 and rng.random() < 0.5          # a duplicated equality row: rank-deficient adjoint system on every instance (flagged, re-solved by LSQR)
        if dup:
            pat[1] = pat[0]; A[:, 1, :] = A[:, 0, :]
            tpl = P.dense_template(n, cones, pattern=pat)
            tag += " dup-row"
        b = np.einsum("bij,bj->bi", A, x0) + s0; c = -np.einsum("bij,bi->bj", A, y0)
        ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=100000)
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, dev)
        A_eval, q_eval = tpl.values_from_dense(A, b, c)
        A_bm = eng.to_batch_major(torch.from_numpy(A_eval).to(dev)); q_t = torch.from_numpy(q_eval).to(dev)
        x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-9, max_iters=100000, acceleration_lookback=0)))
        torch.cuda.synchronize()
        st = status.cpu().numpy(); its = iters.cpu().numpy()
        if not np.isin(st[ref["status"] == 1], (1, 2)).all(): msgs.append(f"status {st} vs {ref['status']}")
        elif not (st == ref["status"]).all(): info_msgs.append(f"status {st} vs {ref['status']}")
        ok = (st == 1) & (ref["status"] == 1)
        if ok.any():
            for nm, got, want in (("x", x, ref["x"]), ("y", y, ref["y"]), ("s", s, ref["s"])):
                err = (np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1)))[ok]
                if err.max() > 2e-6: msgs.append(f"{nm} err {err.max():.2e}")
            if (np.abs(its - ref["iters"])[ok] > np.maximum(25, 0.01 * ref["iters"][ok])).any(): info_msgs.append(f"iters {its[ok]} vs {ref['iters'][ok]}")
        # accelerated run (one-pair history on both sides)
        ref_a = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=100000, aa_mem=1)
        xa, ya, sa, ita, sta, _ = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-9, max_iters=100000, acceleration_lookback=1)))
        torch.cuda.synchronize()
        sta = sta.cpu().numpy()
        oka = (sta == 1) & (ref_a["status"] == 1)
        if not np.isin(sta[ref_a["status"] == 1], (1, 2)).all(): msgs.append(f"accelerated status {sta} vs {ref_a['status']}")
        if oka.any():
            err = (np.abs(xa.cpu().numpy() - ref_a["x"]).max(axis=1) / (1 + np.abs(ref_a["x"]).max(axis=1)))[oka]
            if err.max() > 2e-6: msgs.append(f"accelerated x err {err.max():.2e}")
            if eng.last_acceleration and np.abs(ita.cpu().numpy() - ref_a["iters"])[oka].max() > 50: info_msgs.append("accelerated iteration counts differ")
        # default adjoint at the oracle's point
        nflag = 0
        if ok.sum() >= 2:
            idx = np.nonzero(ok)[0]
            dx = rng.standard_normal((len(idx), n)); dy = rng.standard_normal((len(idx), m))
            xr, yr, sr = (torch.from_numpy(np.ascontiguousarray(ref[k][idx])).to(dev) for k in ("x", "y", "s"))
            A_sub = A_bm[torch.from_numpy(idx).to(dev)].contiguous(); q_sub = q_t[:, torch.from_numpy(idx).to(dev)].contiguous()
            dA, dq, adj = eng.vjp(A_sub, xr, yr, sr, torch.from_numpy(dx).to(dev), torch.from_numpy(dy).to(dev), path="per_instance", lsqr=TIGHT_LSQR, q_eval=q_sub)
            torch.cuda.synchronize()
            a = adj.cpu().numpy(); got = dA.cpu().numpy(); gq = dq.cpu().numpy()
            if not (np.isfinite(got).all() and np.isfinite(gq).all()): msgs.append("non-finite gradients")
            if (a & 2).any(): msgs.append(f"instances without gradient: adj {a}")
            fl = (a & 8) != 0; nflag = int(fl.sum())
            reg = ~fl & ((a & 4) == 0)
            if reg.any():
                gd = oracle.adjoint_batch(A[idx], b[idx], c[idx], cones, ref["x"][idx], ref["y"][idx], ref["s"][idx], dx, dy, mode="dense")
                wd = boundary(tpl, gd, n)
                el = (np.abs(got - wd).max(axis=0) / (1 + np.abs(wd).max(axis=0)))[reg]
                if (el > 1e-5).any(): msgs.append(f"adjoint (regular) err {el.max():.2e} on {int((el > 1e-5).sum())} instances")
                eq = (np.abs(gq[:n] - gd["dc"].T).max(axis=0) / (1 + np.abs(gd["dc"]).max(axis=1)))[reg]
                if (eq > 1e-5).any(): msgs.append(f"dq (regular) err {eq.max():.2e}")
            if fl.any():
                j = np.nonzero(fl)[0]
                gl = oracle.adjoint_batch(A[idx][j], b[idx][j], c[idx][j], cones, ref["x"][idx][j], ref["y"][idx][j], ref["s"][idx][j], dx[j], dy[j], mode="lsqr",
                                          lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
                wl = boundary(tpl, gl, n)
                el = np.abs(got[:, j] - wl).max(axis=0) / (1 + np.abs(wl).max(axis=0))
                if el.max() > 5e-3 or np.median(el) > 1e-4: msgs.append(f"adjoint (re-solved) err max {el.max():.2e} median {np.median(el):.2e}")
        info = eng.launch_info()
        line = (("FAIL " if msgs else "ok   ") + tag + f" | solved {int(ok.sum())}/{B} flagged {nflag} fwd_mode {info.get('fwd_mode')} bwd_mode {info.get('bwd_mode')}"
                + "".join(" | " + mm for mm in msgs) + "".join(" | note: " + mm[:160] for mm in info_msgs))
        lines.append(line)
        if verbose: print(line, flush=True)
        if msgs: fails.append(tag)
        if info_msgs: notes.append(tag)
        del eng
    return fails, notes, lines


def sweep_shared(n_shapes=20, seed0=1, B=12, verbose=True):
    """Shared-A templates (only b, c vary over the batch: the persistent kernels k_sa_fwd / k_sa_lsqr, or the batch-GEMM fallback when a column has no
    single-entry row): canonicalisation-shaped structure -- a few dense rows (equalities / inequalities), a bound on every variable, second-order and PSD blocks
    made of single-entry rows.  Forward against the oracle at eps 1e-9; the adjoint (diffcp's LSQR on the full system) against the oracle's LSQR mode under the
    same tight rule.  CE_CONST_A=1 must be in the environment BEFORE the first engine is built (the caller sets it)."""
    rng = np.random.default_rng(seed0)
    dev = torch.device("cuda", 0)
    fails, notes, lines = [], [], []
    for it in range(n_shapes):
        n = int(rng.choice([6, 10, 16, 24, 40, 64, 100, 150]))
        rz, rl = int(rng.integers(0, 4)), int(rng.integers(0, 5))
        bounded = rng.random() < 0.8                      # a bound row on every variable (else some columns have no single-entry row: fallback path)
        qdims, sdims, used = [], [], 0
        for _ in range(int(rng.integers(0, 4))):
            d = int(rng.integers(2, 9))
            if used + d <= n: qdims.append(d); used += d
        if n - used >= 6 and rng.random() < 0.4:
            k = int(rng.choice([2, 3])); sdims.append(k); used += k * (k + 1) // 2
        nb = n if bounded else max(n // 2, 1)
        cones = {"z": rz, "l": rl + nb, "q": qdims, "s": sdims}
        m = P.cone_rows(cones)
        A0 = np.zeros((m, n)); r = 0
        for _ in range(rz + rl):
            cols = rng.random(n) < rng.choice([1.0, 0.5]); cols[rng.integers(0, n)] = True
            A0[r, cols] = rng.standard_normal(int(cols.sum())) / np.sqrt(n); r += 1
        bcols = np.arange(n) if bounded else rng.choice(n, nb, replace=False)
        for j in bcols: A0[r, j] = -(0.5 + rng.random()); r += 1
        var = 0
        for d in qdims:
            for _ in range(d): A0[r, var] = -(0.5 + rng.random()); r += 1; var += 1
        for k in sdims:
            for _ in range(k * (k + 1) // 2): A0[r, var] = -1.0; r += 1; var += 1
        assert r == m
        seed = int(rng.integers(1 << 30))
        tag = f"shared shape {it}: n={n} m={m} cones={cones} bounded={bounded} seed={seed}"
        msgs, info_msgs = [], []
        r2 = np.random.default_rng(seed)
        x0 = r2.standard_normal((B, n)) * 0.5; s0, y0 = P._interior_point(r2, cones, B)
        A = np.broadcast_to(A0, (B,) + A0.shape).copy()
        b = x0 @ A0.T + s0; c = -(y0 @ A0)
        tpl = P.dense_template(n, cones, pattern=(A0 != 0))
        rank_A = int(np.linalg.matrix_rank(A0))
        if rank_A < n: tag += f" rank(A)={rank_A}"
        ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=200000)
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, dev)
        A_eval, q_eval = tpl.values_from_dense(A, b, c)
        A_bm = eng.to_batch_major(torch.from_numpy(A_eval).to(dev)); q_t = torch.from_numpy(q_eval).to(dev)
        x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-9, max_iters=200000, acceleration_lookback=0)))
        torch.cuda.synchronize()
        path = f"{eng.last_path}/{getattr(eng, 'last_const_a_kernel', None)}"
        st = status.cpu().numpy()
        if eng.last_path != "const_a": msgs.append(f"path {eng.last_path}")
        if not np.isin(st[ref["status"] == 1], (1, 2)).all(): msgs.append(f"status {st} vs {ref['status']}")
        elif not (st == ref["status"]).all(): info_msgs.append(f"status {st} vs {ref['status']}")
        ok = (st == 1) & (ref["status"] == 1)
        if ok.any():
            for nm, got, want in (("x", x, ref["x"]), ("y", y, ref["y"]), ("s", s, ref["s"])):
                if nm == "x" and rank_A < n: continue          # (columns that only meet a few dense rows: A has a null space, the primal solution is not unique)
                err = (np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1)))[ok]
                if err.max() > 2e-6: msgs.append(f"{nm} err {err.max():.2e}")
        if ok.sum() >= 2:
            idx = np.nonzero(ok)[0]
            dx = rng.standard_normal((len(idx), n)); dy = rng.standard_normal((len(idx), m))
            xr, yr, sr = (torch.from_numpy(np.ascontiguousarray(ref[k][idx])).to(dev) for k in ("x", "y", "s"))
            A_sub = A_bm[torch.from_numpy(idx).to(dev)].contiguous(); q_sub = q_t[:, torch.from_numpy(idx).to(dev)].contiguous()
            dA, dq, adj = eng.vjp(A_sub, xr, yr, sr, torch.from_numpy(dx).to(dev), torch.from_numpy(dy).to(dev), path="const_a", lsqr=TIGHT_LSQR, q_eval=q_sub)
            torch.cuda.synchronize()
            a = adj.cpu().numpy(); got = dA.cpu().numpy(); gq = dq.cpu().numpy()
            if not (np.isfinite(got).all() and np.isfinite(gq).all()): msgs.append("non-finite gradients")
            gl = oracle.adjoint_batch(A[idx], b[idx], c[idx], cones, ref["x"][idx], ref["y"][idx], ref["s"][idx], dx, dy, mode="lsqr",
                                      lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
            wl = boundary(tpl, gl, n)
            el = np.maximum(np.abs(got - wl).max(axis=0) / (1 + np.abs(wl).max(axis=0)), np.abs(gq[:n] - gl["dc"].T).max(axis=0) / (1 + np.abs(gl["dc"]).max(axis=1)))
            conv = (a & 1) == 0                               # (LSQR runs that stopped at the iteration limit are reported, not compared)
            if conv.any() and (el[conv].max() > 5e-3 or np.median(el[conv]) > 1e-5): msgs.append(f"adjoint err max {el[conv].max():.2e} median {np.median(el[conv]):.2e}")
            if not conv.all(): info_msgs.append(f"{int((~conv).sum())} LSQR runs at the iteration limit")
        line = (("FAIL " if msgs else "ok   ") + tag + f" | solved {int(ok.sum())}/{B} path {path}" + "".join(" | " + mm for mm in msgs) + "".join(" | note: " + mm[:160] for mm in info_msgs))
        lines.append(line)
        if verbose: print(line, flush=True)
        if msgs: fails.append(tag)
        if info_msgs: notes.append(tag)
        del eng
    return fails, notes, lines
