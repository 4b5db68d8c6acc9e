"""numpy mirror of the symmetric eigen-path of gem_amd/csrc/hope.hip (sym_filter_svd): same steps, same constants, fp32 blocks and
fp64 small matrices.  Test infrastructure only (the product path is the HIP kernel): it was used to design the solver on the CPU and it
pins the algorithm's logic -- two-sided selection by |f(lambda)|, cut-off from the middle of the oversampling columns, degree caps,
locking, in-filter deflation -- against scipy in tests/test_hope_sym_mirror_cpu.py.

For A = A^T:  S = (I - beta A)^-1 beta A = f(A), f(x) = beta x / (1 - beta x)  (gem/embedding/hope.py:28-33 forms S densely and calls
svds): singular value |f(lambda)|, right vector q, left vector sign(f(lambda)) q."""
import numpy as np


def _orth_scaled(Y, passes):
    """CholeskyQR on the column-normalised Gram matrix; eigen fallback drops directions below 1e-6 relative energy (orth_scaled)."""
    for _ in range(passes):
        G = (Y.T @ Y).astype(np.float64)
        d = 1.0 / np.sqrt(np.maximum(np.diag(G), 1e-300))
        G = G * d[:, None] * d[None, :]
        try:
            R = np.linalg.cholesky(G).T
            if (np.diag(R) ** 2 <= 1e-5).any():
                raise np.linalg.LinAlgError
            C = np.linalg.inv(R)
        except np.linalg.LinAlgError:
            w, Z = np.linalg.eigh(G)
            keep = w > 1e-6 * max(w[-1], 0.0)
            C = Z[:, keep][:, ::-1] / np.sqrt(w[keep][::-1])
        Y = Y @ (C * d[:, None]).astype(np.float32)
    return Y


def _chol_coef(G):
    """C with (Y C)^T (Y C) = I for G = Y^T Y, through the column-normalised Gram matrix; None when a pivot is lost (orth_scaled / the fused step)."""
    d = 1.0 / np.sqrt(np.maximum(np.diag(G), 1e-300))
    Gs = G * d[:, None] * d[None, :]
    try:
        R = np.linalg.cholesky(Gs).T
    except np.linalg.LinAlgError:
        return None
    if (np.diag(R) ** 2 <= 1e-5).any():
        return None
    return np.linalg.inv(R) * d[:, None]


def sym_filter_svd(A, beta, k, oversample=16, tol=1e-5, max_cycles=60, amp=1e4, amp0=1e3, max_degree=32, seed=0, L=None, trace=None, fused_rr=True):
    """A: scipy CSR float32, symmetric.  Returns (sigma descending [k], U [n,k], V [n,k], info).
    fused_rr (the kernel's default since round 4): the second CholeskyQR pass is not applied to the block; G2 = Y1^T Y1 and H1 = Y1^T A Y1 come from
    one round trip, C2 = chol(G2)^-1 and C2^T H1 C2 are formed in fp64 and the Ritz rotation uses the coefficients C2 W."""
    n = A.shape[0]
    f = lambda x: beta * x / (1.0 - beta * x)
    b = min(k + oversample, n)
    rng = np.random.RandomState(seed)
    if L is None:                                   # hope_setup: power iteration on A^T A, + 10 %
        x = 1.0 + 0.37 * np.sin(12.9898 * (np.arange(n) + 1.0))
        rho = 0.0
        for it in range(40):
            z = A.T @ (A @ x)
            prev, rho = rho, (np.dot(z, z) / np.dot(x, x)) ** 0.25
            x = z / np.linalg.norm(z)
            if it >= 4 and abs(rho - prev) <= 1e-3 * rho:
                break
        L = 1.1 * rho
    info = {'spmm': 0, 'columns': 0, 'cycles': 0, 'projections': 0, 'L': L}

    def spmm(X):
        info['spmm'] += 1; info['columns'] += X.shape[1]
        return A @ X

    def cheb(X, m, c, e, Q, q):
        Y0, Y1 = X, np.float32(1 / e) * spmm(X) + np.float32(-c / e) * X
        for j in range(2, m + 1):
            if Q.shape[1] and (j - 1) % q == 0:
                Y0 = Y0 - Q @ (Q.T @ Y0); Y1 = Y1 - Q @ (Q.T @ Y1); info['projections'] += 2
            Y0, Y1 = Y1, np.float32(2 / e) * spmm(Y1) + np.float32(-2 * c / e) * Y1 - Y0
        return Y1

    V = _orth_scaled(rng.randn(n, b).astype(np.float32), 2)
    Q = np.zeros((n, 0), np.float32); qlam = []
    lo, hi, tau_prev = -L, 0.5 * L, 0.0
    lock_tol = 0.1 * np.sqrt(max(tol, 1e-12)); b_min = min(b, oversample + 2)
    sig_old = np.zeros(k); th = np.zeros(0); converged = False
    for cyc in range(max_cycles):
        info['cycles'] = cyc + 1
        c, e = 0.5 * (hi + lo), 0.5 * (hi - lo)
        tmax = max(L - c, c + L) / e
        rho = tmax + np.sqrt(max(tmax * tmax - 1, 0.0))
        q = int(max(1, np.floor(np.log(1e5) / np.log(max(rho, 1.0001)))))
        rho_m = rho
        if Q.shape[1] and cyc > 0 and len(th):
            ta = min(tmax, max(1.0, 1.02 * np.abs(th - c).max() / e))
            rho_m = ta + np.sqrt(max(ta * ta - 1, 0.0))
        m = int(max(2, min(max_degree, np.floor(np.log(amp0 if cyc == 0 else amp) / np.log(max(rho_m, 1.0001))))))
        V = cheb(V, m, c, e, Q, q)
        C2 = None
        if fused_rr:
            V = _orth_scaled(V - Q @ (Q.T @ V), 1) if Q.shape[1] else _orth_scaled(V, 1)
            if Q.shape[1]:
                V = V - Q @ (Q.T @ V)
            if Q.shape[1] + V.shape[1] < k + 1:
                break
            B = spmm(V)
            C2 = _chol_coef((V.T @ V).astype(np.float64))
            if C2 is None:
                V = _orth_scaled(V, 1)
        elif Q.shape[1]:
            V = _orth_scaled(V - Q @ (Q.T @ V), 1)
            V = _orth_scaled(V - Q @ (Q.T @ V), 1)
        else:
            V = _orth_scaled(V, 2)
        if Q.shape[1] + V.shape[1] < k + 1:
            break
        if C2 is None:
            B = spmm(V)
        H = (V.T @ B).astype(np.float64); H = 0.5 * (H + H.T)
        if C2 is not None:
            H = C2.T @ H @ C2; H = 0.5 * (H + H.T)
        ev, Z = np.linalg.eigh(H)
        order = np.argsort(-np.abs(f(ev)), kind='stable')
        th, C = ev[order], Z[:, order]
        if C2 is not None:
            C = C2 @ C
        R = B @ C.astype(np.float32) + V @ (-(C * th)).astype(np.float32)
        V = V @ C.astype(np.float32)
        res = np.linalg.norm(R.astype(np.float64), axis=0)
        sig = np.sort(np.abs(f(np.concatenate([np.asarray(qlam, np.float64), th]))))[::-1][:k]
        change = np.abs(sig - sig_old).max() / sig[0]; sig_old = sig
        nl, ma = Q.shape[1], V.shape[1]; want = k - nl
        rmax = (res[:min(want, ma)] / np.maximum(np.abs(th[:min(want, ma)]), 1e-3 * L)).max()
        if trace is not None:
            trace.append(dict(cycle=cyc, degree=m, lo=lo, hi=hi, locked=nl, active=ma, change=change, residual=rmax))
        if cyc > 0 and change < tol and rmax < 1e-2:
            converged = True
            break
        newl = 0
        while newl < want - 1 and newl < ma - b_min and res[newl] < lock_tol * abs(th[newl]):
            newl += 1
        if newl:
            Q = np.concatenate([Q, V[:, :newl]], axis=1); qlam += list(th[:newl]); V, th = V[:, newl:], th[newl:]
        want_left = k - Q.shape[1]
        jc = max(0, min(len(th) - 1, want_left + (len(th) - want_left) // 2 - 1))
        tau = max(tau_prev, abs(f(th[jc]))); tau_prev = tau
        if not tau > 0:
            lo, hi = -L, 0.5 * L
            continue
        hi = min(tau / (abs(beta) * (1 + tau)), 0.98 * L)
        lo = -min(L, tau / (abs(beta) * (1 - tau)) if tau < 1 else L)
    info['converged'] = converged
    lam = np.concatenate([np.asarray(qlam, np.float64), th]); W = np.concatenate([Q, V], axis=1)
    order = np.argsort(-np.abs(f(lam)), kind='stable')[:k]
    s = np.abs(f(lam[order])); Vk = W[:, order].astype(np.float64)
    return s, Vk * np.sign(f(lam[order])), Vk, info
