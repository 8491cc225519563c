"""Test-side Krylov drivers, restating tests/solvers.cpp of the reference (BiCGSTAB :140-239,
Richardson :90-138, GCR :247-352) so that the solve-level known-answer tests read like the reference's own."""
import numpy as np


def bicgstab(A_apply, prec_apply, rhs, tol, maxiter, x0=None):
    """Right-preconditioned BiCGSTAB exactly as tests/solvers.cpp:140-239.  -> (x, iters, relres)"""
    n = rhs.size
    x = np.zeros(n) if x0 is None else x0.copy()
    omega = 1.0
    rhoold = 1.0
    alpha = 1.0
    p = np.zeros(n)
    v = np.zeros(n)
    r = rhs - A_apply(x)
    rhat = r.copy()
    bnorm = np.sqrt(rhs @ rhs)
    step = 0
    resnorm = 100.0
    while step < maxiter:
        rho = rhat @ r
        beta = rho * alpha / (rhoold * omega)
        p = r + beta * p - beta * omega * v
        y = prec_apply(p)
        v = A_apply(y)
        alpha = rho / (rhat @ v)
        r = r - alpha * v
        z = prec_apply(r)
        t = A_apply(z)
        omega = (t @ r) / (t @ t)
        x = x + alpha * y + omega * z
        r = r - omega * t
        resnorm = np.sqrt(r @ r)
        if resnorm / bnorm < tol:
            break
        rhoold = rho
        step += 1
    return x, step + 1, resnorm / bnorm


def richardson(A_apply, prec_apply, rhs, tol, maxiter):
    n = rhs.size
    x = np.zeros(n)
    bnorm = np.sqrt(rhs @ rhs)
    step = 0
    rel = np.inf
    while step < maxiter:
        s = rhs - A_apply(x)
        rel = np.sqrt(s @ s) / bnorm
        if rel < tol:
            break
        x = x + prec_apply(s)
        step += 1
    return x, step, rel


def gcr(A_apply, prec_apply, rhs, tol, maxiter, restart=30, x0=None):
    """Right-preconditioned restarted GCR as tests/solvers.cpp:247-352 -- the reference's FLEXIBLE Krylov
    method: the search direction p_k = M(r_k) is stored next to q_k = A p_k, so the preconditioner may be a
    different operator at every application (asynchronous sweeps).  A cycle keeps at most `restart`
    direction pairs, each made A^T A-orthogonal to the earlier ones of the cycle; `step` counts inner
    iterations as the reference does.  -> (x, iters, relres)"""
    n = rhs.size
    x = np.zeros(n) if x0 is None else x0.copy()
    bnorm = np.sqrt(rhs @ rhs)
    step = 0
    rel = 1.0
    while step < maxiter:
        res = rhs - A_apply(x)
        P = [prec_apply(res)]
        Q = [A_apply(P[0])]
        qq = [Q[0] @ Q[0]]
        for k in range(restart):
            alpha = (res @ Q[k]) / qq[k]
            x = x + alpha * P[k]
            res = res - alpha * Q[k]
            rel = np.sqrt(res @ res) / bnorm
            step += 1
            if rel < tol or k == restart - 1 or step >= maxiter:
                break
            z = prec_apply(res)
            q = A_apply(z)
            p = z.copy()
            # the coefficients all come from the un-orthogonalised q (classical Gram-Schmidt), :316-327
            betas = [-(q @ Q[i]) / qq[i] for i in range(k + 1)]
            for i, b in enumerate(betas):
                p += b * P[i]
                q += b * Q[i]
            P.append(p)
            Q.append(q)
            qq.append(q @ q)
        if rel < tol:
            break
    return x, step, rel
