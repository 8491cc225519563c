"""Test-side Krylov drivers, restating tests/solvers.cpp of the reference (BiCGSTAB :140-239,
Richardson :90-138) so that the solve-level known-answer tests read like the reference's own."""
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
