"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Independent NumPy/SciPy restatement of the reference's DDP solver and BoxQP, written separately from
oracle/ddp_oracle.hpp (different language, different linear-algebra back end: numpy matmul + scipy.linalg
Cholesky) so that the C++ oracle can be cross-checked (SURVEY.md §8 c (iv)): values to <= 1e-10 relative, all
discrete decisions exactly.  Follows /root/reference/nmpc_ddp/include/nmpc_ddp/DDPSolver.hpp:26-560 and
BoxQP.h:141-347.  Only tests/ imports this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.linalg as sla


# ---------------------------------------------------------------------------------------------------
# BoxQP  (BoxQP.h:141-347)
# ---------------------------------------------------------------------------------------------------
@dataclass
class BoxQPResult:
    x: np.ndarray
    retval: int
    free_idxs: list
    chol: object  # scipy cho_factor of H[free, free] at the last factorisation
    iters: int


def boxqp(H, g, lower, upper, x0=None, max_iter=500, grad_thre=1e-8, rel_improve_thre=1e-8, step_factor=0.6,
          min_step=1e-22, armijo=0.1) -> BoxQPResult:
    H = np.asarray(H, float)
    g = np.asarray(g, float)
    lower = np.asarray(lower, float)
    upper = np.asarray(upper, float)
    m = g.size
    x = np.maximum(np.minimum(np.zeros(m) if x0 is None else np.asarray(x0, float), upper), lower)

    def objective(v):
        return v @ g + 0.5 * (v @ (H @ v))

    obj = objective(x)
    old_obj = obj
    retval = 0
    clamped = np.zeros(m, bool)
    free = []
    chol = None
    it = 1
    while True:
        if it > 1 and (old_obj - obj) < rel_improve_thre * abs(old_obj):
            retval = 4
            break
        old_obj = obj
        grad = g + H @ x
        old_clamped = clamped
        clamped = ((x == lower) & (grad > 0)) | ((x == upper) & (grad < 0))
        free = [i for i in range(m) if not clamped[i]]
        cl = [i for i in range(m) if clamped[i]]
        if clamped.all():
            retval = 6
            break
        if it == 1 or (clamped != old_clamped).any():
            Hf = H[np.ix_(free, free)]
            try:
                chol = sla.cho_factor(Hf, lower=True)
            except sla.LinAlgError:
                retval = -1
                break
        if np.sum(grad[free] ** 2) < grad_thre ** 2:
            retval = 5
            break
        rhs = g[free] + H[np.ix_(free, cl)] @ x[cl]
        sd = np.zeros(m)
        sd[free] = -sla.cho_solve(chol, rhs) - x[free]
        sdg = sd @ grad
        if sdg > 1e-10:
            retval = -2
            break
        step = 1.0
        xc = np.maximum(np.minimum(x + step * sd, upper), lower)
        oc = objective(xc)
        while (oc - old_obj) / (step * sdg) < armijo:
            step *= step_factor
            xc = np.maximum(np.minimum(x + step * sd, upper), lower)
            oc = objective(xc)
            if step < min_step:
                retval = 2
                break
        x, obj = xc, oc
        if it == max_iter:
            retval = 1
            break
        it += 1
    return BoxQPResult(x, retval, free, chol, it)


# ---------------------------------------------------------------------------------------------------
# cart-pole model (TestDDPCartPole.cpp:63-227), vector form
# ---------------------------------------------------------------------------------------------------
class CartPole:
    n, m = 4, 1
    g = 9.80665

    def __init__(self, dt=0.01, m1=1.0, m2=0.5, l=2.0, wx=(0.1, 1.0, 0.01, 0.1), wu=0.001, wt=(0.1, 1.0, 0.01, 0.1),
                 ref_pos=0.0):
        self.dt, self.m1, self.m2, self.l = dt, m1, m2, l
        self.wx, self.wu, self.wt = np.array(wx, float), float(wu), np.array(wt, float)
        self.ref = np.array([ref_pos, 0.0, 0.0, 0.0])

    def input_dim(self, t):
        return 1

    def f(self, t, x, u):
        th, om, F = x[1], x[3], u[0]
        s, c = np.sin(th), np.cos(th)
        den = self.m1 + self.m2 * s * s
        xd = np.array([
            x[2], om,
            (F - self.m2 * self.l * om * om * s + self.m2 * self.g * s * c) / den,
            (F * c - self.m2 * self.l * om * om * s * c + self.g * (self.m1 + self.m2) * s) / (self.l * den)])
        return x + self.dt * xd

    def L(self, t, x, u):
        e = x - self.ref
        return 0.5 * (self.wx @ (e * e)) + 0.5 * (self.wu * (u[0] * u[0]))

    def phi(self, t, x):
        e = x - self.ref
        return 0.5 * (self.wt @ (e * e))

    def fd(self, t, x, u):
        m1, m2, l, g = self.m1, self.m2, self.l, self.g
        th, om, F = x[1], x[3], u[0]
        s, c = np.sin(th), np.cos(th)
        den = m1 + m2 * s * s
        A = np.zeros((4, 4))
        A[0, 2] = 1
        A[1, 3] = 1
        num2 = F - m2 * l * om * om * s + m2 * g * s * c
        num3 = F * c - m2 * l * om * om * s * c + g * (m1 + m2) * s
        dden = 2 * m2 * s * c
        A[2, 1] = ((-m2 * l * om * om * c + m2 * g * (1 - 2 * s * s)) * den - num2 * dden) / den ** 2
        A[2, 3] = (-2 * m2 * l * om * s) / den
        A[3, 1] = ((-F * s - m2 * l * om * om * (1 - 2 * s * s) + g * (m1 + m2) * c) * den - num3 * dden) / (l * den ** 2)
        A[3, 3] = (-2 * m2 * l * om * s * c) / (l * den)
        Fx = np.eye(4) + self.dt * A
        Fu = self.dt * np.array([[0.0], [0.0], [1 / den], [c / (l * den)]])
        return Fx, Fu

    def Ld(self, t, x, u):
        return self.wx * (x - self.ref), np.array([self.wu * u[0]]), np.diag(self.wx), np.array([[self.wu]]), \
            np.zeros((4, 1))

    def phid(self, t, x):
        return self.wt * (x - self.ref), np.diag(self.wt)


# ---------------------------------------------------------------------------------------------------
# DDP solver  (DDPSolver.hpp:26-560)
# ---------------------------------------------------------------------------------------------------
@dataclass
class Config:
    with_input_constraint: bool = False
    max_iter: int = 500
    horizon_steps: int = 100
    reg_type: int = 1
    initial_lambda: float = 1e-4
    initial_dlambda: float = 1.0
    lambda_factor: float = 1.6
    lambda_min: float = 1e-6
    lambda_max: float = 1e10
    k_rel_norm_thre: float = 1e-4
    lambda_thre: float = 1e-5
    alpha_list: np.ndarray = field(default_factory=lambda: np.array(
        [10.0 ** (-3.0 if i == 10 else i * ((-3.0 - 0.0) / 10)) for i in range(11)]))
    cost_update_ratio_thre: float = 0.0
    cost_update_thre: float = 1e-7


class DDP:
    def __init__(self, model, cfg: Config, limits=None):
        self.model, self.cfg, self.limits = model, cfg, limits
        self.trace = []

    def solve(self, t0, x0, u_init):
        c, md = self.cfg, self.model
        T = c.horizon_steps
        self.t0 = t0
        self.lam, self.dlam = c.initial_lambda, c.initial_dlambda
        self.U = [np.array(u, float).ravel() for u in u_init]
        self.X = [np.array(x0, float)]
        self.C = np.zeros(T + 1)
        for i in range(T):
            t = t0 + i * md.dt
            self.X.append(md.f(t, self.X[i], self.U[i]))
            self.C[i] = md.L(t, self.X[i], self.U[i])
        self.C[T] = md.phi(t0 + T * md.dt, self.X[T])
        self.trace = [dict(iter=0, cost=self._sum(self.C), lam=self.lam, dlam=self.dlam, alpha_idx=-1, n_bw=0, n_fw=0)]
        self.status = 0
        for it in range(1, c.max_iter + 1):
            self.status = self._proc_once(it)
            if self.status != 0:
                break
        return self.status == 1

    @staticmethod
    def _sum(v):
        s = 0.0
        for e in v:
            s += e
        return s

    def _proc_once(self, it):
        c, md = self.cfg, self.model
        T = c.horizon_steps
        tr = dict(iter=it, cost=0.0, lam=0.0, dlam=0.0, alpha_idx=-1, n_bw=1, n_fw=0, k_rel_norm=0.0, alpha=0.0)
        self.trace.append(tr)
        self.D = []
        for i in range(T):
            t = self.t0 + i * md.dt
            Fx, Fu = md.fd(t, self.X[i], self.U[i])
            self.D.append((Fx, Fu) + tuple(md.Ld(t, self.X[i], self.U[i])))
        self.VxT, self.VxxT = md.phid(self.t0 + T * md.dt, self.X[T])
        while not self._backward():
            self.dlam = max(self.dlam * c.lambda_factor, c.lambda_factor)
            self.lam = max(self.lam * self.dlam, c.lambda_min)
            if self.lam > c.lambda_max:
                return -1
            tr["n_bw"] += 1
        krn = 0.0
        for i in range(T):
            krn = max(krn, np.sqrt(np.sum(self.k[i] ** 2)) / (np.sqrt(np.sum(self.U[i] ** 2)) + 1.0))
        tr["k_rel_norm"] = krn
        if krn < c.k_rel_norm_thre and self.lam < c.lambda_thre:
            return 1
        ok = False
        act = 0.0
        for ai, alpha in enumerate(c.alpha_list):
            Xc, Uc, Cc = self._forward(alpha)
            tr["n_fw"] += 1
            tr["alpha_idx"] = ai
            tr["alpha"] = alpha
            act = self._sum(self.C) - self._sum(Cc)
            exp = -alpha * (self.dV[0] + alpha * self.dV[1])
            ratio = act / exp if exp != 0 else np.inf * np.sign(act)
            if exp < 0:
                ratio = 1.0 if act >= 0 else -1.0
            if ratio > c.cost_update_ratio_thre:
                ok = True
                break
        ret = 0
        if ok:
            self.X, self.U, self.C = Xc, Uc, Cc
            if act < c.cost_update_thre:
                ret = 1
            self.dlam = min(self.dlam / c.lambda_factor, 1 / c.lambda_factor)
            self.lam = self.lam * self.dlam if self.lam >= c.lambda_min else 0.0
        else:
            self.dlam = max(self.dlam * c.lambda_factor, c.lambda_factor)
            self.lam = max(self.lam * self.dlam, c.lambda_min)
            if self.lam > c.lambda_max:
                ret = -1
        tr["cost"], tr["lam"], tr["dlam"] = self._sum(self.C), self.lam, self.dlam
        return ret

    def _backward(self):
        c, md = self.cfg, self.model
        T = c.horizon_steps
        Vx, Vxx = self.VxT.copy(), self.VxxT.copy()
        self.dV = np.zeros(2)
        n = md.n
        if not hasattr(self, "k") or len(self.k) != T:
            self.k = [np.zeros(0)] * T
            self.K = [np.zeros((0, n))] * T
        self.qp_ret = [0] * T
        self.qp_free = [[] for _ in range(T)]
        for i in range(T - 1, -1, -1):
            Fx, Fu, Lx, Lu, Lxx, Luu, Lxu = self.D[i]
            m = Fu.shape[1]
            Qu = Lu + Fu.T @ Vx
            Qx = Lx + Fx.T @ Vx
            Qux = Lxu.T + (Fu.T @ Vxx) @ Fx
            Quu = Luu + (Fu.T @ Vxx) @ Fu
            Qxx = Lxx + (Fx.T @ Vxx) @ Fx
            Vreg = Vxx + (self.lam * np.eye(n) if c.reg_type == 2 else 0)
            Qux_reg = Lxu.T + (Fu.T @ Vreg) @ Fx
            QuuF = Luu + (Fu.T @ Vreg) @ Fu + (self.lam * np.eye(m) if c.reg_type == 1 else 0)
            if m > 0:
                if c.with_input_constraint:
                    k0 = self.k[i + 1] if (i != T - 1 and self.k[i + 1].size == m) else np.zeros(m)
                    lo, up = self.limits
                    r = boxqp(QuuF, Qu, lo[:m] - self.U[i], up[:m] - self.U[i], k0)
                    self.qp_ret[i], self.qp_free[i] = r.retval, list(r.free_idxs)
                    if r.retval < 0:
                        return False
                    k = r.x
                    K = np.zeros((m, n))
                    if len(r.free_idxs) > 0:
                        K[r.free_idxs, :] = -sla.cho_solve(r.chol, Qux_reg[r.free_idxs, :])
                else:
                    try:
                        ch = sla.cho_factor(QuuF, lower=True)
                    except sla.LinAlgError:
                        return False
                    k = -sla.cho_solve(ch, Qu)
                    K = -sla.cho_solve(ch, Qux_reg)
            else:
                k, K = np.zeros(0), np.zeros((0, n))
            self.dV += np.array([k @ Qu, 0.5 * (k @ (Quu @ k))])
            Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qux.T @ k
            Vxx = Qxx + K.T @ Quu @ K + K.T @ Qux + Qux.T @ K
            Vxx = 0.5 * (Vxx + Vxx.T)
            self.k[i], self.K[i] = k, K
        return True

    def _forward(self, alpha):
        c, md = self.cfg, self.model
        T = c.horizon_steps
        Xc, Uc, Cc = [self.X[0].copy()], [], np.zeros(T + 1)
        for i in range(T):
            u = self.U[i] + alpha * self.k[i] + self.K[i] @ (Xc[i] - self.X[i])
            Uc.append(u)
            t = self.t0 + i * md.dt
            Xc.append(md.f(t, Xc[i], u))
            Cc[i] = md.L(t, Xc[i], u)
        Cc[T] = md.phi(self.t0 + T * md.dt, Xc[T])
        return Xc, Uc, Cc
