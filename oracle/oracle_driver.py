"""Python driver of the CPU oracle (oracle/lh_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product path (iq-tree_amd/).  It walks a tree the way
PhyloTree::computeLikelihood does (phylotree.cpp:1031, phylokernel.h:70-157), calling the C
restatement of the reference's pattern loops for every node, with everything in the
reference's host layout.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    src = os.path.join(_HERE, "lh_oracle.c")
    out = os.path.join(_HERE, "liblh_oracle.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-mavx", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared",
                               "-o", out, src, "-lm"])
    return out


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    dp = C.POINTER(C.c_double)
    u8 = C.POINTER(C.c_uint8)
    sp = C.POINTER(C.c_short)
    L.oracle_scaling_threshold.restype = C.c_double
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_set_threads.restype = C.c_int
    L.oracle_set_mixture.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
    L.oracle_set_mixture.restype = None
    L.oracle_log_scaling_threshold.restype = C.c_double
    L.oracle_tip_partial_lh.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp]
    L.oracle_tip_partial_lh.restype = None
    L.oracle_echild.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, dp]
    L.oracle_echild.restype = None
    L.oracle_partial_update.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, dp, dp, C.c_int,
                                        u8, dp, sp, C.c_double, u8, dp, sp, C.c_double, dp, dp, dp, sp]
    L.oracle_partial_update.restype = C.c_double
    L.oracle_partial_update_multi.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, dp, dp, dp, dp, dp, C.c_int,
                                              C.POINTER(u8), C.POINTER(dp), C.POINTER(sp), dp, dp, dp, dp, sp]
    L.oracle_partial_update_multi.restype = C.c_double
    L.oracle_branch_lnl.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, C.c_double, dp, u8, dp, dp,
                                    dp, dp, dp]
    L.oracle_branch_lnl.restype = C.c_double
    L.oracle_theta.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, u8, dp, dp, dp]
    L.oracle_theta.restype = None
    L.oracle_derv.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp]
    L.oracle_derv.restype = None
    L.oracle_lnl_from_theta.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, C.c_double, dp, dp, dp, dp]
    L.oracle_lnl_from_theta.restype = C.c_double
    L.oracle_asc_prob_const_branch.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, C.c_double, dp, u8, dp, sp,
                                               dp, sp, dp]
    L.oracle_asc_prob_const_branch.restype = C.c_double
    fp = C.POINTER(C.c_float)
    L.oracle_pattern_lh_scaled.argtypes = [C.c_size_t, dp, sp, sp, dp]
    L.oracle_pattern_lh_scaled.restype = None
    L.oracle_dot_float8.argtypes = [fp, fp, C.c_size_t]
    L.oracle_dot_float8.restype = C.c_float
    L.oracle_asc_theta_sums.argtypes = [C.c_int, C.c_int, C.c_size_t, dp, dp, dp, C.c_double, dp, sp, dp, dp]
    L.oracle_asc_theta_sums.restype = None
    _LIB = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def _sp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_short))


SEQ_DNA, SEQ_PROTEIN, SEQ_CODON, SEQ_OTHER = 0, 1, 2, 3


def state_unknown_for(nstates, seq_type):
    if seq_type == SEQ_DNA and nstates == 4:
        return 18
    if seq_type == SEQ_PROTEIN and nstates == 20:
        return 23
    return nstates


def _parse_newick(s):
    pos = [0]

    def node():
        kids = []
        if s[pos[0]] == "(":
            pos[0] += 1
            while True:
                kids.append(node())
                if s[pos[0]] == ",":
                    pos[0] += 1
                    continue
                if s[pos[0]] == ")":
                    pos[0] += 1
                    break
        b = pos[0]
        while s[pos[0]] not in ":,();":
            pos[0] += 1
        label = s[b:pos[0]]
        length = 0.0
        if s[pos[0]] == ":":
            pos[0] += 1
            b = pos[0]
            while s[pos[0]] not in ",();":
                pos[0] += 1
            length = float(s[b:pos[0]])
        return (label, length, kids)

    return node()


class OracleTree:
    """Unrooted tree + alignment + model; partial vectors cached per directed edge."""

    def __init__(self, newick, nstates, seq_type, states, ptn_freq, ptn_invar, model, n_unobs=0, nsites=None):
        self.L = lib()
        self.n = nstates
        self.seq_type = seq_type
        self.states = np.ascontiguousarray(states, dtype=np.uint8)
        self.ntaxa, self.nptn = self.states.shape
        # +ASC: the last n_unobs patterns are the unobserved constant patterns (frequency 0)
        self.n_unobs = int(n_unobs)
        self.nobs = self.nptn - self.n_unobs
        self.nsites = float(nsites if nsites is not None else np.sum(ptn_freq))
        self.freq = np.ascontiguousarray(ptn_freq, dtype=np.float64)
        self.invar = (np.zeros(self.nptn) if ptn_invar is None
                      else np.ascontiguousarray(ptn_invar, dtype=np.float64))
        self.set_model(model)
        # adjacency: node ids as the host mirror assigns them (leaves = taxon id, internal nodes
        # numbered from ntaxa in Newick pre-order)
        self.adj = {}
        top = _parse_newick(newick.strip().rstrip(";") + ";")
        self.next_id = self.ntaxa

        def build(nd):
            label, _, kids = nd
            if not kids:
                i = int(label)
                self.adj.setdefault(i, [])
                return i
            i = self.next_id
            self.next_id += 1
            self.adj[i] = []
            for k in kids:
                c = build(k)
                self.connect(i, c, k[1])
            return i

        if len(top[2]) == 2:
            a, b = top[2]
            na, nb = build(a), build(b)
            self.connect(na, nb, a[1] + b[1])
        else:
            build(top)
        self.cache = {}

    def connect(self, a, b, length):
        self.adj[a].append([b, length])
        self.adj[b].append([a, length])

    def set_model(self, model):
        self.model = model
        self.ncat = len(model.rates)
        self.block = self.n * self.ncat
        self.su = state_unknown_for(self.n, self.seq_type)
        self.eval = np.ascontiguousarray(model.eval, dtype=np.float64)
        self.evec = np.ascontiguousarray(model.evec, dtype=np.float64)
        self.inv_evec = np.ascontiguousarray(model.inv_evec, dtype=np.float64)
        self.rates = np.ascontiguousarray(model.rates, dtype=np.float64)
        self.props = np.ascontiguousarray(model.props, dtype=np.float64)
        # mixture models: model.cat_class[c] = eigen-system of component c; eval / evec / inv_evec are the
        # per-class arrays concatenated, tip is [state][class][n] (phylokernelmixture.h:151)
        self.cat_class = getattr(model, "cat_class", None)
        self.nclass = int(getattr(model, "nclass", 1))
        n2 = self.n * self.n
        if self.nclass > 1:
            self.cat_class = np.ascontiguousarray(self.cat_class, dtype=np.int32)
            tip = np.zeros((self.su + 1, self.nclass, self.n))
            for m in range(self.nclass):
                t = np.zeros((self.su + 1) * self.n)
                ie = np.ascontiguousarray(self.inv_evec.reshape(-1)[m * n2:(m + 1) * n2])
                self.L.oracle_tip_partial_lh(self.n, self.seq_type, self.su, _dp(ie), _dp(t))
                tip[:, m, :] = t.reshape(self.su + 1, self.n)
            self.tip = np.ascontiguousarray(tip.reshape(-1))
        else:
            self.tip = np.zeros((self.su + 1) * self.n)
            self.L.oracle_tip_partial_lh(self.n, self.seq_type, self.su, _dp(self.inv_evec), _dp(self.tip))
        self.cache = {}

    def _ctx(self):
        """select this tree's mixture structure in the C library (a process-wide setting there)"""
        if self.nclass > 1:
            self.L.oracle_set_mixture(self.nclass, self.cat_class.ctypes.data_as(C.POINTER(C.c_int)), self.ncat)
        else:
            self.L.oracle_set_mixture(1, None, 0)

    # ---- tree helpers
    def is_leaf(self, v):
        return len(self.adj[v]) == 1

    def length(self, a, b):
        for nb, ln in self.adj[a]:
            if nb == b:
                return ln
        raise KeyError((a, b))

    def set_length(self, a, b, length):
        for e in self.adj[a]:
            if e[0] == b:
                e[1] = length
        for e in self.adj[b]:
            if e[0] == a:
                e[1] = length
        self.cache = {}

    def clear(self):
        self.cache = {}

    def farthest_leaf(self, root=0):
        """mtree.cpp:2052-2070 from the root leaf."""
        import sys
        sys.setrecursionlimit(100000)
        height = {}

        def go(node, dad):
            if dad is not None and self.is_leaf(node):
                height[node] = 0
                return node
            res = None
            height[node] = 0
            for nb, _ in self.adj[node]:
                if nb == dad:
                    continue
                leaf = go(nb, node)
                if height[node] < height[nb] + 1:
                    height[node] = height[nb] + 1
                    res = leaf
            return res

        return go(root, None)

    # ---- kernels
    def partial(self, frm, to):
        """Vector of the subtree at `to` seen from `frm` (the neighbour frm->to).
        Returns (plh[nptn,block], scale_num[nptn] int16, lh_scale_factor)."""
        import sys
        sys.setrecursionlimit(100000)
        key = (frm, to)
        if key in self.cache:
            return self.cache[key]
        self._ctx()
        assert not self.is_leaf(to)
        kids = [(nb, ln) for nb, ln in self.adj[to] if nb != frm]
        # multifurcating node, or a state count outside the SIMD dispatch cases: the reference's scalar kernel
        # (phylotreesse.cpp:702-806; :281-309 sends every state count but 2 / 4 / 20 / 64 there)
        if len(kids) > 2 or self.n not in (2, 4, 20, 64):
            res = self._partial_multi(to, kids)
            self.cache[key] = res
            return res
        assert len(kids) == 2
        (l, ll), (r, rl) = kids
        if not self.is_leaf(l) and self.is_leaf(r):
            (l, ll), (r, rl) = (r, rl), (l, ll)
        args_l = self._child(to, l)
        args_r = self._child(to, r)
        out = np.empty((self.nptn, self.block))
        sc = np.empty(self.nptn, dtype=np.int16)
        sum_scale = self.L.oracle_partial_update(
            self.n, self.ncat, self.nptn, _dp(self.eval), _dp(self.evec), _dp(self.inv_evec),
            _dp(self.rates), _dp(self.tip), self.su,
            _u8(args_l[0]), _dp(args_l[1]), _sp(args_l[2]), ll,
            _u8(args_r[0]), _dp(args_r[1]), _sp(args_r[2]), rl,
            _dp(self.freq), _dp(self.invar), _dp(out), _sp(sc))
        res = (out, sc, args_l[3] + args_r[3] + sum_scale)
        self.cache[key] = res
        return res

    def _partial_multi(self, to, kids):
        args = [self._child(to, k) for k, _ in kids]
        nk = len(kids)
        u8, dp, sp = C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_short)
        st = (u8 * nk)(*[_u8(a[0]) if a[0] is not None else u8() for a in args])
        pl = (dp * nk)(*[_dp(a[1]) if a[1] is not None else dp() for a in args])
        sc = (sp * nk)(*[_sp(a[2]) if a[2] is not None else sp() for a in args])
        lens = np.array([ln for _, ln in kids], dtype=np.float64)
        out = np.empty((self.nptn, self.block))
        osc = np.empty(self.nptn, dtype=np.int16)
        sum_scale = self.L.oracle_partial_update_multi(
            self.n, self.ncat, self.nptn, nk, _dp(self.eval), _dp(self.evec), _dp(self.inv_evec), _dp(self.rates),
            _dp(self.tip), self.su, st, pl, sc, _dp(lens), _dp(self.freq), _dp(self.invar), _dp(out), _sp(osc))
        return (out, osc, sum(a[3] for a in args) + sum_scale)

    def _child(self, node, child):
        if self.is_leaf(child):
            return (self.states[child], None, None, 0.0)
        plh, sc, sf = self.partial(node, child)
        return (None, plh, sc, sf)

    def _ends(self, a, b):
        """(dad_states, dad_plh, node_plh, scale_factor_sum, dad_scale, node_scale), leaf (if any) as dad."""
        if self.is_leaf(b):
            a, b = b, a
        if self.is_leaf(a):
            plh, sc, sf = self.partial(a, b)
            return self.states[a], None, plh, sf, None, sc
        pa, sca, sfa = self.partial(b, a)
        pb, scb, sfb = self.partial(a, b)
        return None, pa, pb, sfa + sfb, sca, scb

    def branch_lnl(self, a, b, length=None):
        self._ctx()
        ds, dp_, np_, sf, dsc, nsc = self._ends(a, b)
        ln = self.length(a, b) if length is None else length
        plh = np.zeros(self.nptn)
        v = self.L.oracle_branch_lnl(self.n, self.ncat, self.nobs, _dp(self.eval), _dp(self.rates),
                                     _dp(self.props), ln, _dp(self.tip), _u8(ds), _dp(dp_), _dp(np_),
                                     _dp(self.freq), _dp(self.invar), _dp(plh))
        total = sf + v
        if self.n_unobs:
            o = self.nobs
            pc = self.L.oracle_asc_prob_const_branch(
                self.n, self.ncat, self.n_unobs, _dp(self.eval), _dp(self.rates), _dp(self.props), ln,
                _dp(self.tip), _u8(None if ds is None else np.ascontiguousarray(ds[o:])),
                _dp(None if dp_ is None else np.ascontiguousarray(dp_[o:])),
                _sp(None if dsc is None else np.ascontiguousarray(dsc[o:])),
                _dp(np.ascontiguousarray(np_[o:])), _sp(np.ascontiguousarray(nsc[o:])),
                _dp(np.ascontiguousarray(self.invar[o:])))
            assert 0.0 <= pc < 1.0
            lp = np.log(1.0 - pc)  # phylokernel.h:1009-1016
            plh[:o] -= lp
            total -= self.nsites * lp
        return total, plh

    def likelihood(self, root=0):
        """clearAllPartialLH(); computeLikelihood(): branch = farthest leaf's pendant branch."""
        leaf = self.farthest_leaf(root)
        nb = self.adj[leaf][0][0]
        return self.branch_lnl(leaf, nb)[0], (leaf, nb)

    def theta(self, a, b):
        self._ctx()
        ds, dp_, np_, sf, dsc, nsc = self._ends(a, b)
        th = np.zeros((self.nptn, self.block))
        self.L.oracle_theta(self.n, self.ncat, self.nptn, _dp(self.tip), _u8(ds), _dp(dp_), _dp(np_), _dp(th))
        self._theta_scale = (nsc if dsc is None else (dsc + nsc)).astype(np.int16)
        return th, sf

    def derv(self, a, b, length=None, theta=None):
        self._ctx()
        if theta is None:
            theta, _ = self.theta(a, b)
        ln = self.length(a, b) if length is None else length
        df, ddf = C.c_double(), C.c_double()
        self.L.oracle_derv(self.n, self.ncat, self.nobs, _dp(self.eval), _dp(self.rates), _dp(self.props),
                           ln, _dp(theta), _dp(self.freq), _dp(self.invar), C.byref(df), C.byref(ddf))
        df, ddf = df.value, ddf.value
        if self.n_unobs:  # phylokernel.h:655-725
            o = self.nobs
            out = np.zeros(3)
            self.L.oracle_asc_theta_sums(self.n, self.ncat, self.n_unobs, _dp(self.eval), _dp(self.rates),
                                         _dp(self.props), ln, _dp(np.ascontiguousarray(theta[o:])), None,
                                         _dp(np.ascontiguousarray(self.invar[o:])), _dp(out))
            prob_const = 1.0 - out[0]
            df_frac, ddf_frac = out[1] / prob_const, out[2] / prob_const
            df += self.nsites * df_frac
            ddf += self.nsites * (ddf_frac + df_frac * df_frac)
        return df, ddf

    def lnl_from_theta(self, a, b, length=None, theta=None, sf=None):
        self._ctx()
        if theta is None:
            theta, sf = self.theta(a, b)
        ln = self.length(a, b) if length is None else length
        plh = np.zeros(self.nptn)
        v = self.L.oracle_lnl_from_theta(self.n, self.ncat, self.nobs, _dp(self.eval), _dp(self.rates),
                                         _dp(self.props), ln, _dp(theta), _dp(self.freq), _dp(self.invar),
                                         _dp(plh))
        total = sf + v
        if self.n_unobs:  # phylokernel.h:1124-1187
            o = self.nobs
            out = np.zeros(3)
            self.L.oracle_asc_theta_sums(self.n, self.ncat, self.n_unobs, _dp(self.eval), _dp(self.rates),
                                         _dp(self.props), ln, _dp(np.ascontiguousarray(theta[o:])),
                                         _sp(np.ascontiguousarray(self._theta_scale[o:])),
                                         _dp(np.ascontiguousarray(self.invar[o:])), _dp(out))
            lp = np.log(1.0 - out[0])
            total -= self.nsites * lp
            plh[:o] -= lp
        return total, plh

    def minimize_newton(self, a, b, x1, xguess, x2, xacc, max_steps, theta=None):
        """Optimization::minimizeNewton (optimization.cpp:388-465) over computeFuncDerv (phylotree.cpp:2135-2146:
        f = -df, df = -ddf of computeLikelihoodDerv, NaN -> 0 as phylokernel.h:647-651) on branch (a, b), theta built
        once (theta_computed, phylokernel.h:535-536).  Returns (optx, d2l, evaluated points, status): the list holds
        every branch length the derivative was evaluated at, in order; status 'ok' | 'nonfinite' | 'maxsteps' for
        the reference's two nrerror() exits."""
        if theta is None:
            theta, _ = self.theta(a, b)
        evaluated = []

        def func_derv(x):
            evaluated.append(x)
            df, ddf = self.derv(a, b, length=x, theta=theta)
            if not np.isfinite(df):
                df, ddf = 0.0, 0.0
            return -df, -ddf

        rts = min(max(xguess, x1), x2)
        f, df = func_derv(rts)
        d2l = df
        if not (np.isfinite(f) and np.isfinite(df)):
            return rts, d2l, evaluated, "nonfinite"
        if df >= 0.0 and abs(f) < xacc:
            return rts, d2l, evaluated, "ok"
        if f < 0.0:
            xl, xh = rts, x2
        else:
            xh, xl = rts, x1
        dx = abs(xh - xl)
        for j in range(1, max_steps + 1):
            rts_old = rts
            if df <= 0.0 or ((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0:
                dx = 0.5 * (xh - xl)
                rts = xl + dx
                d2l = df
                if xl == rts:
                    return rts, d2l, evaluated, "ok"
            else:
                dx = f / df
                temp = rts
                rts -= dx
                d2l = df
                if temp == rts:
                    return rts, d2l, evaluated, "ok"
            if abs(dx) < xacc or j == max_steps:
                return rts_old, d2l, evaluated, "ok"
            f, df = func_derv(rts)
            if not (np.isfinite(f) and np.isfinite(df)):
                return rts_old, d2l, evaluated, "nonfinite"
            if df > 0.0 and abs(f) < xacc:
                return rts, df, evaluated, "ok"
            if f < 0.0:
                xl = rts
            else:
                xh = rts
        return 0.0, 0.0, evaluated, "maxsteps"

    def time_traversals(self, budget_s=15.0, min_reps=1):
        """cpu_baseline: repeat {clear; full traversal; root lnL}; returns (M upd/s, reps, seconds)."""
        reps, t0 = 0, time.perf_counter()
        while True:
            self.clear()
            self.likelihood()
            reps += 1
            dt = time.perf_counter() - t0
            if reps >= min_reps and dt >= budget_s:
                break
        return reps * (self.ntaxa - 2) * self.nptn / dt / 1e6, reps, dt


def dot_float8(x, y):
    """The reference's UFBoot dot product (float, 8 AVX lanes); x, y float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    assert x.shape == y.shape and x.ndim == 1
    fp = C.POINTER(C.c_float)
    return float(lib().oracle_dot_float8(x.ctypes.data_as(fp), y.ctypes.data_as(fp), x.size))


def pattern_lh_scaled(pattern_lh, sc_a, sc_b):
    """PhyloTree::computePatternLikelihood: scaling events of both branch ends put back."""
    p = np.ascontiguousarray(pattern_lh, dtype=np.float64)
    out = np.zeros_like(p)
    a = None if sc_a is None else np.ascontiguousarray(sc_a, dtype=np.int16)
    b = None if sc_b is None else np.ascontiguousarray(sc_b, dtype=np.int16)
    lib().oracle_pattern_lh_scaled(p.size, _dp(p), _sp(a), _sp(b), _dp(out))
    return out
