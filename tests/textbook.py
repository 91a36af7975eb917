"""Independent check of the MATHEMATICS of the oracle: classical Felsenstein pruning in
probability space (conditional likelihoods, P(t) = expm(Q*r*t) by scipy), no eigen-space
vectors, no scaling tricks (log-space rescaling per node instead).  Shares no code with
oracle/lh_oracle.c; used by tests/test_oracle.py."""
import numpy as np
from scipy.linalg import expm

def tip_vector(state, n, seq_type, state_unknown):
    v = np.zeros(n)
    if state < n:
        v[state] = 1.0
    elif state == state_unknown:
        v[:] = 1.0
    elif seq_type == 0 and n == 4:
        mask = state - 3
        for x in range(4):
            if mask & (1 << x):
                v[x] = 1.0
    elif seq_type == 1 and n == 20:
        for x in {20: (2, 3), 21: (5, 6), 22: (9, 10)}[state]:
            v[x] = 1.0
    else:
        raise ValueError(state)
    return v


def site_log_likelihoods(adj, states, model, seq_type, state_unknown, root=None):
    """adj: {node: [[nb, len], ...]}, leaves = taxon ids.  Returns log L per pattern."""
    import sys
    sys.setrecursionlimit(100000)
    n = model.nstates
    ntaxa, nptn = states.shape
    Q = model.Q
    if root is None:
        root = next(v for v in adj if len(adj[v]) > 1)
    tips = {s: tip_vector(s, n, seq_type, state_unknown) for s in np.unique(states)}
    logs = []
    for c, (rate, prop) in enumerate(zip(model.rates, model.props)):
        pcache = {}

        def P(t):
            if t not in pcache:
                pcache[t] = expm(Q * rate * t)
            return pcache[t]

        def cond(node, dad):
            """-> (L[nptn, n], logscale[nptn])"""
            if len(adj[node]) == 1 and dad is not None:
                return np.stack([tips[s] for s in states[node]]), np.zeros(nptn)
            L = np.ones((nptn, n))
            ls = np.zeros(nptn)
            for nb, ln in adj[node]:
                if nb == dad:
                    continue
                Lc, lsc = cond(nb, node)
                L = L * (Lc @ P(ln).T)
                ls = ls + lsc
            m = L.max(axis=1)
            m[m == 0] = 1.0
            return L / m[:, None], ls + np.log(m)

        L, ls = cond(root, None)
        lik = (L * model.freqs[None, :]).sum(axis=1)
        logs.append(np.log(prop * lik) + ls)
    logs = np.stack(logs)  # [ncat, nptn]
    m = logs.max(axis=0)
    return m + np.log(np.exp(logs - m[None, :]).sum(axis=0))
