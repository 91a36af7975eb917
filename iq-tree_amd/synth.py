"""Synthetic inputs for tests and bench.py: reversible substitution models in the eigen form the
kernels consume, discrete-Gamma rates, random trees and sequence simulation.

These are INPUT GENERATORS, not part of the accelerated path.  The decomposition follows the
recipe the reference uses for reversible models (eigendecomposition.cpp:167-296,306-394:
normalise Q to one expected substitution per unit time, symmetrise with sqrt(pi), U = V/sqrt(pi),
U^-1 = V^T*sqrt(pi)) so that the eigen-space vectors look like the reference's (entries of
either sign); parity tests feed the SAME arrays to the oracle and to the HIP engine, so any
valid eigen-system would do.
"""
import numpy as np

try:
    from scipy.special import gammainc
    from scipy.stats import gamma as _gamma_dist
except Exception:  # pragma: no cover
    gammainc = None
    _gamma_dist = None


class Model:
    """eval[n], evec[n*n] (U[x][i] at x*n+i), inv_evec[n*n] (U^-1[i][x] at i*n+x), rates, props."""

    def __init__(self, Q, freqs, eval_, evec, inv_evec, rates, props, pinvar=0.0, name=""):
        self.Q = Q
        self.freqs = np.asarray(freqs, dtype=np.float64)
        self.eval = np.ascontiguousarray(eval_, dtype=np.float64)
        self.evec = np.ascontiguousarray(evec, dtype=np.float64).reshape(-1)
        self.inv_evec = np.ascontiguousarray(inv_evec, dtype=np.float64).reshape(-1)
        self.rates = np.ascontiguousarray(rates, dtype=np.float64)
        self.props = np.ascontiguousarray(props, dtype=np.float64)
        self.pinvar = float(pinvar)
        self.name = name
        self.nstates = len(self.eval)
        self.ncat = len(self.rates)


def discrete_gamma_rates(alpha, ncat, pinvar=0.0):
    """Mean-of-category discrete Gamma (Yang 1994), mean 1, divided by (1-pinvar) as
    model/rategamma.cpp:86-130 does."""
    if ncat == 1:
        return np.array([1.0 / (1.0 - pinvar)])
    if gammainc is None:
        raise RuntimeError("scipy is required for discrete_gamma_rates")
    cuts = _gamma_dist.ppf(np.arange(1, ncat) / ncat, a=alpha, scale=1.0 / alpha)
    upper = np.concatenate([gammainc(alpha + 1.0, cuts * alpha), [1.0]])
    lower = np.concatenate([[0.0], upper[:-1]])
    rates = (upper - lower) * ncat
    rates = rates / rates.mean()
    return rates / (1.0 - pinvar)


def reversible_model(exch, freqs, alpha=None, ncat=1, pinvar=0.0, name=""):
    """exch: symmetric n x n exchangeabilities (diagonal ignored); freqs: stationary distribution."""
    freqs = np.asarray(freqs, dtype=np.float64)
    freqs = freqs / freqs.sum()
    n = len(freqs)
    R = np.array(exch, dtype=np.float64)
    R = (R + R.T) / 2.0
    np.fill_diagonal(R, 0.0)
    Q = R * freqs[None, :]
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q = Q / -(freqs * np.diag(Q)).sum()
    sq = np.sqrt(freqs)
    S = Q * sq[:, None] / sq[None, :]
    S = (S + S.T) / 2.0
    w, V = np.linalg.eigh(S)
    evec = V / sq[:, None]          # U[x][i]
    inv_evec = V.T * sq[None, :]    # U^-1[i][x]
    if alpha is None:
        rates = np.array([1.0 / (1.0 - pinvar)])
        ncat = 1
    else:
        rates = discrete_gamma_rates(alpha, ncat, pinvar)
    props = np.full(ncat, (1.0 - pinvar) / ncat)
    return Model(Q, freqs, w, evec, inv_evec, rates, props, pinvar, name)


def gtr_model(rates6=(1.5, 2.4, 1.8, 1.9, 2.8, 1.0), freqs=(0.25, 0.26, 0.25, 0.24), alpha=0.9, ncat=4,
              pinvar=0.0):
    """GTR{AC,AG,AT,CG,CT,GT}+F{..}+G4{alpha}: the survey's DNA benchmark model."""
    a, b, c, d, e, f = rates6
    R = np.array([[0, a, b, c], [a, 0, d, e], [b, d, 0, f], [c, e, f, 0]], dtype=np.float64)
    return reversible_model(R, freqs, alpha, ncat, pinvar, name="GTR+G%d" % ncat)


def random_reversible_model(n, seed, alpha=0.9, ncat=4, pinvar=0.0, min_freq=None):
    """A random general time-reversible model on n states (stand-in for LG / GY: the reference's
    empirical matrices are constants of its source and are not copied)."""
    rng = np.random.default_rng(seed)
    R = rng.gamma(shape=1.0, scale=1.0, size=(n, n)) + 0.05
    freqs = rng.dirichlet(np.full(n, 5.0))
    if min_freq:
        freqs = np.maximum(freqs, min_freq)
    return reversible_model(R, freqs, alpha, ncat, pinvar, name="GTR%d+G%d" % (n, ncat))


# standard genetic code (NCBI table 1) in T,C,A,G order; codon states are numbered 16a+4b+c with A,C,G,T = 0..3
# (alignment.cpp:470-472), stop codons included: 64 states
_CODE_TCAG = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"


def standard_genetic_code():
    """amino acid (or '*') per codon state 16a+4b+c, A,C,G,T = 0..3"""
    tcag = {0: 2, 1: 1, 2: 3, 3: 0}  # A,C,G,T -> position in T,C,A,G
    return "".join(_CODE_TCAG[16 * tcag[i // 16] + 4 * tcag[(i % 16) // 4] + tcag[i % 4]] for i in range(64))


def gy94_sense_model(kappa=2.0, omega=0.5, nt_freqs=(0.25, 0.26, 0.25, 0.24)):
    """GY94 (Goldman & Yang 1994; the reference's ModelCodon "GY", model/modelcodon.cpp:454-560) on the 61 sense
    codons with F1X4 frequencies: rate 0 between codons that differ at more than one position, x kappa for a
    transition, x omega for a non-synonymous change.  For SIMULATING codon data only (the evaluated model comes
    from the C++ producer, iq-tree_amd/host/model_host.cpp).  -> (61-state Model, codon state of each of the 61)."""
    code = standard_genetic_code()
    sense = [i for i in range(64) if code[i] != "*"]
    nt = np.asarray(nt_freqs, dtype=np.float64)
    n = len(sense)
    R = np.zeros((n, n))
    for x, i in enumerate(sense):
        a = (i // 16, (i % 16) // 4, i % 4)
        for y, j in enumerate(sense):
            if i == j:
                continue
            b = (j // 16, (j % 16) // 4, j % 4)
            diff = [(p, q) for p, q in zip(a, b) if p != q]
            if len(diff) != 1:
                continue
            r = kappa if abs(diff[0][0] - diff[0][1]) == 2 else 1.0   # A<->G, C<->T
            if code[i] != code[j]:
                r *= omega
            R[x, y] = r
    f = np.array([nt[i // 16] * nt[(i % 16) // 4] * nt[i % 4] for i in sense])
    return reversible_model(R, f, None, 1, 0.0, name="GY94sim"), np.array(sense, dtype=np.uint8)


def codon_phylip(states):
    """states[ntaxa, nsites] of codon codes (16a+4b+c) -> PHYLIP text of 3*nsites nucleotide columns"""
    lut = np.array([[ord("ACGT"[i // 16]), ord("ACGT"[(i % 16) // 4]), ord("ACGT"[i % 4])] for i in range(64)], dtype=np.uint8)
    ntaxa, nsites = states.shape
    rows = ["%d %d" % (ntaxa, 3 * nsites)]
    for t in range(ntaxa):
        rows.append("%d %s" % (t, lut[states[t]].reshape(-1).tobytes().decode()))
    return "\n".join(rows) + "\n"


# -------------------------------------------------------------------------------------------
# trees
# -------------------------------------------------------------------------------------------
def random_tree_newick(ntaxa, seed, lo=0.02, hi=0.2, caterpillar=False):
    """Random unrooted binary tree by random pairwise joining, branch lengths U(lo,hi).
    Leaves are labelled with their taxon id."""
    rng = np.random.default_rng(seed)
    items = [str(i) for i in range(ntaxa)]
    if caterpillar:
        order = list(rng.permutation(ntaxa))
        cur = "%d" % order[0]
        first = True
        for t in order[1:-2]:
            l1, l2 = rng.uniform(lo, hi, 2)
            cur = "(%s:%.6f,%d:%.6f)" % (cur, l1, t, l2)
        l = rng.uniform(lo, hi, 3)
        return "(%s:%.6f,%d:%.6f,%d:%.6f);" % (cur, l[0], order[-2], l[1], order[-1], l[2])
    while len(items) > 3:
        i, j = sorted(rng.choice(len(items), 2, replace=False))
        l1, l2 = rng.uniform(lo, hi, 2)
        new = "(%s:%.6f,%s:%.6f)" % (items[i], l1, items[j], l2)
        items = [x for k, x in enumerate(items) if k not in (i, j)] + [new]
    l = rng.uniform(lo, hi, 3)
    return "(%s:%.6f,%s:%.6f,%s:%.6f);" % (items[0], l[0], items[1], l[1], items[2], l[2])


def random_multifurcating_newick(ntaxa, seed, lo=0.02, hi=0.2, max_children=5, p_multi=0.4):
    """Random unrooted tree with polytomies (user trees of `-t` / `-te`, consensus trees): random joining where a join
    takes 3..max_children items with probability p_multi, else 2."""
    rng = np.random.default_rng(seed)
    items = [str(i) for i in range(ntaxa)]
    while len(items) > 3:
        k = 2
        if rng.random() < p_multi:
            k = int(rng.integers(3, max_children + 1))
        k = min(k, len(items) - 2)
        if k < 2:
            break
        idx = sorted(rng.choice(len(items), k, replace=False))
        parts = ["%s:%.6f" % (items[i], rng.uniform(lo, hi)) for i in idx]
        items = [x for q, x in enumerate(items) if q not in idx] + ["(" + ",".join(parts) + ")"]
    return "(" + ",".join("%s:%.6f" % (x, rng.uniform(lo, hi)) for x in items) + ");"


def parse_newick(s):
    """-> nested (label, length, children) tuples; minimal parser for the generator's output."""
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


# -------------------------------------------------------------------------------------------
# simulation
# -------------------------------------------------------------------------------------------
def transition_matrices(model, t):
    """P[c] = U diag(exp(eval*rate_c*t)) U^-1, rows renormalised."""
    n = model.nstates
    U = model.evec.reshape(n, n)
    Ui = model.inv_evec.reshape(n, n)
    out = []
    for r in model.rates:
        P = (U * np.exp(model.eval * r * t)[None, :]) @ Ui
        P = np.clip(P, 0.0, None)
        out.append(P / P.sum(axis=1, keepdims=True))
    return out


def simulate_alignment(newick, model, nsites, seed, missing_frac=0.0, state_unknown=None):
    """Evolve nsites sites down the tree under `model` (category per site drawn from props).
    Returns states[ntaxa][nsites] uint8."""
    rng = np.random.default_rng(seed)
    tree = parse_newick(newick)
    n = model.nstates
    cat = rng.integers(0, model.ncat, nsites)
    root_states = rng.choice(n, size=nsites, p=model.freqs)
    leaves = {}

    def evolve(parent_states, t):
        Ps = transition_matrices(model, t)
        child = np.empty(nsites, dtype=np.int64)
        u = rng.random(nsites)
        for c in range(model.ncat):
            idx = np.nonzero(cat == c)[0]
            if idx.size == 0:
                continue
            cdf = np.cumsum(Ps[c], axis=1)
            rows = cdf[parent_states[idx]]
            child[idx] = np.minimum((u[idx, None] > rows).sum(axis=1), n - 1)
        return child

    def walk(nd, states):
        label, _, kids = nd
        if not kids:
            leaves[int(label)] = states
            return
        for k in kids:
            walk(k, evolve(states, k[1]))

    walk(tree, root_states)
    ntaxa = len(leaves)
    out = np.empty((ntaxa, nsites), dtype=np.uint8)
    for i in range(ntaxa):
        out[i] = leaves[i]
    if missing_frac > 0.0:
        mask = rng.random(out.shape) < missing_frac
        out[mask] = state_unknown if state_unknown is not None else n
    return out


def compress_patterns(states):
    """site columns -> unique patterns (first-occurrence order) + frequencies
    (alignment.cpp:674 addPattern keeps first-occurrence order)."""
    cols = np.ascontiguousarray(states.T)
    view = cols.view([("", cols.dtype)] * cols.shape[1]).reshape(-1)
    _, first, counts = np.unique(view, return_index=True, return_counts=True)
    order = np.argsort(first)
    pat = cols[first[order]]
    return np.ascontiguousarray(pat.T), counts[order].astype(np.float64)


def ptn_invar_for(states, model):
    """phylotreesse.cpp:543-569: p_invar*pi[const state] for constant patterns, else 0."""
    nptn = states.shape[1]
    out = np.zeros(nptn)
    if model.pinvar == 0.0:
        return out
    n = model.nstates
    first = states[0]
    const = np.all(states == first[None, :], axis=0) & (first < n)
    out[const] = model.pinvar * model.freqs[first[const]]
    return out


def make_workload(ntaxa, npatterns, model, seed, missing_frac=0.0, state_unknown=None):
    """Tree + exactly `npatterns` distinct patterns with frequencies (the named BASELINE shapes)."""
    nwk = random_tree_newick(ntaxa, seed)
    nsites = int(npatterns * 1.02) + 64
    for _ in range(8):
        st = simulate_alignment(nwk, model, nsites, seed + 1, missing_frac, state_unknown)
        pat, freq = compress_patterns(st)
        if pat.shape[1] >= npatterns:
            return nwk, np.ascontiguousarray(pat[:, :npatterns]), freq[:npatterns].copy()
        nsites = int(nsites * 1.5)
    raise RuntimeError("could not reach the requested number of distinct patterns")


# the BASELINE.json workloads (bench.py and tests/test_baseline_shapes_gpu.py build the SAME inputs from here):
# name -> (ntaxa, patterns, nstates, ncat, seq_type); seq_type as iqhost::SeqType (0 DNA, 1 protein, 2 codon)
BASELINE_SHAPES = {
    "dna": (50, 100000, 4, 4, 0),        # configs[1]: DNA 50 taxa x 100k patterns, GTR+G4
    "protein": (100, 50000, 20, 4, 1),   # configs[2]: protein 100 x 50k, 20-state +G4
    "dna4": (200, 1000000, 4, 4, 0),     # configs[3]: DNA 200 x 1M, sharded over the GPUs (125k per GPU at N = 8)
    "codon": (50, 20000, 64, 1, 2),      # configs[4]: codon 64-state 50 x 20k
    "mixture": (50, 10000, 20, 40, 1),   # C10+G4 shape (10 classes x 4 rates), not a BASELINE config
}


def baseline_model(workload, ncat=0):
    """The model of a BASELINE workload (and, for mixtures, the class the sites are simulated under)."""
    T0, P0, nst, C0, seq_type = BASELINE_SHAPES[workload]
    ncat = ncat or C0
    if workload == "mixture":
        model = mixture_model(20, 10, 7, alpha=0.9, ncat=4)
        return model, model.classes[0]
    if nst == 4:
        return gtr_model(rates6=(1.5, 2.4, 1.8, 1.9, 2.8, 1.0), freqs=(0.25, 0.26, 0.25, 0.24), alpha=0.9, ncat=ncat), None
    # random reversible 20-/64-state model of the LG+G4 / GY shape (the reference's empirical matrices are
    # constants of its source and are not copied); codon: ncat = 1 as GY+F1X4, stop-codon-like rare states
    m = random_reversible_model(nst, 7, alpha=0.9 if ncat > 1 else None, ncat=ncat, min_freq=1e-4)
    if workload == "protein":
        m.name = "LG-shaped random reversible 20-state matrix +G%d{0.9}" % ncat
    return m, None


class ProducedModel:
    """model arrays as the C++ producers return them (iqtree_amd.Alignment.build_model)"""

    def __init__(self, m, name):
        self.eval, self.evec, self.inv_evec = m.eval, m.evec.reshape(-1), m.inv_evec.reshape(-1)
        self.rates, self.props, self.freqs = m.rates, m.props, m.state_freq
        self.nstates, self.ncat, self.pinvar, self.name = m.nstates, m.ncat, m.p_invar, name
        self.Q = None


def codon_gy94_workload(T, P, shard):
    """BASELINE configs[4] as the reference runs it (`-st CODON -m GY+F1X4`): codon sites simulated under
    GY94 on the 61 sense codons, written as a nucleotide PHYLIP alignment, read back by the package's own reader
    (codon translation, pattern compression: alignment_host.cpp) and given the model the C++ GY94 producer builds from
    that alignment (F1X4 frequencies, rate matrix, eigen-system: model_host.cpp).  kappa 2, omega 0.5 fixed."""
    import sys
    pkg = sys.modules["iqtree_amd"]
    sim, sense = gy94_sense_model(2.0, 0.5)
    nwk = random_tree_newick(T, 1)
    nsites = int(P * 1.02) + 64
    while True:
        st = simulate_alignment(nwk, sim, nsites, 1000 + shard)
        pat, _ = compress_patterns(st)
        if pat.shape[1] >= P:
            break
        nsites = int(nsites * 1.3)
    aln = pkg.Alignment(content=codon_phylip(sense[pat[:, :P]]), seq_type="CODON")
    states, freq, _, _ = aln.arrays()
    if states.shape[1] != P:
        raise RuntimeError("codon reader returned %d patterns, expected %d" % (states.shape[1], P))
    model = ProducedModel(aln.build_model("GY{2.0,0.5}+F1X4"), "GY94{kappa=2,omega=0.5}+F1X4")
    return nwk, np.ascontiguousarray(states), freq.copy(), model


def baseline_workload(workload, ntaxa=0, patterns=0, shard=0, ncat=0):
    """-> (newick, patterns[ntaxa, P] uint8, ptn_freq[P], model).  The tree is the same for every shard
    (seed 1); shard r simulates its own sites (seed 1000 + r), as one rank of a pattern-sharded run does."""
    T0, P0, nst, C0, seq_type = BASELINE_SHAPES[workload]
    T, P = ntaxa or T0, patterns or P0
    if workload == "codon":
        return codon_gy94_workload(T, P, shard)
    model, sim_model = baseline_model(workload, ncat)
    nwk = random_tree_newick(T, 1)
    nsites = int(P * 1.02) + 64
    while True:
        st = simulate_alignment(nwk, sim_model or model, nsites, 1000 + shard)
        pat, freq = compress_patterns(st)
        if pat.shape[1] >= P:
            break
        nsites = int(nsites * 1.3)
    return nwk, np.ascontiguousarray(pat[:, :P]), freq[:P].copy(), model


# -------------------------------------------------------------------------------------------
# mixture models (ModelMixture: phylokernelmixture.h / phylokernelmixrate.h)
# -------------------------------------------------------------------------------------------

class MixtureModel:
    """(class, rate) components in the reference's block order [class][rate]: component q uses the
    eigen-system of class cat_class[q]; eval / evec / inv_evec are the per-class arrays concatenated;
    props[q] = class weight x category proportion.  `classes` are the plain per-class models (with the
    component weights of that class as props) for an independent evaluation."""

    def __init__(self, classes, cat_class, rates, props, name=""):
        self.classes = classes
        self.nclass = len(classes)
        self.nstates = classes[0].nstates
        self.cat_class = np.ascontiguousarray(cat_class, dtype=np.int32)
        self.eval = np.concatenate([m.eval for m in classes])
        self.evec = np.concatenate([m.evec for m in classes])
        self.inv_evec = np.concatenate([m.inv_evec for m in classes])
        self.rates = np.ascontiguousarray(rates, dtype=np.float64)
        self.props = np.ascontiguousarray(props, dtype=np.float64)
        self.ncat = len(self.rates)
        self.freqs = classes[0].freqs
        self.Q = classes[0].Q
        self.pinvar = 0.0
        self.name = name


def mixture_model(n, nclass, seed, alpha=0.9, ncat=4, fused=False):
    """Profile mixture (C10..C60 style: shared exchangeabilities, per-class frequencies) x discrete Gamma,
    or, fused=True, one rate per class (LG4X style, phylokernelmixrate.h): ncomp = nclass."""
    rng = np.random.default_rng(seed)
    R = rng.gamma(shape=1.0, scale=1.0, size=(n, n)) + 0.05
    w = rng.dirichlet(np.full(nclass, 8.0))
    if fused:
        r = rng.gamma(shape=2.0, scale=0.5, size=nclass) + 0.05
        r = r / (w * r).sum()                     # mean rate 1
        rates, props, cat_class = r, w, np.arange(nclass)
        per_class = [(np.array([r[m]]), np.array([w[m]])) for m in range(nclass)]
    else:
        g = discrete_gamma_rates(alpha, ncat)
        rates = np.tile(g, nclass)
        props = np.repeat(w, ncat) / ncat
        cat_class = np.repeat(np.arange(nclass), ncat)
        per_class = [(g, np.full(ncat, w[m] / ncat)) for m in range(nclass)]
    classes = []
    for m in range(nclass):
        f = rng.dirichlet(np.full(n, 3.0))
        f = np.maximum(f, 2e-3)
        base = reversible_model(R, f, None, 1, 0.0)
        classes.append(Model(base.Q, base.freqs, base.eval, base.evec, base.inv_evec, per_class[m][0], per_class[m][1],
                             0.0, "class%d" % m))
    return MixtureModel(classes, cat_class, rates, props, name="MIX%d%s" % (nclass, "fused" if fused else "+G%d" % ncat))
