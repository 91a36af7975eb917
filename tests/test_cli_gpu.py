"""The stand-alone driver (cli/iqhip_lnl.cpp -> iq-tree_amd/lib/iqhip_lnl): alignment file + tree file +
`-m` string in, lnL / df / ddf / .sitelh out -- the reference's `-s A -te T -m M -blfix -n 0 -wsl` evaluation
(SURVEY 8c "CLI-level oracle") with every stage native: C++ readers and model producers, HIP kernels.
Checked against the oracle fed by the same C++ producers through ctypes."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(ROOT, "iq-tree_amd", "lib", "iqhip_lnl")
EXAMPLE = os.path.join(HERE, "golden", "example.phy")
MODEL = "GTR{1.513,2.393,1.769,1.912,2.838}+F{0.249,0.262,0.251,0.238}+G4{0.934}"


def named_tree(nwk, names):
    return re.sub(r"([(,])(\d+):", lambda m: "%s%s:" % (m.group(1), names[int(m.group(2))]), nwk)


def run_cli(args):
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    return r.stdout


def read_report(prefix):
    out = {}
    for line in open(prefix + ".iqhip"):
        k, _, v = line.strip().partition(" ")
        out[k] = v
    return out


def test_fixed_tree_evaluation_matches_oracle(pkg, synth, oracle, tmp_path):
    aln = pkg.Alignment(EXAMPLE)
    st, fr, sp, _ = aln.arrays()
    model = aln.build_model(MODEL)
    nwk = synth.random_tree_newick(44, 12)
    tf = tmp_path / "t.nwk"
    tf.write_text(named_tree(nwk, aln.seq_names) + "\n")
    pre = str(tmp_path / "run")
    out = run_cli(["-s", EXAMPLE, "-te", str(tf), "-m", MODEL, "-blfix", "-n", "0", "-wsl", "-pre", pre])
    assert "44 sequences with 384 columns and 355 patterns" in out
    rep = read_report(pre)
    ot = oracle.OracleTree(nwk, 4, 0, st, fr, None, model)
    ref, (la, lb) = ot.likelihood()
    ptn = ot.branch_lnl(la, lb)[1]
    lnl = float(rep["lnL"])
    assert abs(lnl - ref) <= 1e-9 * abs(ref)
    assert rep["lnL_input_tree"] == rep["lnL"]
    lines = open(pre + ".sitelh").read().splitlines()
    assert lines[0] == "1 384"
    site = np.array([float(x) for x in lines[1].split()[1:]])
    assert site.size == 384 and abs(site.sum() - ref) < 1e-2          # 6 significant digits per site in the file
    assert np.allclose(site, np.asarray(ptn)[sp], rtol=1e-5)
    # the same evaluation through the Python view of the mirror gives the identical number
    t = pkg.PhyloTree(nwk)
    t.set_alignment(4, 0, st, fr)
    t.set_model(model)
    t.attach_engine(0)
    assert t.compute_likelihood() == lnl
    a, b = t.current_branch()
    df, ddf = t.compute_likelihood_derv(a, b)
    assert float(rep["df"]) == df and float(rep["ddf"]) == ddf


def test_branch_length_optimisation_and_asc(pkg, synth, oracle, tmp_path):
    aln = pkg.Alignment(EXAMPLE)
    st, fr, _, _ = aln.arrays()
    model = aln.build_model(MODEL)
    nwk = synth.random_tree_newick(44, 12)
    tf = tmp_path / "t.nwk"
    tf.write_text(named_tree(nwk, aln.seq_names) + "\n")
    pre = str(tmp_path / "opt")
    run_cli(["-s", EXAMPLE, "-te", str(tf), "-m", MODEL, "-pre", pre])
    rep = read_report(pre)
    assert float(rep["lnL"]) > float(rep["lnL_input_tree"]) + 100
    # re-evaluate the written tree with the oracle
    names = aln.seq_names
    back = re.sub(r"([(,])([^(),:]+):", lambda m: "%s%d:" % (m.group(1), names.index(m.group(2))), rep["tree"])
    ot = oracle.OracleTree(back, 4, 0, st, fr, None, model)
    ref, _ = ot.likelihood()
    assert abs(float(rep["lnL"]) - ref) <= 1e-8 * abs(ref)
    # a variable-sites-only alignment with +ASC
    rows = st[:, [p for p in range(st.shape[1]) if len(set(st[:, p].tolist())) > 1 and st[:, p].max() < 4]][:, :60]
    nuc = "ACGT"
    phy = tmp_path / "var.phy"
    phy.write_text(" 44 %d\n" % rows.shape[1] + "".join("%s %s\n" % (names[i], "".join(nuc[s] for s in rows[i])) for i in range(44)))
    pre2 = str(tmp_path / "asc")
    out = run_cli(["-s", str(phy), "-te", str(tf), "-m", "HKY{2.0}+F{0.3,0.2,0.2,0.3}+G4{0.7}+ASC", "-blfix", "-pre", pre2])
    assert "4 unobservable constant patterns" in out
    a2 = pkg.Alignment(str(phy))
    m2 = a2.build_model("HKY{2.0}+F{0.3,0.2,0.2,0.3}+G4{0.7}+ASC")
    nsite = a2.nsite
    a2.append_unobserved_const_patterns()
    s2, f2, _, _ = a2.arrays()
    ot2 = oracle.OracleTree(nwk, 4, 0, s2, f2, None, m2, n_unobs=4, nsites=nsite)
    ref2, _ = ot2.likelihood()
    assert abs(float(read_report(pre2)["lnL"]) - ref2) <= 1e-9 * abs(ref2)


def test_cli_errors(tmp_path):
    r = subprocess.run([BIN, "-s", "/nonexistent.phy", "-te", "/nonexistent.nwk", "-m", "JC"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open alignment file" in r.stderr
    r = subprocess.run([BIN, "-s", EXAMPLE], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_reference_protein_alignment(pkg, synth, oracle, tmp_path):
    """prot_M126_27_269.phy (the protein data file of the reference's test scripts, test_scripts/test_configs.txt):
    27 sequences x 269 amino acids with gaps and X; Poisson+G4 through the stand-alone driver, and a profile
    mixture through the library, both against the oracle."""
    prot = os.path.join(HERE, "golden", "prot_M126_27_269.phy")
    aln = pkg.Alignment(prot)
    assert (aln.nseq, aln.nsite, aln.nstates, aln.seq_type) == (27, 269, 20, pkg.SEQ_PROTEIN)
    st, fr, sp, _ = aln.arrays()
    assert fr.sum() == 269 and st.max() == 23
    nwk = synth.random_tree_newick(27, 3)
    tf = tmp_path / "p.nwk"
    tf.write_text(named_tree(nwk, aln.seq_names) + "\n")
    pre = str(tmp_path / "prot")
    run_cli(["-s", prot, "-te", str(tf), "-m", "POISSON+G4{0.8}", "-blfix", "-wsl", "-pre", pre])
    model = aln.build_model("POISSON+G4{0.8}")
    ot = oracle.OracleTree(nwk, 20, 1, st, fr, None, model)
    ref, _ = ot.likelihood()
    assert abs(float(read_report(pre)["lnL"]) - ref) <= 1e-9 * abs(ref)
    mix = synth.mixture_model(20, 3, 77, ncat=4)
    t = pkg.PhyloTree(nwk)
    t.set_alignment(20, 1, st, fr)
    t.set_model(mix)
    t.attach_engine(0)
    om = oracle.OracleTree(nwk, 20, 1, st, fr, None, mix)
    refm, _ = om.likelihood()
    assert abs(t.compute_likelihood() - refm) <= 1e-9 * abs(refm)
    assert refm > ref - 5000        # sanity: same data, both finite and of the same order
