// iqhip_lnl -- stand-alone driver of the MI355X likelihood path, the counterpart of the reference's
// "evaluate one fixed tree" command line:
//     iqtree -s A.phy -te T.nwk -m 'GTR{a,b,c,d,e}+F{..}+G4{alpha}' [-blfix] -n 0 [-wsl] -pre X
// Everything it does goes through the same host mirror and C ABI as the tests: read the alignment
// (alignment_host), build the model inputs (model_host), read the tree, setLikelihoodKernel(HIP),
// computeLikelihood(), optionally optimizeAllBranches(), write X.iqhip (full-precision numbers) and
// X.sitelh.  There is no CPU path: without a GPU it fails with the engine's error.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../iq-tree_amd/host/alignment_host.h"
#include "../iq-tree_amd/host/model_host.h"
#include "../iq-tree_amd/host/phylo_host.h"

using namespace iqhost;

static void usage() {
    fprintf(stderr,
            "usage: iqhip_lnl -s <alignment> -te <newick file> -m <model> [-st DNA|AA|CODON[n]] [-pre <prefix>]\n"
            "                 [-blfix] [-wsl] [-dev <gpu>] [-reps <n>] [-nolhmemsave]\n"
            "  model: e.g. 'GTR{1.5,2.4,1.8,1.9,2.8}+F{0.25,0.26,0.25,0.24}+I{0.1}+G4{0.9}', 'HKY{2}+G4{0.5}', JC,\n"
            "         POISSON+G4{1}, <paml matrix file>+G4{0.9}, 'GY{kappa,omega}+F1X4', any of them +ASC\n");
}

int main(int argc, char **argv) {
    std::string aln_file, tree_file, model_str, seq_type, prefix;
    bool blfix = false, wsl = false, all_branch = false;
    int dev = 0, reps = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) { usage(); exit(2); }
            return argv[++i];
        };
        if (a == "-s") aln_file = next();
        else if (a == "-te") tree_file = next();
        else if (a == "-m") model_str = next();
        else if (a == "-st") seq_type = next();
        else if (a == "-pre") prefix = next();
        else if (a == "-blfix") blfix = true;
        else if (a == "-wsl") wsl = true;
        else if (a == "-nolhmemsave") all_branch = true;
        else if (a == "-dev") dev = atoi(next().c_str());
        else if (a == "-reps") reps = atoi(next().c_str());
        else if (a == "-n") next();  // accepted for command-line compatibility (-n 0)
        else { usage(); return 2; }
    }
    if (aln_file.empty() || tree_file.empty() || model_str.empty()) { usage(); return 2; }
    if (prefix.empty()) prefix = aln_file;
    try {
        Alignment aln;
        aln.readFile(aln_file, seq_type);
        printf("Alignment has %d sequences with %d columns and %d patterns\n", aln.getNSeq(), aln.getNSite(), aln.getNPattern());
        ModelSpec spec = parseModelString(model_str);
        ModelInputs mi;
        buildModel(spec, aln, mi);  // frequencies from the observed patterns only
        const int nsite = aln.getNSite();
        if (spec.ascertainment) {
            const int k = aln.appendUnobservedConstPatterns();
            printf("Ascertainment bias correction: %d unobservable constant patterns\n", k);
        }
        std::ifstream tin(tree_file.c_str());
        if (!tin) throw std::runtime_error("cannot open tree file " + tree_file);
        std::stringstream tss;
        tss << tin.rdbuf();

        PhyloTree tree;
        tree.readTreeString(tss.str(), aln.seq_names);
        if (tree.leafNum != aln.getNSeq()) throw std::runtime_error("Tree and alignment have different numbers of taxa");
        std::vector<uint8_t> states;
        std::vector<double> freq, invar;
        aln.statesByLeaf(states);
        aln.ptnFreq(freq);
        aln.ptnInvar(mi.p_invar, mi.state_freq.data(), invar);
        tree.setAlignment(aln.num_states, aln.seq_type, aln.getNPattern(), states.data(), freq.data(), invar.data());
        if (spec.ascertainment) tree.setAscertainment(aln.n_unobserved, (double)nsite);
        tree.setModel(mi.ncat, mi.eig.eval.data(), mi.eig.evec.data(), mi.eig.inv_evec.data(), mi.rates.data(), mi.props.data());
        tree.lh_mem_save = all_branch ? LM_ALL_BRANCH : LM_PER_NODE;
        tree.setLikelihoodKernel(LK_EIGEN_HIP);
        tree.attachEngine(dev);
        tree.initializeAllPartialLh();
        tree.clearAllPartialLH();
        std::vector<double> pattern_lh(aln.getNPattern());
        double lnl = tree.computeLikelihood(pattern_lh.data());
        printf("Log-likelihood of the input tree: %.17g\n", lnl);
        const double lnl_input = lnl;
        if (!blfix) {
            lnl = tree.optimizeAllBranches();
            printf("Log-likelihood after branch-length optimisation: %.17g\n", lnl);
            tree.clearAllPartialLH();
            lnl = tree.computeLikelihood(pattern_lh.data());
        }
        double df = 0.0, ddf = 0.0;
        tree.theta_computed = false;
        tree.computeLikelihoodDerv(tree.current_it, tree.current_it_back->node, df, ddf);
        double upd_per_s = 0.0;
        if (reps > 0) {
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) {
                tree.clearAllPartialLH();
                lnl = tree.computeLikelihood();
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            upd_per_s = (double)reps * (tree.leafNum - 2) * (double)aln.getNPattern() / sec;
            printf("%d traversals in %.4f s: %.1f M pattern-node updates/s\n", reps, sec, upd_per_s / 1e6);
        }
        {
            std::ofstream out((prefix + ".iqhip").c_str());
            char buf[128];
            out << "model " << model_str << "\n";
            out << "nseq " << aln.getNSeq() << " nsite " << nsite << " npattern " << aln.getNPattern() - aln.n_unobserved << "\n";
            snprintf(buf, sizeof buf, "%.17g", lnl_input); out << "lnL_input_tree " << buf << "\n";
            snprintf(buf, sizeof buf, "%.17g", lnl);       out << "lnL " << buf << "\n";
            snprintf(buf, sizeof buf, "%.17g", df);        out << "df " << buf << "\n";
            snprintf(buf, sizeof buf, "%.17g", ddf);       out << "ddf " << buf << "\n";
            out << "rates";
            for (double r : mi.rates) { snprintf(buf, sizeof buf, " %.17g", r); out << buf; }
            out << "\ntree " << tree.getTreeString() << "\n";
        }
        if (wsl) {
            writeSiteLh(prefix + ".sitelh", aln, pattern_lh.data());
            printf("Site log-likelihoods printed to %s.sitelh\n", prefix.c_str());
        }
        printf("BEST SCORE FOUND : %.3f\n", lnl);
    } catch (const std::exception &ex) {
        fprintf(stderr, "ERROR: %s\n", ex.what());
        return 2;  // outError(): exit(2), tools.cpp:99-106
    }
    return 0;
}
