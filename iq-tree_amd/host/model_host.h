// model_host.h -- producers of the likelihood kernels' model inputs.
//
// The kernels take an eigen-system (eval, U = evec, U^-1 = inv_evec), category rates and
// proportions.  In the reference these come from (file:line = /root/reference):
//   ModelGTR::decomposeRateMatrix            model/modelgtr.cpp:608-722
//   EigenDecomposition::eigensystem_sym      eigendecomposition.cpp:167-296  (normalise Q to one
//       substitution per unit time :306-346, drop zero-frequency states :348-371, symmetrise with
//       sqrt(pi) :373-394, symmetric eigen-solver, U = V/sqrt(pi), U^-1 = V^T*sqrt(pi))
//   RateGamma::computeRates / computeRatesMean   model/rategamma.cpp:86-150 (+ the four published
//       routines it cites: AS 291 lnGamma, AS 32 incomplete gamma, AS 70 normal point, AS 91 chi2 point)
//   ModelCodon (GY94 rate attributes / kappa-omega scaling)   model/modelcodon.cpp:468-540, 671-700
//   Alignment::computeCodonFreq (F1X4 / F3X4)   alignment.cpp:2990-3080
// This is host-side C++ (the reference's is too); nothing here runs on the likelihood hot path.
// The symmetric eigen-solver is a cyclic Jacobi iteration, not the reference's tred2/tqli pair:
// eigenvector order and signs differ, the likelihood does not depend on either.
#pragma once
#include <string>
#include <vector>

namespace iqhost {

struct EigenSystem {
    int n = 0;
    std::vector<double> eval, evec, inv_evec;  // evec[x*n+i] = U[x][i]; inv_evec[i*n+x] = U^-1[i][x]
};

// rate_matrix: n*n exchangeabilities (symmetric, diagonal ignored) -- or, with
// ignore_state_freq, the complete off-diagonal rates q_ij (codon models that already folded the
// nucleotide frequencies in).  state_freq need not be normalised.
void decomposeRateMatrix(const double *rate_matrix, const double *state_freq, int n, EigenSystem &out,
                         bool ignore_state_freq = false);

// RateGamma::computeRates: mean (default) or median discrete-Gamma rates, divided by (1 - p_invar)
void discreteGammaRates(double gamma_shape, int ncategory, bool cut_median, double p_invar, double *rates);
double cmpLnGamma(double alpha);
double cmpIncompleteGamma(double x, double alpha, double ln_gamma_alpha);
double cmpPointNormal(double prob);
double cmpPointChi2(double prob, double v);

// Standard genetic code in the reference's codon numbering (A,C,G,T = 0..3, codon = 16a+4b+c)
const char *geneticCode(int ncbi_table);

// A parsed `-m` string, e.g. "GTR{1.5,2.4,1.8,1.9,2.8}+F{0.25,0.26,0.25,0.24}+I{0.1}+G4{0.9}",
// "HKY{2.0}+G4{0.5}", "JC", "POISSON+G4{1.0}", "GY{2.0,0.5}+F1X4", "file.paml+G4{0.9}", "...+ASC".
struct ModelSpec {
    std::string name;                 // JC F81 K80 HKY TN GTR POISSON GY <paml file>
    std::vector<double> params;       // rate parameters in the reference's order
    enum FreqType { FREQ_DEFAULT, FREQ_EQUAL, FREQ_USER, FREQ_EMPIRICAL, FREQ_CODON_1x4, FREQ_CODON_3x4 } freq_type = FREQ_DEFAULT;
    std::vector<double> user_freq;
    int ncat = 1;
    double gamma_shape = 1.0;
    bool gamma_median = false;
    double p_invar = 0.0;
    bool ascertainment = false;
};
ModelSpec parseModelString(const std::string &s);

struct ModelInputs {      // what PhyloTree::setModel / iqhip_set_model consume
    int nstates = 0, ncat = 1;
    EigenSystem eig;
    std::vector<double> state_freq, rates, props;
    double p_invar = 0.0;
};

class Alignment;
// builds the rate matrix named by `spec` for the alignment's data type, takes frequencies from the
// spec or the alignment, decomposes, and produces the category rates
void buildModel(const ModelSpec &spec, const Alignment &aln, ModelInputs &out);

}  // namespace iqhost
