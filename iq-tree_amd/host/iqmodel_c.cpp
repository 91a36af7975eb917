// iqmodel_c.cpp -- flat C view of the model / alignment producers (model_host.h, alignment_host.h)
// for ctypes; test and tool plumbing, not part of the drop-in boundary.
#include <string.h>

#include <string>
#include <vector>

#include "alignment_host.h"
#include "model_host.h"

using namespace iqhost;

static thread_local std::string g_model_err;

#define IQM_TRY(body)                    \
    try {                                \
        body;                            \
        return 0;                        \
    } catch (const std::exception &ex) { \
        g_model_err = ex.what();         \
        return 1;                        \
    }

extern "C" {

const char *iqmodel_last_error(void) { return g_model_err.c_str(); }

int iqmodel_decompose(const double *rate_matrix, const double *state_freq, int n, int ignore_state_freq,
                      double *eval, double *evec, double *inv_evec) {
    IQM_TRY({
        EigenSystem es;
        decomposeRateMatrix(rate_matrix, state_freq, n, es, ignore_state_freq != 0);
        memcpy(eval, es.eval.data(), sizeof(double) * n);
        memcpy(evec, es.evec.data(), sizeof(double) * n * n);
        memcpy(inv_evec, es.inv_evec.data(), sizeof(double) * n * n);
    });
}
int iqmodel_gamma_rates(double shape, int ncat, int median, double p_invar, double *rates) {
    IQM_TRY(discreteGammaRates(shape, ncat, median != 0, p_invar, rates));
}
double iqmodel_ln_gamma(double a) { return cmpLnGamma(a); }
double iqmodel_incomplete_gamma(double x, double a) { return cmpIncompleteGamma(x, a, cmpLnGamma(a)); }
double iqmodel_point_normal(double p) { return cmpPointNormal(p); }
double iqmodel_point_chi2(double p, double v) { return cmpPointChi2(p, v); }
const char *iqmodel_genetic_code(int table) { return geneticCode(table); }

// ---- alignment ---------------------------------------------------------------------------
int iqaln_read(void **out, const char *filename_or_null, const char *content_or_null, const char *seq_type) {
    *out = nullptr;
    Alignment *a = new Alignment();
    try {
        if (filename_or_null) a->readFile(filename_or_null, seq_type ? seq_type : "");
        else a->readString(content_or_null, seq_type ? seq_type : "");
        *out = a;
        return 0;
    } catch (const std::exception &ex) {
        g_model_err = ex.what();
        delete a;
        return 1;
    }
}
void iqaln_destroy(void *h) { delete (Alignment *)h; }
int iqaln_nseq(void *h) { return ((Alignment *)h)->getNSeq(); }
int iqaln_nsite(void *h) { return ((Alignment *)h)->getNSite(); }
int iqaln_npattern(void *h) { return ((Alignment *)h)->getNPattern(); }
int iqaln_nstates(void *h) { return ((Alignment *)h)->num_states; }
int iqaln_seq_type(void *h) { return (int)((Alignment *)h)->seq_type; }
int iqaln_state_unknown(void *h) { return ((Alignment *)h)->STATE_UNKNOWN; }
double iqaln_frac_const_sites(void *h) { return ((Alignment *)h)->frac_const_sites; }
const char *iqaln_seq_name(void *h, int i) { return ((Alignment *)h)->seq_names[i].c_str(); }
int iqaln_append_unobserved(void *h, int *n_out) { IQM_TRY(*n_out = ((Alignment *)h)->appendUnobservedConstPatterns()); }
int iqaln_get(void *h, uint8_t *states, double *ptn_freq, int *site_pattern, int *const_char) {
    IQM_TRY({
        Alignment *a = (Alignment *)h;
        std::vector<uint8_t> s;
        std::vector<double> f;
        a->statesByLeaf(s);
        a->ptnFreq(f);
        if (states) memcpy(states, s.data(), s.size());
        if (ptn_freq) memcpy(ptn_freq, f.data(), f.size() * sizeof(double));
        if (site_pattern) memcpy(site_pattern, a->site_pattern.data(), a->site_pattern.size() * sizeof(int));
        if (const_char)
            for (size_t p = 0; p < a->patterns.size(); p++) const_char[p] = a->patterns[p].is_const ? a->patterns[p].const_char : -1;
    });
}
int iqaln_ptn_invar(void *h, double p_invar, const double *state_freq, double *out) {
    IQM_TRY({
        std::vector<double> v;
        ((Alignment *)h)->ptnInvar(p_invar, state_freq, v);
        memcpy(out, v.data(), v.size() * sizeof(double));
    });
}
int iqaln_state_freq(void *h, double *out) { IQM_TRY(((Alignment *)h)->computeStateFreq(out)); }
int iqaln_codon_freq(void *h, int f3x4, double *state_freq, double *ntfreq) {
    IQM_TRY(((Alignment *)h)->computeCodonFreq(f3x4 != 0, state_freq, ntfreq));
}
int iqaln_write_sitelh(void *h, const char *filename, const double *pattern_lh) {
    IQM_TRY(writeSiteLh(filename, *(Alignment *)h, pattern_lh));
}

// ---- -m string -> kernel inputs -------------------------------------------------------------
// out arrays sized by the caller: eval[n], evec[n*n], inv_evec[n*n], state_freq[n], rates[64], props[64]
int iqmodel_build(void *aln, const char *model_string, int *ncat, double *p_invar, int *asc, double *eval,
                  double *evec, double *inv_evec, double *state_freq, double *rates, double *props) {
    IQM_TRY({
        ModelSpec spec = parseModelString(model_string);
        if (spec.ncat > 64) throw std::runtime_error("too many rate categories");
        ModelInputs mi;
        buildModel(spec, *(Alignment *)aln, mi);
        const int n = mi.nstates;
        *ncat = mi.ncat;
        *p_invar = mi.p_invar;
        *asc = spec.ascertainment;
        memcpy(eval, mi.eig.eval.data(), sizeof(double) * n);
        memcpy(evec, mi.eig.evec.data(), sizeof(double) * n * n);
        memcpy(inv_evec, mi.eig.inv_evec.data(), sizeof(double) * n * n);
        memcpy(state_freq, mi.state_freq.data(), sizeof(double) * n);
        memcpy(rates, mi.rates.data(), sizeof(double) * mi.ncat);
        memcpy(props, mi.props.data(), sizeof(double) * mi.ncat);
    });
}

}  // extern "C"
