// model_host.cpp -- see model_host.h for the reference functions this follows.
#include "model_host.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <stdexcept>

#include "alignment_host.h"

namespace iqhost {

// ------------------------------------------------------------------------------------------
// Discrete Gamma (model/rategamma.cpp:86-150).  The four numerical routines are the published
// algorithms the reference cites, restated with the same constants and stopping rules so that
// the category rates come out the same: AS 291 (Pike & Hill 1966), AS 32 (Bhattacharjee 1970),
// AS 70 (Odeh & Evans 1974), AS 91 (Best & Roberts 1975).
// ------------------------------------------------------------------------------------------

double cmpLnGamma(double alpha) {  // rategamma.cpp:282-299
    double x = alpha, f = 0.0;
    if (x < 7.0) {
        double prod = 1.0, z = x;
        for (; z < 7.0; z += 1.0) prod *= z;  // x (x+1) ... up to the first value >= 7
        x = z;
        f = -log(prod);
    }
    const double z = 1.0 / (x * x);
    return f + (x - 0.5) * log(x) - x + .918938533204673 +
           (((-.000595238095238 * z + .000793650793651) * z - .002777777777778) * z + .083333333333333) / x;
}

double cmpIncompleteGamma(double x, double alpha, double ln_gamma_alpha) {  // rategamma.cpp:302-359
    const double p = alpha, accurate = 1e-8, overflow = 1e30;
    if (x == 0.0) return 0.0;
    if (x < 0.0 || p <= 0.0) return -1.0;
    const double factor = exp(p * log(x) - x - ln_gamma_alpha);
    if (!(x > 1.0 && x >= p)) {  // series expansion
        double gin = 1.0, term = 1.0, rn = p;
        do {
            rn += 1.0;
            term *= x / rn;
            gin += term;
        } while (term > accurate);
        return gin * factor / p;
    }
    // continued fraction
    double a = 1.0 - p, b = a + x + 1.0, term = 0.0;
    double pn[6] = {1.0, x, x + 1.0, x * b, 0.0, 0.0};
    double gin = pn[2] / pn[3];
    for (;;) {
        a += 1.0;
        b += 2.0;
        term += 1.0;
        const double an = a * term;
        pn[4] = b * pn[2] - an * pn[0];
        pn[5] = b * pn[3] - an * pn[1];
        if (pn[5] != 0.0) {
            const double rn = pn[4] / pn[5];
            const double dif = fabs(gin - rn);
            if (dif <= accurate && dif <= accurate * rn) break;
            gin = rn;
        }
        for (int i = 0; i < 4; i++) pn[i] = pn[i + 2];
        if (fabs(pn[4]) >= overflow)
            for (int i = 0; i < 4; i++) pn[i] /= overflow;
    }
    return 1.0 - factor * gin;
}

double cmpPointNormal(double prob) {  // rategamma.cpp:366-391
    const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245, a4 = -.453642210148e-4;
    const double b0 = .0993484626060, b1 = .588581570495, b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
    const double p1 = (prob < 0.5 ? prob : 1.0 - prob);
    if (p1 < 1e-20) return -9999.0;
    const double y = sqrt(log(1.0 / (p1 * p1)));
    const double z = y + ((((y * a4 + a3) * y + a2) * y + a1) * y + a0) / ((((y * b4 + b3) * y + b2) * y + b1) * y + b0);
    return prob < 0.5 ? -z : z;
}

double cmpPointChi2(double prob, double v) {  // rategamma.cpp:397-459
    const double e = .5e-6, aa = .6931471805, p = prob;
    if (p < .000002 || p > .999998 || v <= 0.0) return -1.0;
    const double g = cmpLnGamma(v / 2.0);
    const double xx = v / 2.0, c = xx - 1.0;
    double ch;
    if (v < -1.24 * log(p)) {  // small chi-square
        ch = pow(p * xx * exp(g + xx * aa), 1.0 / xx);
        if (ch - e < 0.0) return ch;
    } else if (v <= .32) {
        ch = 0.4;
        const double a = log(1.0 - p);
        double q;
        do {
            q = ch;
            const double p1 = 1.0 + ch * (4.67 + ch);
            const double p2 = ch * (6.73 + ch * (6.66 + ch));
            const double t = -0.5 + (4.67 + 2.0 * ch) / p1 - (6.73 + ch * (13.32 + 3.0 * ch)) / p2;
            ch -= (1.0 - exp(a + g + .5 * ch + c * aa) * p2 / p1) / t;
        } while (fabs(q / ch - 1.0) - .01 > 0.0);
    } else {
        const double x = cmpPointNormal(p);
        const double p1 = 0.222222 / v;
        ch = v * pow(x * sqrt(p1) + 1.0 - p1, 3.0);
        if (ch > 2.2 * v + 6.0) ch = -2.0 * (log(1.0 - p) - c * log(.5 * ch) + g);
    }
    double q;
    do {  // seven-term Taylor refinement
        q = ch;
        const double p1 = .5 * ch;
        double t = cmpIncompleteGamma(p1, xx, g);
        if (t < 0.0) return -1.0;
        const double p2 = p - t;
        t = p2 * exp(xx * aa + g + p1 - c * log(ch));
        const double b = t / ch, a = 0.5 * t - b * c;
        const double s1 = (210 + a * (140 + a * (105 + a * (84 + a * (70 + 60 * a))))) / 420;
        const double s2 = (420 + a * (735 + a * (966 + a * (1141 + 1278 * a)))) / 2520;
        const double s3 = (210 + a * (462 + a * (707 + 932 * a))) / 2520;
        const double s4 = (252 + a * (672 + 1182 * a) + c * (294 + a * (889 + 1740 * a))) / 5040;
        const double s5 = (84 + 264 * a + c * (175 + 606 * a)) / 2520;
        const double s6 = (120 + c * (346 + 127 * c)) / 5040;
        ch += t * (1 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
    } while (fabs(q / ch - 1.0) > e);
    return ch;
}

void discreteGammaRates(double gamma_shape, int ncategory, bool cut_median, double p_invar, double *rates) {
    if (ncategory == 1) {  // rategamma.cpp:90-93 (returns before the p_invar division)
        rates[0] = 1.0;
        return;
    }
    if (!cut_median) {  // computeRatesMean, rategamma.cpp:136-150 (Yang 1994, eqs 9 and 10)
        const double lnga1 = cmpLnGamma(gamma_shape + 1.0);
        std::vector<double> freqK(ncategory);
        for (int i = 0; i < ncategory - 1; i++)
            freqK[i] = cmpPointChi2((i + 1.0) / ncategory, 2.0 * gamma_shape) / (2.0 * gamma_shape);
        for (int i = 0; i < ncategory - 1; i++)
            freqK[i] = cmpIncompleteGamma(freqK[i] * gamma_shape, gamma_shape + 1.0, lnga1);
        rates[0] = freqK[0] * ncategory;
        rates[ncategory - 1] = (1.0 - freqK[ncategory - 2]) * ncategory;
        for (int i = 1; i < ncategory - 1; i++) rates[i] = (freqK[i] - freqK[i - 1]) * ncategory;
    } else {  // rategamma.cpp:98-114
        double sum = 0.0;
        for (int cat = 0; cat < ncategory; cat++) {
            const double prob = (2.0 * cat + 1) / (2.0 * ncategory);
            double r = cmpPointChi2(prob, 2.0 * gamma_shape) / (2.0 * gamma_shape);
            rates[cat] = r < 0.0 ? -r : r;
        }
        for (int cat = 0; cat < ncategory; cat++) sum += rates[cat];
        for (int cat = 0; cat < ncategory; cat++) rates[cat] = rates[cat] * ncategory / sum;
    }
    for (int cat = 0; cat < ncategory; cat++) rates[cat] = rates[cat] / (1.0 - p_invar);  // :120-123
}

// ------------------------------------------------------------------------------------------
// Eigen-system of a reversible rate matrix (eigendecomposition.cpp:167-296)
// ------------------------------------------------------------------------------------------

namespace {

// cyclic Jacobi: a (n*n, symmetric, destroyed) -> eigenvalues d, eigenvectors in the columns of v
void jacobiEigenSym(std::vector<double> &a, int n, std::vector<double> &d, std::vector<double> &v) {
    v.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) v[(size_t)i * n + i] = 1.0;
    double total = 0.0;
    for (int i = 0; i < n * n; i++) total += a[i] * a[i];
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += a[(size_t)p * n + q] * a[(size_t)p * n + q];
        if (off <= 1e-34 * total || off == 0.0) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {  // columns p, q
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq;
                    a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {  // rows p, q
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk;
                    a[(size_t)q * n + k] = s * apk + c * aqk;
                }
                a[(size_t)p * n + q] = a[(size_t)q * n + p] = 0.0;
                for (int k = 0; k < n; k++) {
                    const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                    v[(size_t)k * n + p] = c * vkp - s * vkq;
                    v[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    d.resize(n);
    for (int i = 0; i < n; i++) d[i] = a[(size_t)i * n + i];
}

}  // namespace

void decomposeRateMatrix(const double *rate_matrix, const double *state_freq, int n, EigenSystem &out,
                         bool ignore_state_freq) {
    const double ZERO = 0.000001;  // eigendecomposition.cpp (states with pi <= ZERO are dropped)
    std::vector<double> forg(state_freq, state_freq + n);
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum += forg[i];
    for (int i = 0; i < n; i++) forg[i] *= 1.0 / sum;  // :191-197
    // computeRateMatrix (:306-346): q_ij = pi_j r_ij, diagonal = -row sum, one substitution per unit time
    std::vector<double> q((size_t)n * n), m(n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            q[(size_t)i * n + j] = (i == j) ? 0.0 : (ignore_state_freq ? rate_matrix[(size_t)i * n + j] : forg[j] * rate_matrix[(size_t)i * n + j]);
    double exp_rate = 0.0;
    for (int i = 0; i < n; i++) {
        double t = 0.0;
        for (int j = 0; j < n; j++) t += q[(size_t)i * n + j];
        m[i] = t;
        exp_rate += t * forg[i];
    }
    const double delta = 1.0 / exp_rate;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) q[(size_t)i * n + j] = (i != j) ? delta * q[(size_t)i * n + j] : delta * (-m[i]);
    // eliminateZero (:348-371)
    std::vector<int> keep;
    for (int i = 0; i < n; i++)
        if (forg[i] > ZERO) keep.push_back(i);
    const int nn = (int)keep.size();
    std::vector<double> s((size_t)nn * nn), fs(nn);
    for (int i = 0; i < nn; i++) fs[i] = sqrt(forg[keep[i]]);
    // symmetrizeRateMatrix (:373-394): S_ji = q_ji sqrt(pi_j)/sqrt(pi_i) for j < i, mirrored
    for (int i = 0; i < nn; i++) {
        s[(size_t)i * nn + i] = q[(size_t)keep[i] * n + keep[i]];
        const double tmp = 1.0 / fs[i];
        for (int j = 0; j < i; j++) {
            const double x = q[(size_t)keep[j] * n + keep[i]] * (fs[j] * tmp);
            s[(size_t)j * nn + i] = s[(size_t)i * nn + j] = x;
        }
    }
    std::vector<double> d, v;
    jacobiEigenSym(s, nn, d, v);
    out.n = n;
    out.eval.assign(n, 0.0);
    out.evec.assign((size_t)n * n, 0.0);
    out.inv_evec.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) out.evec[(size_t)i * n + i] = out.inv_evec[(size_t)i * n + i] = 1.0;  // dropped states: identity (:232-246)
    for (int a = 0; a < nn; a++) {
        const int i = keep[a];
        out.eval[i] = d[a];
        for (int b = 0; b < nn; b++) {
            const int j = keep[b];
            out.evec[(size_t)i * n + j] = v[(size_t)a * nn + b] / fs[a];       // U[x][k]   = V[x][k] / sqrt(pi_x)
            out.inv_evec[(size_t)i * n + j] = v[(size_t)b * nn + a] * fs[b];   // U^-1[k][x] = V[x][k] * sqrt(pi_x)
        }
    }
    // eigenvalue equation check (:250-262)
    double error = 0.0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            if (forg[i] <= ZERO || forg[j] <= ZERO) continue;
            double zero = 0.0;
            for (int k = 0; k < n; k++) zero += q[(size_t)i * n + k] * out.evec[(size_t)k * n + j];
            zero -= out.eval[j] * out.evec[(size_t)i * n + j];
            if (fabs(zero) > error) error = fabs(zero);
        }
    if (!(error < 1e-4)) throw std::runtime_error("Eigensystem doesn't satisfy eigenvalue equation");
}

// ------------------------------------------------------------------------------------------
// genetic codes: NCBI tables in the reference's A,C,G,T codon numbering (alignment.cpp:32-49)
// ------------------------------------------------------------------------------------------

const char *geneticCode(int table) {
    // NCBI publishes the tables in T,C,A,G order; re-indexed here to 16a+4b+c with A,C,G,T = 0..3
    static std::string cache[32];
    const char *tcag = nullptr;
    switch (table) {
        case 1: case 11: tcag = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
        case 2: tcag = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG"; break;
        case 3: tcag = "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
        case 4: tcag = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
        case 5: tcag = "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG"; break;
        case 6: tcag = "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; break;
        default: return nullptr;
    }
    if (cache[table].empty()) {
        const int tcag_of_acgt[4] = {2, 1, 3, 0};  // A,C,G,T -> position in T,C,A,G
        std::string code(64, '?');
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++)
                for (int c = 0; c < 4; c++)
                    code[16 * a + 4 * b + c] = tcag[16 * tcag_of_acgt[a] + 4 * tcag_of_acgt[b] + tcag_of_acgt[c]];
        cache[table] = code;
    }
    return cache[table].c_str();
}

// ------------------------------------------------------------------------------------------
// -m string
// ------------------------------------------------------------------------------------------

namespace {

std::vector<double> parseBraces(const std::string &tok, size_t open) {
    std::vector<double> v;
    const size_t close = tok.find('}', open);
    if (close == std::string::npos) throw std::runtime_error("Missing } in model string " + tok);
    std::string body = tok.substr(open + 1, close - open - 1);
    for (char &c : body)
        if (c == ',' || c == '/') c = ' ';
    std::istringstream in(body);
    double x;
    while (in >> x) v.push_back(x);
    return v;
}

std::string upper(std::string s) {
    for (char &c : s) c = (char)toupper((unsigned char)c);
    return s;
}

}  // namespace

ModelSpec parseModelString(const std::string &s) {
    ModelSpec spec;
    std::vector<std::string> toks;
    size_t start = 0;
    int depth = 0;
    for (size_t i = 0; i <= s.size(); i++) {
        if (i < s.size() && s[i] == '{') depth++;
        if (i < s.size() && s[i] == '}') depth--;
        if (i == s.size() || (s[i] == '+' && depth == 0)) {
            toks.push_back(s.substr(start, i - start));
            start = i + 1;
        }
    }
    if (toks.empty() || toks[0].empty()) throw std::runtime_error("empty model string");
    {
        const size_t open = toks[0].find('{');
        spec.name = toks[0].substr(0, open);
        if (open != std::string::npos) spec.params = parseBraces(toks[0], open);
    }
    for (size_t t = 1; t < toks.size(); t++) {
        const std::string &tok = toks[t];
        const std::string u = upper(tok);
        const size_t open = tok.find('{');
        if (u == "ASC") spec.ascertainment = true;
        else if (u == "F1X4") spec.freq_type = ModelSpec::FREQ_CODON_1x4;
        else if (u == "F3X4") spec.freq_type = ModelSpec::FREQ_CODON_3x4;
        else if (u == "FQ") spec.freq_type = ModelSpec::FREQ_EQUAL;
        else if (u == "F" || u == "FO") spec.freq_type = ModelSpec::FREQ_EMPIRICAL;
        else if (u[0] == 'F' && open == 1) {
            spec.freq_type = ModelSpec::FREQ_USER;
            spec.user_freq = parseBraces(tok, open);
        } else if (u[0] == 'I' && (u.size() == 1 || open == 1)) {
            if (open == std::string::npos) throw std::runtime_error("+I needs a value: +I{p}");
            std::vector<double> v = parseBraces(tok, open);
            if (v.size() != 1 || v[0] < 0.0 || v[0] >= 1.0) throw std::runtime_error("Wrong proportion of invariable sites");
            spec.p_invar = v[0];
        } else if (u[0] == 'G') {
            size_t i = 1;
            spec.gamma_median = false;
            if (i < u.size() && u[i] == 'M') { spec.gamma_median = true; i++; }
            int ncat = 0;
            while (i < u.size() && isdigit((unsigned char)u[i])) ncat = 10 * ncat + (u[i++] - '0');
            spec.ncat = ncat ? ncat : 4;
            if (open == std::string::npos) throw std::runtime_error("+G needs a shape: +G4{alpha}");
            std::vector<double> v = parseBraces(tok, open);
            if (v.size() != 1 || v[0] <= 0.0) throw std::runtime_error("Wrong gamma shape");
            spec.gamma_shape = v[0];
        } else
            throw std::runtime_error("Unknown model component +" + tok);
    }
    return spec;
}

namespace {

// PAML-format amino-acid matrix: lower triangle (190 numbers) then 20 frequencies (model/modelprotein.cpp readRates)
void readPamlFile(const std::string &file, std::vector<double> &rates, std::vector<double> &freq) {
    std::ifstream in(file.c_str());
    if (!in) throw std::runtime_error("Unknown model / cannot open model file " + file);
    rates.assign(400, 0.0);
    for (int i = 1; i < 20; i++)
        for (int j = 0; j < i; j++) {
            double x;
            if (!(in >> x)) throw std::runtime_error("model file " + file + ": expecting 190 exchangeabilities");
            rates[i * 20 + j] = rates[j * 20 + i] = x;
        }
    freq.resize(20);
    for (int i = 0; i < 20; i++)
        if (!(in >> freq[i])) throw std::runtime_error("model file " + file + ": expecting 20 frequencies");
}

}  // namespace

void buildModel(const ModelSpec &spec, const Alignment &aln, ModelInputs &out) {
    const int n = aln.num_states;
    const std::string name = upper(spec.name);
    std::vector<double> rates((size_t)n * n, 0.0), freq(n, 1.0 / n), file_freq;
    ModelSpec::FreqType ft = spec.freq_type;
    auto sym = [&](int i, int j, double r) { rates[(size_t)i * n + j] = rates[(size_t)j * n + i] = r; };
    auto need = [&](size_t k) {
        if (spec.params.size() != k) {
            std::ostringstream e;
            e << spec.name << " expects " << k << " rate parameter(s) in {}";
            throw std::runtime_error(e.str());
        }
    };
    if (aln.seq_type == SEQ_DNA) {
        // exchangeabilities in the reference's order A-C, A-G, A-T, C-G, C-T, G-T (model/modeldna.cpp)
        double r[6] = {1, 1, 1, 1, 1, 1};
        if (name == "JC" || name == "JC69") { need(0); if (ft == ModelSpec::FREQ_DEFAULT) ft = ModelSpec::FREQ_EQUAL; }
        else if (name == "F81") need(0);
        else if (name == "K80" || name == "K2P") { need(1); r[1] = r[4] = spec.params[0]; if (ft == ModelSpec::FREQ_DEFAULT) ft = ModelSpec::FREQ_EQUAL; }
        else if (name == "HKY" || name == "HKY85") { need(1); r[1] = r[4] = spec.params[0]; }
        else if (name == "TN" || name == "TRN" || name == "TN93") { need(2); r[1] = spec.params[0]; r[4] = spec.params[1]; }
        else if (name == "GTR") { need(5); for (int k = 0; k < 5; k++) r[k] = spec.params[k]; }
        else throw std::runtime_error("Unknown DNA model " + spec.name);
        int k = 0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) sym(i, j, r[k++]);
    } else if (aln.seq_type == SEQ_PROTEIN) {
        if (name == "POISSON") {
            need(0);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) rates[(size_t)i * n + j] = (i != j);
            if (ft == ModelSpec::FREQ_DEFAULT) ft = ModelSpec::FREQ_EQUAL;
        } else {  // a PAML-format matrix file; its frequencies are the model's default
            readPamlFile(spec.name, rates, file_freq);
        }
    } else if (aln.seq_type == SEQ_CODON) {
        if (name != "GY" && name != "GY94") throw std::runtime_error("Unknown codon model " + spec.name);
        double kappa = 1.0, omega = 1.0;
        if (spec.params.size() == 2) { kappa = spec.params[0]; omega = spec.params[1]; }
        else if (!spec.params.empty()) throw std::runtime_error("GY expects {kappa,omega}");
        // rate attributes (modelcodon.cpp:468-533) and kappa/omega scaling (:671-700)
        for (int i = 0; i < n; i++) {
            if (aln.isStopCodon(i)) continue;
            for (int j = 0; j < n; j++) {
                if (j == i || aln.isStopCodon(j)) continue;
                int ts = 0, tv = 0;
                const int a[3] = {i / 16, (i % 16) / 4, i % 4}, b[3] = {j / 16, (j % 16) / 4, j % 4};
                for (int k = 0; k < 3; k++)
                    if (a[k] != b[k]) (abs(a[k] - b[k]) == 2 ? ts : tv)++;
                if (ts + tv > 1) continue;  // multiple nucleotide changes: rate 0 (:371)
                double rr = 1.0;
                const bool syn = aln.genetic_code[i] == aln.genetic_code[j];
                if (ts == 1) rr *= kappa;
                if (!syn) rr *= omega;
                rates[(size_t)i * n + j] = rr;
            }
        }
        if (ft == ModelSpec::FREQ_DEFAULT) ft = ModelSpec::FREQ_EMPIRICAL;
    } else
        throw std::runtime_error("unsupported data type");

    double ntfreq[12];
    switch (ft) {
        case ModelSpec::FREQ_EQUAL:
            if (aln.seq_type == SEQ_CODON) {
                int nonstop = 0;
                for (int i = 0; i < n; i++) nonstop += !aln.isStopCodon(i);
                for (int i = 0; i < n; i++) freq[i] = aln.isStopCodon(i) ? 0.0001 : (1.0 - 0.0001 * (n - nonstop)) / nonstop;
            } else
                for (int i = 0; i < n; i++) freq[i] = 1.0 / n;
            break;
        case ModelSpec::FREQ_USER:
            if ((int)spec.user_freq.size() != n) throw std::runtime_error("+F{...} needs one frequency per state");
            freq = spec.user_freq;
            break;
        case ModelSpec::FREQ_CODON_1x4: aln.computeCodonFreq(false, freq.data(), ntfreq); break;
        case ModelSpec::FREQ_CODON_3x4: aln.computeCodonFreq(true, freq.data(), ntfreq); break;
        case ModelSpec::FREQ_EMPIRICAL: aln.computeStateFreq(freq.data()); break;
        case ModelSpec::FREQ_DEFAULT:
            if (!file_freq.empty()) freq = file_freq;
            else aln.computeStateFreq(freq.data());
            break;
    }
    {  // normalise (ModelGTR::init / eigensystem_sym :191-197)
        double sum = 0.0;
        for (double f : freq) sum += f;
        for (double &f : freq) f /= sum;
    }
    out.nstates = n;
    out.state_freq = freq;
    decomposeRateMatrix(rates.data(), freq.data(), n, out.eig, false);
    out.ncat = spec.ncat;
    out.p_invar = spec.p_invar;
    out.rates.assign(spec.ncat, 1.0);
    discreteGammaRates(spec.gamma_shape, spec.ncat, spec.gamma_median, spec.p_invar, out.rates.data());
    if (spec.ncat == 1 && spec.p_invar > 0.0) out.rates[0] = 1.0 / (1.0 - spec.p_invar);  // RateInvar::getRate (model/rateinvar.h)
    // category proportions (model/rategamma.h:114, rategammainvar): (1 - p_invar)/ncat
    out.props.assign(spec.ncat, (1.0 - spec.p_invar) / spec.ncat);
}

}  // namespace iqhost
