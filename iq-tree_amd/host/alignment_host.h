// alignment_host.h -- on-disk formats on the input side of the likelihood path:
// PHYLIP / FASTA readers and site -> pattern compression, producing exactly the arrays the kernels
// consume (leaf state bytes per pattern, ptn_freq, ptn_invar) plus the site -> pattern index the
// `.sitelh` writer needs.  Reference behaviour followed (file:line = /root/reference):
//   Alignment::readPhylip / readFasta          alignment.cpp:1382-1465, 1467-1555
//   Alignment::buildPattern (state encoding, codon triplets, first-appearance pattern order)
//                                              alignment.cpp:1204-1378, convertState :924-1006
//   Alignment::addPattern / computeConst       alignment.cpp:674-700, 609-671
//   state numbering: DNA 0..3, ambiguity codes 4..17 (bit mask = code-3), 18 unknown;
//   protein 0..19, B/Z/J = 20..22, 23 unknown; codon 0..63 (16a+4b+c), 64 unknown
//                                              alignment.cpp:470-472
//   Alignment::computeStateFreq / convfreq     alignment.cpp:2714-2785, 3219-3255
//   Alignment::computeCodonFreq                alignment.cpp:2990-3080
//   Alignment::getUnobservedConstPatterns (+ASC)   model/modelfactory.cpp:359-370
// Host-side C++; not on the hot path.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "phylo_host.h"

namespace iqhost {

struct Pattern {
    std::vector<uint8_t> states;  // one state per sequence
    int frequency = 0;
    bool is_const = false;
    int const_char = 0;  // state shared by all sequences (num_states for an all-gap pattern)
};

class Alignment {
public:
    // format by content: '>' starts FASTA, anything else is PHYLIP (sequential or interleaved).
    // sequence_type: "" (detect DNA vs protein), "DNA", "AA", "CODON" or "CODON<ncbi table>"
    void readFile(const std::string &filename, const std::string &sequence_type = "");
    void readString(const std::string &content, const std::string &sequence_type = "");
    void buildPattern(const std::vector<std::string> &sequences, const std::string &sequence_type);

    std::vector<std::string> seq_names;
    std::vector<Pattern> patterns;       // first-appearance order, as the reference
    std::vector<int> site_pattern;       // site (codon site for CODON) -> pattern index
    SeqType seq_type = SEQ_DNA;
    int num_states = 0, STATE_UNKNOWN = 0;
    std::string genetic_code;            // 64 letters, '*' = stop (CODON only)
    double frac_const_sites = 0.0;

    int getNSeq() const { return (int)seq_names.size(); }
    int getNSite() const { return (int)site_pattern.size(); }
    int getNPattern() const { return (int)patterns.size(); }
    bool isStopCodon(int state) const {
        return seq_type == SEQ_CODON && state < 64 && genetic_code[state] == '*';
    }
    // which of the num_states states a (possibly ambiguous) state stands for
    void getAppearance(int state, double *state_app) const;

    // kernel inputs: states[leaf][ptn] (leaf = sequence index), ptn_freq, ptn_invar
    void statesByLeaf(std::vector<uint8_t> &out) const;
    void ptnFreq(std::vector<double> &out) const;
    void ptnInvar(double p_invar, const double *state_freq, std::vector<double> &out) const;
    // +ASC: appends one constant pattern per (non-stop) state that is not observed in the
    // alignment, frequency 0; returns how many were appended.  Call once, before the getters.
    int appendUnobservedConstPatterns();
    int n_unobserved = 0;

    void computeStateFreq(double *state_freq) const;                              // empirical, ambiguity-aware
    void computeCodonFreq(bool f3x4, double *state_freq, double *ntfreq) const;   // F1X4 / F3X4

private:
    void computeConst(Pattern &pat) const;
    void countConstSite();
};

// `.sitelh` (phylotesting.cpp:202-241): "1 nsite" header, then one line of per-site lnL
void writeSiteLh(const std::string &filename, const Alignment &aln, const double *pattern_lh,
                 const char *linename = nullptr);

}  // namespace iqhost
