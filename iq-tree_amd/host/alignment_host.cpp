// alignment_host.cpp -- see alignment_host.h for the reference functions this follows.
#include "alignment_host.h"

#include <ctype.h>
#include <string.h>

#include <fstream>
#include <map>
#include <sstream>

#include "model_host.h"

namespace iqhost {

namespace {

const char kProteinSymbols[] = "ARNDCQEGHILKMFPSTWYVX";  // alignment.cpp:22 (X = unknown)
const int kInvalid = 255;

bool sequenceChar(char c) {  // alignment.cpp:1433, 1498
    return isalnum((unsigned char)c) || c == '-' || c == '?' || c == '.' || c == '*' || c == '~';
}

// alignment.cpp:829-853, DNA vs protein only (the kernels' data types)
SeqType detectSequenceType(const std::vector<std::string> &sequences) {
    long num_nuc = 0, num_ungap = 0, num_alpha = 0;
    for (const std::string &s : sequences)
        for (char c : s) {
            if (c != '?' && c != '-' && c != '.' && c != 'N' && c != 'X' && c != '~') num_ungap++;
            if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'U') num_nuc++;
            if (isalpha((unsigned char)c)) num_alpha++;
        }
    if (num_ungap == 0) return SEQ_OTHER;
    if ((double)num_nuc / num_ungap > 0.9) return SEQ_DNA;
    if ((double)num_alpha / num_ungap > 0.9) return SEQ_PROTEIN;
    return SEQ_OTHER;
}

// alignment.cpp:855-922 (buildStateMap)
void buildStateMap(int *map, SeqType seq_type, int state_unknown) {
    for (int i = 0; i < 256; i++) map[i] = kInvalid;
    map[(int)'?'] = map[(int)'-'] = map[(int)'~'] = map[(int)'.'] = state_unknown;
    if (seq_type == SEQ_DNA || seq_type == SEQ_CODON) {
        const int unk = (seq_type == SEQ_DNA) ? state_unknown : 18;  // nucleotide-level unknown inside a triplet
        map[(int)'?'] = map[(int)'-'] = map[(int)'~'] = map[(int)'.'] = unk;
        map[(int)'A'] = 0; map[(int)'C'] = 1; map[(int)'G'] = 2; map[(int)'T'] = 3; map[(int)'U'] = 3;
        map[(int)'R'] = 1 + 4 + 3;  map[(int)'Y'] = 2 + 8 + 3;
        map[(int)'N'] = map[(int)'X'] = map[(int)'O'] = unk;
        map[(int)'W'] = 1 + 8 + 3;  map[(int)'S'] = 2 + 4 + 3;  map[(int)'M'] = 1 + 2 + 3;  map[(int)'K'] = 4 + 8 + 3;
        map[(int)'B'] = 2 + 4 + 8 + 3;  map[(int)'H'] = 1 + 2 + 8 + 3;  map[(int)'D'] = 1 + 4 + 8 + 3;
        map[(int)'V'] = 1 + 2 + 4 + 3;
    } else if (seq_type == SEQ_PROTEIN) {
        for (int i = 0; i < 20; i++) map[(int)kProteinSymbols[i]] = i;
        map[(int)'X'] = state_unknown;
        map[(int)'B'] = 20; map[(int)'Z'] = 21; map[(int)'J'] = 22;
        map[(int)'*'] = state_unknown; map[(int)'U'] = state_unknown;
    }
}

}  // namespace

void Alignment::readFile(const std::string &filename, const std::string &sequence_type) {
    std::ifstream in(filename.c_str(), std::ios::binary);
    if (!in) throw std::runtime_error("cannot open alignment file " + filename);
    std::ostringstream ss;
    ss << in.rdbuf();
    readString(ss.str(), sequence_type);
}

void Alignment::readString(const std::string &content, const std::string &sequence_type) {
    seq_names.clear();
    std::vector<std::string> sequences;
    std::istringstream in(content);
    std::string line;
    size_t first = content.find_first_not_of(" \t\r\n");
    const bool fasta = first != std::string::npos && content[first] == '>';
    int line_num = 0;
    auto append = [&](std::string &dst, const std::string &src) {
        for (char c : src) {
            if ((unsigned char)c <= ' ') continue;
            if (!sequenceChar(c)) {
                std::ostringstream e;
                e << "Line " << line_num << ": Unrecognized character " << c;
                throw std::runtime_error(e.str());
            }
            dst.push_back((char)toupper((unsigned char)c));
        }
    };
    if (fasta) {  // alignment.cpp:1467-1555 (names cut at the first blank)
        while (std::getline(in, line)) {
            line_num++;
            line = line.substr(0, line.find_first_of("\n\r"));
            if (line.empty()) continue;
            if (line[0] == '>') {
                std::string name = line.substr(1);
                size_t b = name.find_first_not_of(" \t"), e = name.find_first_of(" \t", b == std::string::npos ? 0 : b);
                name = (b == std::string::npos) ? "" : name.substr(b, e == std::string::npos ? std::string::npos : e - b);
                seq_names.push_back(name);
                sequences.push_back("");
                continue;
            }
            if (sequences.empty()) throw std::runtime_error("First line must begin with '>' to define sequence name");
            append(sequences.back(), line);
        }
    } else {  // alignment.cpp:1382-1465 (sequential and interleaved PHYLIP)
        int nseq = 0, nsite = 0, seq_id = 0;
        while (std::getline(in, line)) {
            line_num++;
            line = line.substr(0, line.find_first_of("\n\r"));
            if (line.find_first_not_of(" \t") == std::string::npos) continue;
            if (nseq == 0) {
                std::istringstream li(line);
                if (!(li >> nseq >> nsite))
                    throw std::runtime_error("Invalid PHYLIP format. First line must contain number of sequences and sites");
                if (nseq < 3) throw std::runtime_error("There must be at least 3 sequences");
                if (nsite < 1) throw std::runtime_error("No alignment columns");
                seq_names.assign(nseq, "");
                sequences.assign(nseq, "");
                continue;
            }
            if (seq_names[seq_id].empty()) {
                size_t pos = line.find_first_of(" \t");
                if (pos == std::string::npos) pos = 10;
                seq_names[seq_id] = line.substr(0, pos);
                line.erase(0, pos);
            }
            const size_t old_len = sequences[seq_id].size();
            append(sequences[seq_id], line);
            if (sequences[seq_id].size() != sequences[0].size()) {  // every line extends the next sequence (alignment.cpp:1443)
                std::ostringstream e;
                e << "Line " << line_num << ": Sequence " << seq_names[seq_id] << " has wrong sequence length "
                  << sequences[seq_id].size();
                throw std::runtime_error(e.str());
            }
            if (sequences[seq_id].size() > old_len) seq_id++;
            if (seq_id == nseq) seq_id = 0;
        }
        if (nseq == 0) throw std::runtime_error("empty alignment");
        for (int i = 0; i < nseq; i++)
            if ((int)sequences[i].size() != nsite) {
                std::ostringstream e;
                e << "Sequence " << seq_names[i] << " contains " << ((int)sequences[i].size() < nsite ? "not enough" : "too many")
                  << " characters (" << sequences[i].size() << ")";
                throw std::runtime_error(e.str());
            }
    }
    buildPattern(sequences, sequence_type);
}

void Alignment::buildPattern(const std::vector<std::string> &sequences, const std::string &sequence_type) {
    const int nseq = (int)sequences.size();
    if (nseq != (int)seq_names.size()) throw std::runtime_error("Different number of sequences than specified");
    for (int i = 0; i < nseq; i++) {
        if (seq_names[i].empty()) throw std::runtime_error("A sequence has no name");
        for (int j = 0; j < i; j++)
            if (seq_names[i] == seq_names[j]) throw std::runtime_error("The sequence name " + seq_names[i] + " is duplicated");
        if (sequences[i].size() != sequences[0].size())
            throw std::runtime_error("Sequence " + seq_names[i] + " has a different length");
    }
    const int nsite = (int)sequences[0].size();
    seq_type = detectSequenceType(sequences);
    genetic_code.clear();
    if (sequence_type == "DNA" || sequence_type == "NT") seq_type = SEQ_DNA;
    else if (sequence_type == "AA" || sequence_type == "PROT") seq_type = SEQ_PROTEIN;
    else if (sequence_type.compare(0, 5, "CODON") == 0) {
        int table = 1;
        if (sequence_type.size() > 5) table = atoi(sequence_type.c_str() + 5);
        const char *gc = geneticCode(table);
        if (!gc) throw std::runtime_error("Unsupported genetic code " + sequence_type.substr(5));
        genetic_code = gc;
        seq_type = SEQ_CODON;
    } else if (!sequence_type.empty())
        throw std::runtime_error("Invalid sequence type " + sequence_type);
    switch (seq_type) {
        case SEQ_DNA: num_states = 4; STATE_UNKNOWN = 18; break;       // alignment.cpp:470-472
        case SEQ_PROTEIN: num_states = 20; STATE_UNKNOWN = 23; break;
        case SEQ_CODON: num_states = 64; STATE_UNKNOWN = 64; break;
        default: throw std::runtime_error("Unknown sequence type.");
    }
    int char_to_state[256];
    buildStateMap(char_to_state, seq_type, STATE_UNKNOWN);
    const int step = (seq_type == SEQ_CODON) ? 3 : 1;
    if (nsite % step != 0) throw std::runtime_error("Number of sites is not multiple of 3");
    site_pattern.assign(nsite / step, -1);
    patterns.clear();
    n_unobserved = 0;
    std::map<std::vector<uint8_t>, int> pattern_index;
    std::ostringstream err;
    int num_error = 0;
    Pattern pat;
    pat.states.resize(nseq);
    for (int site = 0; site < nsite; site += step) {
        for (int seq = 0; seq < nseq; seq++) {
            int state = char_to_state[(unsigned char)sequences[seq][site]];
            if (seq_type == SEQ_CODON) {
                const int s2 = char_to_state[(unsigned char)sequences[seq][site + 1]];
                const int s3 = char_to_state[(unsigned char)sequences[seq][site + 2]];
                if (state < 4 && s2 < 4 && s3 < 4) {
                    state = state * 16 + s2 * 4 + s3;
                    if (genetic_code[state] == '*') {
                        err << "Sequence " << seq_names[seq] << " has stop codon at site " << site + 1 << "\n";
                        num_error++;
                        state = STATE_UNKNOWN;
                    }
                } else if (state == kInvalid || s2 == kInvalid || s3 == kInvalid)
                    state = kInvalid;
                else
                    state = STATE_UNKNOWN;  // gaps or ambiguous nucleotides inside the triplet
            }
            if (state == kInvalid) {
                if (num_error < 100)
                    err << "Sequence " << seq_names[seq] << " has invalid character " << sequences[seq][site] << " at site "
                        << site + 1 << "\n";
                num_error++;
            }
            pat.states[seq] = (uint8_t)state;
        }
        if (num_error) continue;
        auto it = pattern_index.find(pat.states);  // alignment.cpp:674-700
        if (it == pattern_index.end()) {
            pat.frequency = 1;
            computeConst(pat);
            patterns.push_back(pat);
            pattern_index[pat.states] = (int)patterns.size() - 1;
            site_pattern[site / step] = (int)patterns.size() - 1;
        } else {
            patterns[it->second].frequency += 1;
            site_pattern[site / step] = it->second;
        }
    }
    if (num_error) throw std::runtime_error(err.str());
    countConstSite();
}

void Alignment::getAppearance(int state, double *state_app) const {  // alignment.cpp:2920-2953
    if (state == STATE_UNKNOWN) {
        for (int i = 0; i < num_states; i++) state_app[i] = 1.0;
        return;
    }
    for (int i = 0; i < num_states; i++) state_app[i] = 0.0;
    if (state < num_states) {
        state_app[state] = 1.0;
        return;
    }
    if (seq_type == SEQ_DNA) {
        const int mask = state - (num_states - 1);
        for (int i = 0; i < num_states; i++)
            if (mask & (1 << i)) state_app[i] = 1.0;
    } else if (seq_type == SEQ_PROTEIN) {
        static const int ambi_aa[3] = {4 + 8, 32 + 64, 512 + 1024};  // B = N|D, Z = Q|E, J = I|L
        for (int i = 0; i < 11; i++)
            if (ambi_aa[state - 20] & (1 << i)) state_app[i] = 1.0;
    } else
        throw std::runtime_error("ambiguous state in a data type without ambiguity codes");
}

void Alignment::computeConst(Pattern &pat) const {  // alignment.cpp:609-671
    pat.is_const = false;
    pat.const_char = (STATE_UNKNOWN == num_states) ? STATE_UNKNOWN + 1 : STATE_UNKNOWN;
    std::vector<char> all(num_states, 1);
    std::vector<double> app(num_states);
    for (uint8_t s : pat.states) {
        getAppearance(s, app.data());
        for (int j = 0; j < num_states; j++) all[j] = all[j] && app[j] != 0.0;
    }
    int count = 0, which = -1;
    for (int j = 0; j < num_states; j++)
        if (all[j]) {
            count++;
            if (which < 0) which = j;
        }
    if (count == 0) return;
    if (count == num_states) {  // all-gap pattern
        pat.is_const = true;
        pat.const_char = num_states;
    } else if (count == 1) {
        pat.is_const = true;
        pat.const_char = which;
    }
}

void Alignment::countConstSite() {  // alignment.cpp:2501-2511
    long num_const = 0;
    for (const Pattern &p : patterns)
        if (p.is_const) num_const += p.frequency;
    frac_const_sites = getNSite() ? (double)num_const / getNSite() : 0.0;
}

int Alignment::appendUnobservedConstPatterns() {  // alignment.cpp:2513-2530, modelfactory.cpp:359-370
    if (n_unobserved) throw std::runtime_error("unobserved constant patterns were already appended");
    const int nseq = getNSeq();
    std::vector<int> missing;
    for (int state = 0; state < num_states; state++) {
        if (isStopCodon(state)) continue;
        bool seen = false;
        for (const Pattern &p : patterns) {
            bool same = true;
            for (int s = 0; s < nseq && same; s++) same = p.states[s] == state;
            if (same) { seen = true; break; }
        }
        if (!seen) missing.push_back(state);
    }
    int nonstop = 0;
    for (int state = 0; state < num_states; state++) nonstop += !isStopCodon(state);
    if ((int)missing.size() < nonstop)
        throw std::runtime_error("Invalid use of +ASC because constant patterns are observed in the alignment");
    for (int state : missing) {
        Pattern p;
        p.states.assign(nseq, (uint8_t)state);
        p.frequency = 0;
        p.is_const = true;
        p.const_char = state;
        patterns.push_back(p);
    }
    n_unobserved = (int)missing.size();
    return n_unobserved;
}

void Alignment::statesByLeaf(std::vector<uint8_t> &out) const {
    const size_t nptn = patterns.size();
    const int nseq = getNSeq();
    out.resize((size_t)nseq * nptn);
    for (size_t p = 0; p < nptn; p++)
        for (int s = 0; s < nseq; s++) out[(size_t)s * nptn + p] = patterns[p].states[s];
}

void Alignment::ptnFreq(std::vector<double> &out) const {  // phylotreesse.cpp:531-541
    out.resize(patterns.size());
    for (size_t p = 0; p < patterns.size(); p++) out[p] = patterns[p].frequency;
}

void Alignment::ptnInvar(double p_invar, const double *state_freq, std::vector<double> &out) const {  // :543-569
    out.assign(patterns.size(), 0.0);
    if (p_invar == 0.0) return;
    for (size_t p = 0; p < patterns.size(); p++) {
        const Pattern &pat = patterns[p];
        if (pat.const_char == num_states) out[p] = p_invar;
        else if (pat.const_char < num_states) out[p] = p_invar * state_freq[pat.const_char];
    }
}

void Alignment::computeStateFreq(double *state_freq) const {  // alignment.cpp:2714-2785
    const int nstate_codes = STATE_UNKNOWN + 1;
    std::vector<double> app((size_t)num_states * nstate_codes), new_freq(num_states), acc(num_states);
    std::vector<double> state_count(nstate_codes, 0.0);
    for (int i = 0; i < nstate_codes; i++) {
        if (i >= num_states && i < STATE_UNKNOWN && seq_type == SEQ_CODON) continue;
        if (seq_type == SEQ_PROTEIN && i > 22 && i < STATE_UNKNOWN) continue;
        getAppearance(i, &app[(size_t)i * num_states]);
    }
    for (const Pattern &p : patterns)
        for (uint8_t s : p.states) state_count[s] += p.frequency;
    for (int j = 0; j < num_states; j++) state_freq[j] = 1.0 / num_states;
    for (int k = 0; k < 8; k++) {  // NUM_TIME
        for (int j = 0; j < num_states; j++) acc[j] = 0.0;
        for (int i = 0; i < nstate_codes; i++) {
            if (state_count[i] == 0.0) continue;
            double sum = 0.0;
            for (int j = 0; j < num_states; j++) {
                new_freq[j] = state_freq[j] * app[(size_t)i * num_states + j];
                sum += new_freq[j];
            }
            sum = 1.0 / sum;
            for (int j = 0; j < num_states; j++) acc[j] += new_freq[j] * sum * state_count[i];
        }
        double sum = 0.0;
        for (int j = 0; j < num_states; j++) sum += acc[j];
        sum = 1.0 / sum;
        for (int j = 0; j < num_states; j++) state_freq[j] = acc[j] * sum;
    }
    // convfreq (alignment.cpp:3219-3241): floor at MIN_FREQUENCY, the surplus leaves the largest
    const double MIN_FREQUENCY = 0.0001;
    double sum = 0.0, maxfreq = 0.0;
    int maxi = 0;
    for (int i = 0; i < num_states; i++) {
        const double f = state_freq[i];
        if (f < MIN_FREQUENCY) state_freq[i] = MIN_FREQUENCY;
        if (f > maxfreq) { maxfreq = f; maxi = i; }
        sum += state_freq[i];
    }
    state_freq[maxi] += 1.0 - sum;
}

void Alignment::computeCodonFreq(bool f3x4, double *state_freq, double *ntfreq) const {  // alignment.cpp:2990-3080
    if (seq_type != SEQ_CODON) throw std::runtime_error("codon frequencies need codon data");
    for (int i = 0; i < 12; i++) ntfreq[i] = 0.0;
    for (const Pattern &p : patterns)
        for (uint8_t s : p.states)
            if (s != STATE_UNKNOWN) {
                const int nt1 = s / 16, nt2 = (s % 16) / 4, nt3 = s % 4;
                if (f3x4) {
                    ntfreq[nt1] += p.frequency; ntfreq[4 + nt2] += p.frequency; ntfreq[8 + nt3] += p.frequency;
                } else {
                    ntfreq[nt1] += p.frequency; ntfreq[nt2] += p.frequency; ntfreq[nt3] += p.frequency;
                }
            }
    for (int j = 0; j < (f3x4 ? 12 : 4); j += 4) {
        double sum = 0.0;
        for (int i = 0; i < 4; i++) sum += ntfreq[i + j];
        for (int i = 0; i < 4; i++) ntfreq[i + j] /= sum;
    }
    if (!f3x4) {
        memcpy(ntfreq + 4, ntfreq, 4 * sizeof(double));
        memcpy(ntfreq + 8, ntfreq, 4 * sizeof(double));
    }
    const double MIN_FREQUENCY = 0.0001;
    double sum_stop = 0.0, sum = 0.0;
    for (int i = 0; i < num_states; i++) {
        state_freq[i] = ntfreq[i / 16] * ntfreq[4 + (i % 16) / 4] * ntfreq[8 + i % 4];
        if (isStopCodon(i)) {
            sum_stop += state_freq[i];
            state_freq[i] = MIN_FREQUENCY;
            sum += MIN_FREQUENCY;
        }
    }
    sum = (1.0 - sum) / (1.0 - sum_stop);
    for (int i = 0; i < num_states; i++)
        if (!isStopCodon(i)) state_freq[i] *= sum;
}

void writeSiteLh(const std::string &filename, const Alignment &aln, const double *pattern_lh, const char *linename) {
    std::ofstream out(filename.c_str());
    if (!out) throw std::runtime_error("cannot write " + filename);
    out << 1 << " " << aln.getNSite() << "\n";
    if (!linename) out << "Site_Lh   ";
    else {
        out.width(10);
        out << std::left << linename;
    }
    for (int i = 0; i < aln.getNSite(); i++) out << " " << pattern_lh[aln.site_pattern[i]];
    out << "\n";
}

}  // namespace iqhost
