// Sanitizer + fuzz driver for the native FASTA loader (crp_fasta.cpp), built with a tiny
// piece size (-DCRP_FASTA_CHUNK_BYTES=...) so that records, headers and lines straddle many
// piece borders.  Every input goes through crp_fasta_table and through the serial
// restatement below (cut at '>', cut at the first newline, drop newlines, decorate).
// Built and run by tests/test_sanitizers.py with -fsanitize=address,undefined and =thread.
#include "cropsr_hip.h"

#include <cstdio>
#include <random>
#include <string>
#include <vector>

namespace {

bool plain_char(unsigned char c) { return c >= 33 && c <= 126 && c != '\'' && c != '\\'; }

// returns false when the input is outside the fast path
bool serial_table(const std::string &d, std::vector<std::string> &heads, std::vector<std::string> &values)
{
    size_t gt = 0, nl = 0;
    for (char c : d) { gt += c == '>'; nl += c == '\n'; }
    if (2 * gt == nl + 1) return false;
    std::vector<std::string> pieces;
    std::string cur;
    for (char c : d) {
        if (c == '>') { if (!cur.empty()) pieces.push_back(cur); cur.clear(); }
        else cur += c;
    }
    if (!cur.empty()) pieces.push_back(cur);
    if (pieces.empty()) return false;
    for (size_t k = 0; k < pieces.size(); ++k) {
        const std::string &p = pieces[k];
        const size_t e = p.find('\n');
        if (e == std::string::npos) return false;
        std::string head = p.substr(0, e), body;
        for (size_t i = e + 1; i < p.size(); ++i)
            if (p[i] != '\n') body += p[i];
        for (unsigned char c : head) if (!plain_char(c)) return false;
        for (unsigned char c : body) if (!plain_char(c)) return false;
        heads.push_back(head);
        values.push_back("'" + body + (k + 1 == pieces.size() ? "')]" : "'),"));
    }
    return true;
}

}  // namespace

int main()
{
    std::mt19937_64 rng(99);
    int failures = 0, fast = 0;
    for (int round = 0; round < 3000 && failures < 5; ++round) {
        std::string d;
        const int kind = round % 4;
        if (kind == 0) {  // raw soup
            const char soup[] = "ACGTN>\n\n\nacgt x'";
            const size_t n = rng() % 400;
            for (size_t k = 0; k < n; ++k) d += soup[rng() % (sizeof soup - 1)];
        } else {  // well-formed records, random line widths, sometimes blank lines / odd endings
            const int n_rec = 1 + (int)(rng() % 9);
            if (kind == 3 && rng() % 2) d += "\n";
            for (int r = 0; r < n_rec; ++r) {
                d += ">c" + std::to_string(rng() % 1000) + (kind == 3 && rng() % 7 == 0 ? " desc" : "") + "\n";
                const size_t len = rng() % 700, width = 1 + rng() % 90;
                for (size_t k = 0; k < len; ++k) {
                    d += "ACGTNacgt"[rng() % 9];
                    if ((k + 1) % width == 0) d += '\n';
                }
                if (rng() % 4) d += '\n';
                if (kind == 3 && rng() % 9 == 0) d += '\n';
            }
            if (kind == 2 && rng() % 3 == 0) d += ">";
        }
        std::vector<std::string> heads, values;
        const bool want_plain = serial_table(d, heads, values);
        const int threads = 1 + (int)(rng() % 5);
        size_t n_gt = 1;
        for (char c : d) n_gt += c == '>';
        std::vector<uint8_t> out(d.size() + 4 * n_gt + 8);  // the bound the header promises
        std::vector<uint64_t> recs(4 * n_gt);
        uint64_t n_recs = 0, out_len = 0;
        int plain = -1;
        // first with too little room: the sizes must come back
        int st = crp_fasta_table((const uint8_t *)d.data(), d.size(), nullptr, 0, nullptr, 0, &n_recs, &out_len, &plain, threads);
        if (want_plain && (st != CRP_ERR_CAPACITY || n_recs != heads.size())) { std::printf("round %d: size probe %d\n", round, st); ++failures; continue; }
        st = crp_fasta_table((const uint8_t *)d.data(), d.size(), out.data(), out.size(), recs.data(), n_gt,
                             &n_recs, &out_len, &plain, threads);
        if (st != CRP_OK || (plain != 0) != want_plain) { std::printf("round %d: status %d plain %d want %d\n", round, st, plain, (int)want_plain); ++failures; continue; }
        if (!want_plain) continue;
        ++fast;
        bool same = n_recs == heads.size();
        uint64_t at = 0;
        for (size_t k = 0; same && k < heads.size(); ++k) {
            same = recs[4 * k + 1] == heads[k].size() && d.compare(recs[4 * k], heads[k].size(), heads[k]) == 0 &&
                   recs[4 * k + 2] == at && recs[4 * k + 3] == values[k].size() &&
                   std::string((const char *)out.data() + at, values[k].size()) == values[k];
            at += values[k].size();
        }
        if (!same || at != out_len) { std::printf("round %d: table differs\n", round); ++failures; }
    }
    std::printf("%d inputs on the fast path\n", fast);
    std::printf(failures || fast < 500 ? "FAILED\n" : "OK\n");
    return failures || fast < 500 ? 1 : 0;
}
