// Sanitizer + fuzz driver for the native annotation builder (crp_annotation.cpp): random GFF soups and well-formed
// GFF3 files with overlapping / nested / degenerate gene and CDS rows (+ an annotation_info file) go through
// crp_annotation_build; the label set of EVERY coordinate, read off the elementary intervals, must equal a direct
// loop over the parsed rows; crp_annotation_track is driven with whole texts, pieces, unknown seqids and bad orders.
// Built and run by tests/test_sanitizers.py with -fsanitize=address,undefined.
#include "cropsr_hip.h"

#include <algorithm>
#include <cstdio>
#include <map>
#include <random>
#include <string>
#include <vector>

namespace {

struct Row {
    std::string seq, label;
    long a, b;
};

std::string strip(const std::string &s)
{
    size_t a = 0, b = s.size();
    auto sp = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (a < b && sp(s[a])) ++a;
    while (b > a && sp(s[b - 1])) --b;
    return s.substr(a, b - a);
}

bool digits(const std::string &s) { return !s.empty() && s.size() <= 18 && std::all_of(s.begin(), s.end(), [](char c) { return c >= '0' && c <= '9'; }); }

// the definition, line by line (the same statement as oracle/annotate_oracle.py)
std::vector<Row> parse(const std::string &gff, const std::map<std::string, std::pair<std::string, std::string>> &info, bool have_info)
{
    std::vector<Row> rows;
    size_t p = 0;
    while (p < gff.size()) {
        size_t q = gff.find('\n', p);
        if (q == std::string::npos) q = gff.size();
        const std::string line = gff.substr(p, q - p);
        p = q + 1;
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> c;
        size_t a = 0;
        for (;;) {
            const size_t b = line.find('\t', a);
            c.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        if (c.size() < 9 || (c[2] != "gene" && c[2] != "CDS") || !digits(c[3]) || !digits(c[4])) continue;
        std::map<std::string, std::string> d;
        a = 0;
        for (;;) {
            const size_t b = c[8].find(';', a);
            const std::string part = strip(c[8].substr(a, b == std::string::npos ? std::string::npos : b - a));
            const size_t eq = part.find('=');
            const std::string k = part.substr(0, eq), v = eq == std::string::npos ? "" : part.substr(eq + 1);
            if ((k == "ID" || k == "Name" || k == "Parent") && !d.count(k)) d[k] = v;
            if (b == std::string::npos) break;
            a = b + 1;
        }
        std::string ident = !d["ID"].empty() ? d["ID"] : !d["Name"].empty() ? d["Name"] : !d["Parent"].empty() ? d["Parent"] : ".";
        std::string label = c[2] + ":" + ident;
        if (have_info && c[2] == "gene") {
            auto hit = info.find(d["Name"]);
            if (hit == info.end()) hit = info.find(d["ID"]);
            if (hit != info.end()) {
                if (!hit->second.first.empty()) label += "|" + hit->second.first;
                if (!hit->second.second.empty()) label += "|" + hit->second.second;
            }
        }
        rows.push_back(Row{c[0], label, std::stol(c[3]), std::stol(c[4])});
    }
    return rows;
}

std::string brute(const std::vector<Row> &rows, const std::string &seq, long x)
{
    std::vector<std::string> labels;
    for (const Row &r : rows)
        if (r.seq == seq && r.a <= x && x <= r.b && std::find(labels.begin(), labels.end(), r.label) == labels.end()) labels.push_back(r.label);
    std::string out;
    for (size_t k = 0; k < labels.size(); ++k) out += (k ? ";" : "") + labels[k];
    return out;
}

}  // namespace

int main()
{
    std::mt19937_64 rng(2024);
    int failures = 0;
    long checked = 0;
    for (int round = 0; round < 400 && failures < 5; ++round) {
        std::string gff, info_text;
        std::map<std::string, std::pair<std::string, std::string>> info;
        const bool have_info = round % 3 == 0;
        if (round % 5 == 4) {  // soup: whatever bytes, the builder must survive and agree with the line-by-line definition
            // (\x1c-\x1f and \v \f: bytes Python's str.strip() drops and the definition's ASCII strip keeps -- ADVICE r04)
            const char soup[] = "s\t\t\tgene\tCDS\t1\t20\t300\t;ID=Name=Parent=x y\n\n#\r.;=\x1c\x1f\v\f";
            const size_t n = rng() % 600;
            for (size_t k = 0; k < n; ++k) gff += soup[rng() % (sizeof soup - 1)];
        } else {
            gff = "##gff-version 3\n";
            const int n_rows = (int)(rng() % 60);
            for (int k = 0; k < n_rows; ++k) {
                const long a = 1 + (long)(rng() % 500), len[] = {0, 1, 7, 60, 400};
                const long b = a + len[rng() % 5] - (rng() % 17 == 0 ? 5 : 0);
                const char *type[] = {"gene", "CDS", "CDS", "mRNA", "exon"};
                const char *seq[] = {"s1", "s1", "s2", "chr with blank"};
                std::string attrs;
                switch (rng() % 6) {
                    case 0: attrs = "ID=f" + std::to_string(rng() % 12); break;
                    case 1: attrs = "ID=g" + std::to_string(k) + ";Name=L" + std::to_string(rng() % 6); break;
                    case 2: attrs = "Parent=p" + std::to_string(rng() % 4); break;
                    case 3: attrs = " Name=L" + std::to_string(rng() % 6) + " ; ID= ;ID=second"; break;
                    case 4: attrs = "note=none"; break;
                    default: attrs = "ID=a,b \"q\";Parent=x";
                }
                gff += std::string(seq[rng() % 4]) + "\tsrc\t" + type[rng() % 5] + "\t" + std::to_string(a) + "\t" + (b >= 0 ? std::to_string(b) : "-1") +
                       "\t.\t+\t.\t" + attrs + (rng() % 9 == 0 ? "\textra" : "") + (rng() % 11 == 0 ? "\r\n" : "\n");
            }
            if (have_info)
                for (int k = 0; k < 6; k += 1 + (int)(rng() % 2)) {
                    const std::string locus = "L" + std::to_string(k), best = rng() % 3 ? "AT" + std::to_string(k) : "", def = rng() % 2 ? "defline " + std::to_string(k) : "";
                    info_text += "1\t" + locus + "\tt\tp\t\t\t\t\t\t\t" + best + "\tsym\t" + def + "\n";
                    info.emplace(locus, std::make_pair(best, def));
                }
        }
        crp_annotation *an = nullptr;
        const uint8_t dummy = 0;
        const int st = crp_annotation_build(gff.empty() ? &dummy : reinterpret_cast<const uint8_t *>(gff.data()), gff.size(),
                                            have_info ? (info_text.empty() ? &dummy : reinterpret_cast<const uint8_t *>(info_text.data())) : nullptr,
                                            info_text.size(), &an);
        if (st != CRP_OK) { printf("round %d: build status %d\n", round, st); ++failures; continue; }
        uint64_t n_seq = 0, n_str = 0, n_blob = 0;
        crp_annotation_stats(an, &n_seq, &n_str, &n_blob, nullptr, nullptr);
        std::vector<uint8_t> blob(n_blob + 1);
        std::vector<uint64_t> off(n_str + 1);
        crp_annotation_strings(an, blob.data(), off.data());
        const std::vector<Row> rows = parse(gff, info, have_info);
        for (uint64_t k = 0; k < n_seq; ++k) {
            const uint8_t *name;
            const int64_t *points;
            const uint32_t *ids;
            uint64_t n_name, n;
            crp_annotation_seqid(an, k, &name, &n_name, &points, &ids, &n);
            const std::string seq(reinterpret_cast<const char *>(name), n_name);
            for (long x = -2; x < 960; ++x) {
                const int64_t *it = std::upper_bound(points, points + n, (int64_t)x);
                std::string got;
                if (it != points && ids[it - points - 1] != CRP_NO_FEATURE) {
                    const uint32_t id = ids[it - points - 1];
                    got.assign(reinterpret_cast<const char *>(blob.data()) + off[id], off[id + 1] - off[id]);
                }
                ++checked;
                if (got != brute(rows, seq, x)) {
                    printf("round %d seq %s x %ld: got '%s' want '%s'\n", round, seq.c_str(), x, got.c_str(), brute(rows, seq, x).c_str());
                    ++failures;
                    break;
                }
            }
            // tracks: the whole text, a piece with an offset, dec 0 / 1; every arena position of the text must name the set of its coordinate
            for (int dec = 0; dec < 2; ++dec) {
                const uint64_t lo = rng() % 300, len = 1 + rng() % 700;
                const uint64_t entries[12] = {k, 0, 900, 64, n_seq + 5, 0, 10, 1024, k, lo, len, 2048};
                uint64_t need = 0;
                crp_annotation_track(an, entries, 3, dec, nullptr, nullptr, 0, &need);
                std::vector<uint32_t> tp(need), ti(need);
                if (crp_annotation_track(an, entries, 3, dec, tp.data(), ti.data(), need, &need) != CRP_OK) { printf("round %d: track failed\n", round); ++failures; break; }
                for (uint64_t q = 1; q < need; ++q) if (tp[q] <= tp[q - 1]) { printf("round %d: track not ascending\n", round); ++failures; break; }
                for (int e = 0; e < 3; e += 2)
                    for (uint64_t pos = entries[4 * e + 3]; pos < entries[4 * e + 3] + entries[4 * e + 2]; pos += 1 + rng() % 5) {
                        const auto it = std::upper_bound(tp.begin(), tp.end(), (uint32_t)pos);
                        const uint32_t id = it == tp.begin() ? CRP_NO_FEATURE : ti[it - tp.begin() - 1];
                        std::string got;
                        if (id != CRP_NO_FEATURE) got.assign(reinterpret_cast<const char *>(blob.data()) + off[id], off[id + 1] - off[id]);
                        const long x = (long)(pos - entries[4 * e + 3]) + (long)entries[4 * e + 1] - dec + 1;
                        if (got != brute(rows, seq, x)) { printf("round %d: track pos %llu\n", round, (unsigned long long)pos); ++failures; break; }
                    }
                const uint64_t bad[8] = {k, 0, 100, 640, k, 0, 100, 64};
                if (crp_annotation_track(an, bad, 2, dec, nullptr, nullptr, 0, &need) != CRP_ERR_INVALID) { printf("round %d: bad order accepted\n", round); ++failures; }
            }
        }
        crp_annotation_destroy(an);
    }
    printf("%ld coordinates checked, %d failures\n%s\n", checked, failures, failures ? "FAILED" : "OK");
    return failures ? 1 : 0;
}
