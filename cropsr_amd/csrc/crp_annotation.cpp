// crp_annotation.cpp -- host side of the opt-in annotation join: GFF3 (+ Phytozome
// annotation_info) bytes -> label-set strings and, per seqid, the elementary-interval track the device look-up
// (crp_annotate.hip) works on.  No GPU needed.
//
// The reference reads the GFF into a DataFrame (CROPSR.py:77-95, called at :375), never uses it, and writes '' into
// `features` (:466, :468); `-p` is only echoed (:364).  The join is therefore this engine's own definition
// (cropsr_amd/annotate.py; oracle/annotate_oracle.py restates it as a loop over every GFF line per CSV row):
//
//   rows    lines that do not start with '#', with >= 9 tab-separated fields, type (field 3) `gene` or `CDS`, start /
//           end (fields 4, 5) made of digits only; seqid = field 1; in file order
//   label   "<type>:<ident>", ident = the ID attribute, else Name, else Parent, else "." (attributes = field 9 split at
//           ';', each part stripped, key = text before the first '=', first occurrence of a key wins); for a gene
//           whose Name (else ID) is a locusName of the annotation_info file: + "|" + Best-hit-arabi-name and
//           + "|" + arabi-defline (empty fields left out)
//   set     for a 1-based coordinate x of a seqid: the labels of the rows with start <= x <= end, file order, each
//           label once, joined with ';'
//
// The interval ends cut a seqid's axis into elementary intervals with a constant set: one sweep over the sorted
// points builds one string per DISTINCT set (interned over the whole file) and the id of every interval.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "cropsr_hip.h"

struct crp_annotation {
    struct Seq {
        std::string name;
        std::vector<int64_t> points;  // ascending 1-based coordinates; interval k = [points[k], points[k+1])
        std::vector<uint32_t> ids;    // its label set (CRP_NO_FEATURE: none)
    };
    std::vector<Seq> seqs;  // in order of first appearance in the GFF
    std::string blob;       // the distinct label-set strings, back to back
    std::vector<uint64_t> off{0};
    uint64_t n_gene = 0, n_cds = 0;
};

namespace {

using sv = std::string_view;

bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

sv strip(sv s)
{
    while (!s.empty() && is_space(s.front())) s.remove_prefix(1);
    while (!s.empty() && is_space(s.back())) s.remove_suffix(1);
    return s;
}

// fields of one line (no newline inside)
void split_tabs(sv line, std::vector<sv> &out)
{
    out.clear();
    size_t a = 0;
    for (;;) {
        const size_t b = line.find('\t', a);
        if (b == sv::npos) {
            out.push_back(line.substr(a));
            return;
        }
        out.push_back(line.substr(a, b - a));
        a = b + 1;
    }
}

bool digits(sv s, int64_t *v)
{
    if (s.empty() || s.size() > 18) return false;
    int64_t x = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        x = x * 10 + (c - '0');
    }
    *v = x;
    return true;
}

struct Info {
    std::string best_hit, defline;
};

// Phytozome annotation_info: tab-separated, columns by name from a '#...' header line that has a locusName column,
// else the usual positions; the first line of a locus wins
void parse_info(sv data, std::unordered_map<std::string, Info> &info)
{
    std::vector<std::string> names = {"pacId", "locusName", "transcriptName", "peptideName", "Pfam", "Panther", "KOG",
                                      "KEGG/ec", "KO", "GO", "Best-hit-arabi-name", "arabi-symbol", "arabi-defline"};
    std::vector<sv> cols;
    size_t p = 0;
    while (p < data.size()) {
        size_t q = data.find('\n', p);
        if (q == sv::npos) q = data.size();
        const sv line = data.substr(p, q - p);
        p = q + 1;
        split_tabs(line, cols);
        if (!line.empty() && line[0] == '#') {
            std::vector<std::string> head;
            bool has = false;
            for (sv c : cols) {
                while (!c.empty() && c[0] == '#') c.remove_prefix(1);
                head.emplace_back(c);
                has = has || c == "locusName";
            }
            if (has) names.swap(head);
            continue;
        }
        if (cols.size() < 2) continue;
        // (a repeated column name: the last one counts, like dict(zip(names, cols)))
        sv locus, best, defline;
        const size_t n = std::min(names.size(), cols.size());
        for (size_t k = 0; k < n; ++k) {
            if (names[k] == "locusName") locus = cols[k];
            else if (names[k] == "Best-hit-arabi-name") best = cols[k];
            else if (names[k] == "arabi-defline") defline = cols[k];
        }
        if (locus.empty()) continue;
        info.try_emplace(std::string(locus), Info{std::string(best), std::string(defline)});
    }
}

struct Feature {
    int64_t start, end;
    uint32_t label;  // index into the label table (equal strings share one)
};

}  // namespace

extern "C" {

int crp_annotation_build(const uint8_t *gff, uint64_t gff_len, const uint8_t *info_text, uint64_t info_len,
                         crp_annotation **out)
{
    if (!out || (gff_len && !gff) || (info_len && !info_text)) return CRP_ERR_INVALID;
    *out = nullptr;
    crp_annotation *an = new (std::nothrow) crp_annotation();
    if (!an) return CRP_ERR_NOMEM;
    try {
        std::unordered_map<std::string, Info> info;
        const bool have_info = info_text != nullptr;
        if (have_info) parse_info(sv(reinterpret_cast<const char *>(info_text), info_len), info);
        const sv data(reinterpret_cast<const char *>(gff), gff_len);
        std::unordered_map<std::string, uint32_t> seq_of, label_of;
        std::vector<std::string> labels;
        std::vector<std::vector<Feature>> feats;  // per seqid, file order
        std::vector<sv> cols;
        std::string label;
        // (sized for a typical GFF -- a gene / CDS row every ~250 bytes -- so the tables do not rehash on the way)
        label_of.reserve(gff_len / 256 + 16);
        labels.reserve(gff_len / 256 + 16);
        uint32_t last_seq = 0;  // consecutive rows usually share their seqid
        bool have_last = false;
        size_t p = 0;
        while (p < data.size()) {
            size_t q = data.find('\n', p);
            if (q == sv::npos) q = data.size();
            const sv line = data.substr(p, q - p);
            p = q + 1;
            if (line.empty() || line[0] == '#') continue;
            split_tabs(line, cols);
            if (cols.size() < 9) continue;
            const bool gene = cols[2] == "gene";
            if (!gene && cols[2] != "CDS") continue;
            int64_t start, end;
            if (!digits(cols[3], &start) || !digits(cols[4], &end)) continue;  // unreadable coordinates join nothing
            sv id, name, parent;
            bool has_id = false, has_name = false, has_parent = false;
            {
                const sv attrs = cols[8];
                size_t a = 0;
                for (;;) {
                    size_t b = attrs.find(';', a);
                    const sv part = strip(attrs.substr(a, b == sv::npos ? sv::npos : b - a));
                    const size_t eq = part.find('=');
                    const sv key = part.substr(0, eq), val = eq == sv::npos ? sv() : part.substr(eq + 1);
                    if (key == "ID" && !has_id) id = val, has_id = true;
                    else if (key == "Name" && !has_name) name = val, has_name = true;
                    else if (key == "Parent" && !has_parent) parent = val, has_parent = true;
                    if (b == sv::npos) break;
                    a = b + 1;
                }
            }
            const sv ident = !id.empty() ? id : !name.empty() ? name : !parent.empty() ? parent : sv(".");
            label.assign(gene ? "gene:" : "CDS:");
            label.append(ident);
            if (have_info && gene) {
                auto hit = info.find(std::string(name));
                if (hit == info.end()) hit = info.find(std::string(id));
                if (hit != info.end()) {
                    if (!hit->second.best_hit.empty()) label.append("|").append(hit->second.best_hit);
                    if (!hit->second.defline.empty()) label.append("|").append(hit->second.defline);
                }
            }
            if (!have_last || an->seqs[last_seq].name != cols[0]) {
                auto s = seq_of.find(std::string(cols[0]));
                if (s == seq_of.end()) {
                    s = seq_of.emplace(std::string(cols[0]), (uint32_t)an->seqs.size()).first;
                    an->seqs.emplace_back();
                    an->seqs.back().name.assign(cols[0]);
                    feats.emplace_back();
                }
                last_seq = s->second;
                have_last = true;
            }
            auto l = label_of.find(label);
            if (l == label_of.end()) {
                l = label_of.emplace(label, (uint32_t)labels.size()).first;
                labels.push_back(label);
            }
            feats[last_seq].push_back(Feature{start, end, l->second});
            (gene ? an->n_gene : an->n_cds) += 1;
        }
        // the sweep, seqid by seqid
        std::unordered_map<std::string, uint32_t> string_of;
        string_of.reserve(2 * labels.size() + 16);
        an->off.reserve(2 * labels.size() + 16);
        std::vector<uint32_t> by_start, active, set;
        std::string text;
        for (size_t k = 0; k < an->seqs.size(); ++k) {
            const std::vector<Feature> &f = feats[k];
            crp_annotation::Seq &seq = an->seqs[k];
            seq.points.reserve(2 * f.size());
            for (const Feature &t : f) {
                seq.points.push_back(t.start);
                seq.points.push_back(t.end + 1);
            }
            std::sort(seq.points.begin(), seq.points.end());
            seq.points.erase(std::unique(seq.points.begin(), seq.points.end()), seq.points.end());
            by_start.resize(f.size());
            for (uint32_t i = 0; i < f.size(); ++i) by_start[i] = i;
            std::stable_sort(by_start.begin(), by_start.end(), [&](uint32_t a, uint32_t b) { return f[a].start < f[b].start; });
            seq.ids.assign(seq.points.size(), CRP_NO_FEATURE);
            active.clear();
            size_t nxt = 0;
            for (size_t i = 0; i < seq.points.size(); ++i) {
                const int64_t x = seq.points[i];
                while (nxt < by_start.size() && f[by_start[nxt]].start <= x) active.push_back(by_start[nxt++]);
                active.erase(std::remove_if(active.begin(), active.end(), [&](uint32_t a) { return f[a].end < x; }), active.end());
                if (active.empty()) continue;
                std::sort(active.begin(), active.end());  // feature index = file order
                set.clear();
                for (uint32_t a : active)
                    if (std::find(set.begin(), set.end(), f[a].label) == set.end()) set.push_back(f[a].label);
                text.clear();
                for (size_t j = 0; j < set.size(); ++j) {
                    if (j) text.push_back(';');
                    text.append(labels[set[j]]);
                }
                auto it = string_of.find(text);
                if (it == string_of.end()) {
                    it = string_of.emplace(text, (uint32_t)(an->off.size() - 1)).first;
                    an->blob.append(text);
                    an->off.push_back(an->blob.size());
                }
                seq.ids[i] = it->second;
            }
        }
    } catch (const std::bad_alloc &) {
        delete an;
        return CRP_ERR_NOMEM;
    }
    *out = an;
    return CRP_OK;
}

int crp_annotation_destroy(crp_annotation *an)
{
    delete an;
    return CRP_OK;
}

int crp_annotation_stats(const crp_annotation *an, uint64_t *n_seqids, uint64_t *n_strings, uint64_t *blob_bytes,
                         uint64_t *n_genes, uint64_t *n_cds)
{
    if (!an) return CRP_ERR_INVALID;
    if (n_seqids) *n_seqids = an->seqs.size();
    if (n_strings) *n_strings = an->off.size() - 1;
    if (blob_bytes) *blob_bytes = an->blob.size();
    if (n_genes) *n_genes = an->n_gene;
    if (n_cds) *n_cds = an->n_cds;
    return CRP_OK;
}

int crp_annotation_strings(const crp_annotation *an, uint8_t *blob, uint64_t *offsets)
{
    if (!an) return CRP_ERR_INVALID;
    if (blob && !an->blob.empty()) std::memcpy(blob, an->blob.data(), an->blob.size());
    if (offsets) std::memcpy(offsets, an->off.data(), an->off.size() * sizeof(uint64_t));
    return CRP_OK;
}

int crp_annotation_seqid(const crp_annotation *an, uint64_t k, const uint8_t **name, uint64_t *name_len,
                         const int64_t **points, const uint32_t **ids, uint64_t *n_points)
{
    if (!an || k >= an->seqs.size()) return CRP_ERR_INVALID;
    const crp_annotation::Seq &s = an->seqs[k];
    if (name) *name = reinterpret_cast<const uint8_t *>(s.name.data());
    if (name_len) *name_len = s.name.size();
    if (points) *points = s.points.data();
    if (ids) *ids = s.ids.data();
    if (n_points) *n_points = s.points.size();
    return CRP_OK;
}

int crp_annotation_track(const crp_annotation *an, const uint64_t *entries, uint64_t n_entries, int dec, uint32_t *points,
                         uint32_t *ids, uint64_t cap, uint64_t *n_out)
{
    if (!an || (n_entries && !entries) || !n_out || (cap && (!points || !ids))) return CRP_ERR_INVALID;
    uint64_t n = 0, prev_base = 0, prev_end = 0;
    for (uint64_t e = 0; e < n_entries; ++e) {
        const uint64_t seq = entries[4 * e], lo = entries[4 * e + 1], len = entries[4 * e + 2], base = entries[4 * e + 3];
        // texts lie in the arena in ascending order and do not overlap: the track comes out strictly ascending
        if (base + len > 0x7fffffffull || (e && (base < prev_end || base <= prev_base))) return CRP_ERR_INVALID;
        prev_base = base;
        prev_end = base + len;
        uint32_t first = CRP_NO_FEATURE;
        size_t k = 0;
        const crp_annotation::Seq *s = seq < an->seqs.size() ? &an->seqs[seq] : nullptr;
        // index of the 1-based genome coordinate p inside the text: p + dec - 1 - lo
        const int64_t shift = (int64_t)dec - 1 - (int64_t)lo;
        if (s) {
            // the points at or before the text's first character: the last of them says what holds there
            k = (size_t)(std::upper_bound(s->points.begin(), s->points.end(), -shift) - s->points.begin());
            if (k) first = s->ids[k - 1];
        }
        if (n < cap) {
            points[n] = (uint32_t)base;
            ids[n] = first;
        }
        n += 1;
        if (s)
            for (; k < s->points.size() && s->points[k] + shift < (int64_t)len; ++k) {
                if (n < cap) {
                    points[n] = (uint32_t)(base + (uint64_t)(s->points[k] + shift));
                    ids[n] = s->ids[k];
                }
                n += 1;
            }
    }
    *n_out = n;
    return n <= cap ? CRP_OK : CRP_ERR_CAPACITY;
}

}  // extern "C"
