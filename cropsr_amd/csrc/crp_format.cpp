// crp_format.cpp -- native CSV row formatter.
//
// Host-side, multi-threaded replacement for the row building + csv.writer.writerows
// of the reference (CROPSR.py:463-474, with the strings of :420-421 / :431-432).  It
// emits the very bytes Python's csv module emits for those tuples:
//   * dialect "excel": ',' delimiter, '"' quote char, QUOTE_MINIMAL, "\r\n" line ends;
//     a field is quoted iff it contains ',', '"', '\r' or '\n', quotes inside doubled;
//   * ints as decimal; floats as Python's repr(float): shortest round-trip digits,
//     exponent notation iff the decimal exponent is < -4 or >= 16, two-digit exponents;
//   * 12 fields for scored rows, 11 (no cutsite, literal -1) for the others (:466-468).
// The strings are rebuilt from the contig text exactly as the reference slices and maps
// them (chained str.replace = per-character maps, Python slice clamping at the end).
#include "cropsr_hip.h"

#include <algorithm>
#include <charconv>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Maps {
    uint8_t rna[256];    // get_gRNA_sequence before its [::-1]          (CROPSR.py:128)
    uint8_t minus[256];  // get_gRNA_sequence(get_reverse_complement(.)) (CROPSR.py:120,128), reversals cancel
    Maps()
    {
        auto push = [](uint8_t c, const char *chain) {
            for (const char *p = chain; p[0]; p += 2)
                if (c == (uint8_t)p[0]) c = (uint8_t)p[1];
            return c;
        };
        for (int c = 0; c < 256; ++c) {
            rna[c] = push((uint8_t)c, "AUCZGCZGTA");
            minus[c] = push(push((uint8_t)c, "AUCZGCZGTAUT"), "AUCZGCZGTA");
        }
    }
};
const Maps kMaps;

inline void put_int(std::string &o, int64_t v)
{
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    o.append(buf, r.ptr);
}

// repr(float) of CPython (float_repr_style 'short')
void put_repr(std::string &o, double x)
{
    if (x != x) { o += "nan"; return; }
    if (x == 1.0 / 0.0) { o += "inf"; return; }
    if (x == -1.0 / 0.0) { o += "-inf"; return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);  // shortest round-trip
    // parse  [-]d[.ddd]e[+-]XX
    const char *p = buf;
    bool neg = false;
    if (*p == '-') { neg = true; ++p; }
    char digits[32];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int e10 = 0;
    if (p < r.ptr && *p == 'e') {
        ++p;
        bool eneg = false;
        if (*p == '+') ++p;
        else if (*p == '-') { eneg = true; ++p; }
        for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
        if (eneg) e10 = -e10;
    }
    if (nd == 1 && digits[0] == '0') {  // +-0.0
        o += neg ? "-0.0" : "0.0";
        return;
    }
    const int decpt = e10 + 1;  // value = 0.d1d2... * 10^decpt
    if (neg) o += '-';
    if (decpt <= -4 || decpt > 16) {
        o += digits[0];
        if (nd > 1) {
            o += '.';
            o.append(digits + 1, nd - 1);
        }
        o += 'e';
        int ex = decpt - 1;
        o += ex < 0 ? '-' : '+';
        if (ex < 0) ex = -ex;
        if (ex < 10) o += '0';
        put_int(o, ex);
    } else if (decpt <= 0) {
        o += "0.";
        o.append((size_t)(-decpt), '0');
        o.append(digits, nd);
    } else if (decpt >= nd) {
        o.append(digits, nd);
        o.append((size_t)(decpt - nd), '0');
        o += ".0";
    } else {
        o.append(digits, decpt);
        o += '.';
        o.append(digits + decpt, nd - decpt);
    }
}

// one csv field, QUOTE_MINIMAL
void put_field(std::string &o, const uint8_t *s, size_t n)
{
    bool quote = false;
    for (size_t k = 0; k < n; ++k) {
        const uint8_t c = s[k];
        if (c == ',' || c == '"' || c == '\r' || c == '\n') { quote = true; break; }
    }
    if (!quote) {
        o.append(reinterpret_cast<const char *>(s), n);
        return;
    }
    o += '"';
    for (size_t k = 0; k < n; ++k) {
        if (s[k] == '"') o += '"';
        o += (char)s[k];
    }
    o += '"';
}

// text[a:b] with Python clamping, mapped; '+' strand output is reversed
size_t mapped_slice(const uint8_t *text, uint64_t len, int64_t a, int64_t b, bool minus, uint8_t *out)
{
    if (a < 0) a = 0;
    if (b > (int64_t)len) b = (int64_t)len;
    if (b <= a) return 0;
    const size_t n = (size_t)(b - a);
    if (minus) {
        for (size_t k = 0; k < n; ++k) out[k] = kMaps.minus[text[a + k]];
    } else {
        for (size_t k = 0; k < n; ++k) out[k] = kMaps.rna[text[b - 1 - k]];
    }
    return n;
}

struct Job {
    const uint8_t *text;
    uint64_t len;
    const uint8_t *chrom;
    uint64_t chrom_len;
    int l;
    const uint32_t *pos;
    const uint8_t *minus;
    const double *score;
    const uint8_t *ids;
};

void format_range(const Job &j, uint64_t r0, uint64_t r1, std::string &o)
{
    uint8_t shortbuf[128], longbuf[160];
    std::string chrom_field;
    put_field(chrom_field, j.chrom, j.chrom_len);
    o.reserve((size_t)(r1 - r0) * (150 + chrom_field.size()));
    const int l = j.l;
    for (uint64_t r = r0; r < r1; ++r) {
        const bool minus = j.minus[r] != 0;
        const int64_t p = j.pos[r];
        int64_t start, end, sa, sb;
        if (minus) {  // pam_location = (j+3, j+3+l); row = [j+3+l, j+3, ...]   (CROPSR.py:429,433)
            sa = p + 3;
            sb = p + 3 + l;
            start = sb;
            end = sa;
        } else {      // pam_location = (i-l, i)                                (CROPSR.py:418,422)
            sa = p - l;
            sb = p;
            start = sa;
            end = sb;
        }
        const size_t ns = mapped_slice(j.text, j.len, sa, sb, minus, shortbuf);
        const size_t nl = mapped_slice(j.text, j.len, sa - 5, sb + 5, minus, longbuf);
        o.append(reinterpret_cast<const char *>(j.ids + 7 * r), 7);
        o += ",cas9,";
        put_field(o, shortbuf, ns);
        o += ',';
        put_field(o, longbuf, nl);
        o += ',';
        o += chrom_field;
        o += ',';
        put_int(o, start);
        o += ',';
        put_int(o, end);
        o += ',';
        if (nl == 30) {  // CROPSR.py:466: 12 fields with cutsite (end - 3) and the score
            put_int(o, end - 3);
            o += minus ? ",-," : ",+,";
            put_repr(o, j.score[r]);
            o += ",,completed\r\n";
        } else {         // 11 fields, literal -1
            o += minus ? "-,-1,,completed\r\n" : "+,-1,,completed\r\n";
        }
    }
}

}  // namespace

extern "C" int crp_format_rows(const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom,
                               uint64_t chrom_len, int guide_len, const uint32_t *pos, const uint8_t *minus,
                               const double *score, const uint8_t *ids, uint64_t n_rows, uint8_t *out,
                               uint64_t out_cap, uint64_t *out_len, int n_threads)
{
    if (!out_len || guide_len < 1 || guide_len > 50) return CRP_ERR_INVALID;
    if (n_rows && (!contig_text || !pos || !minus || !score || !ids)) return CRP_ERR_INVALID;
    if (chrom_len && !chrom) return CRP_ERR_INVALID;
    const Job job{contig_text, contig_len, chrom, chrom_len, guide_len, pos, minus, score, ids};
    int nt = n_threads < 1 ? 1 : n_threads;
    if ((uint64_t)nt > n_rows / 4096 + 1) nt = (int)(n_rows / 4096 + 1);
    std::vector<std::string> parts((size_t)nt);
    const uint64_t per = (n_rows + nt - 1) / nt;
    std::vector<char> failed((size_t)nt, 0);
    auto work = [&](int t) {
        const uint64_t r0 = std::min<uint64_t>(n_rows, per * t), r1 = std::min<uint64_t>(n_rows, r0 + per);
        try {
            format_range(job, r0, r1, parts[(size_t)t]);
        } catch (...) {
            failed[(size_t)t] = 1;  // out of memory: nothing may escape a thread or the C ABI
        }
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    for (char f : failed)
        if (f) return CRP_ERR_NOMEM;
    uint64_t total = 0;
    for (auto &s : parts) total += s.size();
    *out_len = total;
    if (total > out_cap || (total && !out)) return CRP_ERR_CAPACITY;
    uint64_t off = 0;
    for (auto &s : parts) {
        std::memcpy(out + off, s.data(), s.size());
        off += s.size();
    }
    return CRP_OK;
}
