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
//
// Two entry points share one block formatter: crp_format_rows fills a caller buffer,
// crp_write_rows appends to a file descriptor (workers format blocks of rows into
// private buffers and commit them with write(2) in row order, so nothing the size of the
// whole CSV is ever held in memory).
#include "cropsr_hip.h"
#include "crp_roctx.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

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

inline char *put_lit(char *o, const char *s, size_t n)
{
    std::memcpy(o, s, n);
    return o + n;
}
#define PUT_LIT(o, s) put_lit((o), (s), sizeof(s) - 1)

inline char *put_int(char *o, int64_t v)
{
    return std::to_chars(o, o + 24, v).ptr;
}

// repr(float) of CPython (float_repr_style 'short'); at most 24 characters
char *put_repr(char *o, double x)
{
    if (x != x) return PUT_LIT(o, "nan");
    if (x == 1.0 / 0.0) return PUT_LIT(o, "inf");
    if (x == -1.0 / 0.0) return PUT_LIT(o, "-inf");
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
    if (nd == 1 && digits[0] == '0')  // +-0.0
        return neg ? PUT_LIT(o, "-0.0") : PUT_LIT(o, "0.0");
    const int decpt = e10 + 1;  // value = 0.d1d2... * 10^decpt
    if (neg) *o++ = '-';
    if (decpt <= -4 || decpt > 16) {
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            o = put_lit(o, digits + 1, (size_t)(nd - 1));
        }
        *o++ = 'e';
        int ex = decpt - 1;
        *o++ = ex < 0 ? '-' : '+';
        if (ex < 0) ex = -ex;
        if (ex < 10) *o++ = '0';
        o = put_int(o, ex);
    } else if (decpt <= 0) {
        o = PUT_LIT(o, "0.");
        std::memset(o, '0', (size_t)(-decpt));
        o += -decpt;
        o = put_lit(o, digits, (size_t)nd);
    } else if (decpt >= nd) {
        o = put_lit(o, digits, (size_t)nd);
        std::memset(o, '0', (size_t)(decpt - nd));
        o += decpt - nd;
        o = PUT_LIT(o, ".0");
    } else {
        o = put_lit(o, digits, (size_t)decpt);
        *o++ = '.';
        o = put_lit(o, digits + decpt, (size_t)(nd - decpt));
    }
    return o;
}

// one csv field, QUOTE_MINIMAL; at most 2 n + 2 characters
char *put_field(char *o, const uint8_t *s, size_t n)
{
    bool quote = false;
    for (size_t k = 0; k < n; ++k) {
        const uint8_t c = s[k];
        if (c == ',' || c == '"' || c == '\r' || c == '\n') { quote = true; break; }
    }
    if (!quote) return put_lit(o, reinterpret_cast<const char *>(s), n);
    *o++ = '"';
    for (size_t k = 0; k < n; ++k) {
        if (s[k] == '"') *o++ = '"';
        *o++ = (char)s[k];
    }
    *o++ = '"';
    return o;
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
    // opt-in extensions (crp_write_rows_ex); all null for the reference's rows
    const uint8_t *feat_blob = nullptr;
    const uint64_t *feat_off = nullptr;
    const uint32_t *feat_idx = nullptr;
    const uint32_t *offtarget = nullptr;
    std::string chrom_field;  // the chromosome column as csv writes it
    size_t row_bound;         // no row is longer than this

    Job(const uint8_t *text_, uint64_t len_, const uint8_t *chrom_, uint64_t chrom_len_, int l_, const uint32_t *pos_,
        const uint8_t *minus_, const double *score_, const uint8_t *ids_)
        : text(text_), len(len_), chrom(chrom_), chrom_len(chrom_len_), l(l_), pos(pos_), minus(minus_),
          score(score_), ids(ids_)
    {
        chrom_field.resize(2 * chrom_len + 2);
        char *e = put_field(&chrom_field[0], chrom, chrom_len);
        chrom_field.resize((size_t)(e - chrom_field.data()));
        // id 7 + ",cas9," 6 + short <= 2l+2 + long <= 2(l+10)+2 + 3 ints <= 11 each + repr <= 24
        // + strand, commas, ",,completed\r\n" <= 24
        row_bound = 7 + 6 + (2 * (size_t)l + 2) + (2 * ((size_t)l + 10) + 2) + chrom_field.size() + 33 + 24 + 24;
    }
};

// rows [r0, r1) -> o (room for (r1 - r0) * row_bound bytes); returns the end
char *format_range(const Job &j, uint64_t r0, uint64_t r1, char *o)
{
    uint8_t shortbuf[64], longbuf[80];
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
        o = put_lit(o, reinterpret_cast<const char *>(j.ids + 7 * r), 7);
        o = PUT_LIT(o, ",cas9,");
        o = put_field(o, shortbuf, ns);
        *o++ = ',';
        o = put_field(o, longbuf, nl);
        *o++ = ',';
        o = put_lit(o, j.chrom_field.data(), j.chrom_field.size());
        *o++ = ',';
        o = put_int(o, start);
        *o++ = ',';
        o = put_int(o, end);
        *o++ = ',';
        if (nl == 30) {  // CROPSR.py:466: 12 fields with cutsite (end - 3) and the score
            o = put_int(o, end - 3);
            o = minus ? PUT_LIT(o, ",-,") : PUT_LIT(o, ",+,");
            o = put_repr(o, j.score[r]);
            *o++ = ',';
            if (j.feat_idx && j.feat_idx[r] != 0xffffffffu) {  // opt-in: the reference writes ''
                const uint64_t f0 = j.feat_off[j.feat_idx[r]], f1 = j.feat_off[j.feat_idx[r] + 1];
                o = put_field(o, j.feat_blob + f0, (size_t)(f1 - f0));
            }
            o = PUT_LIT(o, ",completed");
        } else {         // 11 fields, literal -1
            o = minus ? PUT_LIT(o, "-,-1,,completed") : PUT_LIT(o, "+,-1,,completed");
        }
        if (j.offtarget) {  // opt-in: four more columns
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = j.offtarget[4 * r + k];
                *o++ = ',';
                o = v == 0xffffffffu ? PUT_LIT(o, "-1") : put_int(o, (int64_t)v);
            }
        }
        o = PUT_LIT(o, "\r\n");
    }
    return o;
}

bool bad_args(const uint8_t *contig_text, const uint8_t *chrom, uint64_t chrom_len, int guide_len, const uint32_t *pos,
              const uint8_t *minus, const double *score, const uint8_t *ids, uint64_t n_rows)
{
    if (guide_len < 1 || guide_len > 50) return true;
    if (n_rows && (!contig_text || !pos || !minus || !score || !ids)) return true;
    return chrom_len && !chrom;
}

int clamp_threads(int n_threads, uint64_t n_rows, uint64_t rows_per_thread_min)
{
    int nt = n_threads < 1 ? 1 : n_threads;
    if ((uint64_t)nt > n_rows / rows_per_thread_min + 1) nt = (int)(n_rows / rows_per_thread_min + 1);
    return nt;
}

template <class F>
void run_threads(int nt, F &&work)
{
    if (nt == 1) {
        work(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" int crp_format_rows(const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom,
                               uint64_t chrom_len, int guide_len, const uint32_t *pos, const uint8_t *minus,
                               const double *score, const uint8_t *ids, uint64_t n_rows, uint8_t *out,
                               uint64_t out_cap, uint64_t *out_len, int n_threads)
{
    if (!out_len || bad_args(contig_text, chrom, chrom_len, guide_len, pos, minus, score, ids, n_rows))
        return CRP_ERR_INVALID;
    try {
        const Job job(contig_text, contig_len, chrom, chrom_len, guide_len, pos, minus, score, ids);
        const int nt = clamp_threads(n_threads, n_rows, 4096);
        const uint64_t per = (n_rows + nt - 1) / nt;
        // pass 1: every thread formats its rows into a private buffer (malloc: untouched pages cost nothing)
        std::vector<char *> part((size_t)nt, nullptr);
        std::vector<uint64_t> used((size_t)nt, 0);
        std::atomic<int> failed{0};
        run_threads(nt, [&](int t) {
            const uint64_t r0 = std::min<uint64_t>(n_rows, per * t), r1 = std::min<uint64_t>(n_rows, r0 + per);
            if (r1 == r0) return;
            char *buf = static_cast<char *>(std::malloc((size_t)(r1 - r0) * job.row_bound));
            if (!buf) {
                failed = 1;
                return;
            }
            part[(size_t)t] = buf;
            used[(size_t)t] = (uint64_t)(format_range(job, r0, r1, buf) - buf);
        });
        uint64_t total = 0;
        std::vector<uint64_t> at((size_t)nt, 0);
        for (int t = 0; t < nt; ++t) {
            at[(size_t)t] = total;
            total += used[(size_t)t];
        }
        *out_len = total;
        int st = CRP_OK;
        if (failed) st = CRP_ERR_NOMEM;
        else if (total > out_cap || (total && !out)) st = CRP_ERR_CAPACITY;
        // pass 2: the parts move to their places in parallel
        if (st == CRP_OK)
            run_threads(nt, [&](int t) {
                if (used[(size_t)t]) std::memcpy(out + at[(size_t)t], part[(size_t)t], used[(size_t)t]);
            });
        for (char *b : part) std::free(b);
        return st;
    } catch (...) {
        return CRP_ERR_NOMEM;  // nothing may escape the C ABI
    }
}

// rows of several SEGMENTS appended to fd in segment order: one team of workers for the whole call.  A block -- the unit a
// worker formats into its buffer and commits with one write(2), in order -- holds up to kBlockRows rows and may span several
// short segments, so a run of small contigs costs what one contig of their total size costs (the CLI used to make one call per
// contig: 626 scaffold passes of 3 500 rows each, formatted by ONE thread and joined, took a fifth of the CSV stage).
extern "C" int crp_write_segments(int fd, int guide_len, const crp_row_segment *segs, uint64_t n_segs, uint64_t *bytes_written, int n_threads)
{
    crp::Range roctx_range("crp: format + write rows");
    if (fd < 0 || (n_segs && !segs)) return CRP_ERR_INVALID;
    if (bytes_written) *bytes_written = 0;
    for (uint64_t k = 0; k < n_segs; ++k) {
        const crp_row_segment &g = segs[k];
        if (bad_args(g.contig_text, g.chrom, g.chrom_len, guide_len, g.pos, g.minus, g.score, g.ids, g.n_rows)) return CRP_ERR_INVALID;
        if (g.feat_idx && (!g.feat_off || !g.feat_blob)) return CRP_ERR_INVALID;
    }
    try {
        std::vector<Job> jobs;
        jobs.reserve((size_t)n_segs);
        struct Part {
            uint32_t seg;
            uint64_t r0, r1;
        };
        std::vector<Part> parts;
        std::vector<size_t> block_at{0};  // block b = parts [block_at[b], block_at[b + 1])
        constexpr uint64_t kBlockRows = 16384;
        size_t block_bytes = 0, cur_bytes = 0;
        uint64_t cur_rows = 0;
        auto close_block = [&]() {
            if (parts.size() == block_at.back()) return;
            block_at.push_back(parts.size());
            block_bytes = std::max(block_bytes, cur_bytes);
            cur_rows = 0;
            cur_bytes = 0;
        };
        for (uint64_t k = 0; k < n_segs; ++k) {
            const crp_row_segment &g = segs[k];
            jobs.emplace_back(g.contig_text, g.contig_len, g.chrom, g.chrom_len, guide_len, g.pos, g.minus, g.score, g.ids);
            Job &job = jobs.back();
            job.feat_blob = g.feat_blob;
            job.feat_off = g.feat_off;
            job.feat_idx = g.feat_idx;
            job.offtarget = g.offtarget;
            if (g.offtarget) job.row_bound += 4 * 11;
            if (g.feat_idx) {  // the longest features entry a row can carry, quoted
                uint64_t longest = 0;
                for (uint64_t r = 0; r < g.n_rows; ++r)
                    if (g.feat_idx[r] != 0xffffffffu) longest = std::max(longest, g.feat_off[g.feat_idx[r] + 1] - g.feat_off[g.feat_idx[r]]);
                job.row_bound += 2 * (size_t)longest + 2;
            }
            for (uint64_t r0 = 0; r0 < g.n_rows;) {
                const uint64_t take = std::min(g.n_rows - r0, kBlockRows - cur_rows);
                parts.push_back(Part{(uint32_t)k, r0, r0 + take});
                cur_rows += take;
                cur_bytes += (size_t)take * job.row_bound;
                r0 += take;
                if (cur_rows == kBlockRows) close_block();
            }
        }
        close_block();
        const uint64_t n_blocks = block_at.size() - 1;
        if (!n_blocks) return CRP_OK;
        const int nt = (int)std::min<uint64_t>((uint64_t)(n_threads < 1 ? 1 : n_threads), n_blocks);
        std::atomic<uint64_t> next_block{0};
        std::mutex mu;
        std::condition_variable turn;
        uint64_t committed = 0;  // blocks written so far (guarded by mu)
        uint64_t total = 0;
        int status = CRP_OK;
        run_threads(nt, [&](int) {
            char *buf = static_cast<char *>(std::malloc(block_bytes));
            for (;;) {
                const uint64_t b = next_block.fetch_add(1);
                if (b >= n_blocks) break;
                size_t n = 0;
                if (buf) {
                    char *o = buf;
                    for (size_t q = block_at[b]; q < block_at[b + 1]; ++q) o = format_range(jobs[parts[q].seg], parts[q].r0, parts[q].r1, o);
                    n = (size_t)(o - buf);
                }
                std::unique_lock<std::mutex> lock(mu);
                turn.wait(lock, [&] { return committed == b; });
                if (!buf && status == CRP_OK) status = CRP_ERR_NOMEM;
                if (status == CRP_OK) {  // blocks go out in row order; a failed write stops the output there
                    size_t done = 0;
                    while (done < n) {
                        const ssize_t w = ::write(fd, buf + done, n - done);
                        if (w < 0) {
                            if (errno == EINTR) continue;
                            status = CRP_ERR_IO;
                            break;
                        }
                        done += (size_t)w;
                    }
                    total += done;
                }
                ++committed;
                lock.unlock();
                turn.notify_all();
            }
            std::free(buf);
        });
        if (bytes_written) *bytes_written = total;
        return status;
    } catch (...) {
        return CRP_ERR_NOMEM;
    }
}

extern "C" int crp_write_rows_ex(int fd, const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom,
                                 uint64_t chrom_len, int guide_len, const uint32_t *pos, const uint8_t *minus,
                                 const double *score, const uint8_t *ids, uint64_t n_rows, const uint8_t *feat_blob,
                                 const uint64_t *feat_off, const uint32_t *feat_idx, const uint32_t *offtarget,
                                 uint64_t *bytes_written, int n_threads)
{
    const crp_row_segment seg{contig_text, contig_len, chrom, chrom_len, pos, minus, score, ids, n_rows, feat_blob, feat_off, feat_idx, offtarget};
    return crp_write_segments(fd, guide_len, &seg, 1, bytes_written, n_threads);
}

extern "C" int crp_write_rows(int fd, const uint8_t *contig_text, uint64_t contig_len, const uint8_t *chrom,
                              uint64_t chrom_len, int guide_len, const uint32_t *pos, const uint8_t *minus,
                              const double *score, const uint8_t *ids, uint64_t n_rows, uint64_t *bytes_written,
                              int n_threads)
{
    return crp_write_rows_ex(fd, contig_text, contig_len, chrom, chrom_len, guide_len, pos, minus, score, ids, n_rows,
                             nullptr, nullptr, nullptr, nullptr, bytes_written, n_threads);
}

// The reference's own id draws, natively.  CROPSR.py:316-318 draws crispr ids with
// np.random.choice(alphanum, [size, 7]) on numpy's global legacy RandomState, which is
// alphanum[randint(0, 36)] = one masked-rejection draw per character: take the next MT19937
// output, keep its low 6 bits, retry while that exceeds 35 (numpy random/src/distributions,
// random_bounded_uint64_fill with use_masked for the legacy generator).  The caller passes the
// MT19937 state it took from np.random.get_state() and puts the advanced state back, so the
// global stream continues exactly as if numpy had made the draws.  reverse != 0 stores row r at
// row n_rows - 1 - r (the order the reference consumes ids in, CROPSR.py:448-449).
//
// Round 6: with the formatter taking whole passes (crp_write_segments) this sequential stream became what the CSV stage waits
// for (0.745 s for 52.4 M rows against 0.744 s of formatting and writing), so the three steps -- state transition, tempering +
// rejection, placing the characters -- also exist as AVX-512 code (VBMI2 byte compress; picked at run time, CRP_IDS_SCALAR=1
// forces the portable loop): 64 outputs per step, same outputs consumed, same state left behind.
namespace {

constexpr int kMtN = 624, kMtM = 397;
alignas(64) const char kIdAlphabet[65] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789????????????????????????????";

void mt_refill_scalar(uint32_t *mt)  // the standard MT19937 state transition
{
    int k = 0;
    for (; k < kMtN - kMtM; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + kMtM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; k < kMtN - 1; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + (kMtM - kMtN)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    const uint32_t y = (mt[kMtN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[kMtN - 1] = mt[kMtM - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// `take` outputs from src: tempered, low six bits, kept when <= 35; the kept ones as characters to out (room for `take`)
int mt_chars_scalar(const uint32_t *src, int take, uint8_t *out)
{
    int n = 0;
    for (int i = 0; i < take; ++i) {  // no data-dependent branch: a conditional increment
        uint32_t y = src[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        const uint32_t v = y & 63u;
        out[n] = (uint8_t)kIdAlphabet[v];
        n += v <= 35u;
    }
    return n;
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define CRP_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi,avx512vbmi2")))

CRP_AVX512 inline __m512i mt_next16(__m512i cur, __m512i nxt, __m512i far)
{
    const __m512i y = _mm512_or_si512(_mm512_and_si512(cur, _mm512_set1_epi32((int)0x80000000u)), _mm512_and_si512(nxt, _mm512_set1_epi32(0x7fffffff)));
    const __m512i mag = _mm512_and_si512(_mm512_sub_epi32(_mm512_setzero_si512(), _mm512_and_si512(y, _mm512_set1_epi32(1))),
                                         _mm512_set1_epi32((int)0x9908b0dfu));
    return _mm512_xor_si512(_mm512_xor_si512(far, _mm512_srli_epi32(y, 1)), mag);
}

// the same transition sixteen words at a time: word k needs the OLD words k, k + 1 and k + 397 (or the NEW word k - 227), and a
// vector of sixteen never reaches a word its own store has touched
CRP_AVX512 void mt_refill_avx512(uint32_t *mt)
{
    int k = 0;
    for (; k + 16 <= kMtN - kMtM; k += 16)
        _mm512_storeu_si512(mt + k, mt_next16(_mm512_loadu_si512(mt + k), _mm512_loadu_si512(mt + k + 1), _mm512_loadu_si512(mt + k + kMtM)));
    for (; k < kMtN - kMtM; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + kMtM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; k + 16 <= kMtN - 1; k += 16)
        _mm512_storeu_si512(mt + k, mt_next16(_mm512_loadu_si512(mt + k), _mm512_loadu_si512(mt + k + 1), _mm512_loadu_si512(mt + k + (kMtM - kMtN))));
    for (; k < kMtN - 1; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
        mt[k] = mt[k + (kMtM - kMtN)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    const uint32_t y = (mt[kMtN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[kMtN - 1] = mt[kMtM - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

CRP_AVX512 inline __m128i mt_temper16_low6(__m512i y)
{
    y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
    y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), _mm512_set1_epi32((int)0x9d2c5680u)));
    y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), _mm512_set1_epi32((int)0xefc60000u)));
    y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
    return _mm512_cvtepi32_epi8(_mm512_and_si512(y, _mm512_set1_epi32(63)));
}

// cnt (1...64) outputs from src -> the kept characters, packed, to out (64 bytes are written whatever the count)
CRP_AVX512 int mt_chars_avx512(const uint32_t *src, int cnt, uint8_t *out)
{
    __m128i part[4];
    for (int g = 0; g < 4; ++g) {
        const int lanes = cnt - 16 * g < 0 ? 0 : (cnt - 16 * g > 16 ? 16 : cnt - 16 * g);
        part[g] = mt_temper16_low6(_mm512_maskz_loadu_epi32((__mmask16)((1u << lanes) - 1u), src + 16 * g));  // (masked lanes are not read)
    }
    __m512i v = _mm512_castsi128_si512(part[0]);
    v = _mm512_inserti32x4(v, part[1], 1);
    v = _mm512_inserti32x4(v, part[2], 2);
    v = _mm512_inserti32x4(v, part[3], 3);
    const __mmask64 valid = cnt >= 64 ? ~(__mmask64)0 : (((__mmask64)1 << cnt) - 1);
    const __mmask64 keep = _mm512_cmple_epu8_mask(v, _mm512_set1_epi8(35)) & valid;
    const __m512i chars = _mm512_permutexvar_epi8(v, _mm512_load_si512(kIdAlphabet));
    _mm512_storeu_si512(out, _mm512_maskz_compress_epi8(keep, chars));
    return (int)__builtin_popcountll((unsigned long long)keep);
}

// r rows of 7 characters (src, first to last) -> dst_end - 7 r ... dst_end, last row first
CRP_AVX512 void rows_reversed_avx512(uint8_t *dst_end, const uint8_t *src, size_t r)
{
    alignas(64) uint8_t idx[64];
    for (int j = 0; j < 64; ++j) idx[j] = (uint8_t)(j < 56 ? 7 * (7 - j / 7) + j % 7 : 0);
    const __m512i perm = _mm512_load_si512(idx);
    const __mmask64 m56 = (((__mmask64)1) << 56) - 1;
    size_t i = 0;
    for (; i + 8 <= r; i += 8)  // eight rows per step: rows i ... i + 7 land, reversed, right below what is already there
        _mm512_mask_storeu_epi8(dst_end - 7 * (i + 8), m56, _mm512_permutexvar_epi8(perm, _mm512_maskz_loadu_epi8(m56, src + 7 * i)));
    for (; i < r; ++i) std::memcpy(dst_end - 7 * (i + 1), src + 7 * i, 7);
}

bool ids_have_avx512()
{
    static const bool yes = [] {
        if (const char *e = std::getenv("CRP_IDS_SCALAR"))
            if (*e && *e != '0') return false;
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
               __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512vbmi2");
    }();
    return yes;
}
#else
bool ids_have_avx512() { return false; }
void mt_refill_avx512(uint32_t *) {}
int mt_chars_avx512(const uint32_t *, int, uint8_t *) { return 0; }
void rows_reversed_avx512(uint8_t *, const uint8_t *, size_t) {}
#endif

}  // namespace

extern "C" int crp_legacy_ids(uint32_t *mt_key, int32_t *mt_pos, uint8_t *ids, uint64_t n_rows, int reverse)
{
    if (!mt_key || !mt_pos || (n_rows && !ids) || *mt_pos < 0 || *mt_pos > 624) return CRP_ERR_INVALID;
    const bool vec = ids_have_avx512();
    int pos = *mt_pos;
    // The characters come out in stream order and are staged a few thousand rows at a time (cache-resident), so that the
    // last-first order costs no second pass over the whole array.  A step never takes more outputs than characters are still
    // missing, so the stream is left exactly where numpy would leave it (the output that yields the last character is the last
    // one consumed): 64 outputs give at most 64 characters, so the vector step runs while 64 or more are missing.
    constexpr size_t kStageRows = 4096;
    alignas(64) uint8_t stage[7 * kStageRows + kMtN + 64];
    size_t fill = 0;
    uint64_t rows_out = 0;
    const uint64_t need = 7 * n_rows;
    uint64_t produced = 0;
    auto flush = [&]() {
        const size_t r = fill / 7;
        if (!r) return;
        if (!reverse) std::memcpy(ids + 7 * rows_out, stage, 7 * r);
        else if (vec) rows_reversed_avx512(ids + 7 * (n_rows - rows_out), stage, r);
        else
            for (size_t i = 0; i < r; ++i) std::memcpy(ids + 7 * (n_rows - 1 - rows_out - i), stage + 7 * i, 7);
        rows_out += r;
        std::memmove(stage, stage + 7 * r, fill - 7 * r);
        fill -= 7 * r;
    };
    while (produced < need) {
        if (pos == kMtN) {
            if (vec) mt_refill_avx512(mt_key);
            else mt_refill_scalar(mt_key);
            pos = 0;
        }
        const uint64_t missing = need - produced;
        int n, took;
        if (vec && missing >= 64) {
            took = kMtN - pos < 64 ? kMtN - pos : 64;
            n = mt_chars_avx512(mt_key + pos, took, stage + fill);
        } else {
            took = (uint64_t)(kMtN - pos) < missing ? kMtN - pos : (int)missing;
            n = mt_chars_scalar(mt_key + pos, took, stage + fill);
        }
        pos += took;
        fill += (size_t)n;
        produced += (uint64_t)n;
        if (fill >= 7 * kStageRows) flush();
    }
    flush();
    *mt_pos = pos;
    return CRP_OK;
}

