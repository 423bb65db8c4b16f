// crp_fasta.cpp -- native FASTA loader.
//
// The reference does not parse FASTA records.  import_fasta_file (CROPSR.py:54-74) asks
// cropsr_functions.formatted (:221-229) for the str() of a Python list of (header, body)
// tuples -- every record cut at '>' and at its first newline, newlines removed -- then
// generate_dictionary (:190-196) splits that printed text on whitespace and pairs the
// tokens up.  For a record whose header and body are "plain" (printable ASCII without
// blank, quote or backslash) the printed form is predictable:
//     key   = [('HEADER',   for the first record,  ('HEADER',   for the others
//     value = 'BODY'),      for every record but the last, which ends in  ')]
// This file builds those value strings (the contig strings the scan runs on) straight
// from the bytes of the file, in parallel, with one read of the input and one write of
// the output.  Everything else -- the already-two-lines-per-record path, headers with
// blanks or quotes, records without a newline -- is reported as "not plain" and the
// Python host falls back to the literal restatement of the reference (fasta.contig_table).
#include "cropsr_hip.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace {

#ifndef CRP_FASTA_CHUNK_BYTES
#define CRP_FASTA_CHUNK_BYTES (4u << 20)  // tests/native/fasta_driver.cpp builds with tiny pieces
#endif
constexpr uint64_t kChunk = CRP_FASTA_CHUNK_BYTES;

struct PlainTable {
    uint8_t ok[256];
    PlainTable()
    {
        for (int c = 0; c < 256; ++c) ok[c] = (c >= 33 && c <= 126 && c != 0x27 && c != 0x5C) ? 1 : 0;
    }
};
const PlainTable kPlain;

inline bool all_plain(const uint8_t *p, size_t n)
{
    uint8_t bad = 0;
    for (size_t k = 0; k < n; ++k) bad |= (uint8_t)(kPlain.ok[p[k]] ^ 1);
    return bad == 0;
}

template <class F>
void run_threads(int nt, F &&work)
{
    if (nt <= 1) {
        work(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
}

struct ChunkInfo {
    uint64_t newlines = 0;
    std::vector<uint64_t> gt;         // positions of '>' in this chunk
    std::vector<uint64_t> nl_before;  // newlines of this chunk before each of them
};

struct Record {
    uint64_t head, head_len;  // header bytes in the input
    uint64_t body, body_end;  // body bytes in the input (newlines included)
    uint64_t nl_at_body;      // newlines of the whole input before `body`
    uint64_t value, value_len;
};

}  // namespace

extern "C" int crp_fasta_table(const uint8_t *data, uint64_t n, uint8_t *out_text, uint64_t out_cap,
                               uint64_t *records, uint64_t records_cap, uint64_t *n_records, uint64_t *out_len,
                               int *plain, int n_threads)
{
    if (!n_records || !out_len || !plain || (n && !data)) return CRP_ERR_INVALID;
    *n_records = 0;
    *out_len = 0;
    *plain = 0;
    try {
        const uint64_t n_chunks = (n + kChunk - 1) / kChunk;
        const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(n_threads < 1 ? 1 : n_threads), n_chunks));
        // pass A: newline counts and '>' positions per chunk
        std::vector<ChunkInfo> info((size_t)n_chunks);
        std::atomic<uint64_t> next{0};
        run_threads(nt, [&](int) {
            for (;;) {
                const uint64_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const uint64_t lo = c * kChunk, hi = std::min(n, lo + kChunk);
                ChunkInfo &ci = info[(size_t)c];
                uint64_t at = lo, nl = 0;
                while (at < hi) {  // count newlines between consecutive '>'
                    const uint8_t *g = static_cast<const uint8_t *>(std::memchr(data + at, '>', (size_t)(hi - at)));
                    const uint64_t stop = g ? (uint64_t)(g - data) : hi;
                    uint64_t cnt = 0;
                    for (uint64_t k = at; k < stop; ++k) cnt += data[k] == '\n';
                    nl += cnt;
                    if (!g) break;
                    ci.gt.push_back(stop);
                    ci.nl_before.push_back(nl);
                    at = stop + 1;
                }
                ci.newlines = nl;
            }
        });
        uint64_t total_nl = 0, total_gt = 0;
        std::vector<uint64_t> nl_prefix((size_t)n_chunks + 1, 0);
        for (uint64_t c = 0; c < n_chunks; ++c) {
            nl_prefix[(size_t)c + 1] = nl_prefix[(size_t)c] + info[(size_t)c].newlines;
            total_gt += info[(size_t)c].gt.size();
        }
        total_nl = nl_prefix[(size_t)n_chunks];
        // CROPSR.py:62-63: a file that is already "two lines per record, no final newline" is not
        // re-formatted; that path is left to the host
        if (2 * total_gt == total_nl + 1) return CRP_OK;

        // records: the non-empty pieces between '>' characters (cropsr_functions.py:226)
        std::vector<Record> recs;
        recs.reserve((size_t)total_gt + 1);
        auto add_piece = [&](uint64_t lo, uint64_t hi, uint64_t nl_at_lo) -> bool {
            if (hi <= lo) return true;  // empty piece: dropped
            const uint8_t *e = static_cast<const uint8_t *>(std::memchr(data + lo, '\n', (size_t)(hi - lo)));
            if (!e) return false;  // a record without a newline prints as a 1-tuple: the pairing shifts
            Record r{};
            r.head = lo;
            r.head_len = (uint64_t)(e - data) - lo;
            r.body = lo + r.head_len + 1;
            r.body_end = hi;
            r.nl_at_body = nl_at_lo + 1;
            if (!all_plain(data + r.head, (size_t)r.head_len)) return false;
            recs.push_back(r);
            return true;
        };
        uint64_t piece_lo = 0, piece_nl = 0;
        for (uint64_t c = 0; c < n_chunks; ++c) {
            const ChunkInfo &ci = info[(size_t)c];
            for (size_t k = 0; k < ci.gt.size(); ++k) {
                if (!add_piece(piece_lo, ci.gt[k], piece_nl)) return CRP_OK;
                piece_lo = ci.gt[k] + 1;
                piece_nl = nl_prefix[(size_t)c] + ci.nl_before[k];
            }
        }
        if (!add_piece(piece_lo, n, piece_nl)) return CRP_OK;
        if (recs.empty()) return CRP_OK;  // str([]) == "[]": one token; the host handles it

        // value k = ' + body without newlines + ') + , or ]
        uint64_t off = 0;
        for (size_t k = 0; k < recs.size(); ++k) {
            Record &r = recs[k];
            const uint64_t nl_at_end = k + 1 < recs.size() ? recs[k + 1].nl_at_body - 1 : total_nl;
            const uint64_t body_nl = nl_at_end - r.nl_at_body;
            r.value = off;
            r.value_len = (r.body_end - r.body) - body_nl + 4;
            off += r.value_len;
        }
        *n_records = recs.size();
        *out_len = off;
        if (recs.size() > records_cap || off > out_cap || !out_text || !records) return CRP_ERR_CAPACITY;

        // pass B: strip-copy, chunk by chunk of the INPUT; a chunk's share of a body lands at
        // value + 1 + (x - body) - newlines in [body, x)
        std::atomic<int> not_plain{0};
        next = 0;
        run_threads(nt, [&](int) {
            for (;;) {
                const uint64_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const uint64_t lo = c * kChunk, hi = std::min(n, lo + kChunk);
                // first record whose body reaches into this chunk
                size_t k = (size_t)(std::upper_bound(recs.begin(), recs.end(), lo,
                                                     [](uint64_t x, const Record &r) { return x < r.body_end; }) -
                                    recs.begin());
                for (; k < recs.size() && recs[k].body < hi; ++k) {
                    const Record &r = recs[k];
                    uint64_t x = std::max(lo, r.body);
                    const uint64_t stop = std::min(hi, r.body_end);
                    if (x >= stop) continue;
                    uint64_t nl_in_body;  // newlines in [r.body, x)
                    if (x == r.body) {
                        nl_in_body = 0;
                    } else {  // x == lo, inside the body: count from the chunk prefix
                        nl_in_body = nl_prefix[(size_t)c] - r.nl_at_body;
                    }
                    uint8_t *dst = out_text + r.value + 1 + (x - r.body) - nl_in_body;
                    bool ok = true;
                    while (x < stop) {
                        const uint8_t *e = static_cast<const uint8_t *>(std::memchr(data + x, '\n', (size_t)(stop - x)));
                        const uint64_t line_end = e ? (uint64_t)(e - data) : stop;
                        const size_t len = (size_t)(line_end - x);
                        ok &= all_plain(data + x, len);
                        std::memcpy(dst, data + x, len);
                        dst += len;
                        x = line_end + 1;
                    }
                    if (!ok) not_plain = 1;
                }
            }
        });
        if (not_plain) return CRP_OK;  // *plain stays 0: the host takes the literal path
        for (size_t k = 0; k < recs.size(); ++k) {
            const Record &r = recs[k];
            uint8_t *v = out_text + r.value;
            v[0] = '\'';
            v[r.value_len - 3] = '\'';
            v[r.value_len - 2] = ')';
            v[r.value_len - 1] = k + 1 == recs.size() ? ']' : ',';
            records[4 * k + 0] = r.head;
            records[4 * k + 1] = r.head_len;
            records[4 * k + 2] = r.value;
            records[4 * k + 3] = r.value_len;
        }
        *plain = 1;
        return CRP_OK;
    } catch (...) {
        return CRP_ERR_NOMEM;
    }
}
