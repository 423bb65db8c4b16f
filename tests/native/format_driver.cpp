// Sanitizer driver for the host-side formatter (crp_format.cpp): random rows, including
// windows that hang over both contig ends and fields that need quoting, through
// crp_format_rows and crp_write_rows; the two must produce the same bytes.
// Built and run by tests/test_sanitizers.py with -fsanitize=address,undefined and =thread.
#include "cropsr_hip.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <unistd.h>

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/format_driver.out";
    std::mt19937_64 rng(7);
    const char alphabet[] = "ACGTacgtNUZ,\"'\r\n)]";
    int failures = 0;
    for (int round = 0; round < 6; ++round) {
        const uint64_t len = round == 0 ? 0 : 40 + rng() % 5000;
        const uint64_t n = round == 0 ? 0 : (round == 5 ? 70000 : 1 + rng() % 3000);
        const int l = round == 3 ? 50 : (round == 4 ? 1 : 20);
        std::vector<uint8_t> text(len);
        for (auto &c : text) c = (uint8_t)alphabet[rng() % (sizeof alphabet - 1)];
        std::vector<uint32_t> pos(n);
        std::vector<uint8_t> minus(n), ids(7 * n);
        std::vector<double> score(n);
        for (uint64_t r = 0; r < n; ++r) {
            pos[r] = (uint32_t)(rng() % (len + 8));
            minus[r] = (uint8_t)(rng() & 1);
            score[r] = (double)(rng() % 1000003) / 1000003.0 * (r % 7 == 0 ? 1e-9 : 1.0);
            for (int k = 0; k < 7; ++k) ids[7 * r + k] = (uint8_t)('A' + rng() % 26);
        }
        const std::string chrom = round % 2 ? "Chr,\"1\"" : "Chr01";
        std::vector<uint8_t> out(n * 512 + 64);
        uint64_t used = 0;
        int st = crp_format_rows(text.data(), len, (const uint8_t *)chrom.data(), chrom.size(), l, pos.data(),
                                 minus.data(), score.data(), ids.data(), n, out.data(), out.size(), &used, 4);
        if (st != CRP_OK) { std::printf("format_rows status %d\n", st); ++failures; continue; }
        uint64_t need = 0;
        st = crp_format_rows(text.data(), len, (const uint8_t *)chrom.data(), chrom.size(), l, pos.data(),
                             minus.data(), score.data(), ids.data(), n, out.data(), used ? used - 1 : 0, &need, 2);
        if (n && (st != CRP_ERR_CAPACITY || need != used)) { std::printf("capacity probe %d\n", st); ++failures; }
        std::FILE *f = std::fopen(path, "wb");
        if (!f) return 2;
        uint64_t written = 0;
        st = crp_write_rows(fileno(f), text.data(), len, (const uint8_t *)chrom.data(), chrom.size(), l, pos.data(),
                            minus.data(), score.data(), ids.data(), n, &written, 6);
        std::fclose(f);
        if (st != CRP_OK || written != used) { std::printf("write_rows status %d\n", st); ++failures; continue; }
        std::vector<uint8_t> back(used);
        f = std::fopen(path, "rb");
        const size_t got = used ? std::fread(back.data(), 1, used, f) : 0;
        std::fclose(f);
        if (got != used || !std::equal(back.begin(), back.end(), out.begin())) { std::printf("bytes differ\n"); ++failures; }
        // the opt-in columns (crp_write_rows_ex): a string table for `features` (entries that need quoting
        // included) with one index per row, and four off-target counts per row (0xFFFFFFFF prints as -1)
        const std::vector<std::string> table = {"gene:G1", "gene:G1;CDS:c,1", "x\"y", ""};
        std::string blob;
        std::vector<uint64_t> off(1, 0);
        for (const auto &t : table) { blob += t; off.push_back(blob.size()); }
        std::vector<uint32_t> fidx(n), ot(4 * n);
        for (uint64_t r = 0; r < n; ++r) {
            fidx[r] = rng() % 3 == 0 ? 0xffffffffu : (uint32_t)(rng() % table.size());
            for (int k = 0; k < 4; ++k) ot[4 * r + k] = rng() % 5 == 0 ? 0xffffffffu : (uint32_t)(rng() % 100000);
        }
        f = std::fopen(path, "wb");
        uint64_t written_ex = 0;
        st = crp_write_rows_ex(fileno(f), text.data(), len, (const uint8_t *)chrom.data(), chrom.size(), l, pos.data(),
                               minus.data(), score.data(), ids.data(), n, (const uint8_t *)blob.data(), off.data(), fidx.data(),
                               ot.data(), &written_ex, 5);
        std::fclose(f);
        if (st != CRP_OK || written_ex < used + 8 * n) { std::printf("write_rows_ex status %d\n", st); ++failures; continue; }
        std::vector<char> ex(written_ex + 1, 0);
        f = std::fopen(path, "rb");
        const size_t got_ex = written_ex ? std::fread(ex.data(), 1, written_ex, f) : 0;
        std::fclose(f);
        // every row ends in its four counts
        const char *p = ex.data(), *end = ex.data() + got_ex;
        for (uint64_t r = 0; r < n && failures == 0; ++r) {
            const char *eol = p;
            // rows may contain quoted \r\n inside fields: the row's own end is the \r\n after its last count
            char want[64];
            const int m = std::snprintf(want, sizeof want, ",%lld,%lld,%lld,%lld\r\n",
                                        ot[4 * r] == 0xffffffffu ? -1ll : (long long)ot[4 * r],
                                        ot[4 * r + 1] == 0xffffffffu ? -1ll : (long long)ot[4 * r + 1],
                                        ot[4 * r + 2] == 0xffffffffu ? -1ll : (long long)ot[4 * r + 2],
                                        ot[4 * r + 3] == 0xffffffffu ? -1ll : (long long)ot[4 * r + 3]);
            eol = std::search(p, end, want, want + m);
            if (eol == end) { std::printf("row %llu: counts not found\n", (unsigned long long)r); ++failures; break; }
            p = eol + m;
        }
        if (failures == 0 && p != end) { std::printf("trailing bytes after the last row\n"); ++failures; }
        // and without the extras the _ex entry point is crp_write_rows
        f = std::fopen(path, "wb");
        uint64_t written_plain = 0;
        st = crp_write_rows_ex(fileno(f), text.data(), len, (const uint8_t *)chrom.data(), chrom.size(), l, pos.data(),
                               minus.data(), score.data(), ids.data(), n, nullptr, nullptr, nullptr, nullptr, &written_plain, 3);
        std::fclose(f);
        if (st != CRP_OK || written_plain != used) { std::printf("write_rows_ex (plain) status %d\n", st); ++failures; }
    }
    {  // crp_write_segments: many segments in one call (short ones sharing a block, empty ones, one spanning several
       // blocks) == the segments' crp_format_rows bytes one after the other, with 1, 3 and 7 workers
        struct Seg {
            std::vector<uint8_t> text, minus, ids;
            std::vector<uint32_t> pos;
            std::vector<double> score;
            std::string chrom;
        };
        std::vector<Seg> segs(40);
        std::vector<crp_row_segment> view(segs.size());
        std::vector<uint8_t> want;
        for (size_t k = 0; k < segs.size(); ++k) {
            Seg &g = segs[k];
            const uint64_t len = 40 + rng() % 3000;
            const uint64_t n = k % 9 == 4 ? 0 : (k == 17 ? 50000 : 1 + rng() % 6000);
            g.text.resize(len);
            for (auto &c : g.text) c = (uint8_t)alphabet[rng() % (sizeof alphabet - 1)];
            g.pos.resize(n);
            g.minus.resize(n);
            g.ids.resize(7 * n);
            g.score.resize(n);
            for (uint64_t r = 0; r < n; ++r) {
                g.pos[r] = (uint32_t)(rng() % (len + 8));
                g.minus[r] = (uint8_t)(rng() & 1);
                g.score[r] = (double)(rng() % 1000003) / 1000003.0;
                for (int c = 0; c < 7; ++c) g.ids[7 * r + c] = (uint8_t)('A' + rng() % 26);
            }
            g.chrom = k % 3 ? "scaffold_" + std::to_string(k) : std::string("s,\"") + std::to_string(k);
            std::vector<uint8_t> out(n * 512 + 64);
            uint64_t used = 0;
            if (crp_format_rows(g.text.data(), len, (const uint8_t *)g.chrom.data(), g.chrom.size(), 20, g.pos.data(), g.minus.data(),
                                g.score.data(), g.ids.data(), n, out.data(), out.size(), &used, 2) != CRP_OK)
                ++failures;
            want.insert(want.end(), out.begin(), out.begin() + (long)used);
            view[k] = crp_row_segment{g.text.data(), len, (const uint8_t *)g.chrom.data(), g.chrom.size(), g.pos.data(), g.minus.data(),
                                      g.score.data(), g.ids.data(), n, nullptr, nullptr, nullptr, nullptr};
        }
        for (int threads : {1, 3, 7}) {
            std::FILE *f = std::fopen(path, "wb");
            if (!f) return 2;
            uint64_t written = 0;
            const int st = crp_write_segments(fileno(f), 20, view.data(), view.size(), &written, threads);
            std::fclose(f);
            std::vector<uint8_t> back(written);
            f = std::fopen(path, "rb");
            const size_t got = written ? std::fread(back.data(), 1, written, f) : 0;
            std::fclose(f);
            if (st != CRP_OK || written != want.size() || got != written || back != want) { std::printf("write_segments (%d threads) differs\n", threads); ++failures; }
        }
        uint64_t none = 7;
        if (crp_write_segments(1, 20, nullptr, 0, &none, 4) != CRP_OK || none != 0) ++failures;
        if (crp_write_segments(-1, 20, view.data(), view.size(), nullptr, 4) != CRP_ERR_INVALID) ++failures;
    }
    {  // crp_legacy_ids: forward and last-first draws from the same MT19937 state agree row for row
        std::vector<uint32_t> key(624), key2;
        for (auto &k : key) k = (uint32_t)rng();
        key2 = key;
        int32_t pos = 77, pos2 = 77;
        const uint64_t n = 5003;
        std::vector<uint8_t> fwd(7 * n), rev(7 * n);
        if (crp_legacy_ids(key.data(), &pos, fwd.data(), n, 0) != CRP_OK || crp_legacy_ids(key2.data(), &pos2, rev.data(), n, 1) != CRP_OK)
            ++failures;
        for (uint64_t r = 0; r < n; ++r)
            if (std::memcmp(fwd.data() + 7 * r, rev.data() + 7 * (n - 1 - r), 7) != 0) { ++failures; break; }
        if (pos != pos2 || key != key2 || crp_legacy_ids(key.data(), &pos, nullptr, 0, 0) != CRP_OK) ++failures;
        int32_t bad = 700;
        if (crp_legacy_ids(key.data(), &bad, fwd.data(), 1, 0) != CRP_ERR_INVALID) ++failures;
    }
    unlink(path);
    std::printf(failures ? "FAILED\n" : "OK\n");
    return failures ? 1 : 0;
}
