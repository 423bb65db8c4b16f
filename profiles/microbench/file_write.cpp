// file_write.cpp -- how fast can ONE output file be filled from memory on this box?
//   (a) write(2) of 4 MiB pieces in order, one thread          (what crp_write_rows does)
//   (b) pwrite(2) of 4 MiB pieces at their offsets, T threads
//   (c) ftruncate + mmap(MAP_SHARED) + memcpy of 4 MiB pieces, T threads
//   (d) posix_fallocate + mmap(MAP_SHARED) + memcpy, T threads             (round 6: blocks allocated ahead)
//   (e) as (d), the mapping populated ahead of the copies by MADV_POPULATE_WRITE in 256 MiB windows on a helper thread
// For the mapped modes the minor page faults of the run are printed: a fresh file mapping takes one fault per 4 KiB page
// (no huge pages in the page cache of an ordinary file system), whoever fills it.
// g++ -O2 -pthread file_write.cpp -o file_write && ./file_write /path/on/the/disk [GiB] [threads]
#include <fcntl.h>
#include <sys/resource.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "file_write.tmp";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 4.0) * (1ull << 30);
    const int T = argc > 3 ? atoi(argv[3]) : 16;
    const size_t piece = 4u << 20, n_pieces = total / piece;
    std::vector<char> src(piece * 8);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (char)('A' + i % 23);
    for (int mode = 0; mode < 5; ++mode) {
        struct rusage ru0;
        getrusage(RUSAGE_SELF, &ru0);
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const double t0 = now();
        if (mode == 0) {
            for (size_t k = 0; k < n_pieces; ++k)
                if (write(fd, src.data() + (k % 8) * piece, piece) != (ssize_t)piece) { perror("write"); return 1; }
        } else {
            char *map = nullptr;
            if (mode >= 2) {
                if (mode == 2 && ftruncate(fd, (off_t)total)) { perror("ftruncate"); return 1; }
                if (mode >= 3) { const int e = posix_fallocate(fd, 0, (off_t)total); if (e) { fprintf(stderr, "posix_fallocate: %s\n", strerror(e)); return 1; } }
                map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (map == MAP_FAILED) { perror("mmap"); return 1; }
            }
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            std::thread populate;
            if (mode == 4) {
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
                populate = std::thread([&] {
                    const size_t win = 256u << 20;
                    for (size_t off = 0; off < total; off += win)
                        if (madvise(map + off, std::min(win, total - off), MADV_POPULATE_WRITE)) { perror("madvise(MADV_POPULATE_WRITE)"); break; }
                });
            }
            for (int t = 0; t < T; ++t)
                th.emplace_back([&] {
                    for (size_t k; (k = next.fetch_add(1)) < n_pieces;) {
                        if (mode == 1) {
                            if (pwrite(fd, src.data() + (k % 8) * piece, piece, (off_t)(k * piece)) != (ssize_t)piece) perror("pwrite");
                        } else {
                            memcpy(map + k * piece, src.data() + (k % 8) * piece, piece);
                        }
                    }
                });
            for (auto &x : th) x.join();
            if (populate.joinable()) populate.join();
            if (map) munmap(map, total);
        }
        const double dt = now() - t0;
        close(fd);
        struct rusage ru1;
        getrusage(RUSAGE_SELF, &ru1);
        static const char *names[5] = {"write, 1 thread", "pwrite, T threads", "ftruncate + mmap + memcpy, T threads", "posix_fallocate + mmap + memcpy, T threads",
                                       "posix_fallocate + mmap + MADV_POPULATE_WRITE ahead + memcpy, T threads"};
        printf("%s: %.2f GiB in %.3f s = %.2f GB/s; %ld minor faults (%.2f M/s), sys %.2f s\n", names[mode], total / 1073741824.0, dt, total / dt / 1e9,
               ru1.ru_minflt - ru0.ru_minflt, (ru1.ru_minflt - ru0.ru_minflt) / dt / 1e6,
               (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + 1e-6 * (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec));
        fflush(stdout);
    }
    unlink(path);
    return 0;
}
