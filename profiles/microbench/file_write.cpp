// file_write.cpp -- how fast can ONE output file be filled from memory on this box?
//   (a) write(2) of 4 MiB pieces in order, one thread          (what crp_write_rows does)
//   (b) pwrite(2) of 4 MiB pieces at their offsets, T threads
//   (c) ftruncate + mmap(MAP_SHARED) + memcpy of 4 MiB pieces, T threads
// g++ -O2 -pthread file_write.cpp -o file_write && ./file_write /path/on/the/disk [GiB] [threads]
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    for (int mode = 0; mode < 3; ++mode) {
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const double t0 = now();
        if (mode == 0) {
            for (size_t k = 0; k < n_pieces; ++k)
                if (write(fd, src.data() + (k % 8) * piece, piece) != (ssize_t)piece) { perror("write"); return 1; }
        } else {
            char *map = nullptr;
            if (mode == 2) {
                if (ftruncate(fd, (off_t)total)) { perror("ftruncate"); return 1; }
                map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (map == MAP_FAILED) { perror("mmap"); return 1; }
            }
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
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
            if (map) munmap(map, total);
        }
        const double dt = now() - t0;
        close(fd);
        printf("%s: %.2f GiB in %.3f s = %.2f GB/s\n", mode == 0 ? "write, 1 thread" : mode == 1 ? "pwrite, T threads" : "mmap + memcpy, T threads",
               total / 1073741824.0, dt, total / dt / 1e9);
        fflush(stdout);
    }
    unlink(path);
    return 0;
}
