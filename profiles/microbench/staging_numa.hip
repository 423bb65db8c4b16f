// staging_numa.hip -- where does the staging copy (caller's pageable pages -> pinned buffer) run fastest?
// T threads pinned to the CPUs of one NUMA node copy 32 MiB chunks out of a 1 GiB resident source into a hipHostMalloc'ed
// buffer; for every node of the box, and unpinned.  Prints the GPU's NUMA node (sysfs) and one JSON line per case.
// Build: hipcc -O2 --offload-arch=gfx950 staging_numa.hip -o staging_numa -lpthread
#include <hip/hip_runtime.h>
#include <sched.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<int> cpulist(const std::string &s)
{
    std::vector<int> out;
    std::stringstream ss(s);
    std::string part;
    while (std::getline(ss, part, ',')) {
        int a, b;
        if (sscanf(part.c_str(), "%d-%d", &a, &b) == 2) for (int k = a; k <= b; ++k) out.push_back(k);
        else if (sscanf(part.c_str(), "%d", &a) == 1) out.push_back(a);
    }
    return out;
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const size_t chunk = 32u << 20, total = 1u << 30;
    void *pin = nullptr;
    if (hipHostMalloc(&pin, chunk, hipHostMallocDefault) != hipSuccess) return 1;
    std::vector<char> src(total);
    memset(src.data(), 7, total);
    memset(pin, 1, chunk);
    {   // the GPU's NUMA node
        char bus[64] = {0};
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, 0);
        for (char *p = bus; *p; ++p) *p = (char)tolower(*p);
        std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
        std::string v;
        std::getline(f, v);
        printf("{\"gpu_pci\": \"%s\", \"gpu_numa_node\": \"%s\"}\n", bus, v.c_str());
    }
    std::vector<std::vector<int>> nodes;
    for (int n = 0; n < 16; ++n) {
        std::ifstream f("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist");
        if (!f) break;
        std::string v;
        std::getline(f, v);
        nodes.push_back(cpulist(v));
        printf("{\"node\": %d, \"cpus\": \"%s\"}\n", n, v.c_str());
    }
    auto run = [&](const std::vector<int> *cpus) -> double {
        const double t0 = now();
        for (size_t off = 0; off < total; off += chunk) {
            std::vector<std::thread> th;
            const size_t per = chunk / T;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    if (cpus && !cpus->empty()) {
                        cpu_set_t set;
                        CPU_ZERO(&set);
                        for (int c : *cpus) CPU_SET(c, &set);
                        (void)sched_setaffinity(0, sizeof set, &set);
                    }
                    memcpy((char *)pin + t * per, src.data() + off + t * per, per);
                });
            for (auto &x : th) x.join();
        }
        return now() - t0;
    };
    run(nullptr);
    double best = 1e9;
    for (int r = 0; r < 3; ++r) best = std::min(best, run(nullptr));
    printf("{\"threads\": %d, \"pinned_to\": \"nothing\", \"GBs\": %.1f}\n", T, total / best / 1e9);
    for (size_t n = 0; n < nodes.size(); ++n) {
        best = 1e9;
        for (int r = 0; r < 3; ++r) best = std::min(best, run(&nodes[n]));
        printf("{\"threads\": %d, \"pinned_to\": \"node %zu\", \"GBs\": %.1f}\n", T, n, total / best / 1e9);
    }
    return 0;
}
