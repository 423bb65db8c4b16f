// crp_roctx.h -- optional roctx ranges around the stages of the path (pack + H2D, scan, D2H, format + write,
// gatherv), for `rocprofv3 --marker-trace`.  Off unless the environment variable
// CROPSR_ROCTX is set: the marker library (librocprofiler-sdk-roctx.so, the one rocprofv3 reads; libroctx64.so
// as a fallback) is dlopen()ed on first use, so the library has no dependency on it.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace crp {

struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (!std::getenv("CROPSR_ROCTX")) return;
        const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char *n : names) {
            void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};

inline const Roctx &roctx()
{
    static const Roctx r;
    return r;
}

// RAII range: crp::Range r("scan_score");
struct Range {
    bool on;
    explicit Range(const char *name) : on(roctx().push != nullptr)
    {
        if (on) roctx().push(name);
    }
    ~Range()
    {
        if (on) roctx().pop();
    }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

}  // namespace crp
