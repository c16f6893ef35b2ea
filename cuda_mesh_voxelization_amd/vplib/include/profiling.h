// profiling.h -- scope timer printing "[label]: <ms> ms" at scope exit; the line grammar is the
// measurement interface of the reference's benchmark script (/root/reference/vplib/src/profiling.h:8-33,
// scripts/benchmarks.py:74-95).  Enabled with -DPROFILING=1.
#ifndef VPLIB_PROFILING_H
#define VPLIB_PROFILING_H

#include <chrono>
#include <cstdio>
#include <string>

class Profiling {
    std::string mLabel;
    std::chrono::steady_clock::time_point mStart;

public:
    explicit Profiling(std::string label = "") : mLabel(std::move(label)), mStart(std::chrono::steady_clock::now()) {}
    Profiling(const Profiling&) = delete;
    Profiling& operator=(const Profiling&) = delete;

    ~Profiling()
    {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - mStart).count();
        if (!mLabel.empty()) std::printf("[%s]: %f ms\n", mLabel.c_str(), ms);
        else std::printf("[PROFILING] Delta Time: %f ms\n", ms);
    }
};

#define VPLIB_CAT2(a, b) a##b
#define VPLIB_CAT(a, b) VPLIB_CAT2(a, b)
#if PROFILING
#define PROFILING_SCOPE(msg) Profiling VPLIB_CAT(vplibTimer, __LINE__)(msg)
#else
#define PROFILING_SCOPE(msg)
#endif

#endif
