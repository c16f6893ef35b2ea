// debug_utils.h -- logging macros and the print-and-exit assertions of the reference
// (/root/reference/vplib/src/debug_utils.h:24-64).  gpuAssert takes a libvphip status code
// (include/vphip.h) instead of a cudaError_t.
#ifndef VPLIB_DEBUG_UTILS_H
#define VPLIB_DEBUG_UTILS_H

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <string>

#include "vphip.h"

inline const char* getCurrentTimestamp()
{
    static thread_local char buf[64];
    const std::time_t t = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    std::tm tmv;
    localtime_r(&t, &tmv);
    std::strftime(buf, sizeof(buf), "%Y-%m-%d %X", &tmv);
    return buf;
}

#if LOGGING
#define LOG_INTERNAL(level, format, ...) \
    std::printf("[%s] [%s] [%s:%d] " format "\n", getCurrentTimestamp(), level, __FILE__, __LINE__, ##__VA_ARGS__)
#else
#define LOG_INTERNAL(level, format, ...) ((void)0)
#endif
#define LOG_ERROR(format, ...) LOG_INTERNAL("ERROR", format, ##__VA_ARGS__)
#define LOG_WARN(format, ...)  LOG_INTERNAL("WARN", format, ##__VA_ARGS__)
#define LOG_INFO(format, ...)  LOG_INTERNAL("INFO", format, ##__VA_ARGS__)
#define LOG_DEBUG(format, ...) LOG_INTERNAL("DEBUG", format, ##__VA_ARGS__)

inline void gpuAssertBase(int code, const char* file, int line)
{
    if (code != 0) {
        std::printf("[%s:%d] HIP Assert: %s", file, line, vp_last_error());
        std::exit(code);
    }
}

inline void cpuAssertBase(bool condition, const std::string& msg, const char* file, int line)
{
    if (!condition) {
        std::printf("[%s:%d] CPU Assert: %s", file, line, msg.c_str());
        std::exit(-1);
    }
}

#define gpuAssert(ans) gpuAssertBase((ans), __FILE__, __LINE__)
#define cpuAssert(ans, msg) cpuAssertBase((ans), msg, __FILE__, __LINE__)

#endif
