// proc_utils.h -- variant selector and small helpers.
// Mirrors /root/reference/vplib/src/proc_utils.h:7-34 (Types values are the CLI's -t argument).
#ifndef VPLIB_PROC_UTILS_H
#define VPLIB_PROC_UTILS_H

#include <cstdint>
#include <string>

enum class Types { SEQUENTIAL, NAIVE, TILED, OPENMP };

// min(next power of two >= n, max) -- the reference's launch-size rule (proc_utils.h:11-16).
// The HIP kernels pick their own wave64-friendly workgroup sizes; kept for source compatibility.
inline unsigned long int NextPow2(const unsigned long int n, const int max)
{
    unsigned long int p = 1;
    while (p < n && p < (unsigned long int)max) p <<= 1;
    return p;
}

inline std::string GetFilename(const std::string& path)
{
    const size_t pos = path.find_last_of('/');
    return pos == std::string::npos ? path : path.substr(pos + 1);
}

inline std::string GetTypesString(const Types type)
{
    switch (type) {
        case Types::SEQUENTIAL: return "sequential";
        case Types::NAIVE:      return "naive";
        case Types::TILED:      return "tiled";
        case Types::OPENMP:     return "openmp";
    }
    return "Unknown";
}

#endif
