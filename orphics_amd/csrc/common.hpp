// Shared host-side helpers for the C-ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/orphics_amd.h"
#include "cx.hpp"

namespace oa {

std::string& last_error_ref();

inline int fail(const std::string& msg) {
    last_error_ref() = msg;
    return 1;
}

#define OA_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return ::oa::fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                              ":" + std::to_string(__LINE__) + ")");                              \
    } while (0)

#define OA_REQUIRE(cond, msg)                         \
    do {                                              \
        if (!(cond)) return ::oa::fail(std::string(msg)); \
    } while (0)

#define OA_LAUNCH_CHECK() OA_HIP(hipGetLastError())

template <typename T, int N>
struct alignas(sizeof(T) * N > 16 ? 16 : sizeof(T) * N) Arr {
    T v[N];
};

inline int flat_grid(long units, int block = 256, int cap = 8192) {
    long g = (units + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace oa

struct oa_plan {
    int ny, nx, logNy, logNx, dtype, device;
    long kp;
    void* tw_x;      // cx<T>[nx]
    void* tw_y;      // cx<T>[ny]
    void* scratch;   // FFT scratch plane(s)
    size_t scratch_bytes;
    void* ly;        // T[ny]
    void* lx;        // T[nx]
    void* lyd;       // T[ny]  derivative axis: Nyquist entry zeroed
    void* lxd;       // T[nx]
    double* ly64;    // double[ny]
    double* lx64;    // double[nx]
    bool have_laxes;
};

namespace oa {
int plan_ensure_scratch(oa_plan* p, size_t bytes);
}
