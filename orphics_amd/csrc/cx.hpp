// Minimal complex arithmetic shared by device kernels and the CPU emulator.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OA_HD __host__ __device__ __forceinline__
#define OA_D __device__ __forceinline__
#else
#define OA_HD inline
#define OA_D inline
#endif

namespace oa {

template <typename T>
struct alignas(2 * sizeof(T)) cx {
    T x, y;
};

template <typename T> OA_HD cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename T> OA_HD cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> OA_HD cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> OA_HD cx<T> operator*(cx<T> a, cx<T> b) {
    return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> OA_HD cx<T> operator*(cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T> OA_HD cx<T> conj(cx<T> a) { return mk<T>(a.x, -a.y); }
// multiply by -i / +i
template <typename T> OA_HD cx<T> mul_mi(cx<T> a) { return mk<T>(a.y, -a.x); }
template <typename T> OA_HD cx<T> mul_pi(cx<T> a) { return mk<T>(-a.y, a.x); }
// re<->im swap: IDFT(x) = swap(DFT(swap(x)))
template <typename T> OA_HD cx<T> swp(cx<T> a) { return mk<T>(a.y, a.x); }

OA_HD int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace oa
