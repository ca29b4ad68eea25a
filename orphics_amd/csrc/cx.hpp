// Minimal complex arithmetic shared by device kernels and the CPU emulator.
#pragma once
#include <cstdint>
#include <cstdlib>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OA_HD __host__ __device__ __forceinline__
#define OA_D __device__ __forceinline__
#else
#define OA_HD inline
#define OA_D inline
#endif

namespace oa {

// Kernel-selection A/B switches (OA_NO_RSPLIT, OA_RS4096_PF, ...) exist only in EXPERIMENT builds (tools/build_variant.sh passes
// -DOA_EXPERIMENTS): the product library never reads the environment, so a stray variable cannot select an untested path.
// Alternate paths that tests compare against each other are explicit plan options (oa_plan_set_option).
inline const char* exp_env(const char* name) {
#ifdef OA_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

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

// a + (-i) b  and  a + (+i) b : the radix-4 building blocks of every butterfly
template <typename T> OA_HD cx<T> add_mi(cx<T> a, cx<T> b) { return mk<T>(a.x + b.y, a.y - b.x); }
template <typename T> OA_HD cx<T> add_pi(cx<T> a, cx<T> b) { return mk<T>(a.x - b.y, a.y + b.x); }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(OA_NO_PK_ASM)
// gfx950 packed-f32 forms with operand swizzles / sign modifiers (op_sel, neg_lo/neg_hi) that hipcc
// does not select on its own (it emits v_xor + v_mov + v_pk_* instead): 1 instruction for a +- i b,
// 2 for a complex product.  Plain VALU -> hardware interlocked, no manual wait states needed.
typedef float oa_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cx<float> add_mi(cx<float> a, cx<float> b) {
    oa_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
        : "=v"(r) : "v"(__builtin_bit_cast(oa_f2, a)), "v"(__builtin_bit_cast(oa_f2, b)));
    return __builtin_bit_cast(cx<float>, r);
}
__device__ __forceinline__ cx<float> add_pi(cx<float> a, cx<float> b) {
    oa_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
        : "=v"(r) : "v"(__builtin_bit_cast(oa_f2, a)), "v"(__builtin_bit_cast(oa_f2, b)));
    return __builtin_bit_cast(cx<float>, r);
}
__device__ __forceinline__ cx<float> operator*(cx<float> a, cx<float> b) {
    oa_f2 t, r;
    const oa_f2 av = __builtin_bit_cast(oa_f2, a), bv = __builtin_bit_cast(oa_f2, b);
#ifdef OA_PK_SPLIT_ASM
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));          // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                  // (-a.y b.y, a.y b.x) + t
        : "=v"(r) : "v"(av), "v"(bv), "v"(t));
#else
    // ONE asm statement for the dependent pair: the compiler pads every inline-asm boundary with an s_nop (it cannot
    // see what the instruction is); the VALU -> VALU read-after-write inside is interlocked by the hardware
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                     // (a.x b.x, a.x b.y)
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                  // (-a.y b.y, a.y b.x) + t
        : "=&v"(t), "=v"(r) : "v"(av), "v"(bv));
#endif
    return __builtin_bit_cast(cx<float>, r);
}
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(OA_NO_F64_FMA)
// complex128 product with fused multiply-adds: 2 v_mul_f64 + 2 v_fma_f64 instead of 4 + 2 (the build runs with
// -ffp-contract=off for the bit-exact digitize kernels, so the compiler does not contract on its own; the f64 FFT and
// row-stage kernels are VALU-bound and a third of their arithmetic is complex products).  One rounding less per component:
// results move in the last bit, well inside every f64 tolerance (1e-9 on bandpowers, 1e-12 on transforms).
__device__ __forceinline__ cx<double> operator*(cx<double> a, cx<double> b) {
    return mk<double>(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
#endif

// load through a pointer the compiler cannot prove global (read from a device table, selected at run time): a generic pointer becomes
// a flat_load, which also counts in lgkmcnt -- every LDS-only barrier (GpuCtx::sync) and wave-level wait would then wait for it.
// All such pointers in these kernels are device-memory planes.
#if defined(__HIP_DEVICE_COMPILE__)
template <class U> __device__ __forceinline__ U ldg(const U* p) {
    typedef const __attribute__((address_space(1))) U* gp;
    return *(gp)(unsigned long long)p;
}
#else
template <class U> inline U ldg(const U* p) { return *p; }
#endif

// Memory order of the last-workgroup TICKET of the fused binning tails (fft_divbin.hpp, bin.hip): acquire-release at agent scope.
// The partial sums are stored with agent-scope atomic stores (write-through), every storing wave waits for vmcnt(0), the workgroup
// meets at a barrier and ONE lane adds to the ticket: the release of that add (cumulative over the barrier) publishes the workgroup's
// partials, the acquire of the add that came last -- followed by the barrier the other lanes of that workgroup join -- lets it read
// everybody's (with agent-scope atomic loads).  Rounds 2-4 ran the add RELAXED, which is the write-through hand-over
// /opt/skills/guides/MI355X_MICROARCH.md lists as valid by measurement ("not an architectural guarantee"); the fence costs 1.0 us per
// float64 and 2.6 us per float32 reconstruction (profiles/r05_ticket_order.txt).  -DOA_TICKET_RELAXED: the old form, for A/B.
#ifdef OA_TICKET_RELAXED
#define OA_TICKET_ORDER __ATOMIC_RELAXED
#else
#define OA_TICKET_ORDER __ATOMIC_ACQ_REL
#endif

OA_HD int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace oa
