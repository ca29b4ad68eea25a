// Shared launch plumbing of the FFT translation units (fft.hip, fft_legs.hip).
#pragma once
#include <mutex>
#include "common.hpp"
#include "fft_plan.hpp"

namespace oa {

struct GpuCtx {
    char* sm;
    OA_D int tid() const { return threadIdx.x; }
    OA_D int nthreads() const { return blockDim.x; }
    OA_D int bid_x() const { return blockIdx.x; }
    OA_D int bid_y() const { return blockIdx.y; }
    OA_D int bid_z() const { return blockIdx.z; }
    OA_D int grid_x() const { return gridDim.x; }
    // Workgroup barrier for LDS traffic only.  __syncthreads() also drains vmcnt: every outstanding GLOBAL load and
    // store of the wave would have to land before the barrier -- the stores of one fused pipeline stage would stall
    // the next stage, and loads prefetched for the next row could not stay in flight across the FFT stages.  These
    // kernels exchange data between threads through LDS only (global data is never re-read inside a launch), so the
    // barrier waits for this wave's LDS operations (lgkmcnt) and nothing else.
#ifdef OA_FULL_BARRIER
    OA_D void sync() const { __syncthreads(); }
#else
    OA_D void sync() const { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
    // Exchange between the lanes of ONE wave through LDS: no s_barrier -- the wave waits until its own LDS operations have
    // completed (lgkmcnt(0): the writes of every lane are in the array before any lane's later read is issued) and the compiler
    // keeps the order.  -DOA_WSYNC_NOWAIT: compiler fence only (the LDS executes a wave's operations in issue order; A/B)
#ifdef OA_WSYNC_NOWAIT
    OA_D void wsync() const { asm volatile("" ::: "memory"); }
#else
    OA_D void wsync() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
    OA_D void* smem() const { return sm; }
};

extern __shared__ __attribute__((aligned(16))) char oa_dyn_smem[];

#ifndef OA_WAVES_PER_EU
#define OA_WAVES_PER_EU 4
#endif
constexpr size_t LDS_MAX = 160 * 1024;

// workgroup size is a function of the transform length: NT = L*C/16, L*C = 4096 up to L = 4096
template <class SEQ>
constexpr int seq_logl() {
    return Log2x<SEQ::r0>::v + Log2x<SEQ::r1>::v + Log2x<SEQ::r2>::v + Log2x<SEQ::r3>::v;
}
template <class SEQ> constexpr int row_maxnt() { return seq_logl<SEQ>() <= 12 ? 256 : (seq_logl<SEQ>() == 13 ? 512 : 1024); }
// column workgroups: NT = L * 2^COL_LOGC / 16 (256 for the short sub-lengths of a 32-column tile)
template <class SEQ> constexpr int col_maxnt() {
    constexpr int nt = (1 << (seq_logl<SEQ>() + COL_LOGC)) / EPT;
    return nt <= 256 ? 256 : (nt >= 1024 ? 1024 : nt);
}
// float kernels fit 128 VGPRs (4 waves/SIMD); double needs the 256-register budget
template <typename T> constexpr int waves_per_eu() { return sizeof(T) == 8 ? 2 : OA_WAVES_PER_EU; }


#ifndef OA_FUSED_COL_WAVES
#define OA_FUSED_COL_WAVES 3
#endif
template <typename T> constexpr int fused_col_waves_per_eu() { return sizeof(T) == 8 ? 2 : OA_FUSED_COL_WAVES; }

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, raised once per (kernel, device) and only when a launch needs more than
// what was set before: the runtime call costs microseconds -- per launch it was a visible part of the host time of the small
// (4096^2) configurations
inline hipError_t ensure_dyn_lds(const void* kern, size_t smem) {
    if (smem <= 48 * 1024) return hipSuccess;
    struct Ent { const void* k; int dev; size_t smem; };
    static Ent tab[256];
    static int n = 0;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n; ++i)
        if (tab[i].k == kern && tab[i].dev == dev) {
            if (tab[i].smem >= smem) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e == hipSuccess) tab[i].smem = smem;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess && n < 256) tab[n++] = Ent{kern, dev, smem};
    return e;
}

template <class K, class A>
inline void launch_go(int& rc, hipStream_t st, K kern, dim3 grid, int nt, size_t smem, const A& a) {
    if (rc) return;
    if (smem > LDS_MAX || nt > 1024 || nt < 1) {
        rc = fail("fft: transform size exceeds the LDS / workgroup budget for this dtype");
        return;
    }
    {
        const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) { rc = fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return; }
    }
    hipLaunchKernelGGL(kern, grid, dim3(nt), smem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(std::string("fft launch: ") + hipGetErrorString(e));
}

// defined in fft_legs.hip (built WITHOUT the packed-asm complex operators: the leg kernel's many
// independent global loads schedule better around compiler-visible arithmetic)
template <typename T>
int launch_col_legs(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColLegsArgs<T>& a);
template <typename T>
int launch_col_legs_sp(hipStream_t st, int gx, int nt, size_t smem, int logL, const ColLegsArgs<T>& a);
template <typename T>
int launch_col_fwdlegs(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsArgs<T>& a);
template <typename T>
int launch_col_fwdlegs_cg(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsCgArgs<T>& a, int gz = 1);

}  // namespace oa
