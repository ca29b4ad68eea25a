// Stand-alone timing probe of the R = 2 single-pass column stage (csrc/fft_fband.hpp, col_fband_body<T, Seq<16,16,16>, 1, LC>) at the
// wide band's geometry (8192 rows on the 4096-row grid, 1138 columns): HIP-event time and, with -DSTAMPS, the cycle counter at the
// phase boundaries of every workgroup's first lane.  Build on the GPU box (tools/r05_fband_probe.sh); not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef STAMPS
#define NSTAMP 8
__device__ unsigned long long g_stamps[4096 * NSTAMP];
__device__ __forceinline__ void fb_stamp(int i) {
    const int b = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0 && b < 4096 && i < NSTAMP) g_stamps[b * NSTAMP + i] = __builtin_readcyclecounter();
}
#define FB_STAMP(i) fb_stamp(i)
#endif
#include "fft_launch.hpp"
#include "fft_plan.hpp"
#include "fft_fband.hpp"
using namespace oa;
#ifndef PREC
#define PREC double
#endif
typedef PREC T;
constexpr int LC = sizeof(T) == 8 ? 1 : 2, NTH = sizeof(T) == 8 ? 512 : 1024;
__global__ __launch_bounds__(NTH, (sizeof(T) == 8 ? 2 : 4)) void probe_kernel(ColFBandArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_fband_body<T, Seq<16, 16, 16>, 1, LC>(c, a);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int ny = 8192, my = 4096, nx = 8192, w = 1138, rband = argc > 2 ? atoi(argv[2]) : 1139;
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const long pitch = 1152, kp = nx / 2 + 16;
    cx<T>*Y, *legs, *tw; T *FG, *FH, *lx, *ly;
    CK(hipMalloc(&Y, 2 * (size_t)my * pitch * sizeof(cx<T>)));
    CK(hipMalloc(&legs, 3 * (size_t)my * pitch * sizeof(cx<T>)));
    CK(hipMalloc(&FG, (size_t)ny * kp * sizeof(T))); CK(hipMalloc(&FH, (size_t)ny * kp * sizeof(T)));
    CK(hipMalloc(&lx, nx * sizeof(T))); CK(hipMalloc(&ly, ny * sizeof(T)));
    CK(hipMemset(Y, 0, 2 * (size_t)my * pitch * sizeof(cx<T>)));
    CK(hipMemset(FG, 0, (size_t)ny * kp * sizeof(T))); CK(hipMemset(FH, 0, (size_t)ny * kp * sizeof(T)));
    CK(hipMemset(lx, 0, nx * sizeof(T))); CK(hipMemset(ly, 0, ny * sizeof(T)));
    auto t1 = make_twiddles<T>(my);
    CK(hipMalloc(&tw, t1.size() * sizeof(cx<T>))); CK(hipMemcpy(tw, t1.data(), t1.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    ColFBandArgs<T> a{};
    a.in = Y; a.kplane = (long)my * pitch; a.pitch = pitch; a.FG = FG; a.FH = FH; a.fpitch = kp; a.lxd = lx; a.lyd = ly;
    a.gx = legs; a.gy = legs + (size_t)my * pitch; a.h = legs + 2 * (size_t)my * pitch; a.opitch = pitch; a.width = w; a.tw = tw; a.ny_full = ny; a.rband = rband;
    const int lt = 12 + LC, Cs = 1 << LC;
    const size_t smem = ((size_t)(1 << lt) + tw_lds_size(12) + tw_lds_size(11)) * sizeof(cx<T>);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const dim3 grid((w + Cs - 1) / Cs, 2, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
        CK(hipMemsetAsync(Y, 0, 2 * (size_t)my * pitch * sizeof(cx<T>), 0));       // the producer's writes: Y sits in the infinity cache as inside the step
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe_kernel, grid, dim3(NTH), smem, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("fband probe %s rband=%d grid %d x 2 lds=%zu : median %.1f us  min %.1f us\n", sizeof(T) == 4 ? "f32" : "f64", rband, grid.x, smem, ts[ts.size() / 2], ts[0]);
#ifdef STAMPS
    std::vector<unsigned long long> st((size_t)4096 * NSTAMP);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * sizeof(unsigned long long)));
    const int ng = std::min<int>(grid.x * 2, 4096);
    unsigned long long t0 = ~0ull, tend = 0;
    for (int g = 0; g < ng; ++g) { t0 = std::min(t0, st[(size_t)g * NSTAMP]); tend = std::max(tend, st[(size_t)g * NSTAMP + 7]); }
    printf("stamps (100 MHz ticks): kernel span %llu; mean / max per phase over %d workgroups\n", tend - t0, ng);
    const char* names[8] = {"", "tables + barrier", "filter reads issued", "tile load + forward", "filters applied, leg buffers written", "inverse round 1 (H, Gx) + stores issued", "Gy to buffer 0", "inverse round 2 (Gy)"};
    for (int i = 1; i < 8; ++i) {
        double d = 0, mx = 0; for (int g = 0; g < ng; ++g) { const double v = (double)(st[(size_t)g * NSTAMP + i] - st[(size_t)g * NSTAMP + i - 1]); d += v; mx = std::max(mx, v); }
        printf("  phase %d %-42s mean %7.0f  max %7.0f\n", i, names[i], d / ng, mx);
    }
    double tot = 0; for (int g = 0; g < ng; ++g) tot += (double)(st[(size_t)g * NSTAMP + 7] - st[(size_t)g * NSTAMP]); printf("  workgroup lifetime mean %.0f ticks\n", tot / ng);
    // start offsets: how the rounds line up
    std::vector<double> so; for (int g = 0; g < ng; ++g) so.push_back((double)(st[(size_t)g * NSTAMP] - t0)); std::sort(so.begin(), so.end());
    printf("  start offsets: p10 %.0f p50 %.0f p90 %.0f max %.0f\n", so[ng / 10], so[ng / 2], so[ng * 9 / 10], so[ng - 1]);
#endif
    return 0;
}
