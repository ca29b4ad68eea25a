// Stand-alone timing probe of the eight-points-per-thread row stage (csrc/fft_rowqe8.hpp): launches the kernel on synthetic leg planes,
// times it with HIP events and, with -DSTAMPS, records s_memtime at the phase boundaries of every workgroup's first lane (a timeline of
// one row pair).  Build on the GPU box (tools/r05_probe.sh); not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef STAMPS
#define NSTAMP 40
__device__ unsigned long long g_stamps[8192 * NSTAMP];
__device__ __forceinline__ void rq8_stamp(int i) {
    if (threadIdx.x == 0 && blockIdx.x < 8192 && i < NSTAMP) g_stamps[blockIdx.x * NSTAMP + i] = __builtin_readcyclecounter();
}
#define RQ8_STAMP(i) rq8_stamp(i)
#endif
#include "fft_launch.hpp"
#include "fft_plan.hpp"
#include "fft_rowqe8.hpp"
using namespace oa;
#ifndef PREC
#define PREC float
#endif
#ifndef GA
#define GA 3
#endif
#ifndef WAVES
#define WAVES 4
#endif
#ifndef LAYQ
#define LAYQ 2
#endif
typedef PREC T;
constexpr int NPQ = 1;
#define STAGED 0
__global__ __launch_bounds__(64 * GA * NPQ, WAVES) void probe_kernel(RowQeArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_qe8_body<T, GA, 1, LAYQ, false>(c, a);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int my = 2048, nx = 8192, M = GA * 512, win = 380, wout = 664;
    const int nmaps = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 50;
    const long pl = 384, pk = 672;
    const size_t legn = (size_t)my * pl, prodn = (size_t)my * pk;
    cx<T>*legs, *prod, *tw, *twm;
    CK(hipMalloc(&legs, 3 * nmaps * legn * sizeof(cx<T>)));
    CK(hipMalloc(&prod, 2 * nmaps * prodn * sizeof(cx<T>)));
    std::vector<cx<T>> h(3 * nmaps * legn);
    srand(1);
    for (auto& v : h) { v.x = (T)(rand() / (double)RAND_MAX - 0.5); v.y = (T)(rand() / (double)RAND_MAX - 0.5); }
    CK(hipMemcpy(legs, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    cx<T>* legs2; CK(hipMalloc(&legs2, 3 * nmaps * legn * sizeof(cx<T>))); CK(hipMemcpy(legs2, legs, 3 * nmaps * legn * sizeof(cx<T>), hipMemcpyDeviceToDevice));
    auto t1 = make_twiddles<T>(nx);
    auto t3 = rq8_make_consts<T>(GA);
    CK(hipMalloc(&tw, t1.size() * sizeof(cx<T>))); CK(hipMemcpy(tw, t1.data(), t1.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    CK(hipMalloc(&twm, t3.size() * sizeof(cx<T>))); CK(hipMemcpy(twm, t3.data(), t3.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    RowQeArgs<T> a{};
    a.gx = legs; a.gy = legs + legn; a.h = legs + 2 * legn; a.px = prod; a.py = prod + prodn;
    a.pitch = pl; a.opitch = pk; a.logL = ilog2(M); a.NT = 64 * GA; a.rowStride = M; a.tw = tw; a.logTw = ilog2(nx); a.scale = (T)1e-3;
    a.win = win; a.wout = wout; a.lr = LAYQ; a.nrows = my; a.rq8c = twm;
    if (nmaps > 1) { a.npairs = my / 2; a.in_moff = 3 * legn; a.h_moff = 3 * legn; a.out_moff = 2 * prodn; }
    const size_t smem = rq8_lds_bytes<T, GA, false>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = my / 2 * nmaps / NPQ;
    // a big unrelated buffer written between repetitions: the leg planes are then read from HBM / MALL as inside the step
    char* junk; const size_t jb = (size_t)600 << 20;
    CK(hipMalloc(&junk, jb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
#ifdef FLUSH
        CK(hipMemsetAsync(junk, r, jb, 0));
#else
        CK(hipMemcpyAsync(legs, legs2, 3 * nmaps * legn * sizeof(cx<T>), hipMemcpyDeviceToDevice, 0));   // the producer kernel's writes: the planes sit in the infinity cache as inside the step
#endif
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(64 * GA * NPQ), smem, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("rq8 probe %s A=%d waves=%d lay=%d staged=%d maps=%d lds=%zu : median %.1f us  min %.1f us\n", sizeof(T) == 4 ? "f32" : "f64", GA, WAVES, LAYQ, STAGED, nmaps, smem,
           ts[ts.size() / 2], ts[0]);
#ifdef STAMPS
    std::vector<unsigned long long> st((size_t)8192 * NSTAMP);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * sizeof(unsigned long long)));
    const int ng = std::min(grid, 8192);
    unsigned long long t0 = ~0ull, tend = 0;
    int last = 0;
    for (int g = 0; g < ng; ++g) { t0 = std::min(t0, st[(size_t)g * NSTAMP]); for (int i = 0; i < NSTAMP; ++i) if (st[(size_t)g * NSTAMP + i]) { tend = std::max(tend, st[(size_t)g * NSTAMP + i]); last = std::max(last, i); } }
    printf("stamps: kernel span (first start -> last stamp) %llu ticks; per-phase mean ticks over %d workgroups (start offset, then deltas):\n", tend - t0, ng);
    double so = 0; for (int g = 0; g < ng; ++g) so += (double)(st[(size_t)g * NSTAMP] - t0); printf("  start offset mean %.0f\n", so / ng);
    for (int i = 1; i <= last; ++i) {
        double d = 0, mx = 0; for (int g = 0; g < ng; ++g) { const double v = (double)(st[(size_t)g * NSTAMP + i] - st[(size_t)g * NSTAMP + i - 1]); d += v; mx = std::max(mx, v); }
        printf("  phase %2d: mean %8.0f  max %8.0f\n", i, d / ng, mx);
    }
    double tot = 0; for (int g = 0; g < ng; ++g) tot += (double)(st[(size_t)g * NSTAMP + last] - st[(size_t)g * NSTAMP]); printf("  workgroup lifetime mean %.0f ticks\n", tot / ng);
#endif
    return 0;
}
