// HIP launchers for the LDS-staged 2-D FFT passes (K1) + oa_fft_* entry points.
#include <cmath>
#include <vector>
#include "fft_launch.hpp"
#include "fft_r2c_w64.hpp"
#include "fft_r2c_rs4096.hpp"
#include "fft_divbin.hpp"

namespace oa {

template <typename T, int MODE, class SEQ>
__global__ __launch_bounds__(row_maxnt<SEQ>(), waves_per_eu<T>()) void row_fft_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_fft_body<T, MODE, SEQ>(c, a);
}

// one wave per row, 64 points per lane: a single wave per SIMD (LDS-limited), so the whole register file is its own
#ifndef OA_W64_OCC
#define OA_W64_OCC 1
#endif
__global__ __launch_bounds__(64, OA_W64_OCC) void row_r2c_w64_kernel(RowW64Args a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_w64_body(c, a);
}

template <typename T, class SEQ, int LR, bool PF>
__global__ __launch_bounds__(row_maxnt<SEQ>(), (sizeof(T) == 8 || PF ? 2 : 3)) void row_r2c_rsplit_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_rsplit_body<T, SEQ, LR, PF>(c, a);
}

// 8192-point rows, <= 512 columns kept: one cross-wave exchange per row (fft_r2c_rs4096.hpp); two workgroups per CU
#ifndef OA_RS4096_F32_OCC
#define OA_RS4096_F32_OCC 3
#endif
template <typename T, bool PF>
__global__ __launch_bounds__(RS4096_NT, (sizeof(T) == 8 ? 2 : OA_RS4096_F32_OCC)) void row_r2c_rs4096_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_rs4096_body<T, 2, PF>(c, a);
}
// 4096-point rows (4096^2 maps), <= 256 columns kept: the same body on 128 threads per row
// the wide band (<= 1280 kept columns), radix-2 column butterfly: 8192^2 maps on the 4096-row column grid
template <typename T, bool PF>
__global__ __launch_bounds__(RS4096_NT, (sizeof(T) == 8 ? 2 : OA_RS4096_F32_OCC)) void row_r2c_rs4096w_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_rs_body<T, 12, 1, PF, 5>(c, a);
}
template <typename T, bool PF>
__global__ __launch_bounds__(128, (sizeof(T) == 8 ? 2 : OA_RS4096_F32_OCC)) void row_r2c_rs2048_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_rs_body<T, 11, 2, PF>(c, a);
}
// 16384-point rows (16384^2 maps: BASELINE config 5), <= 512 columns kept, R = 8 column butterfly on top: the same body on 512
// threads per row (float64: 150 KB of LDS, one workgroup per CU -- the prefetch order is what overlaps its loads with its
// arithmetic; float: 75 KB, two per CU when the registers allow)
#ifndef OA_RS8192_F32_OCC
#define OA_RS8192_F32_OCC 2
#endif
template <typename T, bool PF>
__global__ __launch_bounds__(512, (sizeof(T) == 8 ? 2 : OA_RS8192_F32_OCC)) void row_r2c_rs8192_kernel(RowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_rs_body<T, 13, 3, PF>(c, a);
}

// single-pass column stage of the R-split path: [My][C] tile, all threads forward, R groups inverse (fft_fband.hpp)
template <typename T, class SEQF, int LR, int LOGC>
__global__ __launch_bounds__((sizeof(T) == 8 ? 512 : 1024)) void col_fband_pack_kernel(ColFBandArgs<T> a, cx<T>* out) {
    GpuCtx c{nullptr};
    col_fband_pack_body<T, SEQF, LR, LOGC>(c, a, out);
}
template <typename T, class SEQF, int LR, int LOGC>
__global__ __launch_bounds__((sizeof(T) == 8 ? 512 : 1024), (sizeof(T) == 8 ? 2 : 4)) void col_fband_kernel(ColFBandArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_fband_body<T, SEQF, LR, LOGC>(c, a);
}

// 16384-point rows: two waves per row (even / odd packed samples), one workgroup of 128 threads per row in flight
__global__ __launch_bounds__(128, 1) void row_r2c_w64x2_kernel(RowW64Args a) {
    GpuCtx c{oa_dyn_smem};
    row_r2c_w64x2_body(c, a);
}

#ifndef OA_QE_WAVES_PER_EU
#define OA_QE_WAVES_PER_EU 3
#endif
// rows of >= 16384 points need 512-thread workgroups (2 waves/SIMD each): budget the registers for exactly that
template <typename T, class SEQ> constexpr int qe_waves_per_eu() {
    return sizeof(T) == 8 ? 1 : (row_maxnt<SEQ>() > 256 ? 2 : OA_QE_WAVES_PER_EU);
}

template <typename T, class SEQ, int NZ>
__global__ __launch_bounds__(row_maxnt<SEQ>(), (qe_waves_per_eu<T, SEQ>())) void row_qe_kernel(RowQeArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
#ifdef OA_QE_INPLACE   // in-place DIF/DIT variant: 18 instead of 30 barriers per row, same speed on gfx950 (measured)
    row_qe_body_inplace<T, SEQ>(c, a);
#else
    row_qe_body<T, SEQ, NZ>(c, a);
#endif
}


template <class SEQ> constexpr int pair_nt() { return (1 << seq_logl<SEQ>()) / EPT; }
// register budget of the two-rows-per-transform row stage: grids up to 2048 points keep h, the leg and the
// butterfly temporaries in registers only at 2 waves/SIMD (at 3 they spill 17-45 VGPRs: 117 us vs 87 us at 8192^2,
// profiles/r02e_rowqe_variants.txt); the 4096-point grid fits 3 waves/SIMD without spilling
// float64 keeps one wave per SIMD: -DOA_PAIR_F64_WAVES=2 (two waves for the instances that then spill at most ~30 registers) makes the
// 2048-point NZ = 2 kernel 10 % faster alone (54.9 -> 49.3 us) and leaves every line of the bench where it was
// (gpurun_out r04q: headline 5231 / 5118, kappa_out 5402 / 5534, MV 1210 / 1210): not adopted
#ifndef OA_PAIR_F64_WAVES
#define OA_PAIR_F64_WAVES 1
#endif
template <class SEQ, int NZ, int LR> constexpr int pair_f64_waves() {
    constexpr int ll = seq_logl<SEQ>();
    if (OA_PAIR_F64_WAVES < 2 || SEQ::n > 3) return 1;
    if (LR == 3) return NZ == 1 ? 2 : 1;
    if (LR == 2) return ll == 10 ? (NZ == 1 ? 2 : 1) : ll == 11 ? (NZ <= 2 ? 2 : 1) : (NZ <= 4 ? 2 : 1);
    return 2;
}
#ifdef OA_PAIR_WAVES_PER_EU
template <typename T, class SEQ, int NZ, int LR> constexpr int pair_waves_per_eu() { return sizeof(T) == 8 ? pair_f64_waves<SEQ, NZ, LR>() : OA_PAIR_WAVES_PER_EU; }
#else
template <typename T, class SEQ, int NZ, int LR> constexpr int pair_waves_per_eu() {
    return sizeof(T) == 8 ? pair_f64_waves<SEQ, NZ, LR>() : (seq_logl<SEQ>() == 12 ? 3 : 2);
}
#endif
template <typename T, class SEQ, int NZ, int LR = 0>
__global__ __launch_bounds__(pair_nt<SEQ>(), (pair_waves_per_eu<T, SEQ, NZ, LR>())) void row_qe_pair_kernel(RowQeArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_qe_pair_body<T, SEQ, NZ, LR>(c, a);
}
// estimator chains (oa_qe_mv): two more 16-point register sets (the running products of both legs) -- float64 at one wave per
// SIMD: 256 VGPRs + ~150 AGPRs, nothing spilled; float32 at two: 238-256 VGPRs (NZ = 4 spills 19)
template <typename T, class SEQ, int NZ>
__global__ __launch_bounds__(pair_nt<SEQ>(), (sizeof(T) == 8 ? 1 : 2)) void row_qe_chain_kernel(RowQeArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_qe_pair_body<T, SEQ, NZ, 0, true>(c, a);
}

// eight points per thread, M / 512 waves per row pair (fft_rowqe8.hpp): four waves per SIMD
#ifndef OA_RQ8_WAVES_F64
#define OA_RQ8_WAVES_F64 4
#endif
#ifndef OA_RQ8_WAVES_F32
#define OA_RQ8_WAVES_F32 4
#endif
// estimator chains carry two more register sets (the running products of both legs): three waves per SIMD in float32 (168
// registers: 12 waves = four 3-wave workgroups per CU), two in float64
template <typename T, bool CHAIN> constexpr int rq8_waves_per_eu() {
    return CHAIN ? (sizeof(T) == 8 ? 2 : 3) : (sizeof(T) == 8 ? OA_RQ8_WAVES_F64 : OA_RQ8_WAVES_F32);
}
template <typename T, int A, int NZ, int LAY, bool CHAIN>
__global__ __launch_bounds__(64 * A, (rq8_waves_per_eu<T, CHAIN>())) void row_qe8_kernel(RowQeArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    row_qe8_body<T, A, NZ, LAY, CHAIN>(c, a);
}

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), fused_col_waves_per_eu<T>()) void col_div_kernel(ColDivArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_div_body<T, SEQ>(c, a);
}

// single-pass forward column transform + divergence on short (coarse-grid) columns: 1024 threads hold a whole column tile
// (f64: half the tile width -- the same 128 KB of LDS -- and 512 threads with the 256-register budget)
template <typename T, class SEQ, int LOGC>
__global__ __launch_bounds__((sizeof(T) == 8 ? 512 : 1024), (sizeof(T) == 8 ? 2 : 4)) void col_div_sp_kernel(ColDivArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_div_body<T, SEQ, GpuCtx, LOGC>(c, a);
}

// ... with the radial histogram of |kappa|^2 and the moment update in its tail (fft_divbin.hpp)
template <typename T, class SEQ, int LOGC>
__global__ __launch_bounds__((sizeof(T) == 8 ? 512 : 1024), (sizeof(T) == 8 ? 2 : 4)) void col_div_sp_bin_kernel(ColDivArgs<T> a, DivBinFuse f) {
    GpuCtx c{oa_dyn_smem};
    col_div_body<T, SEQ, GpuCtx, LOGC, DivBinTail<T>>(c, a, DivBinTail<T>{f});
}

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), waves_per_eu<T>()) void col_deriv_kernel(ColDerivArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_deriv_body<T, SEQ>(c, a);
}

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), waves_per_eu<T>()) void col_fft_kernel(ColArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_fft_body<T, SEQ>(c, a);
}

// resident workgroups per CU of a persistent kernel, with its dynamic-LDS attribute set: asked of the runtime ONCE per (kernel,
// workgroup size, LDS bytes) -- both are driver calls of several microseconds, more than a 4096^2 launch leaves the host
static int resident_per_cu(const void* kern, int nt, size_t smem, std::string* err) {
    struct Key { const void* k; int dev, nt; size_t smem; int per_cu; };
    static Key cache[64];
    static int ncache = 0;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        for (int i = 0; i < ncache; ++i)
            if (cache[i].k == kern && cache[i].dev == dev && cache[i].nt == nt && cache[i].smem == smem) return cache[i].per_cu;
    }
    const hipError_t e = ensure_dyn_lds(kern, smem);
    if (e != hipSuccess) { *err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return -1; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nt, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    std::lock_guard<std::mutex> lock(mu);
    if (ncache < 64) cache[ncache++] = Key{kern, dev, nt, smem, per_cu};
    return per_cu;
}

struct HipLauncher {
    hipStream_t st;
    int rc = 0;

    template <class K, class A>
    void go(K kern, dim3 grid, int nt, size_t smem, const A& a) { launch_go(rc, st, kern, grid, nt, smem, a); }
    template <class K, class A>
    void launch_plain(K kern, dim3 grid, int nt, size_t smem, const A& a) {      // (dynamic-LDS attribute already set: resident_per_cu)
        if (rc) return;
        hipLaunchKernelGGL(kern, grid, dim3(nt), smem, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(std::string("fft launch: ") + hipGetErrorString(e));
    }

    template <typename T, int MODE, class S>
    void row_mode(int grid, int nt, size_t smem, const RowArgs<T>& a) {
        if (nt > row_maxnt<S>()) { if (!rc) rc = fail("fft: row workgroup size exceeds its launch bound"); return; }
        go(row_fft_kernel<T, MODE, S>, dim3(grid, a.nz > 0 ? a.nz : 1), nt, smem, a);
    }
    static int r2c_w64_mode() {
        static const int m = [] { const char* e = exp_env("OA_R2C_W64"); return e ? atoi(e) : 1; }();
        return m;
    }
    // band-limited R2C of 8192-point rows (f32): one wave per row (fft_r2c_w64.hpp)
    bool row_w64(int ny, const RowArgs<float>& a) {
        if (a.mode != ROW_R2C || !r2c_w64_mode() || rc) return false;
        const bool one = a.logL == 12 && a.wcols <= 512, two = a.logL == 13 && a.wcols <= 64 * W64X2_KEEP;
        if (!one && !two) return false;
        RowW64Args w{};
        w.in = (const cx<float>*)a.in; w.out = (cx<float>*)a.out; w.in_pitch = a.in_pitch; w.out_pitch = a.out_pitch;
        w.tw = a.tw; w.logTw = a.logTw; w.scale = a.scale; w.wcols = a.wcols; w.ny = ny;
        static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        static const int per_cu = [] { const char* e = exp_env("OA_W64_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
        w.nwg = cus * per_cu;                              // resident waves: one per SIMD (launch bound; 16.6 KB of LDS each)
        if (two) w.nwg = cus * (int)(LDS_MAX / W64X2_LDS_BYTES);
        if (w.nwg > ny) w.nwg = ny;
        if (two) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(row_r2c_w64x2_kernel),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)W64X2_LDS_BYTES);
            if (attr != hipSuccess) { rc = fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(attr)); return true; }
            hipLaunchKernelGGL(row_r2c_w64x2_kernel, dim3(w.nwg), dim3(128), W64X2_LDS_BYTES, st, w);
        } else
        hipLaunchKernelGGL(row_r2c_w64_kernel, dim3(w.nwg), dim3(64), W64_LDS_BYTES, st, w);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(std::string("fft launch: ") + hipGetErrorString(e));
        return true;
    }
    bool row_w64(int, const RowArgs<double>&) { return false; }
    template <typename T>
    void row(int grid, int nt, size_t smem, const RowArgs<T>& a) {
        if (row_w64(grid << a.logC, a)) return;
        const bool ok = dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            switch (a.mode) {
                case ROW_R2C: row_mode<T, ROW_R2C, S>(grid, nt, smem, a); break;
                case ROW_C2R: row_mode<T, ROW_C2R, S>(grid, nt, smem, a); break;
                case ROW_C2C_F: row_mode<T, ROW_C2C_F, S>(grid, nt, smem, a); break;
                case ROW_WIN: row_mode<T, ROW_WIN, S>(grid, nt, smem, a); break;
                default: row_mode<T, ROW_C2C_I, S>(grid, nt, smem, a); break;
            }
        });
        if (!ok && !rc) rc = fail("fft: unsupported row length");
    }
    void fail_rlayout() { if (!rc) rc = fail("fft: the R-layout needs the two-rows-per-transform row stage"); }
    // general R-split row pass: one workgroup per group, loads at the top of each row.  OA_RSPLIT_PF=1: persistent workgroups
    // (each walks groups bid, bid + grid, ...) that prefetch their next row -- measured no faster in float (22.9 vs 23.1 us at
    // 4096^2) and slower in float64 (43.6 vs 39.9 us: the 16 taps in flight push it past 256 registers), kept for A/B
    template <typename T, class S>
    void row_rsplit_seq(int ngroups, int nt, size_t smem, const RowArgs<T>& a) {
        static const bool nopf = [] { const char* e = exp_env("OA_RSPLIT_PF"); return !(e && atoi(e) != 0); }();
        if (nopf) { go(row_r2c_rsplit_kernel<T, S, 2, false>, dim3(ngroups), nt, smem, a); return; }
        if (rc) return;
        auto kern = row_r2c_rsplit_kernel<T, S, 2, true>;
        static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        std::string err;
        const int per_cu = resident_per_cu(reinterpret_cast<const void*>(kern), nt, smem, &err);
        if (per_cu < 0) { rc = fail(err); return; }
        int grid = cus * per_cu;
        if (grid > ngroups) grid = ngroups;
        launch_plain(kern, dim3(grid), nt, smem, a);
    }
    // 8192-point rows, <= 512 kept columns (and 4096-point rows, <= 256: 4096^2 maps, float 18.9 us against 23.5 us of the general
    // pass, float64 37.2 against 41.0): the one-cross-wave-exchange kernel (fft_r2c_rs4096.hpp), both precisions.  Measured
    // at 8192^2 (profiles/r03x_r2c_variants.txt): float64 121 us against 131 us of the general pass, float 65 us against 71 us of
    // the one-wave-per-row kernel.  The prefetch order pays in float (65 vs 68 us); in float64 only with the taps in two halves (below).
    // OA_RS4096_PF=0/1 overrides (experiment builds).  OA_NO_RS4096=1: the older kernels (A/B).
    template <typename T>
    bool row_rs4096(const RowArgs<T>& a) {
        static const bool off = exp_env("OA_NO_RS4096") != nullptr;
        static const int pfenv = [] { const char* e = exp_env("OA_RS4096_PF"); return e ? atoi(e) : -1; }();
        const bool l12 = a.lr == 2 && a.logL == 12 && a.wcols <= 512 && a.logTw >= 13, l11 = a.lr == 2 && a.logL == 11 && a.wcols <= 256 && a.logTw >= 12;
        const bool l13 = sizeof(T) == 8 && a.lr == 3 && a.logL == 13 && a.wcols <= 512 && a.logTw >= 14;      // 16384-point rows, R = 8 (float64)
        const bool l12w = a.lr == 1 && a.logL == 12 && a.wcols <= 1280 && a.logTw >= 13;                       // the wide band, R = 2
        if (rc || !(l12w || (!off && (l12 || l11 || l13)))) return false;
        // (prefetch order everywhere.  8192-point float64 rows: with all 16 taps requested right after stage 0 it cost registers and time
        //  in rounds 3-4 (126 vs 121 us); in two halves of 8 it is 110 vs 116 us, the wide-band kernel 120 vs 130 us, +1 % on the job --
        //  profiles/r05_r2c_prefetch.txt.  4096-point float64 rows: 37.2 us with the prefetch, 39.1 without)
        const bool nopf = pfenv >= 0 ? pfenv == 0 : false;
        const size_t smem = l13 ? rs_lds_bytes<T, 13>() : (l12w ? rs_lds_bytes<T, 12, 5>() : (l12 ? rs_lds_bytes<T, 12>() : rs_lds_bytes<T, 11>()));
        const int NTr = l13 ? 512 : ((l12 || l12w) ? RS4096_NT : 128);
        void (*kern)(RowArgs<T>) = l12 ? (nopf ? row_r2c_rs4096_kernel<T, false> : row_r2c_rs4096_kernel<T, true>)
                                       : (nopf ? row_r2c_rs2048_kernel<T, false> : row_r2c_rs2048_kernel<T, true>);
        // (float at 16384^2 keeps the two-waves-per-row kernel + multi-pass columns: measured 274 us / 2743 recon/s against 306 us /
        //  2698 for the float build of this body -- 194 registers leave one 512-thread workgroup per CU; profiles/r04e_16384_f32_variants.txt)
        if constexpr (sizeof(T) == 8) { if (l13) kern = nopf ? row_r2c_rs8192_kernel<T, false> : row_r2c_rs8192_kernel<T, true>; }
        if (l12w) kern = nopf ? row_r2c_rs4096w_kernel<T, false> : row_r2c_rs4096w_kernel<T, true>;
        static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        std::string err;
        const int per_cu = resident_per_cu(reinterpret_cast<const void*>(kern), NTr, smem, &err);
        if (per_cu < 0) { rc = fail(err); return true; }
        int grid = cus * per_cu;
        // (experiment builds: OA_RS4096_WGS = resident workgroups per CU x 2, e.g. 2 = one per CU, 3 = one and a half: leaves LDS and
        //  registers on every CU for the coarse-grid kernels of another stream)
        static const int halfwgs = [] { const char* e = exp_env("OA_RS4096_WGS"); return e ? atoi(e) : 0; }();
        if (halfwgs > 0 && halfwgs < 2 * per_cu) grid = cus * halfwgs / 2;
        // resident workgroups walk the groups.  OA_RS4096_PERSIST=0: one workgroup per group (A/B: so that the scheduler could place
        // workgroups of another stream's kernels as these retire -- measured 1 % slower in the two-stream job, 5096 vs 5159 /s)
        static const int persist = [] { const char* e = exp_env("OA_RS4096_PERSIST"); return e ? atoi(e) : -1; }();
        if (persist == 0) grid = a.my;
        if (grid > a.my) grid = a.my;
        launch_plain(kern, dim3(grid), NTr, smem, a);
        return true;
    }
    template <typename T>
    void row_rsplit(int grid, int nt, size_t smem, const RowArgs<T>& a) {
        if (row_rs4096(a)) return;
        bool ok = false;
        if (a.lr == 2) {
            ok = true;
            if (a.logL == 10) row_rsplit_seq<T, Seq<16, 16, 4>>(grid, nt, smem, a);
            else if (a.logL == 11) row_rsplit_seq<T, Seq<16, 16, 8>>(grid, nt, smem, a);
            else if (a.logL == 12) row_rsplit_seq<T, Seq<16, 16, 16>>(grid, nt, smem, a);
            else if (a.logL == 13) row_rsplit_seq<T, Seq<16, 16, 16, 2>>(grid, nt, smem, a);
            else ok = false;
        }
        if (!ok && !rc) rc = fail("fft: unsupported R-split row pass");
    }
    // f(SEQF, LR, LOGC as integral constants) for the col_fband variant of this grid; false: not built
    template <typename T, class F>
    static bool dispatch_fband(int gy, int logMy, bool narrow, F&& f) {
        using std::integral_constant;
        constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
        if (gy == 4 && logMy == 11 && !narrow) f(Seq<16, 16, 8>{}, integral_constant<int, 2>{}, integral_constant<int, lc11>{});
        else if (gy == 4 && logMy == 10 && !narrow) f(Seq<16, 8, 8>{}, integral_constant<int, 2>{}, integral_constant<int, lc10>{});
        else if (gy == 4 && logMy == 11) f(Seq<16, 16, 8>{}, integral_constant<int, 2>{}, integral_constant<int, lc11 - 1>{});
        else if (gy == 4 && logMy == 10) f(Seq<16, 8, 8>{}, integral_constant<int, 2>{}, integral_constant<int, lc10 - 1>{});
        else if (sizeof(T) == 8 && gy == 8 && logMy == 11 && !narrow) {      // 16384 rows on the 2048-row grid (float64 only: see row_rs4096)
            if constexpr (sizeof(T) == 8) f(Seq<16, 8, 16>{}, integral_constant<int, 3>{}, integral_constant<int, lc11>{});
        }
        else if (gy == 2 && logMy == 12 && !narrow) f(Seq<16, 16, 16>{}, integral_constant<int, 1>{}, integral_constant<int, lc11 - 1>{});     // the wide band: 8192 rows on the 4096-row grid
        else return false;
        return true;
    }
    template <typename T>
    void col_fband(int gx, int gy, int gz, size_t smem, int logMy, const ColFBandArgs<T>& a) {
        if (rc) return;
        const int lt = Fft2dPlan<T>::fband_lt(), nt = (1 << lt) / EPT;
        const bool narrow = lt < (sizeof(T) == 4 ? 14 : 13);
        const bool ok = dispatch_fband<T>(gy, logMy, narrow, [&](auto seq, auto lrc, auto lcc) {
            go(col_fband_kernel<T, decltype(seq), decltype(lrc)::value, decltype(lcc)::value>, dim3(gx, gy, gz), nt, smem, a);
        });
        if (!ok) rc = fail("fft: unsupported R-split column stage");
    }
    template <typename T>
    void col_fband_pack(int gx, int gy, int logMy, const ColFBandArgs<T>& a, cx<T>* out) {
        if (rc) return;
        const int lt = Fft2dPlan<T>::fband_lt(), nt = (1 << lt) / EPT;
        const bool narrow = lt < (sizeof(T) == 4 ? 14 : 13);
        const bool ok = dispatch_fband<T>(gy, logMy, narrow, [&](auto seq, auto lrc, auto lcc) {
            hipLaunchKernelGGL((col_fband_pack_kernel<T, decltype(seq), decltype(lrc)::value, decltype(lcc)::value>), dim3(gx, gy), dim3(nt), 0, st, a, out);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = fail(std::string("fft launch: ") + hipGetErrorString(e));
        });
        if (!ok) rc = fail("fft: unsupported R-split column stage");
    }
    template <typename T>
    void row_qe(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        // rows of 16384 / 32768 points: put the short radix FIRST so that the reversed (inverse) sequence starts with
        // a radix-16 stage (the plain greedy order would start it with 2 / 4)
        const bool ok = dispatch_seq_qe(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_logl<S>() >= 4) {
                if (nt > row_maxnt<S>()) { if (!rc) rc = fail("fft: row workgroup size exceeds its launch bound"); return; }
                // the first inverse stage gathers its live taps straight from the half-complex rows (untangle on
                // load): NZ live taps per side, a power of two up to R/2 = every tap; NZ = 0: prologue pass through LDS
                if constexpr (S::n >= 2 && sizeof(T) == 4) {
                    dispatch_nz<S>(qe_first_stage_nz(a.logL, S::rget(0), a.win), [&](auto nzc) {
                        go(row_qe_kernel<T, S, decltype(nzc)::value>, dim3(grid), nt, smem, a);
                    });
                } else {
                    go(row_qe_kernel<T, S, 0>, dim3(grid), nt, smem, a);
                }
            } else {
                if (!rc) rc = fail("fft: unsupported row length");
            }
        });
        if (!ok && !rc) rc = fail("fft: unsupported row length");
    }
    template <typename T>
    void row_qe_pair8(int pairs, int M, const RowQeArgs<T>& a) {
        const bool ok = dispatch_rq8(M, a.win, a.lr, a.chain != nullptr, [&](auto ac, auto nzc, auto lay, auto ch) {
            constexpr int A = decltype(ac)::value;
            if (ch.value && !a.tab) { if (!rc) rc = fail("fft: estimator chains need a table"); return; }
            if (!a.rq8c) { if (!rc) rc = fail("fft: the 8-point row stage needs the constants of its grid"); return; }
            go(row_qe8_kernel<T, A, decltype(nzc)::value, decltype(lay)::value, decltype(ch)::value>, dim3(pairs), 64 * A, rq8_lds_bytes<T, A, decltype(ch)::value>(), a);
        });
        if (!ok && !rc) rc = fail("fft: unsupported row grid / layout for the 8-point row stage");
    }
    template <typename T>
    void row_qe_pair(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        const bool ok = dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_logl<S>() >= 10 && seq_logl<S>() <= 13) {
                if (nt != pair_nt<S>()) { if (!rc) rc = fail("fft: pair row stage launched with the wrong workgroup size"); return; }
                dispatch_pair_nz<S>(pair_first_stage_nz(a.logL, S::rget(0), a.win), [&](auto nzc) {
                    if (a.chain) {
                        if constexpr (seq_logl<S>() <= 12) { if (a.lr == 0 && a.tab) go(row_qe_chain_kernel<T, S, decltype(nzc)::value>, dim3(grid), nt, smem, a); else if (!rc) rc = fail("fft: estimator chains need natural-order planes and a table"); }
                        else if (!rc) rc = fail("fft: estimator chains: unsupported row grid");
                    } else
                    if (a.lr == 2) go(row_qe_pair_kernel<T, S, decltype(nzc)::value, 2>, dim3(grid), nt, smem, a);
                    else if (a.lr == 3) {
                        if constexpr (seq_logl<S>() == 11 && sizeof(T) == 8) go(row_qe_pair_kernel<T, S, decltype(nzc)::value, 3>, dim3(grid), nt, smem, a);
                        else if (!rc) rc = fail("fft: the R = 8 layout of the pair row stage is built for float64 on 2048-point row grids");
                    } else if (a.lr == 0) go(row_qe_pair_kernel<T, S, decltype(nzc)::value, 0>, dim3(grid), nt, smem, a);
                    else if (!rc) rc = fail("fft: unsupported R-layout of the pair row stage");
                });
            } else if (!rc) rc = fail("fft: unsupported row grid for the pair row stage");
        });
        if (!ok && !rc) rc = fail("fft: unsupported row grid");
    }
    template <typename T>
    void col_legs(int gx, int gy, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
        if (rc) return;
        rc = launch_col_legs<T>(st, gx, gy, nt, smem, logL, a);   // separate translation unit (fft_legs.hip)
    }
    template <typename T>
    void col_legs_sp(int gx, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
        if (rc) return;
        rc = launch_col_legs_sp<T>(st, gx, nt, smem, logL, a);
    }
    template <typename T>
    void col_fwdlegs(int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsArgs<T>& a) {
        if (rc) return;
        rc = launch_col_fwdlegs<T>(st, gx, gy, nt, smem, logL, a);
    }
    template <typename T>
    void col_fwdlegs_cg(int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsCgArgs<T>& a, int gz = 1) {
        if (rc) return;
        rc = launch_col_fwdlegs_cg<T>(st, gx, gy, nt, smem, logL, a, gz);
    }
    template <typename T>
    void col_div(int gx, int gy, int nt, size_t smem, int logL, const ColDivArgs<T>& a, int gz = 1) {
        const bool ok = dispatch_seq(logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_logl<S>() <= 8) {
                if (nt > col_maxnt<S>()) { if (!rc) rc = fail("fft: column workgroup size exceeds its launch bound"); return; }
                go(col_div_kernel<T, S>, dim3(gx, gy, gz), nt, smem, a);
            } else if (!rc) rc = fail("fft: unsupported column sub-length");
        });
        if (!ok && !rc) rc = fail("fft: unsupported column length");
    }
    // single pass: logL = whole column length (10 or 11), tile of 2^(14 - logL) columns, 1024 threads (f32) / 2^(13 - logL)
    // columns, 512 threads (f64): 128 KB of LDS either way
    DivBinFuse* fuse = nullptr;     // != nullptr: the caller wants binning + moments in the divergence launch (common.hpp)
    template <class K, typename T>
    void go_fused(K kern, int gx, int gz, int nt, size_t smem, ColDivArgs<T> a) {
        {
            const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem);
            if (e != hipSuccess) { rc = fail(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return; }
        }
        fuse->gx = gx;
        if (!fuse->store) a.out = nullptr;
        // tile-major copies of Fn / ids (pipeline.hip) -- valid for THIS tile shape and one Fn for every map of the launch
        const bool tabs = fuse->tab_logc == a.logC && fuse->tab_rows == a.ny;
        a.Fn_t = (tabs && a.fn_moff == 0) ? (const T*)fuse->fn_t : nullptr;
        if (!tabs) fuse->ids_t = nullptr;
        hipLaunchKernelGGL(kern, dim3(gx, 1, gz), dim3(nt), smem, st, a, *fuse);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(std::string("fft launch: ") + hipGetErrorString(e));
        fuse->done = true;
    }
    template <typename T>
    bool col_div_sp(int gx, size_t smem, int logL, const ColDivArgs<T>& a, int gz = 1) {
        if (rc) return true;
        const int lt = Fft2dPlan<T>::div_lt(), nt = (1 << lt) / EPT;
        const bool narrow = lt < (sizeof(T) == 4 ? 14 : 13);
        constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
        // (the wave-private histogram rows of the tail live in the column tile: waves x nids doubles must fit it)
        if (fuse && !a.accumulate && (long)gx * gz * fuse->nids <= fuse->part_cap && (logL == 10 || logL == 11 || (logL == 12 && !narrow)) &&
            (size_t)(nt / 64) * fuse->nids * sizeof(double) <= ((size_t)1 << lt) * sizeof(cx<T>)) {
            if (logL == 12) go_fused(col_div_sp_bin_kernel<T, Seq<16, 16, 16>, lc11 - 1>, gx, gz, nt, smem, a);      // 4096-row column grids (the wide band at 8192^2)
            else if (logL == 11 && !narrow) go_fused(col_div_sp_bin_kernel<T, Seq<16, 16, 8>, lc11>, gx, gz, nt, smem, a);
            else if (logL == 10 && !narrow) go_fused(col_div_sp_bin_kernel<T, Seq<16, 16, 4>, lc10>, gx, gz, nt, smem, a);
            else if (logL == 11) go_fused(col_div_sp_bin_kernel<T, Seq<16, 16, 8>, lc11 - 1>, gx, gz, nt, smem, a);
            else go_fused(col_div_sp_bin_kernel<T, Seq<16, 16, 4>, lc10 - 1>, gx, gz, nt, smem, a);
            return true;
        }
        if (logL == 12 && !narrow) { go(col_div_sp_kernel<T, Seq<16, 16, 16>, lc11 - 1>, dim3(gx, 1, gz), nt, smem, a); return true; }
        if (logL == 11 && !narrow) { go(col_div_sp_kernel<T, Seq<16, 16, 8>, lc11>, dim3(gx, 1, gz), nt, smem, a); return true; }
        if (logL == 10 && !narrow) { go(col_div_sp_kernel<T, Seq<16, 16, 4>, lc10>, dim3(gx, 1, gz), nt, smem, a); return true; }
        if (logL == 11) { go(col_div_sp_kernel<T, Seq<16, 16, 8>, lc11 - 1>, dim3(gx, 1, gz), nt, smem, a); return true; }
        if (logL == 10) { go(col_div_sp_kernel<T, Seq<16, 16, 4>, lc10 - 1>, dim3(gx, 1, gz), nt, smem, a); return true; }
        return false;
    }
    template <typename T>
    void col_deriv(int gx, int gy, int nt, size_t smem, const ColDerivArgs<T>& a, int nz) {
        const bool ok = dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_logl<S>() <= 8) {
                if (nt > col_maxnt<S>()) { if (!rc) rc = fail("fft: column workgroup size exceeds its launch bound"); return; }
                go(col_deriv_kernel<T, S>, dim3(gx, gy, nz), nt, smem, a);
            } else if (!rc) rc = fail("fft: unsupported column sub-length");
        });
        if (!ok && !rc) rc = fail("fft: unsupported column length");
    }
    template <typename T>
    void col(int gx, int gy, int nt, size_t smem, const ColArgs<T>& a, int nz = 1) {
        const bool ok = dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_logl<S>() <= 8) {
                if (nt > col_maxnt<S>()) { if (!rc) rc = fail("fft: column workgroup size exceeds its launch bound"); return; }
                go(col_fft_kernel<T, S>, dim3(gx, gy, nz), nt, smem, a);
            } else {
                if (!rc) rc = fail("fft: unsupported column sub-length");
            }
        });
        if (!ok && !rc) rc = fail("fft: unsupported column length");
    }
};

template <typename T>
static Fft2dPlan<T> view(const oa_plan* p) {
    Fft2dPlan<T> f;
    f.ny = p->ny; f.nx = p->nx; f.logNy = p->logNy; f.logNx = p->logNx; f.kp = p->kp;
    f.tw_x = (const cx<T>*)p->tw_x; f.tw_y = (const cx<T>*)p->tw_y; for (int i = 0; i < RQ8_NGRIDS; ++i) f.rq8c[i] = (const cx<T>*)p->rq8c[i];
    return f;
}
// COLUMN GRID view: the same map transformed on my < ny rows (plan_ensure_col_grid made the W_my table)
template <typename T>
static Fft2dPlan<T> coarse_view(const oa_plan* p, int my) {
    Fft2dPlan<T> f = view<T>(p);
    if (my > 0 && my < p->ny && is_pow2(my) && p->tw_y_small[ilog2(my)]) {
        f.ny = my; f.logNy = ilog2(my); f.tw_y = (const cx<T>*)p->tw_y_small[ilog2(my)]; f.ny_full = p->ny;
    }
    return f;
}
int plan_ensure_col_grid(oa_plan* p, int my) {
    if (my <= 0 || my >= p->ny) return 0;
    if (!is_pow2(my) || my < 32) return fail("column grid must be a power of two >= 32");
    void*& slot = p->tw_y_small[ilog2(my)];
    if (slot) return 0;                       // tables are kept: no allocation / synchronisation after the first use of a grid
    if (p->dtype == OA_F32) {
        auto t = make_twiddles<float>(my);
        OA_HIP(hipMalloc(&slot, t.size() * sizeof(cx<float>)));
        OA_HIP(hipMemcpy(slot, t.data(), t.size() * sizeof(cx<float>), hipMemcpyHostToDevice));
    } else {
        auto t = make_twiddles<double>(my);
        OA_HIP(hipMalloc(&slot, t.size() * sizeof(cx<double>)));
        OA_HIP(hipMemcpy(slot, t.data(), t.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
    }
    return 0;
}

template <typename T>
static int r2c_impl(oa_plan* p, const void* in, void* out, double scale, int width, int rband, hipStream_t st) {
    if (int rc = plan_ensure_scratch(p, (size_t)p->ny * p->kp * sizeof(cx<T>))) return rc;
    HipLauncher q{st};
    view<T>(p).r2c(q, (const T*)in, (cx<T>*)out, (cx<T>*)p->scratch, (T)scale, width, rband);
    return q.rc;
}
template <typename T>
static int c2r_impl(oa_plan* p, const void* in, void* out, double scale, int width, hipStream_t st, const void* mul = nullptr) {
    if (int rc = plan_ensure_scratch(p, (size_t)p->ny * p->kp * sizeof(cx<T>))) return rc;
    HipLauncher q{st};
    view<T>(p).c2r(q, (const cx<T>*)in, (T*)out, (cx<T>*)p->scratch, (T)scale, width, (const T*)mul);
    return q.rc;
}
template <typename T>
static int c2c_impl(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st) {
    if (int rc = plan_ensure_scratch(p, (size_t)p->ny * p->nx * sizeof(cx<T>))) return rc;
    HipLauncher q{st};
    view<T>(p).c2c(q, (const cx<T>*)in, (cx<T>*)out, (cx<T>*)p->scratch, inverse != 0, (T)scale);
    return q.rc;
}

template <typename T>
static int pass_impl(oa_plan* p, int pass_id, const void* in, void* out, int width, hipStream_t st) {
    HipLauncher q{st};
    auto f = view<T>(p);
    const int w = f.clampw(width);
    switch (pass_id) {
        case 0: f.rows(q, ROW_R2C, in, p->nx / 2, out, p->kp, (T)1, w); break;
        case 1: f.cols(q, (const cx<T>*)in, p->kp, (cx<T>*)out, p->kp, w, false, (T)1, 1); break;
        case 2: f.cols(q, (const cx<T>*)in, p->kp, (cx<T>*)out, p->kp, w, false, (T)1, 2); break;
        case 3: f.rows(q, ROW_C2R, in, p->kp, out, p->nx / 2, (T)1, w); break;
        default: return fail("oa_fft_pass: pass_id must be 0..3");
    }
    return q.rc;
}

template <typename T>
static int cols_impl(oa_plan* p, const void* in, void* out, int inverse, double scale, int width, hipStream_t st) {
    HipLauncher q{st};
    auto f = view<T>(p);
    f.cols(q, (const cx<T>*)in, p->kp, (cx<T>*)out, p->kp, f.clampw(width), inverse != 0, (T)scale);
    return q.rc;
}
template <typename T>
static int qe_rows_impl(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale,
                        int accumulate, int win, int wout, int mrow, hipStream_t st, long pin = 0, long pout = 0, int my = 0, int lr = 0) {
    HipLauncher q{st};
    auto f = coarse_view<T>(p, my);
    const int wi = f.clampw(win), wo = f.clampw(wout);
    if (mrow < 0) {                                                         // auto: smallest alias-free grid
        mrow = Fft2dPlan<T>::row_grid_min(p->nx, wi, wo);
        if (2L * wi + wo > mrow) mrow = 0;                                  // no band limit to exploit: full-length path
    } else if (mrow > 0) {
        if (mrow > p->nx) mrow = p->nx;
        if (!(is_pow2(mrow) && mrow >= 64) && !(rq8_is_m3(mrow) && f.rows_qe_is_pair(wi, wo, mrow)))
            return fail("oa_qe_rows: mrow must be a power of two >= 64 (or 1536 for band limits that fit it)");
        if (2L * wi + wo > mrow) return fail("oa_qe_rows: mrow < 2*win + wout would alias the leg products into the kept columns");
    }
    if (lr && (!f.rows_qe_is_pair(wi, wo, mrow) || (lr == 1 && !f.rows_qe_lr1(wi, wo, mrow))))
        return fail("oa_qe_rows: leg planes in the R-layout need the two-rows-per-transform row stage");
    f.rows_qe(q, (const cx<T>*)gx, (const cx<T>*)gy, (const cx<T>*)h, (cx<T>*)px, (cx<T>*)py, (T)scale, accumulate, wi, wo, mrow,
              pin, pout, 1, 0, 0, -1, nullptr, lr);
    return q.rc;
}
// does the from-map path of this geometry run the R-split row pass + single-pass column stage (leg planes in the R-LAYOUT)?
template <typename T>
static int rsplit_lr(const oa_plan* p, int my, int width, int wout, int mrow) {
    auto f = view<T>(p);
    const int w = f.clampw(width), wo = f.clampw(wout);
    if (!(my > 0 && my < p->ny && is_pow2(my) && p->tw_y_small[ilog2(my)]) || !Fft2dPlan<T>::has_rsplit(p->logNy, p->logNx, my, w)) return 0;
    if (mrow < 0) { mrow = Fft2dPlan<T>::row_grid_min(p->nx, w, wo); if (2L * w + wo > mrow) mrow = 0; }
    const auto cv = coarse_view<T>(p, my);
    const int lr = p->logNy - ilog2(my);
    if (lr == 1 && !cv.rows_qe_lr1(w, wo, mrow)) return 0;
    return cv.rows_qe_is_pair(w, wo, mrow) ? lr : 0;
}
int qe_rsplit_lr(const oa_plan* p, int my, int width, int wout, int mrow) {
    return p->dtype == OA_F32 ? rsplit_lr<float>(p, my, width, wout, mrow) : rsplit_lr<double>(p, my, width, wout, mrow);
}

template <typename T>
static int legs_cols_impl(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* gx, void* gy,
                          void* h, int width, int rband, hipStream_t st, long pout = 0, int my = 0, long pin = 0) {
    HipLauncher q{st};
    coarse_view<T>(p, my).legs_cols(q, (const cx<T>*)kX, (const cx<T>*)kY, (const T*)FG, (const T*)FH, (const T*)p->lxd,
                                    (const T*)p->lyd, (cx<T>*)gx, (cx<T>*)gy, (cx<T>*)h, width, rband, pin, pout, true);
    return q.rc;
}
// real map -> the three column-transformed leg planes (both legs from this one map)
template <typename T>
static int map_legs_cols_impl(oa_plan* p, const void* map, const void* FG, const void* FH, void* gx, void* gy, void* h,
                              int width, int rband, hipStream_t st, long pwork = 0, long pout = 0, int stages = 7, int my = 0, int lr = 0,
                              const void* fgh = nullptr) {
    const size_t plane = (size_t)p->ny * p->kp * sizeof(cx<T>);
    if (int rc = plan_ensure_scratch(p, 2 * plane)) return rc;
    HipLauncher q{st};
    auto f = view<T>(p);
    cx<T>* tA = (cx<T>*)p->scratch;
    cx<T>* tB = tA + (size_t)p->ny * p->kp;
    const int w = f.clampw(width);
    const long pw = pwork > 0 ? pwork : p->kp;          // pitch of the two scratch planes
    // stages (per-kernel timing, oa_qe_tt_stage): 1 = row R2C, 2 = forward column pass 1, 4 = fused legs + inverse pass 2
    if (lr > 0) {
        // R-SPLIT: the row pass carries the first radix-R butterfly of the column transform; ONE column kernel to the leg planes
        const auto cv = coarse_view<T>(p, my);
        const long kplane = (long)my * pw;
        if (stages & 1) f.rows_rsplit(q, map, tA, pw, kplane, w, my);
        if (stages & 4) f.legs_fband(q, cv, tA, kplane, pw, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)gx, (cx<T>*)gy,
                                     (cx<T>*)h, width, rband, pout > 0 ? pout : p->kp, 1, 0, 0, (const cx<T>*)fgh);
        return q.rc;
    }
    if (stages & 1) f.rows(q, ROW_R2C, map, p->nx / 2, tA, pw, (T)1, w);
    if (my > 0 && my < p->ny) {
        // COLUMN GRID: the map's transform is needed on the leg band only (forward pass 2 stores just those rows, at
        // their full-resolution positions); legs and inverse transform then run on my rows
        if (stages & 2) f.cols(q, tA, pw, tB, pw, w, false, (T)1, 1);
        if (stages & 4) {
            static const bool nofuse = exp_env("OA_NO_FWDLEGS_CG") != nullptr;        // A/B switch
            const auto cv = coarse_view<T>(p, my);
            if (nofuse || !f.legs_cols_from_pass1_cg(q, cv, tB, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd,
                                                     (cx<T>*)gx, (cx<T>*)gy, (cx<T>*)h, width, pw, pout)) {
                f.cols(q, tA, pw, tB, pw, w, false, (T)1, 2, 1, nullptr, nullptr, rband);
                cv.legs_cols(q, tB, tB, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)gx, (cx<T>*)gy,
                             (cx<T>*)h, width, rband, pw, pout, true);
            }
        }
    } else if (Fft2dPlan<T>::has_fwdlegs(p->logNy)) {
        if (stages & 2) f.cols(q, tA, pw, tB, pw, w, false, (T)1, 1);       // forward pass 1 only
        if (stages & 4)
            f.legs_cols_from_pass1(q, tB, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)gx,
                                   (cx<T>*)gy, (cx<T>*)h, width, rband, pw, pout);
    } else {
        // (short columns: both forward column passes count as stage 2, the leg kernel + inverse pass 2 as stage 4)
        if (stages & 2) f.cols(q, tA, pw, tB, pw, w, false, (T)1, 0, 1, nullptr, nullptr, rband);
        if (stages & 4)
            f.legs_cols(q, tB, tB, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)gx, (cx<T>*)gy,
                        (cx<T>*)h, width, rband, pw, pout);
    }
    return q.rc;
}

template <typename T>
static int cols_div_impl(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, int accumulate,
                         int width, int rband, hipStream_t st, long pin = 0, int my = 0, DivBinFuse* fuse = nullptr) {
    const size_t plane = (size_t)p->ny * p->kp * sizeof(cx<T>);
    if (int rc = plan_ensure_scratch(p, 2 * plane)) return rc;
    HipLauncher q{st};
    q.fuse = fuse;
    cx<T>* tA = (cx<T>*)p->scratch;
    cx<T>* tB = tA + (size_t)p->ny * p->kp;
    coarse_view<T>(p, my).cols_div(q, (const cx<T>*)pa, (const cx<T>*)pb, (const T*)Fn, (const T*)p->lxd, (const T*)p->lyd,
                                   (cx<T>*)out, tA, tB, accumulate, width, rband, pin);
    return q.rc;
}

// oa_qe_mv: the divergence of `nmaps` estimators in one launch (grid z = estimator): product planes in_moff apart, Fn planes
// fn_moff apart, each estimator's kappa into its own plane (out_moff apart); tmp: 2 nmaps compact planes for the two-pass path
template <typename T>
static int cols_div_batch_impl(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, void* tmp, int nmaps, long in_moff,
                               long fn_moff, long out_moff, int width, int rband, hipStream_t st, long pin, int my, DivBinFuse* fuse = nullptr) {
    HipLauncher q{st};
    q.fuse = fuse;
    cx<T>* tA = (cx<T>*)tmp;
    cx<T>* tB = tA + (in_moff >> 1);
    coarse_view<T>(p, my).cols_div(q, (const cx<T>*)pa, (const cx<T>*)pb, (const T*)Fn, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)out, tA,
                                   tB, 0, width, rband, pin, nmaps, in_moff, in_moff, out_moff, fn_moff);
    return q.rc;
}

// ---- TWO real maps -> two kappa_hat planes with every coarse-grid stage launched once for both (pipeline.hip,
//      oa_qe_tt_moments2).  The small launches behind the row R2C are latency-bound and far from filling the chip: doing
//      two realisations' worth of work per launch costs little more than one.  Returns -1 when this geometry lacks one
//      of the pieces (column grid with the fused forward-legs kernel, two-rows-per-transform row stage): the caller
//      then runs the maps one after the other.
template <typename T>
static int qe_tt_pair_impl(oa_plan* p, const void* map0, const void* map1, const void* FG, const void* FH, const void* Fn, void* c0,
                           void* c1, void* c2, void* g0, void* g1, void* out0, void* out1, int wl, int wk, int rl, int rk, int mrow,
                           int my, long pl, long pk, hipStream_t st, DivBinFuse* fuse, const void* fgh) {
    auto f = view<T>(p);
    const int lr = rsplit_lr<T>(p, my, wl, wk, mrow);
    if (!(my > 0 && my < p->ny && is_pow2(my) && p->tw_y_small[ilog2(my)]) || !(lr || Fft2dPlan<T>::has_fwdlegs_cg(p->logNy, my))) return -1;
    const auto cv = coarse_view<T>(p, my);
    const int wi = f.clampw(wl), wo = f.clampw(wk);
    if (mrow < 0) { mrow = Fft2dPlan<T>::row_grid_min(p->nx, wi, wo); if (2L * wi + wo > mrow) mrow = 0; }
    if (!cv.rows_qe_is_pair(wi, wo, mrow)) return -1;
    const long ms = (long)p->ny * pl, cms = (long)my * pl, gms = (long)my * pk;      // second-map offsets of the plane families
    if (2 * ms > (long)p->ny * p->kp || 2 * gms > (long)p->ny * p->kp) return -1;
    const size_t plane = (size_t)p->ny * p->kp * sizeof(cx<T>);
    if (int rc = plan_ensure_scratch(p, 2 * plane)) return rc;
    HipLauncher q{st};
    q.fuse = fuse;
    cx<T>* tA = (cx<T>*)p->scratch;
    cx<T>* tB = tA + (size_t)p->ny * p->kp;
    const double s = 1.0 / ((double)p->ny * p->nx), sy = (double)p->ny / my;
    if (lr) {
        // R-SPLIT: two row passes, then ONE column launch, ONE row-stage launch and ONE divergence launch for both maps
        const long kplane = (long)my * pl;
        f.rows_rsplit(q, map0, tA, pl, kplane, wi, my);
        f.rows_rsplit(q, map1, tA + ms, pl, kplane, wi, my);
        f.legs_fband(q, cv, tA, kplane, pl, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)c0, (cx<T>*)c1, (cx<T>*)c2, wl, rl,
                     pl, 2, ms, cms, (const cx<T>*)fgh);
        cv.rows_qe(q, (const cx<T>*)c0, (const cx<T>*)c1, (const cx<T>*)c2, (cx<T>*)g0, (cx<T>*)g1, (T)(s * s * sy), 0, wi, wo, mrow, pl, pk, 2, cms, gms,
                   -1, nullptr, lr);
        cv.cols_div(q, (const cx<T>*)g0, (const cx<T>*)g1, (const T*)Fn, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)out0, tA, tB, 0, wk, rk, pk, 2,
                    gms, gms, (long)((cx<T>*)out1 - (cx<T>*)out0));
        return q.rc;
    }
    f.rows(q, ROW_R2C, map0, p->nx / 2, tA, pl, (T)1, wi);
    f.rows(q, ROW_R2C, map1, p->nx / 2, tA + ms, pl, (T)1, wi);
    const cx<T>* ins[2] = {tA, tA + ms};
    cx<T>* outs[2] = {tB, tB + ms};
    f.cols(q, tA, pl, tB, pl, wi, false, (T)1, 1, 2, ins, outs);                      // forward pass 1 of both maps, one launch
    if (!f.legs_cols_from_pass1_cg(q, cv, tB, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)c0, (cx<T>*)c1,
                                   (cx<T>*)c2, wl, pl, pl, 2, ms, cms))
        return -1;
    cv.rows_qe(q, (const cx<T>*)c0, (const cx<T>*)c1, (const cx<T>*)c2, (cx<T>*)g0, (cx<T>*)g1, (T)(s * s * sy), 0, wi, wo, mrow, pl, pk, 2, cms, gms);
    cv.cols_div(q, (const cx<T>*)g0, (const cx<T>*)g1, (const T*)Fn, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)out0, tA, tB, 0, wk, rk, pk, 2,
                gms, gms, (long)((cx<T>*)out1 - (cx<T>*)out0));
    return q.rc;
}
int qe_tt_pair_w(oa_plan* p, const void* map0, const void* map1, const void* FG, const void* FH, const void* Fn, void* c0, void* c1,
                 void* c2, void* g0, void* g1, void* out0, void* out1, int wl, int wk, int rl, int rk, int mrow, int my, long pl, long pk,
                 hipStream_t st, DivBinFuse* fuse, const void* fgh) {
    return p->dtype == OA_F32 ? qe_tt_pair_impl<float>(p, map0, map1, FG, FH, Fn, c0, c1, c2, g0, g1, out0, out1, wl, wk, rl, rk, mrow, my, pl, pk, st, fuse, fgh)
                              : qe_tt_pair_impl<double>(p, map0, map1, FG, FH, Fn, c0, c1, c2, g0, g1, out0, out1, wl, wk, rl, rk, mrow, my, pl, pk, st, fuse, fgh);
}
// the packed filter table of the R-split column stage (ColFBandArgs::fgh) for this binding: entries of it, and the launch that fills it
template <typename T>
static int fband_pack_impl(oa_plan* p, const void* FG, const void* FH, void* out, int width, int rband, int my, hipStream_t st) {
    HipLauncher q{st};
    auto f = view<T>(p);
    const auto cv = coarse_view<T>(p, my);
    f.legs_fband(q, cv, nullptr, 0, 0, (const T*)FG, (const T*)FH, (const T*)p->lxd, (const T*)p->lyd, nullptr, nullptr, nullptr, width, rband, 0, 1, 0, 0,
                 nullptr, (cx<T>*)out);
    return q.rc;
}
long qe_fband_table_entries(const oa_plan* p, int width, int my) {
    return p->dtype == OA_F32 ? view<float>(p).fband_table_entries(coarse_view<float>(p, my), width)
                              : view<double>(p).fband_table_entries(coarse_view<double>(p, my), width);
}
int qe_fband_pack_w(oa_plan* p, const void* FG, const void* FH, void* out, int width, int rband, int my, hipStream_t st) {
    return p->dtype == OA_F32 ? fband_pack_impl<float>(p, FG, FH, out, width, rband, my, st) : fband_pack_impl<double>(p, FG, FH, out, width, rband, my, st);
}

// ---- the same passes on the plan's COMPACT work planes (pipeline.hip): pl = pitch of the leg planes and of the
//      scratch planes while they hold the input transform, pk = pitch of the product planes (Fft2dPlan::work_pitch)
long work_pitch(const oa_plan* p, int w) {
    return p->dtype == OA_F32 ? view<float>(p).work_pitch(w) : view<double>(p).work_pitch(w);
}
int qe_map_legs_cols_w(oa_plan* p, const void* map, const void* FG, const void* FH, void* gx, void* gy, void* h, int width,
                       int rband, long pl, hipStream_t st, int stages, int my, int lr, const void* fgh) {
    return p->dtype == OA_F32 ? map_legs_cols_impl<float>(p, map, FG, FH, gx, gy, h, width, rband, st, pl, pl, stages, my, lr, fgh)
                              : map_legs_cols_impl<double>(p, map, FG, FH, gx, gy, h, width, rband, st, pl, pl, stages, my, lr, fgh);
}
// Windowed simulation front end (oa_mc_run_windowed): full-plane hc spectrum -> inverse column transform (into `cols_tmp`, a full
// hc plane) -> ONE fused row pass C2R x window -> R2C (ROW_WIN) that leaves the row-transformed windowed map on the plan's first
// scratch plane at the compact pitch `pl`, active columns only -- exactly what the row R2C of qe_map_legs_cols_w (stage 1) leaves
// there, so the caller continues with stages 2 | 4.  The real map never exists in HBM.
template <typename T>
static int windowed_rows_impl(oa_plan* p, const void* hc_in, void* cols_tmp, const void* window, int width, long pl, double scale, hipStream_t st,
                              void* out) {
    const size_t plane = (size_t)p->ny * p->kp * sizeof(cx<T>);
    if (int rc = plan_ensure_scratch(p, 2 * plane)) return rc;
    HipLauncher q{st};
    auto f = view<T>(p);
    f.cols(q, (const cx<T>*)hc_in, p->kp, (cx<T>*)cols_tmp, p->kp, p->nx / 2 + 1, true, (T)1);
    f.rows(q, ROW_WIN, cols_tmp, p->kp, out ? out : p->scratch, pl > 0 ? pl : p->kp, (T)scale, f.clampw(width), window);
    return q.rc;
}
// out = nullptr: the plan's first scratch plane (what qe_map_legs_cols_w stages 2 | 4 read); else a caller plane of pitch pl
int qe_windowed_rows_w(oa_plan* p, const void* hc_in, void* cols_tmp, const void* window, int width, long pl, double scale, hipStream_t st, void* out) {
    return p->dtype == OA_F32 ? windowed_rows_impl<float>(p, hc_in, cols_tmp, window, width, pl, scale, st, out)
                              : windowed_rows_impl<double>(p, hc_in, cols_tmp, window, width, pl, scale, st, out);
}
// forward column transform of B row-transformed compact planes (pitch pin, in_moff apart) onto the leg band (columns < width, rows
// |ky index| < rband, natural order) of B full-pitch hc planes out_moff apart: two launches for the batch
template <typename T>
static int fwd_cols_batch_impl(oa_plan* p, const void* in, long pin, void* out, int B, long in_moff, long out_moff, int width, int rband, hipStream_t st) {
    HipLauncher q{st};
    auto f = view<T>(p);
    f.cols(q, (const cx<T>*)in, pin, (cx<T>*)out, p->kp, f.clampw(width), false, (T)1, 0, 1, nullptr, nullptr, rband, false, -1, B, in_moff, out_moff);
    return q.rc;
}
int qe_fwd_cols_batch_w(oa_plan* p, const void* in, long pin, void* out, int B, long in_moff, long out_moff, int width, int rband, hipStream_t st) {
    return p->dtype == OA_F32 ? fwd_cols_batch_impl<float>(p, in, pin, out, B, in_moff, out_moff, width, rband, st)
                              : fwd_cols_batch_impl<double>(p, in, pin, out, B, in_moff, out_moff, width, rband, st);
}
// flat-sky Taylor lensing, FFT part (oa_lens_maps): nmaps real maps -> their transforms (k0: nmaps hc planes) -> the nd derivative
// fields of each, inverse-transformed: ONE pass-1 launch with the derivative factor at the load, ONE pass-2 launch, ONE row C2R
// launch per (map, y-derivative order) (hc_pool: one hc plane at least; real_pool: nmaps * nd real planes, 1 / Npix applied)
template <typename T>
static int lens_derivs_impl(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd,
                            hipStream_t st, const void* hc_in, long hc_stride, double hc_scale) {
    const size_t plane = (size_t)p->ny * p->kp * sizeof(cx<T>);
    if (int rc = plan_ensure_scratch(p, plane)) return rc;
    HipLauncher q{st};
    auto f = view<T>(p);
    const long hcp = (long)p->ny * p->kp, rp = (long)p->ny * (p->nx / 2);       // plane strides in complex elements
    // hc_in: the maps' transforms are the caller's (oa_lens_maps_hc) -- no R2C, and the map itself ((a, b) = (0, 0)) is one more plane
    // of the b = 0 row launch: a map then owns nd + 1 planes of the real pool, D_00 first
    const cx<T>* src = hc_in ? (const cx<T>*)hc_in : (const cx<T>*)k0;
    const long sstride = hc_in ? hc_stride : hcp;
    const int d00 = hc_in ? 1 : 0;
    if (!hc_in)
        for (int m = 0; m < nmaps; ++m)
            f.r2c(q, (const T*)real_in + (long)m * in_stride, (cx<T>*)k0 + (long)m * hcp, (cx<T>*)p->scratch, (T)1);
    // SEPARABLE derivatives: (i lx)^a (i ly)^b k0 -- the column transform of (i ly)^b k0 does not depend on a, so a map needs `order`
    // column transforms (b = 0 .. order - 1; col_deriv_body, b-only mode) instead of nd = order (order + 1) / 2 - 1, and every
    // x-derivative is a row C2R of that column-transformed plane with (i lx)^a applied at its load (RowArgs::dlx): per b ONE launch
    // for a = (b == 0) .. order - 1 - b.  The column-transformed plane (one hc plane: the chunk buffer) is written by pass 1,
    // transformed in place by pass 2 and read by the row launch back to back: it stays in the 256 MB infinity cache (a column pass
    // over one cache-resident float64 plane at 4096^2 takes 49-52 us, over planes streamed from HBM 71-83 us:
    // profiles/r04f_lensloop_kernel_stats_f64_batched.txt, r04_lensloop_step.txt).
    const int order = (int)((std::sqrt(8.0 * (nd + 1) + 1.0) - 1.0) / 2.0 + 0.5);      // nd = order (order + 1) / 2 - 1
    const double cscale = hc_in ? hc_scale : 1.0 / ((double)p->ny * p->nx);
    for (int m = 0; m < nmaps; ++m)
        for (int b = 0; b < order; ++b) {
            const int a0 = (b == 0 && !d00) ? 1 : 0, na = order - b - a0;
            if (na <= 0) continue;
            f.cols_derivs(q, src, sstride, (cx<T>*)hc_pool, hcp, nmaps, order, (const T*)p->lxd, (const T*)p->lyd, m * order + b, 1, 1);
            f.rows(q, ROW_C2R, hc_pool, p->kp, (cx<T>*)real_pool + ((long)m * (nd + d00) + d00) * rp, p->nx / 2, (T)cscale, 0x7fffffff,
                   nullptr, na, 0, rp, (const T*)p->lxd, a0, b);
        }
    return q.rc;
}
int div_tile_logc(const oa_plan* p, int rows) {
    const int lt = p->dtype == OA_F32 ? Fft2dPlan<float>::div_lt() : Fft2dPlan<double>::div_lt();
    return lt - ilog2(rows);
}
int qe_lens_derivs_w(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd, hipStream_t st,
                     const void* hc_in, long hc_stride, double hc_scale) {
    if (p->mixed) return mixed_lens_derivs(p, nmaps, real_in, in_stride, k0, hc_pool, real_pool, nd, st, hc_in, hc_stride, hc_scale);
    return p->dtype == OA_F32 ? lens_derivs_impl<float>(p, nmaps, real_in, in_stride, k0, hc_pool, real_pool, nd, st, hc_in, hc_stride, hc_scale)
                              : lens_derivs_impl<double>(p, nmaps, real_in, in_stride, k0, hc_pool, real_pool, nd, st, hc_in, hc_stride, hc_scale);
}
int qe_legs_cols_w(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* gx, void* gy, void* h,
                   int width, int rband, long pl, hipStream_t st, int my) {
    return p->dtype == OA_F32 ? legs_cols_impl<float>(p, kX, kY, FG, FH, gx, gy, h, width, rband, st, pl, my)
                              : legs_cols_impl<double>(p, kX, kY, FG, FH, gx, gy, h, width, rband, st, pl, my);
}
// oa_qe_mv: one filtered field per call -- subset 1: src, F -> a (H plane); subset 2: src, F -> a, b (gradient pair) -- pass 1
// only; then ONE inverse pass 2 over `nplanes` compact planes `stride` elements apart
template <typename T>
static int legs_subset_impl(oa_plan* p, const void* src, const void* F, void* a, void* b, int subset, int width, int rband,
                            hipStream_t st, long pout, int my) {
    HipLauncher q{st};
    const cx<T>* k = (const cx<T>*)src;
    const T* f = (const T*)F;
    coarse_view<T>(p, my).legs_cols(q, k, k, f, f, (const T*)p->lxd, (const T*)p->lyd, (cx<T>*)a, (cx<T>*)b, (cx<T>*)a, width, rband, 0,
                                    pout, true, subset);
    return q.rc;
}
template <typename T>
static int legs_batch_impl(oa_plan* p, const void* src0, long off1, long off2, unsigned long long srcsel, const void* const* ftab,
                           int ngrad, int nh, void* pool, long ostride, int width, int rband, hipStream_t st, long pout, int my, int selbits, int* finished) {
    HipLauncher q{st};
    const bool done = coarse_view<T>(p, my).legs_cols_batch(q, (const cx<T>*)src0, off1, off2, srcsel, (const T* const*)ftab, ngrad, nh, (const T*)p->lxd,
                                                            (const T*)p->lyd, (cx<T>*)pool, ostride, width, rband, 0, pout, selbits);
    if (finished) *finished = done ? 1 : 0;
    return q.rc;
}
template <typename T>
static int legs_pass2_impl(oa_plan* p, void* pool, int nplanes, long stride, int width, hipStream_t st, long pout, int my) {
    HipLauncher q{st};
    const auto f = coarse_view<T>(p, my);
    const long po = pout > 0 ? pout : p->kp;
    f.cols(q, (const cx<T>*)pool, po, (cx<T>*)pool, po, f.clampw(width), true, (T)1, 2, 1, nullptr, nullptr, 0, false, -1, nplanes, stride,
           stride);
    return q.rc;
}
int qe_legs_subset_w(oa_plan* p, const void* src, const void* F, void* a, void* b, int subset, int width, int rband, long pl,
                     hipStream_t st, int my) {
    return p->dtype == OA_F32 ? legs_subset_impl<float>(p, src, F, a, b, subset, width, rband, st, pl, my)
                              : legs_subset_impl<double>(p, src, F, a, b, subset, width, rband, st, pl, my);
}
int qe_cols_div_batch_w(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, void* tmp, int nmaps, long in_moff,
                        long fn_moff, long out_moff, int width, int rband, long pk, hipStream_t st, int my, DivBinFuse* fuse) {
    return p->dtype == OA_F32 ? cols_div_batch_impl<float>(p, pa, pb, Fn, out, tmp, nmaps, in_moff, fn_moff, out_moff, width, rband, st, pk, my, fuse)
                              : cols_div_batch_impl<double>(p, pa, pb, Fn, out, tmp, nmaps, in_moff, fn_moff, out_moff, width, rband, st, pk, my, fuse);
}
int qe_legs_batch_w(oa_plan* p, const void* src0, long off1, long off2, unsigned long long srcsel, const void* const* ftab,
                    int ngrad, int nh, void* pool, long ostride, int width, int rband, long pl, hipStream_t st, int my, int selbits, int* finished) {
    return p->dtype == OA_F32 ? legs_batch_impl<float>(p, src0, off1, off2, srcsel, ftab, ngrad, nh, pool, ostride, width, rband, st, pl, my, selbits, finished)
                              : legs_batch_impl<double>(p, src0, off1, off2, srcsel, ftab, ngrad, nh, pool, ostride, width, rband, st, pl, my, selbits, finished);
}
// the row stage of `nmaps` maps in one launch (two-rows-per-transform kernel only): -1 when this geometry runs another kernel
template <typename T>
static int qe_rows_batch_impl(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale, int win, int wout,
                              int mrow, hipStream_t st, long pin, long pout, int my, int nmaps, long in_moff, long h_moff, long out_moff) {
    HipLauncher q{st};
    auto f = coarse_view<T>(p, my);
    const int wi = f.clampw(win), wo = f.clampw(wout);
    if (mrow < 0) { mrow = Fft2dPlan<T>::row_grid_min(p->nx, wi, wo); if (2L * wi + wo > mrow) mrow = 0; }
    if (!f.rows_qe_is_pair(wi, wo, mrow)) return -1;
    f.rows_qe(q, (const cx<T>*)gx, (const cx<T>*)gy, (const cx<T>*)h, (cx<T>*)px, (cx<T>*)py, (T)scale, 0, wi, wo, mrow, pin, pout, nmaps, in_moff,
              out_moff, h_moff);
    return q.rc;
}
// row stage of n maps with per-map planes and scales (device table `dev_tab`, n entries of RowQeMap<T>, rewritten from the host
// arrays when `upload`); -1 when this geometry's row stage is not the two-rows-per-transform kernel
template <typename T>
static int qe_rows_table_impl(oa_plan* p, int n, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                              void* const* py, const double* scales, void* dev_tab, int upload, int accumulate, int win, int wout, int mrow,
                              hipStream_t st, long pin, long pout, int my) {
    HipLauncher q{st};
    auto f = coarse_view<T>(p, my);
    const int wi = f.clampw(win), wo = f.clampw(wout);
    if (mrow < 0) { mrow = Fft2dPlan<T>::row_grid_min(p->nx, wi, wo); if (2L * wi + wo > mrow) mrow = 0; }
    if (!f.rows_qe_is_pair(wi, wo, mrow)) return -1;
    if (upload) {
        std::vector<RowQeMap<T>> tab(n);
        const double fac = f.row_grid_scale(mrow);
        for (int i = 0; i < n; ++i)
            tab[i] = RowQeMap<T>{(const cx<T>*)gx[i], (const cx<T>*)gy[i], (const cx<T>*)h[i], (cx<T>*)px[i], (cx<T>*)py[i], (T)(scales[i] * fac)};
        OA_HIP(hipMemcpyAsync(dev_tab, tab.data(), n * sizeof(RowQeMap<T>), hipMemcpyHostToDevice, st));     // pageable: staged before return
    }
    f.rows_qe(q, (const cx<T>*)gx[0], (const cx<T>*)gy[0], (const cx<T>*)h[0], (cx<T>*)px[0], (cx<T>*)py[0], (T)scales[0], accumulate, wi, wo, mrow,
              pin, pout, n, 0, 0, 0, (const RowQeMap<T>*)dev_tab);
    return q.rc;
}
int qe_rows_table_w(oa_plan* p, int n, const void* const* gx, const void* const* gy, const void* const* h, void* const* px, void* const* py,
                    const double* scales, void* dev_tab, int upload, int accumulate, int win, int wout, int mrow, long pl, long pk, hipStream_t st,
                    int my) {
    return p->dtype == OA_F32 ? qe_rows_table_impl<float>(p, n, gx, gy, h, px, py, scales, dev_tab, upload, accumulate, win, wout, mrow, st, pl, pk, my)
                              : qe_rows_table_impl<double>(p, n, gx, gy, h, px, py, scales, dev_tab, upload, accumulate, win, wout, mrow, st, pl, pk, my);
}
size_t qe_rows_table_entry_bytes(const oa_plan* p) { return p->dtype == OA_F32 ? sizeof(RowQeMap<float>) : sizeof(RowQeMap<double>); }
// estimator chains: `total` pieces (planes, scales as in qe_rows_table_w) grouped into nest estimators (first[e], count[e]); the
// device table holds the pieces followed by the 2 nest chain integers.  -1: this geometry's row stage is another kernel
template <typename T>
static int qe_rows_chain_impl(oa_plan* p, int nest, int total, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                              void* const* py, const double* scales, const int* first, const int* count, void* dev_tab, int upload, int win, int wout,
                              int mrow, hipStream_t st, long pin, long pout, int my) {
    HipLauncher q{st};
    auto f = coarse_view<T>(p, my);
    const int wi = f.clampw(win), wo = f.clampw(wout);
    if (mrow < 0) { mrow = Fft2dPlan<T>::row_grid_min(p->nx, wi, wo); if (2L * wi + wo > mrow) mrow = 0; }
    if (!f.rows_qe_is_pair(wi, wo, mrow) || (mrow > 0 && ilog2(mrow) > 12)) return -1;
    const size_t tb = ((size_t)total * sizeof(RowQeMap<T>) + 15) / 16 * 16;
    if (upload) {
        std::vector<char> buf(tb + 2 * (size_t)nest * sizeof(int));
        RowQeMap<T>* tab = reinterpret_cast<RowQeMap<T>*>(buf.data());
        const double fac = f.row_grid_scale(mrow);
        for (int i = 0; i < total; ++i)
            tab[i] = RowQeMap<T>{(const cx<T>*)gx[i], (const cx<T>*)gy[i], (const cx<T>*)h[i], (cx<T>*)px[i], (cx<T>*)py[i], (T)(scales[i] * fac)};
        int* ch = reinterpret_cast<int*>(buf.data() + tb);
        for (int e = 0; e < nest; ++e) { ch[2 * e] = first[e]; ch[2 * e + 1] = count[e]; }
        OA_HIP(hipMemcpyAsync(dev_tab, buf.data(), buf.size(), hipMemcpyHostToDevice, st));                 // pageable: staged before return
    }
    f.rows_qe(q, (const cx<T>*)gx[0], (const cx<T>*)gy[0], (const cx<T>*)h[0], (cx<T>*)px[0], (cx<T>*)py[0], (T)scales[0], 0, wi, wo, mrow, pin, pout,
              nest, 0, 0, 0, (const RowQeMap<T>*)dev_tab, 0, reinterpret_cast<const int*>((const char*)dev_tab + tb));
    return q.rc;
}
int qe_rows_chain_w(oa_plan* p, int nest, int total, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                    void* const* py, const double* scales, const int* first, const int* count, void* dev_tab, int upload, int win, int wout, int mrow,
                    long pl, long pk, hipStream_t st, int my) {
    return p->dtype == OA_F32 ? qe_rows_chain_impl<float>(p, nest, total, gx, gy, h, px, py, scales, first, count, dev_tab, upload, win, wout, mrow, st, pl, pk, my)
                              : qe_rows_chain_impl<double>(p, nest, total, gx, gy, h, px, py, scales, first, count, dev_tab, upload, win, wout, mrow, st, pl, pk, my);
}
int qe_rows_batch_w(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale, int win, int wout, int mrow,
                    long pl, long pk, hipStream_t st, int my, int nmaps, long in_moff, long h_moff, long out_moff) {
    return p->dtype == OA_F32 ? qe_rows_batch_impl<float>(p, gx, gy, h, px, py, scale, win, wout, mrow, st, pl, pk, my, nmaps, in_moff, h_moff, out_moff)
                              : qe_rows_batch_impl<double>(p, gx, gy, h, px, py, scale, win, wout, mrow, st, pl, pk, my, nmaps, in_moff, h_moff, out_moff);
}
int qe_legs_pass2_w(oa_plan* p, void* pool, int nplanes, long stride, int width, long pl, hipStream_t st, int my) {
    return p->dtype == OA_F32 ? legs_pass2_impl<float>(p, pool, nplanes, stride, width, st, pl, my)
                              : legs_pass2_impl<double>(p, pool, nplanes, stride, width, st, pl, my);
}
int qe_rows_w(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale, int accumulate,
              int win, int wout, int mrow, long pl, long pk, hipStream_t st, int my, int lr) {
    return p->dtype == OA_F32 ? qe_rows_impl<float>(p, gx, gy, h, px, py, scale, accumulate, win, wout, mrow, st, pl, pk, my, lr)
                              : qe_rows_impl<double>(p, gx, gy, h, px, py, scale, accumulate, win, wout, mrow, st, pl, pk, my, lr);
}
int qe_cols_div_w(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, int accumulate, int width, int rband,
                  long pk, hipStream_t st, int my, DivBinFuse* fuse) {
    return p->dtype == OA_F32 ? cols_div_impl<float>(p, pa, pb, Fn, out, accumulate, width, rband, st, pk, my, fuse)
                              : cols_div_impl<double>(p, pa, pb, Fn, out, accumulate, width, rband, st, pk, my, fuse);
}

}  // namespace oa

using namespace oa;

extern "C" {

int oa_qe_legs_cols(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* gx, void* gy, void* h,
                    int width, int rband, void* stream) {
    OA_REQUIRE(p && kX && kY && FG && FH && gx && gy && h, "oa_qe_legs_cols: NULL argument");
    OA_NEED_POW2(p, "oa_qe_legs_cols");
    OA_REQUIRE(p->have_laxes, "oa_qe_legs_cols: call oa_plan_set_laxes first");
    OA_REQUIRE(gx != kX && gy != kX && h != kX && gx != kY && gy != kY && h != kY, "oa_qe_legs_cols: outputs alias inputs");
    return p->dtype == OA_F32 ? legs_cols_impl<float>(p, kX, kY, FG, FH, gx, gy, h, width, rband, (hipStream_t)stream)
                              : legs_cols_impl<double>(p, kX, kY, FG, FH, gx, gy, h, width, rband, (hipStream_t)stream);
}

int oa_qe_map_legs_cols(oa_plan* p, const void* real_map, const void* FG, const void* FH, void* gx, void* gy, void* h,
                        int width, int rband, void* stream) {
    OA_REQUIRE(p && real_map && FG && FH && gx && gy && h, "oa_qe_map_legs_cols: NULL argument");
    OA_NEED_POW2(p, "oa_qe_map_legs_cols");
    OA_REQUIRE(p->have_laxes, "oa_qe_map_legs_cols: call oa_plan_set_laxes first");
    OA_REQUIRE(gx != gy && gx != h && gy != h, "oa_qe_map_legs_cols: outputs alias each other");
    return p->dtype == OA_F32 ? map_legs_cols_impl<float>(p, real_map, FG, FH, gx, gy, h, width, rband, (hipStream_t)stream)
                              : map_legs_cols_impl<double>(p, real_map, FG, FH, gx, gy, h, width, rband, (hipStream_t)stream);
}

int oa_qe_cols_div(oa_plan* p, const void* px_rows, const void* py_rows, const void* Fnorm, void* out, int accumulate,
                   int width, int rband, void* stream) {
    OA_REQUIRE(p && px_rows && py_rows && Fnorm && out, "oa_qe_cols_div: NULL argument");
    OA_NEED_POW2(p, "oa_qe_cols_div");
    OA_REQUIRE(p->have_laxes, "oa_qe_cols_div: call oa_plan_set_laxes first");
    return p->dtype == OA_F32 ? cols_div_impl<float>(p, px_rows, py_rows, Fnorm, out, accumulate, width, rband, (hipStream_t)stream)
                              : cols_div_impl<double>(p, px_rows, py_rows, Fnorm, out, accumulate, width, rband, (hipStream_t)stream);
}

int oa_fft_cols(oa_plan* p, const void* hc_in, void* hc_out, int inverse, double scale, int width, void* stream) {
    OA_REQUIRE(p && hc_in && hc_out, "oa_fft_cols: NULL argument");
    OA_NEED_POW2(p, "oa_fft_cols");
    OA_REQUIRE(hc_in != hc_out, "oa_fft_cols: in-place not supported");
    return p->dtype == OA_F32 ? cols_impl<float>(p, hc_in, hc_out, inverse, scale, width, (hipStream_t)stream)
                              : cols_impl<double>(p, hc_in, hc_out, inverse, scale, width, (hipStream_t)stream);
}

int oa_qe_rows(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale,
               int accumulate, int win, int wout, int mrow, void* stream) {
    OA_REQUIRE(p && gx && gy && h && px && py, "oa_qe_rows: NULL argument");
    OA_NEED_POW2(p, "oa_qe_rows");
    return p->dtype == OA_F32 ? qe_rows_impl<float>(p, gx, gy, h, px, py, scale, accumulate, win, wout, mrow, (hipStream_t)stream)
                              : qe_rows_impl<double>(p, gx, gy, h, px, py, scale, accumulate, win, wout, mrow, (hipStream_t)stream);
}

int oa_fft_pass(oa_plan* p, int pass_id, const void* in, void* out, int width, void* stream) {
    OA_REQUIRE(p && in && out, "oa_fft_pass: NULL argument");
    OA_NEED_POW2(p, "oa_fft_pass");
    return p->dtype == OA_F32 ? pass_impl<float>(p, pass_id, in, out, width, (hipStream_t)stream)
                              : pass_impl<double>(p, pass_id, in, out, width, (hipStream_t)stream);
}

int oa_fft_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, int width, int rband, void* stream) {
    OA_REQUIRE(p && real_in && hc_out, "oa_fft_r2c: NULL argument");
    OA_REQUIRE(real_in != hc_out, "oa_fft_r2c: in-place not supported");
    if (p->mixed) return mixed_r2c(p, real_in, hc_out, scale, (hipStream_t)stream);     // width / rband hints do not apply
    if (!p->pow2) return czt_r2c(p, real_in, hc_out, scale, (hipStream_t)stream);
    return p->dtype == OA_F32 ? r2c_impl<float>(p, real_in, hc_out, scale, width, rband, (hipStream_t)stream)
                              : r2c_impl<double>(p, real_in, hc_out, scale, width, rband, (hipStream_t)stream);
}

int oa_fft_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, int width, void* stream) {
    OA_REQUIRE(p && hc_in && real_out, "oa_fft_c2r: NULL argument");
    OA_REQUIRE(hc_in != real_out, "oa_fft_c2r: in-place not supported");
    if (p->mixed) return mixed_c2r(p, hc_in, real_out, scale, (hipStream_t)stream);
    if (!p->pow2) return czt_c2r(p, hc_in, real_out, scale, (hipStream_t)stream);
    return p->dtype == OA_F32 ? c2r_impl<float>(p, hc_in, real_out, scale, width, (hipStream_t)stream)
                              : c2r_impl<double>(p, hc_in, real_out, scale, width, (hipStream_t)stream);
}

int oa_fft_c2r_windowed(oa_plan* p, const void* hc_in, void* real_out, double scale, const void* window_real, void* stream) {
    OA_REQUIRE(p && hc_in && real_out && window_real, "oa_fft_c2r_windowed: NULL argument");
    OA_REQUIRE(hc_in != real_out && window_real != real_out, "oa_fft_c2r_windowed: in-place not supported");
    OA_NEED_POW2(p, "oa_fft_c2r_windowed");
    return p->dtype == OA_F32 ? c2r_impl<float>(p, hc_in, real_out, scale, 0, (hipStream_t)stream, window_real)
                              : c2r_impl<double>(p, hc_in, real_out, scale, 0, (hipStream_t)stream, window_real);
}

int oa_fft_c2c(oa_plan* p, const void* full_in, void* full_out, int inverse, double scale, void* stream) {
    OA_REQUIRE(p && full_in && full_out, "oa_fft_c2c: NULL argument");
    OA_REQUIRE(full_in != full_out, "oa_fft_c2c: in-place not supported");
    if (p->mixed) return mixed_c2c(p, full_in, full_out, inverse, scale, (hipStream_t)stream);
    if (!p->pow2) return czt_c2c(p, full_in, full_out, inverse, scale, (hipStream_t)stream);
    return p->dtype == OA_F32 ? c2c_impl<float>(p, full_in, full_out, inverse, scale, (hipStream_t)stream)
                              : c2c_impl<double>(p, full_in, full_out, inverse, scale, (hipStream_t)stream);
}

}  // extern "C"
