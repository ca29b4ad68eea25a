// Fused leg-filter + inverse column pass-1 kernel (own translation unit, see fft_launch.hpp).
#define OA_NO_PK_ASM 1
#include "fft_launch.hpp"

namespace oa {

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), fused_col_waves_per_eu<T>()) void col_legs_kernel(ColLegsArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_legs_body<T, SEQ>(c, a);
}


template <typename T>
int launch_col_legs(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
    int rc = 0;
    const bool ok = dispatch_seq(logL, [&](auto seq) {
        using S = decltype(seq);
        if constexpr (seq_logl<S>() <= 8) {
            if (nt > col_maxnt<S>()) { rc = fail("fft: column workgroup size exceeds its launch bound"); return; }
            launch_go(rc, st, col_legs_kernel<T, S>, dim3(gx, gy, a.batch ? a.batch : (a.split ? (a.zcount ? a.zcount : 3) : 1)), nt, smem, a);
        } else {
            rc = fail("fft: unsupported column sub-length");
        }
    });
    if (!ok && !rc) rc = fail("fft: unsupported column length");
    return rc;
}

// single-pass filtered inverse column transform of a batch of leg planes on short (coarse-grid) columns: a whole column of an
// 8- / 16-column (f64: 4- / 8-column) tile in 128 KB of LDS, 1024 (f64: 512) threads; grid z = leg plane
template <typename T, class SEQ, int LOGC>
__global__ __launch_bounds__((sizeof(T) == 8 ? 512 : 1024), (sizeof(T) == 8 ? 2 : 4)) void col_legs_sp_kernel(ColLegsArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_legs_body<T, SEQ, GpuCtx, LOGC>(c, a);
}
template <typename T>
int launch_col_legs_sp(hipStream_t st, int gx, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
    int rc = 0;
    constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
    if (!a.batch || nt != (sizeof(T) == 4 ? 1024 : 512)) return fail("fft: col_legs_sp is the batch-mode kernel of 1024 / 512 threads");
    if (logL == 11) launch_go(rc, st, col_legs_sp_kernel<T, Seq<16, 16, 8>, lc11>, dim3(gx, 1, a.batch), nt, smem, a);
    else if (logL == 10) launch_go(rc, st, col_legs_sp_kernel<T, Seq<16, 16, 4>, lc10>, dim3(gx, 1, a.batch), nt, smem, a);
    else rc = fail("fft: col_legs_sp handles 1024- and 2048-row column grids");
    return rc;
}
template int launch_col_legs_sp<float>(hipStream_t, int, int, size_t, int, const ColLegsArgs<float>&);
template int launch_col_legs_sp<double>(hipStream_t, int, int, size_t, int, const ColLegsArgs<double>&);

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), fused_col_waves_per_eu<T>()) void col_fwdlegs_kernel(ColFwdLegsArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_fwdlegs_body<T, SEQ>(c, a);
}

template <typename T>
int launch_col_fwdlegs(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsArgs<T>& a) {
    int rc = 0;
    const bool ok = dispatch_seq(logL, [&](auto seq) {
        using S = decltype(seq);
        if constexpr (seq_logl<S>() >= 5 && seq_logl<S>() <= 7) {
            if (nt > col_maxnt<S>()) { rc = fail("fft: column workgroup size exceeds its launch bound"); return; }
            launch_go(rc, st, col_fwdlegs_kernel<T, S>, dim3(gx, gy), nt, smem, a);
        } else {
            rc = fail("fft: unsupported column sub-length");
        }
    });
    if (!ok && !rc) rc = fail("fft: unsupported column length");
    return rc;
}

template <typename T, class SEQ>
__global__ __launch_bounds__(col_maxnt<SEQ>(), fused_col_waves_per_eu<T>()) void col_fwdlegs_cg_kernel(ColFwdLegsCgArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    col_fwdlegs_cg_body<T, SEQ>(c, a);
}

template <typename T>
int launch_col_fwdlegs_cg(hipStream_t st, int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsCgArgs<T>& a, int gz) {
    int rc = 0;
    if (logL != 6) return fail("fft: col_fwdlegs_cg is built for 64-point forward sub-lengths");
    using S = Seq<16, 4>;
    if (nt != (1 << (6 + COL_LOGC)) / EPT) return fail("fft: col_fwdlegs_cg launched with the wrong workgroup size");
    launch_go(rc, st, col_fwdlegs_cg_kernel<T, S>, dim3(gx, gy, gz), nt, smem, a);
    return rc;
}
template int launch_col_fwdlegs_cg<float>(hipStream_t, int, int, int, size_t, int, const ColFwdLegsCgArgs<float>&, int);
template int launch_col_fwdlegs_cg<double>(hipStream_t, int, int, int, size_t, int, const ColFwdLegsCgArgs<double>&, int);

template int launch_col_fwdlegs<float>(hipStream_t, int, int, int, size_t, int, const ColFwdLegsArgs<float>&);
template int launch_col_fwdlegs<double>(hipStream_t, int, int, int, size_t, int, const ColFwdLegsArgs<double>&);
template int launch_col_legs<float>(hipStream_t, int, int, int, size_t, int, const ColLegsArgs<float>&);
template int launch_col_legs<double>(hipStream_t, int, int, int, size_t, int, const ColLegsArgs<double>&);

}  // namespace oa
