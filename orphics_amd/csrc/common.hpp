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

// Request to fold the radial histogram of |kappa|^2 and the moment update (n += maps, S += b, C += b b^T, b = sums / mode
// counts) into the single-pass divergence launch (fft_divbin.hpp).  Filled by pipeline.hip, honoured by HipLauncher::col_div_sp
// when the geometry runs that kernel (`done` set); otherwise the caller runs bin_power_moments as before.  Device pointers.
struct DivBinFuse {
    const int32_t* ids; long ipitch;      // radial ids of the full-resolution half plane, row pitch
    const int32_t* ids_t;                 // tile-major copy on the coarse grid of the single-pass divergence launch, or nullptr
    const void* fn_t;                     // ... and of Fnorm (handed to ColDivArgs::Fn_t by the launcher)
    int tab_logc, tab_rows;               // tile shape the two copies were made for: log2 columns per tile, coarse rows
    double pnorm; int nids, nxh;
    double* part;                         // [maps][workgroups][nids] partial sums (capacity part_cap doubles)
    long part_cap;
    double* sums;                         // [maps][nids]
    unsigned* ticket;                     // zero before the launch; reset by the last workgroup
    const int64_t* mcounts; int64_t* n; double* S; double* C;
    int store;                            // != 0: kappa is also written to `out`
    int gx;                               // (set by the launcher) workgroups per map
    bool done;                            // (set by the launcher)
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
    // sides that are not powers of two: chirp-z (Bluestein) state, see czt.hip.  pow2 == false then.
    bool pow2;
    oa_plan* inner;            // (My, Mx) power-of-two plan
    int My, Mx;
    void* chirp_y; void* chirp_x;            // cx<T>[ny], cx<T>[nx]
    void* cz_bhat; void* cz_a; void* cz_f;   // cx<T>[My*Mx]: chirp-kernel transform, two work planes
    void* cz_full;                           // cx<T>[ny*nx]
    bool mixed;                              // sides 2^a 3^b 5^c: mixed-radix transforms (mixed.hip) instead of the chirp-z path
    void* mr_twx; void* mr_twxh; void* mr_twy;   // cx<T>[nx + 1], [nx / 2], [ny]
    void* pipe;                              // oa::Pipeline* (pipeline.hip): filters, bins, work planes of the one-call entries
    void* rq8c[5];                           // per-thread constants of the fused row stage's grids of 1024, 1536, 2048, 4096, 8192 points (fft_rowqe8.hpp: RQ8_NGRIDS)
    void* tw_y_small[16];                    // COLUMN GRID: cx<T>[my] = W_my^k for my = 2^i (made on first use, kept: estimators
                                             // with different row bands may alternate on one plan)
};

namespace oa {
int plan_ensure_scratch(oa_plan* p, size_t bytes);
int czt_setup(oa_plan* p);
void czt_release(oa_plan* p);
int czt_c2c(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st);
int czt_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st);
int czt_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st);
bool mixed_sides_ok(int ny, int nx);
int mixed_setup(oa_plan* p);
void mixed_release(oa_plan* p);
int mixed_c2c(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st);
int mixed_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st);
int mixed_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st);
int mixed_lens_derivs(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd, hipStream_t st,
                      const void* hc_in, long hc_stride, double hc_scale);
void pipeline_release(oa_plan* p);
// fused estimator passes on the plan's compact work planes (fft.hip)
long work_pitch(const oa_plan* p, int w);
int bin_power_moments(int dtype, const void* k, double norm, const int32_t* ids, long n, int nids, long hp, int nxh, double* sums,
                      int64_t* counts, void* scratch, int active_cols, int active_rows, unsigned* ticket,
                      const int64_t* mcounts, int64_t* mn, double* S, double* C, hipStream_t st, int nbatch = 1, long kstride = 0);
int stack_add_region(int dtype, const void* x, double* acc, int ny, long kp, int w, int rb, hipStream_t st, int nbatch = 1, long xstride = 0);
// my > 0: COLUMN GRID -- legs, row stage and divergence run on my < ny rows (plan_ensure_col_grid(p, my) first)
int plan_ensure_col_grid(oa_plan* p, int my);
// lr > 0 (qe_rsplit_lr): R-SPLIT path -- row R2C with the first radix-R column butterfly, then one single-pass column kernel;
// the leg planes are then in the R-LAYOUT and qe_rows_w must be told so (same lr)
int qe_rsplit_lr(const oa_plan* p, int my, int width, int wout, int mrow);
// fgh (lr > 0 only): the packed (FG, FH) table of this binding made by qe_fband_pack_w (qe_fband_table_entries complex entries), or
// nullptr: the column kernel reads the filter planes
int qe_map_legs_cols_w(oa_plan* p, const void* map, const void* FG, const void* FH, void* gx, void* gy, void* h, int width,
                       int rband, long pl, hipStream_t st, int stages = 7, int my = 0, int lr = 0, const void* fgh = nullptr);
long qe_fband_table_entries(const oa_plan* p, int width, int my);
int qe_fband_pack_w(oa_plan* p, const void* FG, const void* FH, void* out, int width, int rband, int my, hipStream_t st);
int qe_legs_cols_w(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* gx, void* gy, void* h,
                   int width, int rband, long pl, hipStream_t st, int my = 0);
// flat-sky Taylor lensing, FFT part: R2C of nmaps maps, then all nmaps * nd derivative fields inverse-transformed in three launches
// tile-major copy of a full-pitch real / int32 hc-layout plane on the coarse grid of `rows` rows: dst[(tile * rows + k) * C + c] =
// src[(k + (k >= rows / 2 ? ny - rows : 0)) * kp + tile * C + c], zero beyond `width` columns (elem_bytes 4 or 8)
int pack_tiles(const oa_plan* p, const void* src, void* dst, int rows, int logc, int width, int elem_bytes, hipStream_t st);
int div_tile_logc(const oa_plan* p, int rows);          // log2 columns per tile of the single-pass divergence launch on `rows` coarse rows
int qe_lens_derivs_w(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd, hipStream_t st,
                     const void* hc_in = nullptr, long hc_stride = 0, double hc_scale = 1.0);
// windowed simulation front end: hc spectrum -> inverse columns -> fused C2R x window -> R2C rows onto the plan's scratch plane
// (then qe_map_legs_cols_w with stages = 6, lr = 0)
int qe_windowed_rows_w(oa_plan* p, const void* hc_in, void* cols_tmp, const void* window, int width, long pl, double scale, hipStream_t st,
                       void* out = nullptr);
int qe_fwd_cols_batch_w(oa_plan* p, const void* in, long pin, void* out, int B, long in_moff, long out_moff, int width, int rband, hipStream_t st);
// one filtered field (subset 1: H plane into a; 2: gradient pair into a, b), inverse pass 1 only; then pass 2 over a pool of planes
int qe_legs_subset_w(oa_plan* p, const void* src, const void* F, void* a, void* b, int subset, int width, int rband, long pl,
                     hipStream_t st, int my = 0);
// divergence of nmaps estimators in one launch: product planes (A_e, B_e adjacent: B_e = A_e + in_moff / 2) in_moff apart, Fn planes
// fn_moff apart, outputs out_moff apart (offsets in elements of the respective plane type); tmp: 2 nmaps compact planes
int qe_cols_div_batch_w(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, void* tmp, int nmaps, long in_moff,
                        long fn_moff, long out_moff, int width, int rband, long pk, hipStream_t st, int my = 0, DivBinFuse* fuse = nullptr);
int sum_region(int dtype, const void* parts, long part_stride, int nparts, void* out, int accumulate, int ny, long kp, int w, int rb,
               hipStream_t st);
// all leg planes of several estimators in one inverse pass-1 launch (ColLegsArgs::batch); offsets in complex elements
int qe_legs_batch_w(oa_plan* p, const void* src0, long off1, long off2, unsigned long long srcsel, const void* const* ftab,
                    int ngrad, int nh, void* pool, long ostride, int width, int rband, long pl, hipStream_t st, int my = 0, int selbits = 2,
                    int* finished = nullptr);      // *finished = 1: single pass (col_legs_sp), the planes need no inverse pass 2
// row stage of nmaps maps in one launch; -1: this geometry's row stage is not the two-rows-per-transform kernel
int qe_rows_batch_w(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale, int win, int wout, int mrow,
                    long pl, long pk, hipStream_t st, int my, int nmaps, long in_moff, long h_moff, long out_moff);
int qe_rows_table_w(oa_plan* p, int n, const void* const* gx, const void* const* gy, const void* const* h, void* const* px, void* const* py,
                    const double* scales, void* dev_tab, int upload, int accumulate, int win, int wout, int mrow, long pl, long pk, hipStream_t st,
                    int my);
size_t qe_rows_table_entry_bytes(const oa_plan* p);
int qe_rows_chain_w(oa_plan* p, int nest, int total, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                    void* const* py, const double* scales, const int* first, const int* count, void* dev_tab, int upload, int win, int wout, int mrow,
                    long pl, long pk, hipStream_t st, int my);
int grf_hc_band_batch(oa_plan* p, uint64_t seed, uint64_t stream_id, int nreal, const void* covsqrt_hc, void* hc_out, long zstride,
                      int width, int rband, hipStream_t stream);
int qe_legs_pass2_w(oa_plan* p, void* pool, int nplanes, long stride, int width, long pl, hipStream_t st, int my = 0);
int qe_rows_w(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale, int accumulate,
              int win, int wout, int mrow, long pl, long pk, hipStream_t st, int my = 0, int lr = 0);
int qe_cols_div_w(oa_plan* p, const void* pa, const void* pb, const void* Fn, void* out, int accumulate, int width, int rband,
                  long pk, hipStream_t st, int my = 0, DivBinFuse* fuse = nullptr);
// two maps at once (fft.hip); -1 = not available for this geometry
int qe_tt_pair_w(oa_plan* p, const void* map0, const void* map1, const void* FG, const void* FH, const void* Fn, void* c0, void* c1,
                 void* c2, void* g0, void* g1, void* out0, void* out1, int wl, int wk, int rl, int rk, int mrow, int my, long pl, long pk,
                 hipStream_t st, DivBinFuse* fuse = nullptr, const void* fgh = nullptr);
}
#define OA_NEED_POW2(p, what) \
    OA_REQUIRE((p)->pow2, what ": needs power-of-two map sides (other sizes: oa_fft_r2c / oa_fft_c2r / oa_fft_c2c and the modular oa_qe_legs / oa_mul_real / oa_qe_div calls)")
