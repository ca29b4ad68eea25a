// Host side of the mixed-radix 2-D FFT for map sides 2^a 3^b 5^c that are not powers of two (fft_mixed.hpp): plan tables and the three
// transforms behind oa_fft_r2c / oa_fft_c2r / oa_fft_c2c on such plans (reference: FourierCalc.fft / .ifft, maps.py:1609-1636, on the
// notebooks' own 600^2, 1200^2, 2400^2 patches).
#include <cmath>
#include <vector>
#include "fft_launch.hpp"
#include "fft_mixed.hpp"

namespace oa {

template <typename T>
__global__ __launch_bounds__(256) void mr_row_kernel(MrRowArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    mr_row_body<T>(c, a);
}
template <typename T>
__global__ __launch_bounds__(256) void mr_col_kernel(MrColArgs<T> a) {
    GpuCtx c{oa_dyn_smem};
    mr_col_body<T>(c, a);
}

bool mixed_sides_ok(int ny, int nx) { return ny % 2 == 0 && nx % 2 == 0 && mixed_ok(ny) && mixed_ok(nx) && mixed_ok(nx / 2); }

template <typename T>
static int upload(void** dst, int N, int extra = 0) {
    std::vector<cx<T>> t((size_t)N + extra);
    const long double tau = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N + extra; ++k) {
        const long double x = tau * (long double)k / (long double)N;
        t[(size_t)k] = mk<T>((T)cosl(x), (T)(-sinl(x)));
    }
    OA_HIP(hipMalloc(dst, t.size() * sizeof(cx<T>)));
    OA_HIP(hipMemcpy(*dst, t.data(), t.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    return 0;
}
template <typename T>
static int setup_t(oa_plan* p) {
    if (int rc = upload<T>(&p->mr_twx, p->nx, 1)) return rc;       // W_nx^k: complex rows, and the (un)tangle factors of real rows (k <= nx / 2)
    if (int rc = upload<T>(&p->mr_twxh, p->nx / 2)) return rc;     // W_(nx/2)^k: packed real rows
    if (int rc = upload<T>(&p->mr_twy, p->ny)) return rc;          // W_ny^k: columns
    return plan_ensure_scratch(p, (size_t)p->ny * p->kp * sizeof(cx<T>));
}
int mixed_setup(oa_plan* p) { return p->dtype == OA_F32 ? setup_t<float>(p) : setup_t<double>(p); }
void mixed_release(oa_plan* p) {
    void* bufs[] = {p->mr_twx, p->mr_twxh, p->mr_twy};
    for (void* b : bufs) if (b) (void)hipFree(b);
}

template <typename T>
static int rows(oa_plan* p, int mode, const void* in, long ipitch, void* out, long opitch, int N, const void* tw, double scale, hipStream_t st,
                const T* dlx = nullptr, int dpow0 = 0, int dcol_b = 0, int nz = 1, long out_zoff = 0) {
    MrRowArgs<T> a{};
    a.in = in; a.out = out; a.in_pitch = ipitch; a.out_pitch = opitch; a.N = N; a.f = mixed_factor(N);
    a.tw = (const cx<T>*)tw; a.tw2 = (const cx<T>*)p->mr_twx; a.scale = (T)scale; a.mode = mode;
    a.dlx = dlx; a.dpow0 = dpow0; a.dcol_b = dcol_b; a.out_zoff = out_zoff;
    int rc = 0;
    launch_go(rc, st, mr_row_kernel<T>, dim3(p->ny, nz), 256, 2 * ((size_t)N + 1) * sizeof(cx<T>), a);
    return rc;
}
// columns per tile: the largest power of two whose two [N][C] buffers fit 96 KB
template <typename T> static int col_logc(int N) {
    int lc = 0;
    while (lc < 4 && 2 * ((size_t)N << (lc + 1)) * sizeof(cx<T>) <= 96 * 1024) ++lc;
    return lc;
}
template <typename T>
static int cols(oa_plan* p, const void* in, long ipitch, void* out, long opitch, int width, bool inverse, double scale, hipStream_t st,
                const T* dly = nullptr, int dpow = 0) {
    MrColArgs<T> a{};
    a.dly = dly; a.dpow = dpow;
    a.in = (const cx<T>*)in; a.out = (cx<T>*)out; a.in_pitch = ipitch; a.out_pitch = opitch; a.N = p->ny; a.width = width;
    a.logC = col_logc<T>(p->ny); a.f = mixed_factor(p->ny); a.tw = (const cx<T>*)p->mr_twy; a.scale = (T)scale; a.inverse = inverse ? 1 : 0;
    const int C = 1 << a.logC;
    int rc = 0;
    launch_go(rc, st, mr_col_kernel<T>, dim3((width + C - 1) / C), 256, 2 * ((size_t)p->ny << a.logC) * sizeof(cx<T>), a);
    return rc;
}

template <typename T>
static int r2c_t(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st) {
    if (int rc = rows<T>(p, MR_R2C, real_in, p->nx, hc_out, p->kp, p->nx / 2, p->mr_twxh, 1.0, st)) return rc;
    return cols<T>(p, hc_out, p->kp, hc_out, p->kp, p->nx / 2 + 1, false, scale, st);
}
template <typename T>
static int c2r_t(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st) {
    if (int rc = cols<T>(p, hc_in, p->kp, p->scratch, p->kp, p->nx / 2 + 1, true, 1.0, st)) return rc;
    return rows<T>(p, MR_C2R, p->scratch, p->kp, real_out, p->nx, p->nx / 2, p->mr_twxh, scale, st);
}
template <typename T>
static int c2c_t(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st) {
    if (int rc = rows<T>(p, inverse ? MR_C2C_I : MR_C2C_F, in, p->nx, out, p->nx, p->nx, p->mr_twx, 1.0, st)) return rc;
    return cols<T>(p, out, p->nx, out, p->nx, p->nx, inverse != 0, scale, st);
}
// flat-sky Taylor lensing, FFT part, on these sides (the mixed-radix counterpart of lens_derivs_impl, fft.hip): per map and
// y-derivative order b ONE inverse column transform of (i ly)^b k (factor at the load) onto the one-plane hc pool, then ONE row launch
// that takes every x-derivative (i lx)^a at its load (grid y = a) into the real pool
template <typename T>
static int lens_derivs_t(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd, hipStream_t st,
                         const void* hc_in, long hc_stride, double hc_scale) {
    const long hcp = (long)p->ny * p->kp, rp = (long)p->ny * p->nx;
    const cx<T>* src = hc_in ? (const cx<T>*)hc_in : (const cx<T>*)k0;
    const long sstride = hc_in ? hc_stride : hcp;
    const int d00 = hc_in ? 1 : 0;
    if (!hc_in)
        for (int m = 0; m < nmaps; ++m)
            if (int rc = r2c_t<T>(p, (const T*)real_in + (long)m * in_stride, (cx<T>*)k0 + (long)m * hcp, 1.0, st)) return rc;
    const int order = (int)((std::sqrt(8.0 * (nd + 1) + 1.0) - 1.0) / 2.0 + 0.5);      // nd = order (order + 1) / 2 - 1
    const double cscale = hc_in ? hc_scale : 1.0 / ((double)p->ny * p->nx);
    for (int m = 0; m < nmaps; ++m)
        for (int b = 0; b < order; ++b) {
            const int a0 = (b == 0 && !d00) ? 1 : 0, na = order - b - a0;
            if (na <= 0) continue;
            if (int rc = cols<T>(p, src + (long)m * sstride, p->kp, hc_pool, p->kp, p->nx / 2 + 1, true, 1.0, st, (const T*)p->lyd, b)) return rc;
            if (int rc = rows<T>(p, MR_C2R, hc_pool, p->kp, (T*)real_pool + ((long)m * (nd + d00) + d00) * rp, p->nx, p->nx / 2, p->mr_twxh, cscale, st,
                                 (const T*)p->lxd, a0, b, na, rp)) return rc;
        }
    return 0;
}
int mixed_lens_derivs(oa_plan* p, int nmaps, const void* real_in, long in_stride, void* k0, void* hc_pool, void* real_pool, int nd, hipStream_t st,
                      const void* hc_in, long hc_stride, double hc_scale) {
    return p->dtype == OA_F32 ? lens_derivs_t<float>(p, nmaps, real_in, in_stride, k0, hc_pool, real_pool, nd, st, hc_in, hc_stride, hc_scale)
                              : lens_derivs_t<double>(p, nmaps, real_in, in_stride, k0, hc_pool, real_pool, nd, st, hc_in, hc_stride, hc_scale);
}
int mixed_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? r2c_t<float>(p, real_in, hc_out, scale, st) : r2c_t<double>(p, real_in, hc_out, scale, st);
}
int mixed_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? c2r_t<float>(p, hc_in, real_out, scale, st) : c2r_t<double>(p, hc_in, real_out, scale, st);
}
int mixed_c2c(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? c2c_t<float>(p, in, out, inverse, scale, st) : c2c_t<double>(p, in, out, inverse, scale, st);
}

}  // namespace oa
