// Map sides that are not powers of two (the reference notebooks use 600^2, 750^2, 2400^2 patches): the 2-D DFT is
// evaluated exactly as a chirp-z (Bluestein) convolution on an inner power-of-two plan,
//   X[k,l] = w_y[k] w_x[l] * sum_{m,n} (x[m,n] w_y[m] w_x[n]) conj(w_y)[k-m] conj(w_x)[l-n],   w_N[j] = e^{-i pi j^2 / N},
// i.e. pre-multiply, one forward and one inverse C2C transform of size (My, Mx) >= (2ny-1, 2nx-1) around a
// multiplication with the (precomputed) transform of the chirp kernel, post-multiply.  This path exists for drop-in
// completeness (FourierCalc / filter_map / MapGen / the modular estimator on any even-sided map); the fused
// estimator kernels and every performance figure of this library are for power-of-two sides.
#include <cmath>
#include <vector>
#include "common.hpp"

namespace oa {

template <typename T>
__global__ __launch_bounds__(256) void czt_pre_kernel(const cx<T>* __restrict__ x, int ny, int nx, const cx<T>* __restrict__ wy,
                                                      const cx<T>* __restrict__ wx, cx<T>* __restrict__ a, int My, int Mx,
                                                      int conj_in) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= Mx) return;
    cx<T> v = mk<T>((T)0, (T)0);
    if (m < ny && n < nx) {
        cx<T> s = x[(long)m * nx + n];
        if (conj_in) s = conj(s);
        v = s * (wy[m] * wx[n]);
    }
    a[(long)m * Mx + n] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void czt_post_kernel(const cx<T>* __restrict__ a, int My, int Mx, const cx<T>* __restrict__ wy,
                                                       const cx<T>* __restrict__ wx, cx<T>* __restrict__ out, int ny, int nx,
                                                       T scale, int conj_out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (l >= nx) return;
    cx<T> v = a[(long)k * Mx + l] * (wy[k] * wx[l]);
    if (conj_out) v = conj(v);
    out[(long)k * nx + l] = v * scale;
}

// circular chirp kernel b[m,n] = conj(w_y[|m|]) conj(w_x[|n|]) on the (My, Mx) grid, zero where |m| >= ny or |n| >= nx
template <typename T>
__global__ __launch_bounds__(256) void czt_kernel_fill(const cx<T>* __restrict__ wy, const cx<T>* __restrict__ wx, int ny, int nx,
                                                       cx<T>* __restrict__ b, int My, int Mx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= Mx) return;
    const int am = m < ny ? m : (My - m < ny ? My - m : -1);
    const int an = n < nx ? n : (Mx - n < nx ? Mx - n : -1);
    cx<T> v = mk<T>((T)0, (T)0);
    if (am >= 0 && an >= 0) v = conj(wy[am] * wx[an]);
    b[(long)m * Mx + n] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void cmul_kernel(cx<T>* __restrict__ a, const cx<T>* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = a[i] * b[i];
}

template <typename T>
__global__ __launch_bounds__(256) void real_to_cx_kernel(const T* __restrict__ x, cx<T>* __restrict__ z, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) z[i] = mk<T>(x[i], (T)0);
}
template <typename T>
__global__ __launch_bounds__(256) void cx_to_real_kernel(const cx<T>* __restrict__ z, T* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = z[i].x;
}

static int next_pow2(int v) {
    int p = 32;
    while (p < v) p <<= 1;
    return p;
}

// w_N[j] = exp(-i pi j^2 / N) with the phase reduced exactly: j^2 mod 2N in integers
template <typename T>
static std::vector<cx<T>> make_chirp(int N) {
    std::vector<cx<T>> w((size_t)N);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (long j = 0; j < N; ++j) {
        const long r = (j * j) % (2L * N);
        const long double a = pi * (long double)r / (long double)N;
        w[(size_t)j].x = (T)cosl(a);
        w[(size_t)j].y = (T)(-sinl(a));
    }
    return w;
}

template <typename T>
static int czt_setup_t(oa_plan* p) {
    const int My = next_pow2(2 * p->ny - 1), Mx = next_pow2(2 * p->nx - 1);
    OA_REQUIRE(My <= 32768 && Mx <= 32768, "oa_plan_create: non power-of-two sides must be <= 16384");
    p->My = My; p->Mx = Mx;
    if (int rc = oa_plan_create(My, Mx, p->dtype, &p->inner)) return rc;
    auto wy = make_chirp<T>(p->ny), wx = make_chirp<T>(p->nx);
    OA_HIP(hipMalloc(&p->chirp_y, wy.size() * sizeof(cx<T>)));
    OA_HIP(hipMalloc(&p->chirp_x, wx.size() * sizeof(cx<T>)));
    OA_HIP(hipMemcpy(p->chirp_y, wy.data(), wy.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    OA_HIP(hipMemcpy(p->chirp_x, wx.data(), wx.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    const size_t big = (size_t)My * Mx * sizeof(cx<T>);
    OA_HIP(hipMalloc(&p->cz_bhat, big));
    OA_HIP(hipMalloc(&p->cz_a, big));
    OA_HIP(hipMalloc(&p->cz_f, big));
    OA_HIP(hipMalloc(&p->cz_full, (size_t)p->ny * p->nx * sizeof(cx<T>)));
    const dim3 grid((Mx + 255) / 256, My);
    hipLaunchKernelGGL(czt_kernel_fill<T>, grid, dim3(256), 0, 0, (const cx<T>*)p->chirp_y, (const cx<T>*)p->chirp_x, p->ny, p->nx,
                       (cx<T>*)p->cz_a, My, Mx);
    OA_LAUNCH_CHECK();
    if (int rc = oa_fft_c2c(p->inner, p->cz_a, p->cz_bhat, 0, 1.0, nullptr)) return rc;
    OA_HIP(hipDeviceSynchronize());
    return 0;
}

int czt_setup(oa_plan* p) { return p->dtype == OA_F32 ? czt_setup_t<float>(p) : czt_setup_t<double>(p); }

void czt_release(oa_plan* p) {
    if (p->inner) oa_plan_destroy(p->inner);
    void* bufs[] = {p->chirp_y, p->chirp_x, p->cz_bhat, p->cz_a, p->cz_f, p->cz_full};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
}

// full (ny,nx) complex -> full complex, out may equal in
template <typename T>
static int czt_c2c_t(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st) {
    const int My = p->My, Mx = p->Mx;
    const dim3 gbig((Mx + 255) / 256, My), gsmall((p->nx + 255) / 256, p->ny);
    hipLaunchKernelGGL(czt_pre_kernel<T>, gbig, dim3(256), 0, st, (const cx<T>*)in, p->ny, p->nx, (const cx<T>*)p->chirp_y,
                       (const cx<T>*)p->chirp_x, (cx<T>*)p->cz_a, My, Mx, inverse ? 1 : 0);
    OA_LAUNCH_CHECK();
    if (int rc = oa_fft_c2c(p->inner, p->cz_a, p->cz_f, 0, 1.0, st)) return rc;
    const long nb = (long)My * Mx;
    hipLaunchKernelGGL(cmul_kernel<T>, dim3(flat_grid(nb)), dim3(256), 0, st, (cx<T>*)p->cz_f, (const cx<T>*)p->cz_bhat, nb);
    OA_LAUNCH_CHECK();
    if (int rc = oa_fft_c2c(p->inner, p->cz_f, p->cz_a, 1, 1.0 / ((double)My * (double)Mx), st)) return rc;
    hipLaunchKernelGGL(czt_post_kernel<T>, gsmall, dim3(256), 0, st, (const cx<T>*)p->cz_a, My, Mx, (const cx<T>*)p->chirp_y,
                       (const cx<T>*)p->chirp_x, (cx<T>*)out, p->ny, p->nx, (T)scale, inverse ? 1 : 0);
    OA_LAUNCH_CHECK();
    return 0;
}

int czt_c2c(oa_plan* p, const void* in, void* out, int inverse, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? czt_c2c_t<float>(p, in, out, inverse, scale, st) : czt_c2c_t<double>(p, in, out, inverse, scale, st);
}

template <typename T>
static int czt_r2c_t(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st) {
    const long n = (long)p->ny * p->nx;
    hipLaunchKernelGGL(real_to_cx_kernel<T>, dim3(flat_grid(n)), dim3(256), 0, st, (const T*)real_in, (cx<T>*)p->cz_full, n);
    OA_LAUNCH_CHECK();
    if (int rc = czt_c2c_t<T>(p, p->cz_full, p->cz_full, 0, scale, st)) return rc;
    return oa_full_to_hc(p, p->cz_full, hc_out, st);
}
template <typename T>
static int czt_c2r_t(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st) {
    if (int rc = oa_hc_to_full(p, hc_in, p->cz_full, st)) return rc;
    if (int rc = czt_c2c_t<T>(p, p->cz_full, p->cz_full, 1, scale, st)) return rc;
    const long n = (long)p->ny * p->nx;
    hipLaunchKernelGGL(cx_to_real_kernel<T>, dim3(flat_grid(n)), dim3(256), 0, st, (const cx<T>*)p->cz_full, (T*)real_out, n);
    OA_LAUNCH_CHECK();
    return 0;
}
int czt_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? czt_r2c_t<float>(p, real_in, hc_out, scale, st) : czt_r2c_t<double>(p, real_in, hc_out, scale, st);
}
int czt_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, hipStream_t st) {
    return p->dtype == OA_F32 ? czt_c2r_t<float>(p, hc_in, real_out, scale, st) : czt_c2r_t<double>(p, hc_in, real_out, scale, st);
}

}  // namespace oa
