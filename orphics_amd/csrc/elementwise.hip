// Memory-bound per-mode kernels (K2 f2power, K3 filter multiply / QE legs,
// K4 real-space product, K5 divergence, K6 QU<->EB rotation) + layout helpers.
// All are streaming kernels: 16-byte vector accesses, grid-stride, no LDS.
#include "common.hpp"

namespace oa {

constexpr int EW = 4;  // elements per thread per iteration

// ---------------------------------------------------------------- flat kernels
template <typename T>
__global__ __launch_bounds__(256) void f2power_kernel(const cx<T>* __restrict__ k1, const cx<T>* __restrict__ k2,
                                                      T* __restrict__ out, T norm, long n4, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const Arr<T, 8> a = reinterpret_cast<const Arr<T, 8>*>(k1)[i];
        const Arr<T, 8> b = reinterpret_cast<const Arr<T, 8>*>(k2)[i];
        Arr<T, 4> o;
#pragma unroll
        for (int j = 0; j < EW; ++j) o.v[j] = (a.v[2 * j] * b.v[2 * j] + a.v[2 * j + 1] * b.v[2 * j + 1]) * norm;
        reinterpret_cast<Arr<T, 4>*>(out)[i] = o;
    }
    // scalar tail
    const long t0 = n4 * EW + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 < n) out[t0] = (k1[t0].x * k2[t0].x + k1[t0].y * k2[t0].y) * norm;
}

// out = k * f, both complex (filter_map with a complex k-space filter, maps.py:1923)
template <typename T>
__global__ __launch_bounds__(256) void cmul_kernel(const cx<T>* __restrict__ k, const cx<T>* __restrict__ f, cx<T>* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const cx<T> a = k[i], b = f[i];
        out[i] = mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cmul_real_kernel(const cx<T>* __restrict__ k, const T* __restrict__ f,
                                                        cx<T>* __restrict__ out, long n4, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const Arr<T, 8> a = reinterpret_cast<const Arr<T, 8>*>(k)[i];
        const Arr<T, 4> w = reinterpret_cast<const Arr<T, 4>*>(f)[i];
        Arr<T, 8> o;
#pragma unroll
        for (int j = 0; j < EW; ++j) {
            o.v[2 * j] = a.v[2 * j] * w.v[j];
            o.v[2 * j + 1] = a.v[2 * j + 1] * w.v[j];
        }
        reinterpret_cast<Arr<T, 8>*>(out)[i] = o;
    }
    const long t0 = n4 * EW + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 < n) out[t0] = k[t0] * f[t0];
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                    T alpha, T beta, int mul, long n4, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const Arr<T, 4> x = reinterpret_cast<const Arr<T, 4>*>(a)[i];
        const Arr<T, 4> y = reinterpret_cast<const Arr<T, 4>*>(b)[i];
        Arr<T, 4> o;
#pragma unroll
        for (int j = 0; j < EW; ++j) o.v[j] = mul ? x.v[j] * y.v[j] : alpha * x.v[j] + beta * y.v[j];
        reinterpret_cast<Arr<T, 4>*>(out)[i] = o;
    }
    const long t0 = n4 * EW + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 < n) out[t0] = mul ? a[t0] * b[t0] : alpha * a[t0] + beta * b[t0];
}

template <typename T>
__global__ __launch_bounds__(256) void rot2_kernel(const T* __restrict__ c, const T* __restrict__ s,
                                                   const cx<T>* __restrict__ i1, const cx<T>* __restrict__ i2,
                                                   cx<T>* __restrict__ o1, cx<T>* __restrict__ o2, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T cc = c[i], ss = s[i];
        const cx<T> a = i1[i], b = i2[i];
        o1[i] = a * cc - b * ss;
        o2[i] = a * ss + b * cc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stack_add_kernel(const T* __restrict__ x, double* __restrict__ acc, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc[i] += (double)x[i];
}

// the same over the active region of an hc plane only (columns < w, band rows): kappa_hat is exactly zero elsewhere
template <typename T>
__global__ __launch_bounds__(256) void stack_add_region_kernel(const T* __restrict__ x, double* __restrict__ acc, int ny, long kp,
                                                               int w, int rb, int nbatch, long xstride) {
    int y = blockIdx.y;
    if (rb > 0 && y >= rb) y += ny - (2 * rb - 1);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;      // real-valued element within the row (2 per complex column)
    if (c >= 2 * w) return;
    const long i = 2 * (long)y * kp + c;
    double a = acc[i];
    for (int b = 0; b < nbatch; ++b) a += (double)x[i + b * xstride];      // planes of a batch, added in plane order
    acc[i] = a;
}

// out (+)= sum of `nparts` planes `stride` elements apart, over the active region of an hc plane (oa_qe_mv: the estimators'
// weighted kappa planes, summed in estimator order)
template <typename T>
__global__ __launch_bounds__(256) void sum_region_kernel(const cx<T>* __restrict__ parts, long stride, int nparts, cx<T>* __restrict__ out,
                                                         int accumulate, int ny, long kp, int w, int rb) {
    int y = blockIdx.y;
    if (rb > 0 && y >= rb) y += ny - (2 * rb - 1);
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const long at = (long)y * kp + x;
    cx<T> acc = accumulate ? out[at] : mk<T>((T)0, (T)0);
    for (int e = 0; e < nparts; ++e) acc = acc + parts[at + e * stride];
    out[at] = acc;
}

// ---------------------------------------------------------------- split-based 4-point combination
// SplitLensing.cross_estimator (lensing.py:980-1003) per Fourier mode from the N^2 pairwise reconstructions
// K[i*N+j] = QE(X leg from split i, Y leg from split j).  The QE is bilinear, so with s = mean of the splits
//   QE(s,s) = mean_ij K_ij,   (QE(m_i,s) + QE(s,m_i))/2 = sum_j (K_ij + K_ji) / (2N),
// and every term of the estimator is a linear combination of the K's: one pass, arithmetic in f64.
struct SplitPlanes { const void* k[64]; };

template <typename T, int N>
__global__ __launch_bounds__(256) void split_cross_power_kernel(SplitPlanes P, T* __restrict__ out, double norm, int ny, long kp,
                                                                int w, int rb) {
    int y = blockIdx.y;
    if (rb > 0 && y >= rb) y += ny - (2 * rb - 1);
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const long at = (long)y * kp + x;
    double rcr[N], rci[N], dr[N], di[N];           // rc_i = sum_j (K_ij + K_ji), d_i = K_ii
#pragma unroll
    for (int i = 0; i < N; ++i) rcr[i] = rci[i] = 0.0;
    double tr = 0.0, ti = 0.0, pij = 0.0;         // sum of all K, sum_{i<j} |K_ij + K_ji|^2
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const cx<T> d = ((const cx<T>*)P.k[i * N + i])[at];
        dr[i] = (double)d.x; di[i] = (double)d.y;
        rcr[i] += 2.0 * dr[i]; rci[i] += 2.0 * di[i];
        tr += dr[i]; ti += di[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            const cx<T> a = ((const cx<T>*)P.k[i * N + j])[at], b = ((const cx<T>*)P.k[j * N + i])[at];
            const double sr = (double)a.x + (double)b.x, si = (double)a.y + (double)b.y;
            rcr[i] += sr; rci[i] += si; rcr[j] += sr; rci[j] += si;
            tr += sr; ti += si;
            pij += sr * sr + si * si;
        }
    }
    const double n = (double)N, n2 = n * n;
    double sdr = 0.0, sdi = 0.0, pic = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sdr += dr[i]; sdi += di[i];
        const double cr = rcr[i] / (2.0 * n) - dr[i] / n, ci = rci[i] / (2.0 * n) - di[i] / n;    // k_i - k_ii / N
        pic += cr * cr + ci * ci;
    }
    const double kcr = (tr - sdr) / n2, kci = (ti - sdi) / n2;                                   // QE(s,s) - sum_i k_ii / N^2
    const double v = (n2 * n2 * (kcr * kcr + kci * kci) - 4.0 * n2 * pic + pij) * norm / (n * (n - 1.0) * (n - 2.0) * (n - 3.0));
    out[at] = (T)v;
}

template <typename T>
static int split_cross_power_launch(int n, const SplitPlanes& P, T* out, double norm, int ny, long kp, int w, int rb, hipStream_t st) {
    dim3 grid((w + 255) / 256, rb ? 2 * rb - 1 : ny);
#define OA_SPLIT_CASE(NN) case NN: hipLaunchKernelGGL((split_cross_power_kernel<T, NN>), grid, dim3(256), 0, st, P, out, norm, ny, kp, w, rb); break
    switch (n) {
        OA_SPLIT_CASE(4); OA_SPLIT_CASE(5); OA_SPLIT_CASE(6); OA_SPLIT_CASE(7); OA_SPLIT_CASE(8);
        default: return fail("oa_split_cross_power: 4 <= nsplits <= 8");
    }
#undef OA_SPLIT_CASE
    OA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- layout helpers
// hc -> full complex plane by X(-l) = conj X(l)
template <typename T>
__global__ __launch_bounds__(256) void hc_to_full_kernel(const cx<T>* __restrict__ hc, cx<T>* __restrict__ full, int ny,
                                                         int nx, long kp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const int nxh = nx / 2;
    cx<T> v;
    if (x <= nxh) v = hc[(long)y * kp + x];
    else {
        const int ym = y ? ny - y : 0;
        v = conj(hc[(long)ym * kp + (nx - x)]);
    }
    full[(long)y * nx + x] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void full_to_hc_kernel(const cx<T>* __restrict__ full, cx<T>* __restrict__ hc, int ny,
                                                         int nx, long kp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x > nx / 2) return;
    hc[(long)y * kp + x] = full[(long)y * nx + x];
}

// real-valued planes on the hc grid <-> full grid (even symmetry f(-l) = f(l))
template <typename T>
__global__ __launch_bounds__(256) void hcreal_to_full_kernel(const T* __restrict__ hc, T* __restrict__ full, int ny, int nx,
                                                             long kp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const int nxh = nx / 2;
    T v;
    if (x <= nxh) v = hc[(long)y * kp + x];
    else v = hc[(long)(y ? ny - y : 0) * kp + (nx - x)];
    full[(long)y * nx + x] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void fullreal_to_hc_kernel(const T* __restrict__ full, T* __restrict__ hc, int ny, int nx,
                                                             long kp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= kp) return;
    hc[(long)y * kp + x] = (x <= nx / 2) ? full[(long)y * nx + x] : (T)0;
}

// ---------------------------------------------------------------- QE legs
// spin-2 phase e^{i sgn 2 psi}, psi = atan2(-lx, ly) (pixell queb_rotmat angle)
template <typename T>
OA_D cx<T> phase2(T lx, T ly, int sgn) {
    const T u = -lx, v = ly;
    const T r2 = u * u + v * v;
    if (r2 == (T)0) return mk<T>((T)1, (T)0);
    const T inv = (T)1 / r2;
    return mk<T>((v * v - u * u) * inv, (T)sgn * (T)2 * u * v * inv);
}

template <typename T>
__global__ __launch_bounds__(256) void qe_legs_kernel(const cx<T>* __restrict__ kX, const cx<T>* __restrict__ kY,
                                                      const T* __restrict__ FG, const T* __restrict__ FH,
                                                      cx<T>* __restrict__ Gx, cx<T>* __restrict__ Gy, cx<T>* __restrict__ H,
                                                      const T* __restrict__ lxv, const T* __restrict__ lyv,
                                                      const T* __restrict__ lxd, const T* __restrict__ lyd, int nxh, long kp,
                                                      int phase_g, int phase_h, int h_times_i) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x > nxh) return;
    const long i = (long)y * kp + x;
    const T lx = lxv[x], ly = lyv[y];
    cx<T> g = kX[i] * FG[i];
    cx<T> h = kY[i] * FH[i];
    if (phase_g) g = g * phase2<T>(lx, ly, phase_g);
    if (phase_h) h = h * phase2<T>(lx, ly, phase_h);
    if (h_times_i) h = mul_pi(h);
    Gx[i] = mul_pi(g) * lxd[x];
    Gy[i] = mul_pi(g) * lyd[y];
    H[i] = h;
}

template <typename T>
__global__ __launch_bounds__(256) void qe_div_kernel(const cx<T>* __restrict__ Px, const cx<T>* __restrict__ Py,
                                                     const T* __restrict__ Fn, cx<T>* __restrict__ out,
                                                     const T* __restrict__ lxv, const T* __restrict__ lyv, int nxh, long kp,
                                                     int accumulate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x > nxh) return;
    const long i = (long)y * kp + x;
    const cx<T> d = mul_pi(Px[i] * lxv[x] + Py[i] * lyv[y]) * Fn[i];
    out[i] = accumulate ? out[i] + d : d;
}


// ---------------------------------------------------------------- flat-sky lensing op (lensing.py:395-440)
// displacement (coordinate units) -> nearest-pixel shift + sub-pixel remainder
template <typename T>
__global__ __launch_bounds__(256) void lens_split_kernel(const T* __restrict__ alpha, T step, int* __restrict__ shift,
                                                         T* __restrict__ delta, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T a = alpha[i];
        const T r = rint(a / step);
        shift[i] = (int)r;
        delta[i] = a - r * step;
    }
}

template <typename T>
OA_D T ipow(T x, int p) {
    T r = (T)1;
    for (int i = 0; i < p; ++i) r *= x;
    return r;
}

// out[y,x] (+)= coef * src[(y+sy)%ny, (x+sx)%nx] * dx^px * dy^py   (Taylens gather + Taylor term)
template <typename T>
__global__ __launch_bounds__(256) void lens_gather_kernel(const T* __restrict__ src, const int* __restrict__ sx,
                                                          const int* __restrict__ sy, const T* __restrict__ dx,
                                                          const T* __restrict__ dy, int px, int py, T coef,
                                                          T* __restrict__ out, int ny, int nx, int accumulate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const long i = (long)y * nx + x;
    int xs = (x + sx[i]) % nx, ys = (y + sy[i]) % ny;
    if (xs < 0) xs += nx;
    if (ys < 0) ys += ny;
    T v = coef * src[(long)ys * nx + xs];
    if (px) v *= ipow(dx[i], px);
    if (py) v *= ipow(dy[i], py);
    out[i] = accumulate ? out[i] + v : v;
}


// All Fourier-space derivatives of one map for the Taylor series of the lensing op in ONE pass over its transform:
// plane idx(a, b) = n (n + 1) / 2 - 1 + b, n = a + b = 1 .. order - 1, holds (i lx)^a (i ly)^b k  (derivative axes: the
// self-conjugate Nyquist row / column carries no odd derivative).  14 planes for the reference's order 5.
template <typename T>
__global__ __launch_bounds__(256) void hc_derivs_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, long ostride,
                                                        const T* __restrict__ lxd, const T* __restrict__ lyd, int nxh, long kp,
                                                        int order) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x > nxh) return;
    const long i = (long)y * kp + x;
    const cx<T> k = in[i];
    const T lx = lxd[x], ly = lyd[y];
    cx<T> kn = k;                                 // i^n k
    T xp[8];                                      // lx^a
    xp[0] = (T)1;
    for (int n = 1; n < order; ++n) {
        kn = mul_pi(kn);
        xp[n] = xp[n - 1] * lx;
        T yb = (T)1;                              // ly^b
        cx<T>* o = out + (long)(n * (n + 1) / 2 - 1) * ostride + i;
        for (int b = 0; b <= n; ++b) {
            o[(long)b * ostride] = kn * (xp[n - b] * yb);
            yb *= ly;
        }
    }
}

// flat_taylens in one gather pass: out = sum_{a + b < order} dx^a dy^b / (a! b!) D_ab[(y + sy) % ny, (x + sx) % nx], D_00 = src
// and D_ab (n >= 1) = real plane idx(a, b) of `planes` (the C2R of hc_derivs_kernel's output).
// ORDER is compile-time: the term loops unroll, the powers dx^a / a!, dy^b / b! are formed once (the same operations in the same order
// as the run-time-order loop this replaces -- identical values -- which recomputed dy^b / b! for every total order and indexed
// dx^a / a! through scratch: 385 us per 4096^2 float32 map against 215 us of traffic) and all gathers of a pixel are in flight together
template <typename T, int ORDER>
__global__ __launch_bounds__(256) void lens_taylor_kernel(const T* __restrict__ src, const T* __restrict__ planes, long pstride,
                                                          const int* __restrict__ sx, const int* __restrict__ sy,
                                                          const T* __restrict__ dx, const T* __restrict__ dy, T* __restrict__ out,
                                                          int ny, int nx) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const long i = (long)y * nx + x;
    int xs = (x + sx[i]) % nx, ys = (y + sy[i]) % ny;
    if (xs < 0) xs += nx;
    if (ys < 0) ys += ny;
    const long g = (long)ys * nx + xs;
    const T ddx = dx[i], ddy = dy[i];
    T v[ORDER * (ORDER + 1) / 2];
    v[0] = src[g];
#pragma unroll
    for (int k = 1; k < ORDER * (ORDER + 1) / 2; ++k) v[k] = planes[(long)(k - 1) * pstride + g];
    T xa[ORDER], ya[ORDER];                       // dx^a / a!, dy^b / b!
    xa[0] = (T)1; ya[0] = (T)1;
#pragma unroll
    for (int n = 1; n < ORDER; ++n) { xa[n] = xa[n - 1] * ddx / (T)n; ya[n] = ya[n - 1] * ddy / (T)n; }
    T acc = v[0];
#pragma unroll
    for (int n = 1; n < ORDER; ++n) {
#pragma unroll
        for (int b = 0; b <= n; ++b) acc += v[n * (n + 1) / 2 + b] * (xa[n - b] * ya[b]);
    }
    out[i] = acc;
}

// tile-major repack of a full-pitch hc-layout plane onto the coarse grid (common.hpp pack_tiles)
template <typename E>
__global__ __launch_bounds__(256) void pack_tiles_kernel(const E* __restrict__ src, E* __restrict__ dst, int rows, int ny, long kp, int logc, int width,
                                                         long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    const int C = 1 << logc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i & (C - 1));
        const long r = i >> logc;
        const int k = (int)(r % rows);
        const long tile = r / rows;
        const long col = tile * C + c;
        const long y = k + (k >= rows / 2 ? ny - rows : 0);
        dst[i] = col < width ? src[y * kp + col] : (E)0;
    }
}

// ---- HBM bandwidth probes (bench.py: the ceiling the roofline fractions are read against, measured in the same run) -----
// 16-byte accesses, grid-stride, the layout of the guide's float4 copy; the read probe keeps a per-thread checksum so the
// loads cannot be elided and writes one word per thread at the end.
__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
typedef unsigned oa_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe_read_kernel(const oa_u4* __restrict__ src, unsigned* __restrict__ sink, long n16) {
    const long stride = (long)gridDim.x * blockDim.x;
    oa_u4 acc = {0u, 0u, 0u, 0u};
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {          // four independent 16-byte loads in flight per thread
        const oa_u4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const oa_u4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= src[i];
    sink[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

}  // namespace oa

namespace oa {
int sum_region(int dtype, const void* parts, long part_stride, int nparts, void* out, int accumulate, int ny, long kp, int w, int rb,
               hipStream_t st) {
    if (w <= 0 || w > kp) w = (int)kp;
    if (!(rb > 0 && 2L * rb - 1 < ny)) rb = 0;
    dim3 grid((w + 255) / 256, rb ? 2 * rb - 1 : ny);
    if (dtype == OA_F32)
        hipLaunchKernelGGL(sum_region_kernel<float>, grid, dim3(256), 0, st, (const cx<float>*)parts, part_stride, nparts, (cx<float>*)out, accumulate, ny, kp, w, rb);
    else
        hipLaunchKernelGGL(sum_region_kernel<double>, grid, dim3(256), 0, st, (const cx<double>*)parts, part_stride, nparts, (cx<double>*)out, accumulate, ny, kp, w, rb);
    OA_LAUNCH_CHECK();
    return 0;
}
// mean-field stack of the one-call Monte-Carlo driver (pipeline.hip)
int stack_add_region(int dtype, const void* x, double* acc, int ny, long kp, int w, int rb, hipStream_t st, int nbatch, long xstride) {
    // xstride: real elements between the planes of a batch
    if (w <= 0 || w > kp) w = (int)kp;
    if (!(rb > 0 && 2L * rb - 1 < ny)) rb = 0;
    dim3 grid((2 * w + 255) / 256, rb ? 2 * rb - 1 : ny);
    if (dtype == OA_F32) hipLaunchKernelGGL(stack_add_region_kernel<float>, grid, dim3(256), 0, st, (const float*)x, acc, ny, kp, w, rb, nbatch, xstride);
    else hipLaunchKernelGGL(stack_add_region_kernel<double>, grid, dim3(256), 0, st, (const double*)x, acc, ny, kp, w, rb, nbatch, xstride);
    OA_LAUNCH_CHECK();
    return 0;
}
}  // namespace oa

using namespace oa;

#define DISPATCH(dtype, CALL_F, CALL_D) \
    if ((dtype) == OA_F32) { CALL_F; } else if ((dtype) == OA_F64) { CALL_D; } else return fail("bad dtype")

extern "C" {

int oa_f2power(int dtype, const void* k1, const void* k2, void* out, double norm, long n, void* stream) {
    OA_REQUIRE(k1 && k2 && out && n >= 0, "oa_f2power: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / EW;
    const int g = flat_grid(n4 > 0 ? n4 : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(f2power_kernel<float>, dim3(g), dim3(256), 0, st, (const cx<float>*)k1, (const cx<float>*)k2,
                                (float*)out, (float)norm, n4, n),
             hipLaunchKernelGGL(f2power_kernel<double>, dim3(g), dim3(256), 0, st, (const cx<double>*)k1,
                                (const cx<double>*)k2, (double*)out, norm, n4, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_split_cross_power(int dtype, int nsplits, const void* const* host_kappa, void* out_hcreal, double norm, int ny, long kp,
                         int active_cols, int active_rows, void* stream) {
    OA_REQUIRE(host_kappa && out_hcreal && ny > 0 && kp > 0, "oa_split_cross_power: bad argument");
    OA_REQUIRE(nsplits >= 4 && nsplits <= 8, "oa_split_cross_power: 4 <= nsplits <= 8");
    SplitPlanes P;
    for (int i = 0; i < nsplits * nsplits; ++i) {
        OA_REQUIRE(host_kappa[i], "oa_split_cross_power: NULL plane");
        P.k[i] = host_kappa[i];
    }
    int w = active_cols, rb = active_rows;
    if (w <= 0 || w > kp) w = (int)kp;
    if (!(rb > 0 && 2L * rb - 1 < ny)) rb = 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(dtype, return split_cross_power_launch<float>(nsplits, P, (float*)out_hcreal, norm, ny, kp, w, rb, st),
             return split_cross_power_launch<double>(nsplits, P, (double*)out_hcreal, norm, ny, kp, w, rb, st));
}

int oa_cmul_real(int dtype, const void* k, const void* f, void* out, long n, void* stream) {
    OA_REQUIRE(k && f && out && n >= 0, "oa_cmul_real: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / EW;
    const int g = flat_grid(n4 > 0 ? n4 : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(cmul_real_kernel<float>, dim3(g), dim3(256), 0, st, (const cx<float>*)k, (const float*)f,
                                (cx<float>*)out, n4, n),
             hipLaunchKernelGGL(cmul_real_kernel<double>, dim3(g), dim3(256), 0, st, (const cx<double>*)k, (const double*)f,
                                (cx<double>*)out, n4, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_cmul(int dtype, const void* k, const void* f, void* out, long n, void* stream) {
    OA_REQUIRE(k && f && out && n >= 0, "oa_cmul: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = flat_grid(n > 0 ? n : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(cmul_kernel<float>, dim3(g), dim3(256), 0, st, (const cx<float>*)k, (const cx<float>*)f, (cx<float>*)out, n),
             hipLaunchKernelGGL(cmul_kernel<double>, dim3(g), dim3(256), 0, st, (const cx<double>*)k, (const cx<double>*)f, (cx<double>*)out, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_mul_real(int dtype, const void* a, const void* b, void* out, long n, void* stream) {
    OA_REQUIRE(a && b && out && n >= 0, "oa_mul_real: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / EW;
    const int g = flat_grid(n4 > 0 ? n4 : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(axpby_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out,
                                0.f, 0.f, 1, n4, n),
             hipLaunchKernelGGL(axpby_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)a, (const double*)b,
                                (double*)out, 0., 0., 1, n4, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_axpby_real(int dtype, const void* a, const void* b, void* out, double alpha, double beta, long n, void* stream) {
    OA_REQUIRE(a && b && out && n >= 0, "oa_axpby_real: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / EW;
    const int g = flat_grid(n4 > 0 ? n4 : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(axpby_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out,
                                (float)alpha, (float)beta, 0, n4, n),
             hipLaunchKernelGGL(axpby_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)a, (const double*)b,
                                (double*)out, alpha, beta, 0, n4, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_rot2(int dtype, const void* c, const void* s, const void* i1, const void* i2, void* o1, void* o2, long n,
            void* stream) {
    OA_REQUIRE(c && s && i1 && i2 && o1 && o2 && n >= 0, "oa_rot2: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = flat_grid(n > 0 ? n : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(rot2_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)c, (const float*)s,
                                (const cx<float>*)i1, (const cx<float>*)i2, (cx<float>*)o1, (cx<float>*)o2, n),
             hipLaunchKernelGGL(rot2_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)c, (const double*)s,
                                (const cx<double>*)i1, (const cx<double>*)i2, (cx<double>*)o1, (cx<double>*)o2, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_stack_add(int dtype, const void* x, double* acc, long n, void* stream) {
    OA_REQUIRE(x && acc && n >= 0, "oa_stack_add: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = flat_grid(n > 0 ? n : 1);
    DISPATCH(dtype, hipLaunchKernelGGL(stack_add_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, acc, n),
             hipLaunchKernelGGL(stack_add_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)x, acc, n));
    OA_LAUNCH_CHECK();
    return 0;
}

#define PLANE_GRID(p, w) dim3(((w) + 255) / 256, (p)->ny)

int oa_hc_to_full(oa_plan* p, const void* hc, void* full, void* stream) {
    OA_REQUIRE(p && hc && full, "oa_hc_to_full: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(hc_to_full_kernel<float>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const cx<float>*)hc,
                                (cx<float>*)full, p->ny, p->nx, p->kp),
             hipLaunchKernelGGL(hc_to_full_kernel<double>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const cx<double>*)hc,
                                (cx<double>*)full, p->ny, p->nx, p->kp));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_full_to_hc(oa_plan* p, const void* full, void* hc, void* stream) {
    OA_REQUIRE(p && hc && full, "oa_full_to_hc: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(full_to_hc_kernel<float>, PLANE_GRID(p, p->nx / 2 + 1), dim3(256), 0, st,
                                (const cx<float>*)full, (cx<float>*)hc, p->ny, p->nx, p->kp),
             hipLaunchKernelGGL(full_to_hc_kernel<double>, PLANE_GRID(p, p->nx / 2 + 1), dim3(256), 0, st,
                                (const cx<double>*)full, (cx<double>*)hc, p->ny, p->nx, p->kp));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_hcreal_to_full(oa_plan* p, const void* hc, void* full, void* stream) {
    OA_REQUIRE(p && hc && full, "oa_hcreal_to_full: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(hcreal_to_full_kernel<float>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const float*)hc,
                                (float*)full, p->ny, p->nx, p->kp),
             hipLaunchKernelGGL(hcreal_to_full_kernel<double>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const double*)hc,
                                (double*)full, p->ny, p->nx, p->kp));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_fullreal_to_hc(oa_plan* p, const void* full, void* hc, void* stream) {
    OA_REQUIRE(p && hc && full, "oa_fullreal_to_hc: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(fullreal_to_hc_kernel<float>, PLANE_GRID(p, (int)p->kp), dim3(256), 0, st, (const float*)full,
                                (float*)hc, p->ny, p->nx, p->kp),
             hipLaunchKernelGGL(fullreal_to_hc_kernel<double>, PLANE_GRID(p, (int)p->kp), dim3(256), 0, st,
                                (const double*)full, (double*)hc, p->ny, p->nx, p->kp));
    OA_LAUNCH_CHECK();
    return 0;
}


int oa_lens_split(int dtype, const void* alpha, double step, int32_t* shift, void* delta, long n, void* stream) {
    OA_REQUIRE(alpha && shift && delta && n >= 0 && step != 0.0, "oa_lens_split: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = flat_grid(n > 0 ? n : 1);
    DISPATCH(dtype,
             hipLaunchKernelGGL(lens_split_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)alpha, (float)step, shift,
                                (float*)delta, n),
             hipLaunchKernelGGL(lens_split_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)alpha, step, shift,
                                (double*)delta, n));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_lens_gather(oa_plan* p, const void* src, const int32_t* shift_x, const int32_t* shift_y, const void* dx, const void* dy,
                   int pow_x, int pow_y, double coef, void* out, int accumulate, void* stream) {
    OA_REQUIRE(p && src && shift_x && shift_y && dx && dy && out, "oa_lens_gather: NULL argument");
    OA_REQUIRE(pow_x >= 0 && pow_y >= 0 && pow_x + pow_y <= 16, "oa_lens_gather: bad Taylor powers");
    OA_REQUIRE(src != out, "oa_lens_gather: in-place not supported");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(lens_gather_kernel<float>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const float*)src, shift_x,
                                shift_y, (const float*)dx, (const float*)dy, pow_x, pow_y, (float)coef, (float*)out, p->ny,
                                p->nx, accumulate),
             hipLaunchKernelGGL(lens_gather_kernel<double>, PLANE_GRID(p, p->nx), dim3(256), 0, st, (const double*)src,
                                shift_x, shift_y, (const double*)dx, (const double*)dy, pow_x, pow_y, coef, (double*)out,
                                p->ny, p->nx, accumulate));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_hc_derivs(oa_plan* p, const void* hc_in, int order, void* hc_out_planes, long plane_stride, void* stream) {
    OA_REQUIRE(p && hc_in && hc_out_planes, "oa_hc_derivs: NULL argument");
    OA_REQUIRE(p->have_laxes, "oa_hc_derivs: call oa_plan_set_laxes first");
    OA_REQUIRE(order >= 2 && order <= 8, "oa_hc_derivs: order must be 2..8");
    OA_REQUIRE(plane_stride >= (long)p->ny * p->kp, "oa_hc_derivs: plane_stride smaller than a plane");
    hipStream_t st = (hipStream_t)stream;
    const int nxh = p->nx / 2;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(hc_derivs_kernel<float>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<float>*)hc_in,
                                (cx<float>*)hc_out_planes, plane_stride, (const float*)p->lxd, (const float*)p->lyd, nxh, p->kp, order),
             hipLaunchKernelGGL(hc_derivs_kernel<double>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<double>*)hc_in,
                                (cx<double>*)hc_out_planes, plane_stride, (const double*)p->lxd, (const double*)p->lyd, nxh, p->kp, order));
    OA_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
namespace oa {
int pack_tiles(const oa_plan* p, const void* src, void* dst, int rows, int logc, int width, int elem_bytes, hipStream_t st) {
    const long tiles = ((long)width + (1 << logc) - 1) >> logc, total = tiles * rows << logc;
    const int g = flat_grid(total);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(pack_tiles_kernel<uint32_t>, dim3(g), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, rows, p->ny, p->kp, logc, width, total);
    else
        hipLaunchKernelGGL(pack_tiles_kernel<uint64_t>, dim3(g), dim3(256), 0, st, (const uint64_t*)src, (uint64_t*)dst, rows, p->ny, p->kp, logc, width, total);
    OA_LAUNCH_CHECK();
    return 0;
}
}  // namespace oa
extern "C" {
int oa_lens_taylor(oa_plan* p, const void* src, const void* deriv_planes, long plane_stride, int order, const int32_t* shift_x,
                   const int32_t* shift_y, const void* dx, const void* dy, void* out, void* stream) {
    OA_REQUIRE(p && src && shift_x && shift_y && dx && dy && out, "oa_lens_taylor: NULL argument");
    OA_REQUIRE(order >= 1 && order <= 8 && (order == 1 || deriv_planes), "oa_lens_taylor: order must be 1..8 (derivative planes for order > 1)");
    OA_REQUIRE(order == 1 || plane_stride >= (long)p->ny * p->nx, "oa_lens_taylor: plane_stride smaller than a plane");
    OA_REQUIRE(src != out, "oa_lens_taylor: in-place not supported");
    hipStream_t st = (hipStream_t)stream;
#define OA_TAYLOR(T, O) \
    hipLaunchKernelGGL((lens_taylor_kernel<T, O>), PLANE_GRID(p, p->nx), dim3(256), 0, st, (const T*)src, (const T*)deriv_planes, plane_stride, \
                       shift_x, shift_y, (const T*)dx, (const T*)dy, (T*)out, p->ny, p->nx)
#define OA_TAYLOR_ORDERS(T) \
    switch (order) { \
        case 1: OA_TAYLOR(T, 1); break; case 2: OA_TAYLOR(T, 2); break; case 3: OA_TAYLOR(T, 3); break; case 4: OA_TAYLOR(T, 4); break; \
        case 5: OA_TAYLOR(T, 5); break; case 6: OA_TAYLOR(T, 6); break; case 7: OA_TAYLOR(T, 7); break; default: OA_TAYLOR(T, 8); break; \
    }
    if (p->dtype == OA_F32) { OA_TAYLOR_ORDERS(float) } else { OA_TAYLOR_ORDERS(double) }
#undef OA_TAYLOR_ORDERS
#undef OA_TAYLOR
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_qe_legs(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* Gx, void* Gy, void* H,
               int phase_g, int phase_h, int h_times_i, void* stream) {
    OA_REQUIRE(p && kX && kY && FG && FH && Gx && Gy && H, "oa_qe_legs: NULL argument");
    OA_REQUIRE(p->have_laxes, "oa_qe_legs: call oa_plan_set_laxes first");
    hipStream_t st = (hipStream_t)stream;
    const int nxh = p->nx / 2;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(qe_legs_kernel<float>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<float>*)kX,
                                (const cx<float>*)kY, (const float*)FG, (const float*)FH, (cx<float>*)Gx, (cx<float>*)Gy,
                                (cx<float>*)H, (const float*)p->lx, (const float*)p->ly, (const float*)p->lxd,
                                (const float*)p->lyd, nxh, p->kp, phase_g, phase_h,
                                h_times_i),
             hipLaunchKernelGGL(qe_legs_kernel<double>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<double>*)kX,
                                (const cx<double>*)kY, (const double*)FG, (const double*)FH, (cx<double>*)Gx,
                                (cx<double>*)Gy, (cx<double>*)H, (const double*)p->lx, (const double*)p->ly,
                                (const double*)p->lxd, (const double*)p->lyd, nxh, p->kp,
                                phase_g, phase_h, h_times_i));
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_qe_div(oa_plan* p, const void* Px, const void* Py, const void* Fnorm, void* out, int accumulate, void* stream) {
    OA_REQUIRE(p && Px && Py && Fnorm && out, "oa_qe_div: NULL argument");
    OA_REQUIRE(p->have_laxes, "oa_qe_div: call oa_plan_set_laxes first");
    hipStream_t st = (hipStream_t)stream;
    const int nxh = p->nx / 2;
    DISPATCH(p->dtype,
             hipLaunchKernelGGL(qe_div_kernel<float>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<float>*)Px,
                                (const cx<float>*)Py, (const float*)Fnorm, (cx<float>*)out, (const float*)p->lxd,
                                (const float*)p->lyd, nxh, p->kp, accumulate),
             hipLaunchKernelGGL(qe_div_kernel<double>, PLANE_GRID(p, nxh + 1), dim3(256), 0, st, (const cx<double>*)Px,
                                (const cx<double>*)Py, (const double*)Fnorm, (cx<double>*)out, (const double*)p->lxd,
                                (const double*)p->lyd, nxh, p->kp, accumulate));
    OA_LAUNCH_CHECK();
    return 0;
}

/* Bandwidth probes: device copy / read of `bytes` (multiple of 16) with 16-byte accesses.  `sink` (read probe): at least
 * 4 * 256 * 8192 bytes of scratch. */
int oa_probe_copy(void* dst, const void* src, size_t bytes, void* stream) {
    OA_REQUIRE(dst && src && bytes % 16 == 0, "oa_probe_copy: bad argument");
    const long n16 = (long)(bytes / 16);
    hipLaunchKernelGGL(probe_copy_kernel, dim3(flat_grid(n16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
    OA_LAUNCH_CHECK();
    return 0;
}
int oa_probe_read(const void* src, size_t bytes, void* sink, void* stream) {
    OA_REQUIRE(src && sink && bytes % 16 == 0, "oa_probe_read: bad argument");
    const long n16 = (long)(bytes / 16);
    hipLaunchKernelGGL(probe_read_kernel, dim3(flat_grid(n16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const oa_u4*)src, (unsigned*)sink, n16);
    OA_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
