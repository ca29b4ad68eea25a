// K8: on-device Gaussian random fields (Philox4x32-10 counter RNG + Box-Muller)
// K9: device-side moment accumulation for Monte-Carlo ensembles.
#include "common.hpp"

namespace oa {

struct U4 { uint32_t x, y, z, w; };

OA_D U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one: the quarter-rate integer
        // multiplies are what the draw kernels spend their time on
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)ctr.x, p1 = (uint64_t)M1 * (uint64_t)ctr.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        U4 n;
        n.x = hi1 ^ ctr.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ ctr.w ^ k1;
        n.w = lo0;
        ctr = n;
        k0 += W0;
        k1 += W1;
    }
    return ctr;
}

// two N(0,1) from two 32-bit words
OA_D void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;  // (0,1]
    const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
#ifdef OA_FAST_BOXMULLER     // experiment: hardware log2 / sqrt / sin / cos (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32 take revolutions)
    const float l2 = __builtin_amdgcn_logf(u1);                    // log2(u1) <= 0
    const float r = __builtin_amdgcn_sqrtf(fmaxf(-1.3862943611198906f * l2, 0.0f));
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
#else
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincospif(2.0f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
#endif
}

// 4 normals for counter index `idx` of stream (seed, sid)
OA_D void normals4(uint64_t seed, uint64_t sid, uint64_t idx, float* n) {
    U4 c;
    c.x = (uint32_t)idx; c.y = (uint32_t)(idx >> 32); c.z = (uint32_t)sid; c.w = (uint32_t)(sid >> 32);
    const U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, n[0], n[1]);
    box_muller(r.z, r.w, n[2], n[3]);
}

// Hermitian-consistent unit white noise on the hc grid, times covsqrt.
// One Philox call serves 2 modes: (row y', column pair p) -> columns 2p, 2p+1.
template <typename T>
__global__ __launch_bounds__(256) void grf_hc_kernel(uint64_t seed, uint64_t sid, const T* __restrict__ cs,
                                                     cx<T>* __restrict__ out, int ny, int nx, long kp, int wpairs, int rband,
                                                     long zstride) {
    // grid z = realisation: stream id sid + z into the plane z * zstride elements behind `out` (oa_mc_run's batches)
    sid += blockIdx.z;
    out += (long)blockIdx.z * zstride;
    const int nxh = nx / 2;
    const int npair = nxh / 2 + 1;  // pairs cover columns 0..nxh(+1)
    const int pr = blockIdx.x * blockDim.x + threadIdx.x;
    // region draws (oa_grf_hc_band): grid y enumerates the band rows y < rband, y > ny - rband; every mode keeps the
    // Philox counter of the full-plane draw, so the region is a bit-identical subset of it
    int y = blockIdx.y;
    if (rband > 0 && y >= rband) y += ny - (2 * rband - 1);
    if (pr >= (wpairs > 0 ? wpairs : npair)) return;
    const T rs2 = (T)0.70710678118654752440;
    float n[4];
    int cur_ys = -1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = 2 * pr + j;
        if (x > nxh) break;
        // self-conjugate columns: rows y and ny-y must be conjugates
        const bool edgecol = (x == 0 || x == nxh);
        int ys = y;
        bool cj = false;
        if (edgecol && y > ny / 2) { ys = ny - y; cj = true; }
        // columns 2p, 2p + 1 share a counter (and its four normals) unless one of them is a self-conjugate column read at the
        // mirrored row: ONE Philox evaluation per thread, not one per column
        if (ys != cur_ys) { normals4(seed, sid, (uint64_t)ys * (uint64_t)npair + (uint64_t)pr, n); cur_ys = ys; }
        T re = (T)n[2 * j], im = (T)n[2 * j + 1];
        if (edgecol && (ys == 0 || ys == ny / 2)) { im = (T)0; }  // real mode, variance 1
        else { re *= rs2; im *= rs2; }
        if (cj) im = -im;
        const long i = (long)y * kp + x;
        const T s = cs ? cs[i] : (T)1;
        out[i] = mk<T>(re * s, im * s);
    }
}

// MapGen.get_map in one pass (maps.py:1579-1587): up to three white fields of streams (seed, sid0 + j) -- the counters and the
// Hermitian rules of grf_hc_kernel, so every w_j is bit-identical to a separate oa_grf_hc draw -- mixed by the covariance square
// root, v_i = sum_j cs[i][j] w_j (terms formed and added in the order of the per-plane path: cmul_real, then +), optionally
// rotated (E, B <-> Q, U on components 1, 2) and optionally ADDED to filtered input planes:
//   in == nullptr : out_i = rot(v)_i * scale                          (the unlensed T, Q, U transforms; kappa)
//   in != nullptr : out_i = rot(in * filt)_i + scale * v_i            (beam x lensed transforms -> E, B, + noise)
template <typename T>
struct GrfMixArgs {
    uint64_t seed, sid0;
    const T* cs[9];
    const T* rc;
    const T* rs;
    const cx<T>* in[3];
    const T* filt;
    cx<T>* out[3];
    T scale;
    int ncomp, ny, nx;
    long kp;
};
// NC and the input mode are compile-time: every loop unrolls, the pointer tables and the per-component values stay in registers
// (with run-time indices they went through scratch: 400-660 us per call at 4096^2 float64 instead of 100-250).  A thread owns the
// column pair (2p, 2p + 1) of one row and moves it with ONE load / store per plane (32 B float64, 16 B float32: kp is a multiple
// of 16, so pairs are aligned and in bounds, row padding included); all loads are issued before the Philox / Box-Muller arithmetic.
template <typename T> struct alignas(2 * sizeof(T)) Pair { T a, b; };
template <typename T, int NC, bool HAS_IN>
__global__ __launch_bounds__(256) void grf_mix_kernel(GrfMixArgs<T> a) {
    const int nxh = a.nx / 2, npair = nxh / 2 + 1, ny = a.ny;
    const int pr = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (2L * pr >= a.kp) return;
    const long i0 = (long)y * a.kp + 2 * pr;
    typedef Pair<T> R2;
    typedef Pair<cx<T>> C2;
    // ---- loads ----
    R2 sv[NC * NC];
#pragma unroll
    for (int k = 0; k < NC * NC; ++k) { sv[k].a = (T)0; sv[k].b = (T)0; if (a.cs[k]) sv[k] = *reinterpret_cast<const R2*>(a.cs[k] + i0); }
    R2 rc{(T)1, (T)1}, rs{(T)0, (T)0}, fl{(T)1, (T)1};
    if (NC == 3 && a.rc) { rc = *reinterpret_cast<const R2*>(a.rc + i0); rs = *reinterpret_cast<const R2*>(a.rs + i0); }
    C2 u[NC];
    if constexpr (HAS_IN) {
        if (a.filt) fl = *reinterpret_cast<const R2*>(a.filt + i0);
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = *reinterpret_cast<const C2*>(a.in[c] + i0);
    }
    // ---- draws: columns 2p, 2p + 1 share a counter unless one of them is a self-conjugate column read at the mirrored row ----
    const T rs2 = (T)0.70710678118654752440;
    cx<T> w[2][NC];
    float n[NC][4];
    int cur_ys = -1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = 2 * pr + j;
        const bool edgecol = (x == 0 || x == nxh);
        int ys = y;
        bool cj = false;
        if (edgecol && y > ny / 2) { ys = ny - y; cj = true; }
        if (ys != cur_ys && x <= nxh) {
#pragma unroll
            for (int c = 0; c < NC; ++c) normals4(a.seed, a.sid0 + c, (uint64_t)ys * (uint64_t)npair + (uint64_t)pr, n[c]);
            cur_ys = ys;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            T re = (T)n[c][2 * j], im = (T)n[c][2 * j + 1];
            if (edgecol && (ys == 0 || ys == ny / 2)) { im = (T)0; }
            else { re *= rs2; im *= rs2; }
            if (cj) im = -im;
            w[j][c] = mk<T>(re, im);
        }
    }
    // ---- mix, rotate, add: per column of the pair ----
    C2 o[NC];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = 2 * pr + j;
        cx<T> v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bool any = false;
            v[c] = mk<T>((T)0, (T)0);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (!a.cs[c * NC + k]) continue;
                const T q = j ? sv[c * NC + k].b : sv[c * NC + k].a;
                const cx<T> term = mk<T>(w[j][k].x * q, w[j][k].y * q);
                v[c] = any ? v[c] + term : term;
                any = true;
            }
        }
        const T cc = j ? rc.b : rc.a, ss = j ? rs.b : rs.a;
        cx<T> r[NC];
        if constexpr (HAS_IN) {
            const T f = j ? fl.b : fl.a;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const cx<T> t = j ? u[c].b : u[c].a; r[c] = a.filt ? mk<T>(t.x * f, t.y * f) : t; }
            if constexpr (NC == 3) {
                if (a.rc) { const cx<T> p = r[1], q = r[2]; r[1] = p * cc - q * ss; r[2] = p * ss + q * cc; }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = r[c] + v[c] * a.scale;
        } else {
            if constexpr (NC == 3) {
                if (a.rc) { const cx<T> p = v[1], q = v[2]; v[1] = p * cc - q * ss; v[2] = p * ss + q * cc; }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = a.scale == (T)1 ? v[c] : v[c] * a.scale;
        }
        // the row padding of the outputs (columns nx/2 + 1 .. kp - 1) is written as zero: callers pass torch.empty planes
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const cx<T> z = x > nxh ? mk<T>((T)0, (T)0) : r[c];
            if (j) o[c].b = z; else o[c].a = z;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<C2*>(a.out[c] + i0) = o[c];
}

template <typename T>
__global__ __launch_bounds__(256) void randn_kernel(uint64_t seed, uint64_t sid, T* __restrict__ out, long n) {
    const long n4 = (n + 3) / 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float v[4];
        normals4(seed, sid, (uint64_t)i, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * i + j < n) out[4 * i + j] = (T)v[j];
    }
}

__global__ __launch_bounds__(256) void moments_add_kernel(const double* __restrict__ x, int d, int64_t* __restrict__ n,
                                                          double* __restrict__ S, double* __restrict__ C) {
    const long tot = (long)d * d;
    const long stride = (long)gridDim.x * blockDim.x;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = gid; i < tot; i += stride) {
        const int a = (int)(i / d), b = (int)(i % d);
        C[i] += x[a] * x[b];
    }
    for (long i = gid; i < d; i += stride) S[i] += x[i];
    if (gid == 0) n[0] += 1;
}

// x_a = sums[a] / counts[a] formed on the fly: bin means (bin2D.bin) straight into the ensemble moments
__global__ __launch_bounds__(256) void moments_add_binned_kernel(const double* __restrict__ sums, const int64_t* __restrict__ counts,
                                                                 int d, int64_t* __restrict__ n, double* __restrict__ S,
                                                                 double* __restrict__ C) {
    const long tot = (long)d * d;
    const long stride = (long)gridDim.x * blockDim.x;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = gid; i < tot; i += stride) {
        const int a = (int)(i / d), b = (int)(i % d);
        C[i] += (sums[a] / (double)counts[a]) * (sums[b] / (double)counts[b]);
    }
    for (long i = gid; i < d; i += stride) S[i] += sums[i] / (double)counts[i];
    if (gid == 0) n[0] += 1;
}

template <typename T>
static int grf_mix_launch(oa_plan* p, uint64_t seed, uint64_t sid0, int ncomp, const void* const* cs, const void* rc, const void* rs,
                          const void* const* in, const void* filt, double scale, void* const* out, hipStream_t st) {
    GrfMixArgs<T> a;
    a.seed = seed; a.sid0 = sid0;
    for (int i = 0; i < 9; ++i) a.cs[i] = i < ncomp * ncomp ? (const T*)cs[i] : nullptr;
    a.rc = (const T*)rc; a.rs = (const T*)rs;
    for (int i = 0; i < 3; ++i) { a.in[i] = (in && i < ncomp) ? (const cx<T>*)in[i] : nullptr; a.out[i] = i < ncomp ? (cx<T>*)out[i] : nullptr; }
    a.filt = (const T*)filt;
    a.scale = (T)scale;
    a.ncomp = ncomp; a.ny = p->ny; a.nx = p->nx; a.kp = p->kp;
    const int npair = (int)((p->kp + 1) / 2), bs = npair >= 256 ? 256 : 64;          // pairs of columns, the row padding included
    const dim3 grid((npair + bs - 1) / bs, p->ny);
#define OA_MIX(NC) \
    do { \
        if (in) hipLaunchKernelGGL((grf_mix_kernel<T, NC, true>), grid, dim3(bs), 0, st, a); \
        else hipLaunchKernelGGL((grf_mix_kernel<T, NC, false>), grid, dim3(bs), 0, st, a); \
    } while (0)
    if (ncomp == 1) OA_MIX(1);
    else if (ncomp == 2) OA_MIX(2);
    else OA_MIX(3);
#undef OA_MIX
    OA_LAUNCH_CHECK();
    return 0;
}

}  // namespace oa

using namespace oa;

extern "C" {

int oa_grf_hc_band(oa_plan* p, uint64_t seed, uint64_t stream_id, const void* covsqrt_hc, void* hc_out, int width, int rband,
                   void* stream) {
    return oa::grf_hc_band_batch(p, seed, stream_id, 1, covsqrt_hc, hc_out, 0, width, rband, (hipStream_t)stream);
}
}  // extern "C"

namespace oa {
int grf_hc_band_batch(oa_plan* p, uint64_t seed, uint64_t stream_id, int nreal, const void* covsqrt_hc, void* hc_out, long zstride,
                      int width, int rband, hipStream_t stream) {
    OA_REQUIRE(p && hc_out && nreal >= 1, "oa_grf_hc: NULL argument");
    const int npair = p->nx / 4 + 1;
    int wpairs = (width > 0 && width < p->nx / 2 + 1) ? (width + 1) / 2 : 0;          // column pairs drawn (0 = all)
    if (wpairs >= npair) wpairs = 0;
    const int rb = (rband > 0 && 2L * rband - 1 < p->ny) ? rband : 0;
    const int np = wpairs ? wpairs : npair;
    const int bs = np >= 256 ? 256 : 64;
    dim3 grid((np + bs - 1) / bs, rb ? 2 * rb - 1 : p->ny, nreal);
    if (p->dtype == OA_F32)
        hipLaunchKernelGGL(grf_hc_kernel<float>, grid, dim3(bs), 0, stream, seed, stream_id,
                           (const float*)covsqrt_hc, (cx<float>*)hc_out, p->ny, p->nx, p->kp, wpairs, rb, zstride);
    else
        hipLaunchKernelGGL(grf_hc_kernel<double>, grid, dim3(bs), 0, stream, seed, stream_id,
                           (const double*)covsqrt_hc, (cx<double>*)hc_out, p->ny, p->nx, p->kp, wpairs, rb, zstride);
    OA_LAUNCH_CHECK();
    return 0;
}
}  // namespace oa

extern "C" {

int oa_grf_hc(oa_plan* p, uint64_t seed, uint64_t stream_id, const void* covsqrt_hc, void* hc_out, void* stream) {
    return oa_grf_hc_band(p, seed, stream_id, covsqrt_hc, hc_out, 0, 0, stream);
}

int oa_grf_mix(oa_plan* p, uint64_t seed, uint64_t stream_id0, int ncomp, const void* const* covsqrt_hc, const void* rot_c, const void* rot_s,
               const void* const* hc_in, const void* filt_hcreal, double scale, void* const* hc_out, void* stream) {
    OA_REQUIRE(p && covsqrt_hc && hc_out && ncomp >= 1 && ncomp <= 3, "oa_grf_mix: bad argument (1 <= ncomp <= 3)");
    OA_REQUIRE((rot_c == nullptr) == (rot_s == nullptr), "oa_grf_mix: rotation needs both planes");
    OA_REQUIRE(!rot_c || ncomp == 3, "oa_grf_mix: the rotation acts on components 1, 2 of three");
    for (int i = 0; i < ncomp; ++i) {
        OA_REQUIRE(hc_out[i], "oa_grf_mix: NULL output plane");
        OA_REQUIRE(!hc_in || hc_in[i], "oa_grf_mix: NULL input plane");
    }
    OA_REQUIRE(hc_in || !filt_hcreal, "oa_grf_mix: a filter without input planes");
    return p->dtype == OA_F32 ? grf_mix_launch<float>(p, seed, stream_id0, ncomp, covsqrt_hc, rot_c, rot_s, hc_in, filt_hcreal, scale, hc_out, (hipStream_t)stream)
                              : grf_mix_launch<double>(p, seed, stream_id0, ncomp, covsqrt_hc, rot_c, rot_s, hc_in, filt_hcreal, scale, hc_out, (hipStream_t)stream);
}

int oa_randn(int dtype, uint64_t seed, uint64_t stream_id, void* out, long n, void* stream) {
    OA_REQUIRE(out && n >= 0, "oa_randn: bad argument");
    const int g = flat_grid((n + 3) / 4 > 0 ? (n + 3) / 4 : 1);
    if (dtype == OA_F32)
        hipLaunchKernelGGL(randn_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, seed, stream_id, (float*)out, n);
    else if (dtype == OA_F64)
        hipLaunchKernelGGL(randn_kernel<double>, dim3(g), dim3(256), 0, (hipStream_t)stream, seed, stream_id, (double*)out, n);
    else
        return fail("oa_randn: bad dtype");
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_moments_add(const double* x, int d, int64_t* n, double* S, double* C, void* stream) {
    OA_REQUIRE(x && n && S && C && d >= 1, "oa_moments_add: bad argument");
    const int g = flat_grid((long)d * d, 256, 64);
    hipLaunchKernelGGL(moments_add_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, d, n, S, C);
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_moments_add_binned(const double* sums, const int64_t* counts, int d, int64_t* n, double* S, double* C, void* stream) {
    OA_REQUIRE(sums && counts && n && S && C && d >= 1, "oa_moments_add_binned: bad argument");
    const int g = flat_grid((long)d * d, 256, 64);
    hipLaunchKernelGGL(moments_add_binned_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, sums, counts, d, n, S, C);
    OA_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
