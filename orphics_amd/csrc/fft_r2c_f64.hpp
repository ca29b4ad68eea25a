// Two-waves-per-row R2C pass for band-limited consumers in float64 / complex128 (the reference's arithmetic, maps.py:1613):
// rows of 8192 reals = 4096 packed complex points, <= 512 columns kept.
//
// The general row pass keeps a whole complex128 row (64 KB) in LDS and goes through it three times (three radix-16 stages,
// seven workgroup barriers): 128 us per 8192^2 map = 4.6 TB/s.  Here, as in the f32 one-wave-per-row kernel
// (fft_r2c_w64.hpp), the points live in registers -- 32 complex128 per lane = 128 VGPRs, so a row takes TWO waves:
// wave w transforms the packed samples of parity w,
//     Zw[k] = sum_n' z[2 n' + w] W2048^(n' k),   n' = j + 64 t  (lane j, t < 32),   k = k1 + 32 k2:
//   stage 1 (lane j):        B_j[k1] = W2048^(j k1) DFT32_t( z_w[j + 64 t] )                              -> LDS[k1][j]
//   stage 2 (lane k1 + 32 p): the 64-point transform over j of row k1, split radix-2 (decimation in frequency): lane p
//                            takes the outputs k2 = 2 m + p:  u[jj] = (B_jj + (-1)^p B_(jj+32)) W64^(jj p),
//                            Zw[k1 + 32 p + 64 m] = DFT32_jj(u)[m]     -- lane l = k1 + 32 p holds the bins l + 64 m
//   only m < 8 (columns < 512) and m >= 24 (their mirror images, for the real-transform untangle) are computed.
// The transposes go through one 8-byte plane per wave twice (real parts, then imaginary parts; the transposed values
// land in the slots just vacated), then the two waves exchange their 16 kept bins and each finishes half of the columns:
//     Z[k] = Ze[k] + W4096^k Zo[k],   Z[4096 - k] = Ze[2048 - k] + conj(W4096^k) Zo[2048 - k],   X[k] = E + W8192^k O.
// LR = 2 adds the R-SPLIT of the column transform (RowArgs::lr; fft_r2c_w64.hpp): the workgroup walks the four rows
// g + my n of its group and keeps the radix-4 butterfly in LDS.
#pragma once
#include "fft_kernels.hpp"

namespace oa {

struct RowF64Args {
    const cx<double>* in;     // real rows viewed as packed complex: z[n] = x[2n] + i x[2n+1]
    cx<double>* out;
    long in_pitch, out_pitch; // complex elements
    const cx<double>* tw;     // W_M^k, M = 2^logTw >= 8192
    int logTw;
    double scale;
    int wcols;                // columns produced (<= 512)
    int ny, nwg;
    long kplane;              // R-split: output plane k1 sits kplane elements behind plane 0
    const cx<double>* twy;    // R-split: W_ny^k
};

constexpr int F64_STRIDE = 65;                                                  // doubles per row of the transpose plane
constexpr size_t F64_PLANE_BYTES = (size_t)32 * F64_STRIDE * sizeof(double);     // per wave
constexpr size_t F64_EX_BYTES = (size_t)2 * 16 * 64 * sizeof(cx<double>);       // the two waves' 16 kept bins per lane (aliases the planes)
constexpr size_t F64_ACC_BYTES = (size_t)2 * 3 * 4 * 64 * sizeof(cx<double>);    // R-split: [wave][slot][i][lane]
constexpr size_t F64_LDS_BYTES = 2 * F64_PLANE_BYTES > F64_EX_BYTES ? 2 * F64_PLANE_BYTES : F64_EX_BYTES;

// W32^m = exp(-2 pi i m / 32), W64^m likewise (compile-time arguments after unrolling: the table reads fold to literals)
struct W64dTab {
    static constexpr double c[17] = {1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494, 0.92387953251128675613,
                                     0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081, 0.70710678118654752440,
                                     0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173,
                                     0.29028467725446236764, 0.19509032201612826785, 0.098017140329560601994, 0.0};
};
OA_HD cx<double> w64d(int m) {
    m &= 63;
    const int q = m >> 4, r = m & 15;
    const double x = W64dTab::c[r], y = -W64dTab::c[16 - r];      // quadrant symmetry: W^(m + 16) = -i W^m
    return q == 0 ? mk<double>(x, y) : (q == 1 ? mk<double>(y, -x) : (q == 2 ? mk<double>(-x, -y) : mk<double>(-y, x)));
}
OA_HD cx<double> w32d(int m) { return w64d(2 * m); }

// In-register DFT of 32 points, in place.  Input natural order; output bin m = a + 8 b (a < 8, b < 4) is left in v[4 a + b].
// PRUNE: only the bins b = 0 (m < 8) and b = 3 (m >= 24) are produced.
template <bool PRUNE>
OA_HD void dft32(cx<double>* v) {
    // inner layer: for each s0 < 4, DFT-8 over s1 of v[4 s1 + s0] -> bin a, times W32^(s0 a), stored at v[4 a + s0]
#pragma unroll
    for (int s0 = 0; s0 < 4; ++s0) {
        cx<double> t[8];
#pragma unroll
        for (int s1 = 0; s1 < 8; ++s1) t[s1] = v[4 * s1 + s0];
        Dft<double, 8>::run(t);
#pragma unroll
        for (int a = 0; a < 8; ++a) v[4 * a + s0] = (s0 * a) ? t[a] * w32d(s0 * a) : t[a];
    }
    // outer layer: for each a, DFT-4 over s0 of v[4 a + s0] -> bin b, i.e. output m = a + 8 b, stored at v[4 a + b]
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        cx<double>* t = v + 4 * a;
        const cx<double> p02 = t[0] + t[2], m02 = t[0] - t[2], p13 = t[1] + t[3], m13 = t[1] - t[3];
        if (!PRUNE) {
            t[0] = p02 + p13;
            t[1] = add_mi(m02, m13);
            t[2] = p02 - p13;
            t[3] = add_pi(m02, m13);
        } else {
            t[0] = p02 + p13;
            t[3] = add_pi(m02, m13);
        }
    }
}

// one row by the two waves of the workgroup: X[i] = untangled output column l + 64 (4 w + i), i < 4 (times scale)
template <class Ctx>
OA_HD void f64_row(Ctx& ctx, const RowF64Args& a, const cx<double>* src, cx<double> P1, cx<double> Q1, const cx<double>* Wk, const cx<double>* U,
                   cx<double>* X) {
    const int tid = ctx.tid(), w = tid >> 6, l = tid & 63;
    double* plane = reinterpret_cast<double*>(ctx.smem()) + (size_t)w * 32 * F64_STRIDE;
    cx<double>* EX = reinterpret_cast<cx<double>*>(ctx.smem());
    cx<double> v[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 x = __builtin_nontemporal_load(reinterpret_cast<const d2*>(src + 128 * t));     // packed sample 2 (l + 64 t) + w
        v[t] = mk<double>(x.x, x.y);
#else
        v[t] = src[128 * t];
#endif
    }
    dft32<false>(v);                           // bin k1 = aa + 8 b sits in v[4 aa + b]
    {   // W2048^(l k1) = P1^aa Q1^b, P1 = W2048^l, Q1 = W2048^(8 l)
        cx<double> P[8], Q[4];
        P[1] = P1; P[2] = P1 * P1; P[3] = P[2] * P1; P[4] = P[2] * P[2]; P[5] = P[4] * P1; P[6] = P[4] * P[2]; P[7] = P[4] * P[3];
        Q[1] = Q1; Q[2] = Q1 * Q1; Q[3] = Q[2] * Q1;
#pragma unroll
        for (int aa = 0; aa < 8; ++aa)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (aa + 8 * b) v[4 * aa + b] = v[4 * aa + b] * ((aa && b) ? P[aa] * Q[b] : (aa ? P[aa] : Q[b]));
    }
    const int k1 = l & 31, p = l >> 5;
    const double sg = p ? -1.0 : 1.0;
    // transpose + first radix-2 of stage 2, real parts then imaginary parts: lane l writes [k1'][l] for its 32 bins k1',
    // lane (k1, p) reads row k1 and forms u[jj] = B_jj + sg B_(jj+32) in the slots just vacated
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
        for (int b = 0; b < 4; ++b) plane[(aa + 8 * b) * F64_STRIDE + l] = v[4 * aa + b].x;
    ctx.sync();
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) v[jj].x = plane[k1 * F64_STRIDE + jj] + sg * plane[k1 * F64_STRIDE + jj + 32];
    ctx.sync();
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
        for (int b = 0; b < 4; ++b) plane[(aa + 8 * b) * F64_STRIDE + l] = v[4 * aa + b].y;
    ctx.sync();
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) v[jj].y = plane[k1 * F64_STRIDE + jj] + sg * plane[k1 * F64_STRIDE + jj + 32];
    ctx.sync();
    // odd outputs (p = 1): twiddle W64^jj
#pragma unroll
    for (int jj = 1; jj < 32; ++jj) {
        const cx<double> wj = w64d(jj);
        v[jj] = v[jj] * mk<double>(p ? wj.x : 1.0, p ? wj.y : 0.0);      // (a per-lane factor, not a select between aggregates)
    }
    dft32<true>(v);                            // Zw[l + 64 m]: m = aa in v[4 aa], m = 24 + aa in v[4 aa + 3]
    // exchange: EX[wave][slot][lane], slots 0..7 = bins m, 8..15 = bins 24 + m
#pragma unroll
    for (int aa = 0; aa < 8; ++aa) {
        EX[(w * 16 + aa) * 64 + l] = v[4 * aa];
        EX[(w * 16 + 8 + aa) * 64 + l] = v[4 * aa + 3];
    }
    ctx.sync();
    const int lm = (64 - l) & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = 4 * w + i;                                             // this wave finishes columns l + 64 m
        const cx<double> Zk = EX[m * 64 + l] + Wk[i] * EX[(16 + m) * 64 + l];
        // partner bin 2048 - k of the half transforms: lane (64 - l) & 63, bin m' = 31 - m (l > 0) or 32 - m (l = 0; m = 0: Z[0] itself)
        const int mp = l ? (31 - m) : (32 - m);
        const int sp = 8 + ((mp - 24) & 7);                                  // its slot (8..15); (m = 0, l = 0 never reads it)
        cx<double> Zm = Zk;
        if (m > 0 || l) Zm = EX[sp * 64 + lm] + conj(Wk[i]) * EX[(16 + sp) * 64 + lm];
        const cx<double> E = (Zk + conj(Zm)) * 0.5;
        const cx<double> O = mul_mi(Zk - conj(Zm)) * 0.5;
        X[i] = (E + U[i] * O) * a.scale;
    }
    ctx.sync();                                // the exchange reads precede the next row's plane writes
}

template <int LR, class Ctx>
OA_HD void row_r2c_f64_body(Ctx& ctx, const RowF64Args& a) {
    const int tid = ctx.tid(), w = tid >> 6, l = tid & 63;
    const int sh = a.logTw - 11;                                             // W2048^e = tw[e << sh]
    const cx<double> P1 = a.tw[(unsigned)l << sh], Q1 = a.tw[(unsigned)((8 * l) & 2047) << sh];
    cx<double> Wk[4], U[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = l + 64 * (4 * w + i);
        Wk[i] = a.tw[(unsigned)k << (sh - 1)];                               // W4096^k
        U[i] = a.tw[(unsigned)k << (sh - 2)];                                // W8192^k
    }
    if constexpr (LR == 0) {
        for (long row = ctx.bid_x(); row < a.ny; row += a.nwg) {
            cx<double> X[4];
            f64_row(ctx, a, a.in + row * a.in_pitch + w + 2 * l, P1, Q1, Wk, U, X);
            cx<double>* dst = a.out + row * a.out_pitch + l + 256 * w;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (l + 64 * (4 * w + i) < a.wcols) dst[64 * i] = X[i];
        }
    } else {
        static_assert(LR == 2, "R = 4");
        const long ngroups = a.ny >> LR;
        cx<double>* accl = reinterpret_cast<cx<double>*>(reinterpret_cast<char*>(ctx.smem()) + F64_LDS_BYTES) + (size_t)w * 3 * 4 * 64 + l;
        for (long g = ctx.bid_x(); g < ngroups; g += a.nwg) {
            cx<double> wy[4];
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) wy[k1] = a.twy[((unsigned)g * (unsigned)k1) & (unsigned)(a.ny - 1)];
            // rows in the order n = 0, 2, 1, 3: a = X0 + X2, b = X0 - X2, then Y0 = a + c, Y2 = a - c, Y1 = b - i d, Y3 = b + i d
#pragma unroll 1
            for (int step = 0; step < 4; ++step) {
                const int n = ((step & 1) << 1) | (step >> 1);
                cx<double> X[4];
                f64_row(ctx, a, a.in + (g + n * ngroups) * a.in_pitch + w + 2 * l, P1, Q1, Wk, U, X);
                if (step == 0 || step == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) accl[((step ? 2 : 0) * 4 + i) * 64] = X[i];
                } else if (step == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const cx<double> x0 = accl[i * 64];
                        accl[i * 64] = x0 + X[i];
                        accl[(4 + i) * 64] = x0 - X[i];
                    }
                } else {
                    cx<double>* dst = a.out + g * a.out_pitch + l + 256 * w;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (l + 64 * (4 * w + i) < a.wcols) {
                            const cx<double> aa = accl[i * 64], bb = accl[(4 + i) * 64], x1 = accl[(8 + i) * 64];
                            const cx<double> c = x1 + X[i], d = x1 - X[i];
                            dst[64 * i] = (aa + c) * wy[0];
                            dst[a.kplane + 64 * i] = add_mi(bb, d) * wy[1];
                            dst[2 * a.kplane + 64 * i] = (aa - c) * wy[2];
                            dst[3 * a.kplane + 64 * i] = add_pi(bb, d) * wy[3];
                        }
                }
            }
        }
    }
}

}  // namespace oa
