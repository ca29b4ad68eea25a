// 2-D FFT passes for map sides of the form 2^a 3^b 5^c that are not powers of two (round 5).
//
// The reference's own geometries are such sides: tutorials/tt_verification.ipynb cell 1 is a 10 degree patch at 0.5' = 1200^2
// (2^4 3 5^2), mapwork.ipynb cell 3 is 2400^2, demo-grf.ipynb cell 5 is 600^2; its FFT (maps.py:1613, pixell -> FFTW / ducc) takes any
// size at full speed.  Until round 4 these sides went through a 2-D chirp-z (Bluestein) convolution on an inner 4096^2 complex plan
// (czt.hip): two complex 4096^2 transforms + four elementwise passes per transform of a 1200^2 real map.  Here they are ordinary
// mixed-radix Stockham transforms staged in LDS:
//   * one sequence (row passes) or a tile of C adjacent columns (column passes) per workgroup, ping-pong between two LDS buffers,
//     one workgroup barrier per stage; radices 4, 2, 3, 5 chosen on the host (mixed_factor);
//   * stage of radix R after sub-length Ns:  v_t = in[j + t N/R] W_N^(k t N/(Ns R)),  k = j mod Ns;  out[(j - k) R + k + t Ns] = DFT_R(v)_t
//     (the autosort form of fft_kernels.hpp with a run-time radix list); factors from the W_N table (L1/L2-resident);
//   * real rows as packed N/2-point transforms with the (un)tangle of fft_kernels.hpp (N even: every map side is);
//   * inverse transforms as the forward transform of the swapped data, IDFT(x) = swap(DFT(swap(x))).
// Sides with another prime factor keep the chirp-z path.  The fused estimator kernels remain power-of-two; on these sides the
// estimators run the modular calls (oa_qe_legs / oa_mul_real / oa_qe_div) over these transforms.
#pragma once
#include "cx.hpp"

namespace oa {

constexpr int MR_MAXSTAGES = 14;

struct MrFactors {
    int n, r[MR_MAXSTAGES];
};
// N = product of r[i], r[i] in {4, 2, 3, 5}; n = 0: N has another prime factor (or is too long)
inline MrFactors mixed_factor(int N) {
    MrFactors f{};
    int m = N;
    const int rad[4] = {4, 2, 3, 5};
    for (int q = 0; q < 4; ++q)
        while (m % rad[q] == 0 && f.n < MR_MAXSTAGES) { f.r[f.n++] = rad[q]; m /= rad[q]; }
    if (m != 1) f.n = 0;
    return f;
}
inline bool mixed_ok(int N) { return N >= 2 && mixed_factor(N).n > 0; }

// multiply by i^q
template <typename T> OA_HD cx<T> mr_rot_i(cx<T> x, int q) {
    switch (q & 3) {
        case 0: return x;
        case 1: return mk<T>(-x.y, x.x);
        case 2: return mk<T>(-x.x, -x.y);
        default: return mk<T>(x.y, -x.x);
    }
}

template <typename T, int R> struct MrDft;
template <typename T> struct MrDft<T, 2> {
    static OA_HD void run(cx<T>* v) { const cx<T> a = v[0] + v[1], b = v[0] - v[1]; v[0] = a; v[1] = b; }
};
template <typename T> struct MrDft<T, 3> {
    static OA_HD void run(cx<T>* v) {
        const T hs = (T)0.86602540378443864676L;
        const cx<T> s = v[1] + v[2], d = (v[1] - v[2]) * hs, m = v[0] - s * (T)0.5;
        v[0] = v[0] + s; v[1] = add_mi(m, d); v[2] = add_pi(m, d);
    }
};
template <typename T> struct MrDft<T, 4> {
    static OA_HD void run(cx<T>* v) {
        const cx<T> a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = v[1] - v[3];
        v[0] = a + c; v[2] = a - c; v[1] = add_mi(b, d); v[3] = add_pi(b, d);
    }
};
template <typename T> struct MrDft<T, 5> {
    static OA_HD void run(cx<T>* v) {
        // y_k = x0 + sum_{t=1..4} x_t w^(t k), w = exp(-2 pi i / 5): with s1 = x1 + x4, s2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3,
        // y_{1,4} = x0 + c1 s1 + c2 s2 -/+ i (n1 d1 + n2 d2),  y_{2,3} = x0 + c2 s1 + c1 s2 -/+ i (n2 d1 - n1 d2)
        const T c1 = (T)0.30901699437494742410L, c2 = (T)-0.80901699437494742410L;      // cos(2 pi / 5), cos(4 pi / 5)
        const T n1 = (T)0.95105651629515357212L, n2 = (T)0.58778525229247312917L;       // sin(2 pi / 5), sin(4 pi / 5)
        const cx<T> s1 = v[1] + v[4], s2 = v[2] + v[3], d1 = v[1] - v[4], d2 = v[2] - v[3];
        const cx<T> a1 = v[0] + s1 * c1 + s2 * c2, a2 = v[0] + s1 * c2 + s2 * c1;
        const cx<T> b1 = d1 * n1 + d2 * n2, b2 = d1 * n2 - d2 * n1;
        v[0] = v[0] + s1 + s2;
        v[1] = add_mi(a1, b1); v[4] = add_pi(a1, b1);
        v[2] = add_mi(a2, b2); v[3] = add_pi(a2, b2);
    }
};

// one Stockham stage over a [N][C] tile (C = 1 << logC sequences side by side, element n of sequence c at n C + c), A -> B
template <typename T, int R>
OA_HD void mr_stage(const cx<T>* A, cx<T>* B, int N, int Ns, int logC, const cx<T>* tw, int tid, int NT) {
    const int nb = N / R, step = N / (Ns * R), total = nb << logC, cm = (1 << logC) - 1;
    for (int i = tid; i < total; i += NT) {
        const int c = i & cm, j = i >> logC, k = j % Ns;
        cx<T> v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = A[((j + t * nb) << logC) + c];
        if (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = v[t] * tw[k * t * step];        // W_(Ns R)^(k t); k t step < N
        }
        MrDft<T, R>::run(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) B[((j0 + t * Ns) << logC) + c] = v[t];
    }
}

// all stages; data in buf0 (caller synced); returns the buffer that holds the result (synced)
template <typename T, class Ctx>
OA_HD cx<T>* mr_transform(Ctx& ctx, cx<T>* buf0, cx<T>* buf1, int N, const MrFactors& f, int logC, const cx<T>* tw, int tid, int NT) {
    cx<T>* A = buf0;
    cx<T>* B = buf1;
    int Ns = 1;
    for (int s = 0; s < f.n; ++s) {
        switch (f.r[s]) {
            case 2: mr_stage<T, 2>(A, B, N, Ns, logC, tw, tid, NT); break;
            case 3: mr_stage<T, 3>(A, B, N, Ns, logC, tw, tid, NT); break;
            case 4: mr_stage<T, 4>(A, B, N, Ns, logC, tw, tid, NT); break;
            default: mr_stage<T, 5>(A, B, N, Ns, logC, tw, tid, NT); break;
        }
        ctx.sync();
        Ns *= f.r[s];
        cx<T>* t = A; A = B; B = t;
    }
    return A;
}

enum MrMode { MR_C2C_F = 0, MR_C2C_I = 1, MR_R2C = 2, MR_C2R = 3 };

template <typename T>
struct MrRowArgs {
    const void* in;
    void* out;
    long in_pitch, out_pitch;       // elements of the respective type (real rows: reals; complex rows: complex)
    int N;                           // complex transform length (real modes: nx / 2)
    MrFactors f;
    const cx<T>* tw;                 // W_N^e, e < N (length of THIS transform)
    const cx<T>* tw2;                // real modes: W_(2N)^e, e <= N (the (un)tangle factors)
    T scale;
    int mode;
    // X-DERIVATIVE C2R (oa_lens_maps on these sides; RowArgs::dlx of fft_kernels.hpp): dlx != nullptr -> launch plane z (grid y) is the
    // C2R of (i lx)^(dpow0 + z) x the SAME input (a column-transformed field that already carries (i ly)^dcol_b) and goes to output
    // plane n (n + 1) / 2 - 1 + dcol_b, n = dpow0 + z + dcol_b (the order lens_taylor_kernel reads), out_zoff reals per plane
    const T* dlx;
    int dpow0, dcol_b;
    long out_zoff;
};

// one row per workgroup
template <typename T, class Ctx>
OA_HD void mr_row_body(Ctx& ctx, const MrRowArgs<T>& a) {
    cx<T>* b0 = reinterpret_cast<cx<T>*>(ctx.smem());
    cx<T>* b1 = b0 + a.N + 1;
    const int tid = ctx.tid(), NT = ctx.nthreads(), N = a.N;
    const long row = ctx.bid_x();
    if (a.mode == MR_C2C_F || a.mode == MR_C2C_I) {
        const bool inv = a.mode == MR_C2C_I;
        const cx<T>* src = reinterpret_cast<const cx<T>*>(a.in) + row * a.in_pitch;
        cx<T>* dst = reinterpret_cast<cx<T>*>(a.out) + row * a.out_pitch;
        for (int n = tid; n < N; n += NT) b0[n] = inv ? swp(src[n]) : src[n];
        ctx.sync();
        const cx<T>* r = mr_transform<T>(ctx, b0, b1, N, a.f, 0, a.tw, tid, NT);
        for (int n = tid; n < N; n += NT) { const cx<T> v = r[n] * a.scale; dst[n] = inv ? swp(v) : v; }
    } else if (a.mode == MR_R2C) {
        // packed: z[n] = x[2 n] + i x[2 n + 1]; X[k] = E[k] + W_2N^k O[k], E = (Z[k] + conj Z[N - k]) / 2, O = (Z[k] - conj Z[N - k]) / 2i
        const cx<T>* src = reinterpret_cast<const cx<T>*>(reinterpret_cast<const T*>(a.in) + row * a.in_pitch);
        cx<T>* dst = reinterpret_cast<cx<T>*>(a.out) + row * a.out_pitch;
        for (int n = tid; n < N; n += NT) b0[n] = src[n];
        ctx.sync();
        const cx<T>* r = mr_transform<T>(ctx, b0, b1, N, a.f, 0, a.tw, tid, NT);
        for (int k = tid; k <= N; k += NT) {
            const cx<T> Zk = r[k == N ? 0 : k], Zm = conj(r[k == 0 ? 0 : N - k]);
            const cx<T> E = (Zk + Zm) * (T)0.5, O = mul_mi(Zk - Zm) * (T)0.5;
            dst[k] = (E + a.tw2[k] * O) * a.scale;
        }
    } else {
        // C2R: Z'[k] = (X[k] + conj X[N - k]) + i conj(W_2N^k) (X[k] - conj X[N - k]), k < N; inverse of the packed transform; the
        // self-conjugate columns k = 0 and k = N keep their Hermitian (real) part only, as ifft(...).real does
        const cx<T>* src = reinterpret_cast<const cx<T>*>(a.in) + row * a.in_pitch;
        T* outp = reinterpret_cast<T*>(a.out);
        int apow = 0;
        if (a.dlx) {
            apow = a.dpow0 + ctx.bid_y();
            const int nn = apow + a.dcol_b;
            outp += (long)(nn * (nn + 1) / 2 - 1 + a.dcol_b) * a.out_zoff;
        }
        cx<T>* dst = reinterpret_cast<cx<T>*>(outp + row * a.out_pitch);
        for (int k = tid; k < N; k += NT) {
            cx<T> A = src[k], B = src[N - k];
            if (a.dlx) {                                    // (i lx)^apow at the load; lx = 0 at the self-conjugate Nyquist column
                T fa = (T)1, fb = (T)1;
                const T la = a.dlx[k], lb = a.dlx[N - k];
                for (int i = 0; i < apow; ++i) { fa = fa * la; fb = fb * lb; }
                A = mr_rot_i(A, apow) * fa;
                B = mr_rot_i(B, apow) * fb;
            }
            if (k == 0) { A.y = (T)0; B.y = (T)0; }
            const cx<T> z = (A + conj(B)) + mul_pi(conj(a.tw2[k]) * (A - conj(B)));
            b0[k] = swp(z);
        }
        ctx.sync();
        const cx<T>* r = mr_transform<T>(ctx, b0, b1, N, a.f, 0, a.tw, tid, NT);
        for (int n = tid; n < N; n += NT) dst[n] = swp(r[n]) * a.scale;
    }
}

template <typename T>
struct MrColArgs {
    const cx<T>* in;
    cx<T>* out;
    long in_pitch, out_pitch;
    int N, width, logC;
    MrFactors f;
    const cx<T>* tw;
    T scale;
    int inverse;
    const T* dly;                    // != nullptr: the input is multiplied by (i ly[n])^dpow at the load (y-derivative of a field's transform)
    int dpow;
};

// a tile of C adjacent columns per workgroup, transformed along y
template <typename T, class Ctx>
OA_HD void mr_col_body(Ctx& ctx, const MrColArgs<T>& a) {
    cx<T>* b0 = reinterpret_cast<cx<T>*>(ctx.smem());
    const int C = 1 << a.logC, N = a.N;
    cx<T>* b1 = b0 + (long)N * C;
    const int tid = ctx.tid(), NT = ctx.nthreads();
    const int c0 = ctx.bid_x() << a.logC;
    int ncols = a.width - c0;
    if (ncols > C) ncols = C;
    const bool inv = a.inverse != 0;
    for (int i = tid; i < N * C; i += NT) {
        const int c = i & (C - 1), n = i >> a.logC;
        cx<T> v = mk<T>((T)0, (T)0);
        if (c < ncols) v = a.in[(long)n * a.in_pitch + c0 + c];
        if (a.dly) {
            T f = (T)1;
            const T l = a.dly[n];
            for (int q = 0; q < a.dpow; ++q) f = f * l;
            v = mr_rot_i(v, a.dpow) * f;
        }
        b0[i] = inv ? swp(v) : v;
    }
    ctx.sync();
    const cx<T>* r = mr_transform<T>(ctx, b0, b1, N, a.f, a.logC, a.tw, tid, NT);
    for (int i = tid; i < N * C; i += NT) {
        const int c = i & (C - 1), n = i >> a.logC;
        if (c < ncols) { const cx<T> v = r[i] * a.scale; a.out[(long)n * a.out_pitch + c0 + c] = inv ? swp(v) : v; }
    }
}

}  // namespace oa
