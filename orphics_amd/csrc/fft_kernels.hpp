// LDS-staged Stockham FFT passes for gfx950 (wave64), written against an
// abstract execution context so the identical bodies run (a) as HIP kernels
// and (b) under the CPU thread emulator in tests/emul (index-math validation
// without a GPU).
//
// Design (DESIGN.md section "K1"):
//  * every thread owns EPT=16 complex points per stage: one radix-16 butterfly
//    or 16/R radix-R butterflies, register resident, fully unrolled;
//  * a stage = LDS->regs, twiddle (table lookup, exact to 0.5 ulp), in-register
//    DFT-R, barrier, regs->LDS at the Stockham (autosort) position, barrier;
//  * row passes keep a whole row (or C short rows) in LDS, padded 1 element per
//    16 so the stride-R first-stage writes are bank-conflict free;
//  * column passes work on [L points][C=32 columns] tiles so every global
//    access is a >=256-byte contiguous segment; a length-Ny column transform is
//    split four-step style Ny = N1*N2 into two such passes (pass 1 applies the
//    inter-pass twiddle and writes transposed blocks, pass 2 is in place);
//  * inverse transforms use IDFT(x) = swap(DFT(swap(x))) so one forward
//    butterfly/twiddle set serves both directions;
//  * real transforms use the packed N/2-point trick with an in-LDS
//    (un)tangle step, so a real row costs half the LDS and flops.
#pragma once
#include "cx.hpp"

namespace oa {

constexpr int EPT = 16;        // complex points per thread per stage
constexpr int MAX_STAGES = 8;

struct Stages {
    int n;
    int radix[MAX_STAGES];
};

enum RowMode { ROW_R2C = 0, ROW_C2R = 1, ROW_C2C_F = 2, ROW_C2C_I = 3 };

// ---- constant twiddles W16^k = exp(-2 pi i k / 16), k = 0..7 -------------
template <typename T>
OA_HD cx<T> w16(int k) {
    const T c1 = (T)0.92387953251128673848L;  // cos(pi/8)
    const T s1 = (T)0.38268343236508978178L;  // sin(pi/8)
    const T h = (T)0.70710678118654752440L;   // sqrt(1/2)
    switch (k) {
        case 0: return mk<T>((T)1, (T)0);
        case 1: return mk<T>(c1, -s1);
        case 2: return mk<T>(h, -h);
        case 3: return mk<T>(s1, -c1);
        case 4: return mk<T>((T)0, (T)-1);
        case 5: return mk<T>(-s1, -c1);
        case 6: return mk<T>(-h, -h);
        default: return mk<T>(-c1, -s1);
    }
}

// ---- in-register forward DFT of R points, natural order in and out --------
template <typename T, int R>
struct Dft {
    static OA_HD void run(cx<T>* v) {
        cx<T> e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = v[2 * k];
            o[k] = v[2 * k + 1];
        }
        Dft<T, R / 2>::run(e);
        Dft<T, R / 2>::run(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            cx<T> t;
            if (k == 0) t = o[k];
            else if (4 * k == R) t = mul_mi(o[k]);
            else t = o[k] * w16<T>(k * (16 / R));
            v[k] = e[k] + t;
            v[k + R / 2] = e[k] - t;
        }
    }
};
template <typename T>
struct Dft<T, 1> {
    static OA_HD void run(cx<T>*) {}
};

template <int R> struct Log2c;
template <> struct Log2c<2> { static constexpr int v = 1; };
template <> struct Log2c<4> { static constexpr int v = 2; };
template <> struct Log2c<8> { static constexpr int v = 3; };
template <> struct Log2c<16> { static constexpr int v = 4; };

// LDS address of point n of sequence c.
//  ROWMAJOR: sequences are rows, contiguous in n, padded 1 per 16.
//  else    : tile [L][C], contiguous in c.
template <bool ROWMAJOR>
OA_HD int lds_addr(int n, int c, int logC, int rowStride) {
    if (ROWMAJOR) return c * rowStride + n + (n >> 4);
    return (n << logC) + c;
}

// ---- one Stockham stage, part A: LDS -> regs, twiddle, DFT-R --------------
template <typename T, int R, bool ROWMAJOR>
OA_HD void stage_a(const cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int rowStride,
                   int logNs, const cx<T>* tw, int logTw) {
    constexpr int LR = Log2c<R>::v;
    constexpr int NB = EPT / R;
    const int logLR = logL - LR;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        int j, c;
        if (ROWMAJOR) { j = b & ((1 << logLR) - 1); c = b >> logLR; }
        else { c = b & ((1 << logC) - 1); j = b >> logC; }
#pragma unroll
        for (int t = 0; t < R; ++t)
            v[u * R + t] = s[lds_addr<ROWMAJOR>(j + (t << logLR), c, logC, rowStride)];
        if (logNs > 0) {
            const int k = j & ((1 << logNs) - 1);
            const int sh = logTw - logNs - LR;
#pragma unroll
            for (int t = 1; t < R; ++t) v[u * R + t] = v[u * R + t] * tw[(t * k) << sh];
        }
        Dft<T, R>::run(v + u * R);
    }
}

// ---- part B: regs -> LDS at the autosort position -------------------------
template <typename T, int R, bool ROWMAJOR>
OA_HD void stage_b(cx<T>* s, const cx<T>* v, int tid, int NT, int logL, int logC, int rowStride, int logNs) {
    constexpr int LR = Log2c<R>::v;
    constexpr int NB = EPT / R;
    const int logLR = logL - LR;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        int j, c;
        if (ROWMAJOR) { j = b & ((1 << logLR) - 1); c = b >> logLR; }
        else { c = b & ((1 << logC) - 1); j = b >> logC; }
        const int k = j & ((1 << logNs) - 1);
        const int base = ((j - k) << LR) + k;
#pragma unroll
        for (int t = 0; t < R; ++t)
            s[lds_addr<ROWMAJOR>(base + (t << logNs), c, logC, rowStride)] = v[u * R + t];
    }
}

// all stages of one batch of LDS-resident sequences (forward DFT, in place).
// Entry: data in LDS, barrier already passed.  Exit: results in LDS, barrier passed.
template <typename T, bool ROWMAJOR, class Ctx>
OA_HD void lds_fft(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int rowStride,
                   const Stages& st, const cx<T>* tw, int logTw) {
    cx<T> v[EPT];
    int logNs = 0;
    for (int i = 0; i < st.n; ++i) {
        const int R = st.radix[i];
        switch (R) {
            case 16: stage_a<T, 16, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs, tw, logTw); break;
            case 8: stage_a<T, 8, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs, tw, logTw); break;
            case 4: stage_a<T, 4, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs, tw, logTw); break;
            default: stage_a<T, 2, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs, tw, logTw); break;
        }
        ctx.sync();
        switch (R) {
            case 16: stage_b<T, 16, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs); break;
            case 8: stage_b<T, 8, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs); break;
            case 4: stage_b<T, 4, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs); break;
            default: stage_b<T, 2, ROWMAJOR>(s, v, tid, NT, logL, logC, rowStride, logNs); break;
        }
        ctx.sync();
        logNs += ilog2(R);
    }
}

// ===========================================================================
// Row pass: contiguous sequences.  One workgroup transforms C rows.
// ===========================================================================
template <typename T>
struct RowArgs {
    const void* in;
    void* out;
    long in_pitch, out_pitch;  // in COMPLEX elements (a real row of 2L reals has pitch L')
    int logL;                  // complex transform length L (= N/2 for the real modes)
    int logC;                  // rows per workgroup
    int NT;                    // threads per workgroup = L*C/EPT
    int rowStride;             // LDS complex elements per row
    Stages st;
    const cx<T>* tw;           // master table W_M^k, k < M, M = 2^logTw >= 2L (real modes) or L
    int logTw;
    T scale;
    int mode;
};

template <typename T, class Ctx>
OA_HD void row_fft_body(Ctx& ctx, const RowArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    const int logL = a.logL, L = 1 << logL, C = 1 << a.logC, RS = a.rowStride;
    const long r0 = (long)ctx.bid_x() * C;
    const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in);
    cx<T>* out = reinterpret_cast<cx<T>*>(a.out);
    const bool inv = (a.mode == ROW_C2C_I);

    // ---- global -> LDS
    for (int i = tid; i < (C << logL); i += NT) {
        const int c = i >> logL, n = i & (L - 1);
        cx<T> x = in[(r0 + c) * a.in_pitch + n];
        if (inv) x = swp(x);
        s[lds_addr<true>(n, c, 0, RS)] = x;
    }
    if (a.mode == ROW_C2R && tid < C)
        s[lds_addr<true>(L, tid, 0, RS)] = in[(r0 + tid) * a.in_pitch + L];
    ctx.sync();

    if (a.mode == ROW_C2R) {
        // Z'[k] = (X[k]+conj X[L-k]) + i W_N^{-k} (X[k]-conj X[L-k]), stored swapped
        const int sh = a.logTw - (logL + 1);
        for (int i = tid; i < (C << (logL - 1)); i += NT) {
            const int c = i >> (logL - 1), k = i & ((L >> 1) - 1);
            for (int rep = 0; rep < 2; ++rep) {
                const int kk = rep ? (L >> 1) : k;
                if (rep && k != 0) break;
                const cx<T> A = s[lds_addr<true>(kk, c, 0, RS)];
                const cx<T> B = s[lds_addr<true>(L - kk, c, 0, RS)];
                const cx<T> w = a.tw[kk << sh];  // W_N^k
                const cx<T> d1 = A - conj(B), d2 = B - conj(A);
                const cx<T> z1 = (A + conj(B)) + mul_pi(conj(w) * d1);
                const cx<T> z2 = (B + conj(A)) - mul_pi(w * d2);
                s[lds_addr<true>(kk, c, 0, RS)] = swp(z1);
                if (kk != 0 && 2 * kk != L) s[lds_addr<true>(L - kk, c, 0, RS)] = swp(z2);
            }
        }
        ctx.sync();
    }

    lds_fft<T, true>(ctx, s, tid, NT, logL, a.logC, RS, a.st, a.tw, a.logTw);

    if (a.mode == ROW_R2C) {
        // X[k] = E + W_N^k O ; X[L-k] = conj(E - W_N^k O)
        const int sh = a.logTw - (logL + 1);
        for (int i = tid; i < (C << (logL - 1)); i += NT) {
            const int c = i >> (logL - 1), k = i & ((L >> 1) - 1);
            for (int rep = 0; rep < 2; ++rep) {
                const int kk = rep ? (L >> 1) : k;
                if (rep && k != 0) break;
                const cx<T> Zk = s[lds_addr<true>(kk, c, 0, RS)];
                const cx<T> Zm = s[lds_addr<true>((L - kk) & (L - 1), c, 0, RS)];
                const cx<T> E = (Zk + conj(Zm)) * (T)0.5;
                const cx<T> O = mul_mi(Zk - conj(Zm)) * (T)0.5;
                const cx<T> wO = a.tw[kk << sh] * O;
                s[lds_addr<true>(kk, c, 0, RS)] = E + wO;
                s[lds_addr<true>(L - kk, c, 0, RS)] = conj(E - wO);
            }
        }
        ctx.sync();
    }

    // ---- LDS -> global
    const bool swap_out = (a.mode == ROW_C2C_I || a.mode == ROW_C2R);
    for (int i = tid; i < (C << logL); i += NT) {
        const int c = i >> logL, n = i & (L - 1);
        cx<T> x = s[lds_addr<true>(n, c, 0, RS)];
        if (swap_out) x = swp(x);
        out[(r0 + c) * a.out_pitch + n] = x * a.scale;
    }
    if (a.mode == ROW_R2C && tid < C)
        out[(r0 + tid) * a.out_pitch + L] = s[lds_addr<true>(L, tid, 0, RS)] * a.scale;
}

// ===========================================================================
// Column pass: [L][C] tiles of strided rows.  grid = (column tiles, groups).
//   input  row of point n in group g : g*in_gs  + n*in_ns
//   output row of bin   k in group g : g*out_gs + k*out_ks
// ===========================================================================
template <typename T>
struct ColArgs {
    const cx<T>* in;
    cx<T>* out;
    long in_pitch, out_pitch;  // complex elements
    int width;                 // valid columns
    int logL, logC, NT;
    Stages st;
    const cx<T>* tw;           // master table of length 2^logTw (= Ny)
    int logTw;
    long in_gs, in_ns, out_gs, out_ks;
    int twiddle;               // multiply bin k of group g by W_{Ny}^{g k}
    int inverse;
    T scale;
};

template <typename T, class Ctx>
OA_HD void col_fft_body(Ctx& ctx, const ColArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    const int logL = a.logL, logC = a.logC, C = 1 << logC;
    const int c0 = ctx.bid_x() << logC;
    const long g = ctx.bid_y();
    const int tot = 1 << (logL + logC);

    for (int i = tid; i < tot; i += NT) {
        const int c = i & (C - 1), n = i >> logC;
        cx<T> x = mk<T>((T)0, (T)0);
        if (c0 + c < a.width) x = a.in[(g * a.in_gs + n * a.in_ns) * a.in_pitch + c0 + c];
        if (a.inverse) x = swp(x);
        s[i] = x;
    }
    ctx.sync();

    lds_fft<T, false>(ctx, s, tid, NT, logL, logC, 0, a.st, a.tw, a.logTw);

    for (int i = tid; i < tot; i += NT) {
        const int c = i & (C - 1), k = i >> logC;
        if (c0 + c >= a.width) continue;
        cx<T> x = s[i];
        if (a.twiddle) x = x * a.tw[(int)(g * k)];
        if (a.inverse) x = swp(x);
        a.out[(g * a.out_gs + k * a.out_ks) * a.out_pitch + c0 + c] = x * a.scale;
    }
}

}  // namespace oa
