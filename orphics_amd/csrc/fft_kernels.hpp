// LDS-staged Stockham FFT passes for gfx950 (wave64), written against an
// abstract execution context so the identical bodies run (a) as HIP kernels
// and (b) under the CPU thread emulator in tests/emul (index-math validation
// without a GPU).
//
// Design (DESIGN.md section "K1"):
//  * every thread owns EPT=16 complex points per stage: one radix-16 butterfly
//    or 16/R radix-R butterflies, register resident, fully unrolled;
//  * the FIRST stage gathers its butterfly inputs straight from global memory
//    (per t the lanes read consecutive addresses -> coalesced) and the LAST stage
//    scatters straight to global memory: an L = R1*R2 transform costs ONE LDS
//    round trip and ONE barrier; load/store are functors so filters, twiddles,
//    scales and layout changes fuse into the pass;
//  * twiddles: 4 table loads (w^1, w^2, w^4, w^8; exact to 0.5 ulp) + <= 3
//    complex products each, instead of 15 scattered gathers;
//  * row passes keep whole rows in LDS, padded 1 element per 16 so the
//    stride-R first-stage writes are bank-conflict free;
//  * column passes work on [L points][C=32 columns] tiles so every global
//    access is a >=256-byte contiguous segment; a length-Ny column transform is
//    split four-step style Ny = N1*N2 into two such passes (pass 1 applies the
//    inter-pass twiddle and writes transposed blocks, pass 2 is in place);
//  * inverse transforms use IDFT(x) = swap(DFT(swap(x))) so one forward
//    butterfly/twiddle set serves both directions;
//  * real transforms use the packed N/2-point trick with an in-LDS
//    (un)tangle step, so a real row costs half the LDS and flops.
#pragma once
#include <type_traits>
#include "cx.hpp"

namespace oa {

constexpr int EPT = 16;        // complex points per thread per stage
#ifndef OA_COL_LOGC
#define OA_COL_LOGC 5
#endif
constexpr int COL_LOGC = OA_COL_LOGC;  // log2 columns per column tile (compile-time: index math folds to masks/shifts)

// ROW_WIN: half-complex rows -> C2R -> x real-space window -> R2C -> half-complex rows in ONE pass: the real rows exist in LDS only
// (oa_mc_run_windowed: every simulated map is multiplied by the apodisation taper before its transform, maps.py:1350-1361)
enum RowMode { ROW_R2C = 0, ROW_C2R = 1, ROW_C2C_F = 2, ROW_C2C_I = 3, ROW_WIN = 4 };

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
            if (4 * k == R) {   // twiddle -i: one swizzled add each, no product
                v[k] = add_mi(e[k], o[k]);
                v[k + R / 2] = add_pi(e[k], o[k]);
            } else {
                const cx<T> t = (k == 0) ? o[k] : o[k] * w16<T>(k * (16 / R));
                v[k] = e[k] + t;
                v[k + R / 2] = e[k] - t;
            }
        }
    }
};
template <typename T>
struct Dft<T, 1> {
    static OA_HD void run(cx<T>*) {}
};

// multiply by i^q (q uniform across the workgroup)
template <typename T> OA_HD cx<T> rot_i(cx<T> x, int q) {
    switch (q & 3) {
        case 0: return x;
        case 1: return mk<T>(-x.y, x.x);
        case 2: return mk<T>(-x.x, -x.y);
        default: return mk<T>(x.y, -x.x);
    }
}

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

// Stage twiddles come from a tiny two-level table in LDS (no global gathers on the critical path):
//   tab[0 .. 2^h)            = W_L^i          (low digits)
//   tab[2^h .. 2^h + L/2^h)  = W_L^(q 2^h)    (high digits),   h = (logL+1)/2
//   W_L^e = tab[2^h + (e >> h)] * tab[e & (2^h - 1)]           (1 product, ~1 ulp)
template <typename T>
OA_HD cx<T> tw_lds(const cx<T>* tab, int h, int e) {
    return tab[(1 << h) + (e >> h)] * tab[e & ((1 << h) - 1)];
}
OA_HD int tw_lds_h(int logL) { return (logL + 1) >> 1; }
OA_HD int tw_lds_size(int logL) { return (1 << tw_lds_h(logL)) + (1 << (logL - tw_lds_h(logL))); }

template <typename T, class Ctx>
OA_HD void tw_lds_fill(Ctx& ctx, cx<T>* tab, const cx<T>* gtw, int logG, int logL, int NT) {
    const int h = tw_lds_h(logL), nlo = 1 << h, nhi = 1 << (logL - h), sh = logG - logL;
    for (int i = ctx.tid(); i < nlo + nhi; i += NT) tab[i] = gtw[(i < nlo ? i : ((i - nlo) << h)) << sh];
}

// multiply v[1..R-1] by w^t, w = W_L^(k << sh); base powers w^1,w^2,w^4,w^8 from the LDS table, rest by products
template <typename T, int R>
OA_HD void apply_twiddles(cx<T>* v, const cx<T>* tab, int k, int sh, int h) {
    const cx<T> w1 = tw_lds(tab, h, k << sh);
    v[1] = v[1] * w1;
    if (R >= 4) {
        const cx<T> w2 = tw_lds(tab, h, (2 * k) << sh);
        const cx<T> w3 = w1 * w2;
        v[2] = v[2] * w2;
        v[3] = v[3] * w3;
        if (R >= 8) {
            const cx<T> w4 = tw_lds(tab, h, (4 * k) << sh);
            v[4] = v[4] * w4;
            v[5] = v[5] * (w4 * w1);
            v[6] = v[6] * (w4 * w2);
            const cx<T> w7 = w4 * w3;
            v[7] = v[7] * w7;
            if (R >= 16) {
                const cx<T> w8 = tw_lds(tab, h, (8 * k) << sh);
                v[8] = v[8] * w8;
                v[9] = v[9] * (w8 * w1);
                v[10] = v[10] * (w8 * w2);
                v[11] = v[11] * (w8 * w3);
                const cx<T> w12 = w8 * w4;
                v[12] = v[12] * w12;
                v[13] = v[13] * (w12 * w1);
                v[14] = v[14] * (w12 * w2);
                v[15] = v[15] * (w8 * w7);
            }
        }
    }
}

struct NoLoad {
    static constexpr bool reads_lds = false;
    template <typename T> OA_HD cx<T> get(int, int) const { return cx<T>{}; }
};
struct NoStore {
    template <typename T> OA_HD void put(int, int, cx<T>) const {}
};

// One Stockham stage of radix R for this thread's EPT/R butterflies, in two halves:
//   stage_in : gather inputs (global functor or LDS), twiddle, in-register DFT-R
//   stage_out: scatter to the autosort position (global functor or LDS)
template <typename T, int R, bool ROWMAJOR, bool SRC_G, class Ld>
OA_HD void stage_in(const cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int rowStride, int logNs,
                    const cx<T>* tw, int logTw, const Ld& ld) {
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
        for (int t = 0; t < R; ++t) {
            const int n = j + (t << logLR);
            if (SRC_G) v[u * R + t] = ld.template get<T>(n, c);
            else v[u * R + t] = s[lds_addr<ROWMAJOR>(n, c, logC, rowStride)];
        }
        if (logNs > 0) apply_twiddles<T, R>(v + u * R, tw, j & ((1 << logNs) - 1), logL - logNs - LR, tw_lds_h(logL));
        Dft<T, R>::run(v + u * R);
    }
}

template <typename T, int R, bool ROWMAJOR, bool DST_G, class St>
OA_HD void stage_out(cx<T>* s, const cx<T>* v, int tid, int NT, int logL, int logC, int rowStride, int logNs,
                     const St& st) {
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
        for (int t = 0; t < R; ++t) {
            const int n = base + (t << logNs);
            if (DST_G) st.template put<T>(n, c, v[u * R + t]);
            else s[lds_addr<ROWMAJOR>(n, c, logC, rowStride)] = v[u * R + t];
        }
    }
}

template <typename T, int R, bool ROWMAJOR, bool SRC_G, bool DST_G, class Ctx, class Ld, class St>
OA_HD void stage(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int rowStride, int logNs,
                 const cx<T>* tw, int logTw, const Ld& ld, const St& st) {
    cx<T> v[EPT];
    stage_in<T, R, ROWMAJOR, SRC_G>(s, v, tid, NT, logL, logC, rowStride, logNs, tw, logTw, ld);
    if ((!SRC_G || Ld::reads_lds) && !DST_G) ctx.sync();  // in-place: every read of this stage precedes any write
    stage_out<T, R, ROWMAJOR, DST_G>(s, v, tid, NT, logL, logC, rowStride, logNs, st);
}

// compile-time radix sequence (unused slots = 1)
template <int A, int B = 1, int C = 1, int D = 1>
struct Seq {
    static constexpr int r0 = A, r1 = B, r2 = C, r3 = D;
    static constexpr int n = (A > 1) + (B > 1) + (C > 1) + (D > 1);
    static constexpr int get(int i) { return i == 0 ? A : (i == 1 ? B : (i == 2 ? C : D)); }
    // radix of stage i of the REVERSED sequence and the log2 of the product of the stages before it
    static constexpr int rget(int i) { return get(n - 1 - i); }
};

constexpr int clog2(int r) { return r >= 16 ? 4 : (r >= 8 ? 3 : (r >= 4 ? 2 : (r >= 2 ? 1 : 0))); }
template <class SEQ> constexpr int fwd_logns(int i) { int a = 0; for (int m = 0; m < i; ++m) a += clog2(SEQ::get(m)); return a; }
template <class SEQ> constexpr int rev_logns(int i) { int a = 0; for (int m = 0; m < i; ++m) a += clog2(SEQ::rget(m)); return a; }

template <int R> struct Log2x { static constexpr int v = Log2c<R>::v; };
template <> struct Log2x<1> { static constexpr int v = 0; };
template <class SEQ> constexpr int seq_total_log() { return clog2(SEQ::r0) + clog2(SEQ::r1) + clog2(SEQ::r2) + clog2(SEQ::r3); }

// Forward DFT of the C sequences of this workgroup, radix sequence SEQ.
//   SRC_G: the first stage reads ld.get(n,c); otherwise data is in LDS (caller synced).
//   DST_G: the last stage writes st.put(k,c,v); otherwise results end in LDS (synced on exit).
template <typename T, bool ROWMAJOR, bool SRC_G, bool DST_G, class SEQ, class Ctx, class Ld, class St>
OA_HD void fft_pipeline(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int rowStride, const cx<T>* tw,
                        int logTw, const Ld& ld, const St& st) {
    constexpr int n = SEQ::n;
    constexpr int l0 = Log2x<SEQ::r0>::v, l1 = l0 + Log2x<SEQ::r1>::v, l2 = l1 + Log2x<SEQ::r2>::v;
    if constexpr (n == 1) {
        stage<T, SEQ::r0, ROWMAJOR, SRC_G, DST_G>(ctx, s, tid, NT, logL, logC, rowStride, 0, tw, logTw, ld, st);
        if (!DST_G) ctx.sync();
    } else {
        stage<T, SEQ::r0, ROWMAJOR, SRC_G, false>(ctx, s, tid, NT, logL, logC, rowStride, 0, tw, logTw, ld, NoStore{});
        ctx.sync();
        if constexpr (n == 2) {
            stage<T, SEQ::r1, ROWMAJOR, false, DST_G>(ctx, s, tid, NT, logL, logC, rowStride, l0, tw, logTw, NoLoad{}, st);
        } else {
            stage<T, SEQ::r1, ROWMAJOR, false, false>(ctx, s, tid, NT, logL, logC, rowStride, l0, tw, logTw, NoLoad{}, NoStore{});
            ctx.sync();
            if constexpr (n == 3) {
                stage<T, SEQ::r2, ROWMAJOR, false, DST_G>(ctx, s, tid, NT, logL, logC, rowStride, l1, tw, logTw, NoLoad{}, st);
            } else {
                stage<T, SEQ::r2, ROWMAJOR, false, false>(ctx, s, tid, NT, logL, logC, rowStride, l1, tw, logTw, NoLoad{}, NoStore{});
                ctx.sync();
                stage<T, SEQ::r3, ROWMAJOR, false, DST_G>(ctx, s, tid, NT, logL, logC, rowStride, l2, tw, logTw, NoLoad{}, st);
            }
        }
        if (!DST_G) ctx.sync();
    }
}

// Stages 1 .. n-1 of the forward sequence, LDS to LDS: the first stage's outputs are in LDS and the workgroup has synced
// (callers that run the first stage themselves, e.g. on prefetched registers); results end in LDS, synced on exit.
template <typename T, bool ROWMAJOR, class SEQ, class Ctx>
OA_HD void fft_pipeline_rest(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int rowStride, const cx<T>* tw, int logTw) {
    constexpr int n = SEQ::n;
    constexpr int l0 = Log2x<SEQ::r0>::v, l1 = l0 + Log2x<SEQ::r1>::v, l2 = l1 + Log2x<SEQ::r2>::v;
    if constexpr (n >= 2) {
        stage<T, SEQ::r1, ROWMAJOR, false, false>(ctx, s, tid, NT, logL, logC, rowStride, l0, tw, logTw, NoLoad{}, NoStore{});
        ctx.sync();
    }
    if constexpr (n >= 3) {
        stage<T, SEQ::r2, ROWMAJOR, false, false>(ctx, s, tid, NT, logL, logC, rowStride, l1, tw, logTw, NoLoad{}, NoStore{});
        ctx.sync();
    }
    if constexpr (n >= 4) {
        stage<T, SEQ::r3, ROWMAJOR, false, false>(ctx, s, tid, NT, logL, logC, rowStride, l2, tw, logTw, NoLoad{}, NoStore{});
        ctx.sync();
    }
}

// host-side dispatch: logL -> radix sequence (greedy 16s, remainder last)
template <class F>
inline bool dispatch_seq(int logL, F&& f) {
    switch (logL) {
        case 1: f(Seq<2>{}); return true;
        case 2: f(Seq<4>{}); return true;
        case 3: f(Seq<8>{}); return true;
        case 4: f(Seq<16>{}); return true;
        case 5: f(Seq<16, 2>{}); return true;
        case 6: f(Seq<16, 4>{}); return true;
        case 7: f(Seq<16, 8>{}); return true;
        case 8: f(Seq<16, 16>{}); return true;
        case 9: f(Seq<16, 16, 2>{}); return true;
        case 10: f(Seq<16, 16, 4>{}); return true;
        case 11: f(Seq<16, 16, 8>{}); return true;
        case 12: f(Seq<16, 16, 16>{}); return true;
        case 13: f(Seq<16, 16, 16, 2>{}); return true;
        case 14: f(Seq<16, 16, 16, 4>{}); return true;
        default: return false;
    }
}

// the same table at compile time
template <int LOGL> struct SeqOf;
template <> struct SeqOf<4> { using type = Seq<16>; };
template <> struct SeqOf<5> { using type = Seq<16, 2>; };
template <> struct SeqOf<6> { using type = Seq<16, 4>; };
template <> struct SeqOf<7> { using type = Seq<16, 8>; };
template <> struct SeqOf<8> { using type = Seq<16, 16>; };
template <> struct SeqOf<9> { using type = Seq<16, 16, 2>; };
template <> struct SeqOf<10> { using type = Seq<16, 16, 4>; };
template <> struct SeqOf<11> { using type = Seq<16, 16, 8>; };

// radix sequences of the fused row stage: as dispatch_seq, except that 4-stage lengths lead with the short radix
// (3-stage lengths with a short radix -- 1024..4096-point rows -- were measured faster in the greedy order: their
// active-column variants spill)
template <class F>
inline bool dispatch_seq_qe(int logL, F&& f) {
    switch (logL) {
        case 13: f(Seq<2, 16, 16, 16>{}); return true;
        case 14: f(Seq<4, 16, 16, 16>{}); return true;
        default: return dispatch_seq(logL, f);
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
    const cx<T>* tw;           // master table W_M^k, k < M, M = 2^logTw >= 2L (real modes) or L
    int logTw;
    T scale;
    int mode;
    int wcols;                 // R2C: columns produced; C2R: columns read (the rest are zero).  >= L+1: all
    const void* mul;           // C2R only, may be NULL: real plane (layout and pitch of `out`) multiplied into the result at the
                               // store -- a real-space window applied without another pass over the map (oa_mc_run_windowed)
    // R-SPLIT R2C (row_r2c_rsplit_body): the first radix-R butterfly of the COLUMN transform rides on the row pass.  The
    // workgroup transforms the R rows g + my n (n < R = 2^lr) of its group g one after the other and keeps, per kept
    // column, Y[k1][g] = W_ny^(g k1) sum_n X_n W_R^(n k1) in registers: output plane k1 (kplane elements apart) row g.
    // The column transform is then X[k1 + R k2] = DFT_my over g of Y[k1][g] -- one single-pass kernel (col_fband_body).
    int lr, my;
    long kplane;
    const cx<T>* twy;          // W_ny^k, ny = my << lr
    // SEVERAL PLANES PER LAUNCH (row_fft_body, grid y = plane): plane z reads z * in_zoff and writes z * out_zoff complex elements
    // behind the first (oa_lens_maps: the C2R of every derivative field of every map in one launch).  nz = 0: one plane.
    int nz;
    long in_zoff, out_zoff;
    // X-DERIVATIVE C2R (oa_lens_maps): dlx != nullptr -> plane z of the launch is the C2R of (i lx)^(dpow0 + z) x the SAME input
    // (a column-transformed field that already carries (i ly)^dcol_b) and goes to output plane idx(a, b) = n (n + 1) / 2 - 1 + b,
    // n = a + b (the order lens_taylor_kernel reads), out_zoff elements per plane
    const void* dlx;
    int dpow0, dcol_b;
};

template <typename T, bool SWAP>
struct RowLoad {
    static constexpr bool reads_lds = false;
    const cx<T>* in;   // pre-offset to the first row of this workgroup
    unsigned pitch;
    template <typename U> OA_HD cx<U> get(int n, int c) const {
        const cx<U> x = in[(unsigned)c * pitch + (unsigned)n];
        return SWAP ? swp(x) : x;
    }
};
// R2C input rows: read exactly once per transform -> non-temporal (does not displace the small planes in L2 / MALL)
template <typename T>
struct RowLoadOnce {
    static constexpr bool reads_lds = false;
    const cx<T>* in;
    unsigned pitch;
    template <typename U> OA_HD cx<U> get(int n, int c) const {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OA_W64_PLAIN_LOADS)
        typedef U v2 __attribute__((ext_vector_type(2)));
        const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2*>(in + (unsigned)c * pitch + (unsigned)n));
        return mk<U>(v.x, v.y);
#else
        return in[(unsigned)c * pitch + (unsigned)n];
#endif
    }
};
template <typename T, bool SWAP>
struct RowStore {
    cx<T>* out;        // pre-offset to the first row of this workgroup
    unsigned pitch;
    T scale;
    const cx<T>* mul = nullptr;   // optional: per-element real factors (two packed reals per complex slot), same offsets as out
    template <typename U> OA_HD void put(int n, int c, cx<U> v) const {
        if (SWAP) v = swp(v);
        cx<U> r = v * scale;
        if (mul) { const cx<U> m = mul[(unsigned)c * pitch + (unsigned)n]; r = mk<U>(r.x * m.x, r.y * m.y); }
        out[(unsigned)c * pitch + (unsigned)n] = r;
    }
};

// C2R prologue: half-complex rows (global) -> packed Z'[k] = (X[k]+conj X[L-k]) + i W_N^{-k} (X[k]-conj X[L-k]),
// stored SWAPPED in LDS (the inverse runs as a forward transform of the swapped data).  Caller syncs.
// GUARD: columns >= win are zero and never read (active-column mode); the unguarded body keeps every load
// unconditional so the compiler batches them (dense mode is HBM-latency bound).
// dlx != nullptr: the x-derivative (i lx)^apow is applied to the spectrum at the load (dlx = the derivative axis lx per column, Nyquist
// entry zero; apow uniform) -- the Taylor lensing op takes every x-derivative of a column-transformed field in the row pass
// The global reads of U = 4 (k, L - k) pairs -- spectrum, twiddle, derivative axis -- are issued together before any of them is used
// (clamped indices, values selected afterwards): the loop used to wait for the two or three dependent reads of ONE pair per trip, at
// two waves per SIMD (L1w1 L4w9 L1w2 ... in the ISA; 57 us per 4096^2 float64 plane for 268 MB).
template <typename T, bool GUARD, class Ctx>
OA_HD void c2r_prologue_impl(Ctx& ctx, cx<T>* s, const cx<T>* in, long pitch, long r0, int logL, int logC, int NT, int RS,
                             const cx<T>* tw, int logTw, int win, const T* dlx = nullptr, int apow = 0) {
    const int tid = ctx.tid(), L = 1 << logL, C = 1 << logC;
    const int sh = logTw - (logL + 1);
    const int total = C << (logL - 1);
    constexpr int U = 4;
    // one (A, B) pair -> Z'[kk], Z'[L - kk] in LDS
    auto emit = [&](int kk, int c, cx<T> A, cx<T> B, cx<T> w, T la, T lb) {
        if (GUARD && kk >= win && L - kk >= win) {      // both partners beyond the band: Z' = 0, no arithmetic
            const cx<T> z = mk<T>((T)0, (T)0);
            s[lds_addr<true>(kk, c, 0, RS)] = z;
            if (kk != 0 && 2 * kk != L) s[lds_addr<true>(L - kk, c, 0, RS)] = z;
            return;
        }
        // the columns kx = 0 and kx = nx/2 of a REAL field's transform are real once the column transform has been inverted;
        // whatever imaginary part arrives here comes from a non-Hermitian input column (e.g. the Nyquist column of Q, U =
        // R^-1 (E, B): the rotation's sine is odd there) and is DROPPED -- the reference's `ifft(...).real` (maps.py:1585)
        // symmetrises each column on its own; packed into Z'[0] it would leak from the Nyquist column into kx = 0
        if (dlx) {
            T fa = (T)1, fb = (T)1;
            for (int i = 0; i < apow; ++i) { fa = fa * la; fb = fb * lb; }
            A = rot_i(A, apow) * fa;
            B = rot_i(B, apow) * fb;
        }
        if (kk == 0) { A.y = (T)0; B.y = (T)0; }
        const cx<T> d1 = A - conj(B), d2 = B - conj(A);
        const cx<T> z1 = (A + conj(B)) + mul_pi(conj(w) * d1);
        const cx<T> z2 = (B + conj(A)) - mul_pi(w * d2);
        s[lds_addr<true>(kk, c, 0, RS)] = swp(z1);
        if (kk != 0 && 2 * kk != L) s[lds_addr<true>(L - kk, c, 0, RS)] = swp(z2);
    };
    for (int i0 = tid; i0 < total; i0 += U * NT) {
        cx<T> A[U], B[U], W[U];
        T la[U], lb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT, ii = i < total ? i : tid;
            const int c = ii >> (logL - 1), k = ii & ((L >> 1) - 1);
            const cx<T>* row = in + r0 * pitch + (unsigned)c * (unsigned)pitch;
            if (GUARD) {                                  // reads beyond the band are replaced by zero (clamped address, select)
                const cx<T> va = row[k < win ? k : 0], vb = row[L - k < win ? L - k : 0];
                A[u] = k < win ? va : mk<T>((T)0, (T)0);
                B[u] = L - k < win ? vb : mk<T>((T)0, (T)0);
            } else {
                A[u] = row[k];
                B[u] = row[L - k];
            }
            W[u] = tw[k << sh];  // W_N^k
            la[u] = dlx ? dlx[k] : (T)0;
            lb[u] = dlx ? dlx[L - k] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT;
            if (i < total) emit(i & ((L >> 1) - 1), i >> (logL - 1), A[u], B[u], W[u], la[u], lb[u]);
        }
    }
    // the middle element kk = L/2 of every row (its own partner)
    for (int c = tid; c < C; c += NT) {
        const int kk = L >> 1;
        const cx<T>* row = in + r0 * pitch + (unsigned)c * (unsigned)pitch;
        cx<T> A = mk<T>((T)0, (T)0);
        if (!GUARD || kk < win) A = row[kk];
        emit(kk, c, A, A, tw[kk << sh], dlx ? dlx[kk] : (T)0, dlx ? dlx[kk] : (T)0);
    }
}
template <typename T, class Ctx>
OA_HD void c2r_prologue(Ctx& ctx, cx<T>* s, const cx<T>* in, long pitch, long r0, int logL, int logC, int NT, int RS,
                        const cx<T>* tw, int logTw, int win = 0x7fffffff, const T* dlx = nullptr, int apow = 0) {
    if (win > (1 << logL)) c2r_prologue_impl<T, false>(ctx, s, in, pitch, r0, logL, logC, NT, RS, tw, logTw, win, dlx, apow);
    else c2r_prologue_impl<T, true>(ctx, s, in, pitch, r0, logL, logC, NT, RS, tw, logTw, win, dlx, apow);
}

// R2C epilogue: packed transform Z in LDS -> X[k] = E + W_N^k O, X[L-k] = conj(E - W_N^k O), straight to global.
// GUARD: only columns < wout are produced (the caller's consumers never look at the others).
template <typename T, bool GUARD, class Ctx>
OA_HD void r2c_epilogue_impl(Ctx& ctx, const cx<T>* s, cx<T>* out, long pitch, long r0, int logL, int logC, int NT, int RS,
                             const cx<T>* tw, int logTw, T scale, bool accumulate, int wout) {
    const int tid = ctx.tid(), L = 1 << logL;
    const int sh = logTw - (logL + 1);
    // L/16 threads per row (NT = L C / 16): this thread's row c and first column k0 are fixed and it owns the 8
    // column pairs (k, L - k), k = k0 + m L/16 -- a compile-time trip count once logL is, 32-bit offsets from two
    // per-thread bases, so the loop body is loads + butterfly + stores
    const int tpr = (NT >> logC) > 0 ? (NT >> logC) : 1;
    const int c = tid / tpr, k0 = tid - c * tpr;
    cx<T>* row = out + (r0 + c) * pitch;
    const cx<T>* sr = s + c * RS;
    auto pair = [&](int kk) {
        const bool w1 = !GUARD || kk < wout, w2 = !GUARD || (L - kk) < wout;
        if (GUARD && !w1 && !w2) return;
        const int km = (L - kk) & (L - 1);
        const cx<T> Zk = sr[kk + (kk >> 4)];
        const cx<T> Zm = sr[km + (km >> 4)];
        const cx<T> E = (Zk + conj(Zm)) * (T)0.5;
        const cx<T> O = mul_mi(Zk - conj(Zm)) * (T)0.5;
        const cx<T> wO = tw[kk << sh] * O;
        cx<T> o1 = (E + wO) * scale, o2 = conj(E - wO) * scale;
        if (accumulate) {
            if (w1) o1 = o1 + row[kk];
            if (2 * kk != L) { if (w2) o2 = o2 + row[L - kk]; } else o2 = o1;
        }
        if (w1) row[kk] = o1;
        if (w2) row[L - kk] = o2;
    };
    if (c < (1 << logC)) {
        for (int kk = k0; kk < (L >> 1); kk += tpr) pair(kk);
        if (k0 == 0) pair(L >> 1);
    }
}
template <typename T, class Ctx>
OA_HD void r2c_epilogue(Ctx& ctx, const cx<T>* s, cx<T>* out, long pitch, long r0, int logL, int logC, int NT, int RS,
                        const cx<T>* tw, int logTw, T scale, bool accumulate = false, int wout = 0x7fffffff) {
    if (wout > (1 << logL)) r2c_epilogue_impl<T, false>(ctx, s, out, pitch, r0, logL, logC, NT, RS, tw, logTw, scale, accumulate, wout);
    else r2c_epilogue_impl<T, true>(ctx, s, out, pitch, r0, logL, logC, NT, RS, tw, logTw, scale, accumulate, wout);
}

template <typename T, int MODE, class SEQ, class Ctx>
OA_HD void row_fft_body(Ctx& ctx, const RowArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    constexpr int logL = seq_total_log<SEQ>();   // compile-time: LDS offsets of the 16 taps become immediates
    const int C = 1 << a.logC, RS = a.rowStride;
    const long r0 = (long)ctx.bid_x() * C;
    const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + (a.nz ? (long)ctx.bid_y() * a.in_zoff : 0L);
    cx<T>* out = reinterpret_cast<cx<T>*>(a.out) + (a.nz ? (long)ctx.bid_y() * a.out_zoff : 0L);
    int apow = 0;
    if (a.dlx) {
        apow = a.dpow0 + (a.nz ? ctx.bid_y() : 0);
        const int nn = apow + a.dcol_b;
        in = reinterpret_cast<const cx<T>*>(a.in);
        out = reinterpret_cast<cx<T>*>(a.out) + (long)(nn * (nn + 1) / 2 - 1 + a.dcol_b) * a.out_zoff;
    }
    cx<T>* twl = s + C * RS;                      // two-level stage-twiddle table (LDS)
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    ctx.sync();

    if constexpr (MODE == ROW_C2C_F) {
        fft_pipeline<T, true, true, true, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL,
                                               RowLoad<T, false>{in + r0 * a.in_pitch, (unsigned)a.in_pitch},
                                               RowStore<T, false>{out + r0 * a.out_pitch, (unsigned)a.out_pitch, a.scale});
    } else if constexpr (MODE == ROW_C2C_I) {
        fft_pipeline<T, true, true, true, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL,
                                               RowLoad<T, true>{in + r0 * a.in_pitch, (unsigned)a.in_pitch},
                                               RowStore<T, true>{out + r0 * a.out_pitch, (unsigned)a.out_pitch, a.scale});
    } else if constexpr (MODE == ROW_R2C) {
        fft_pipeline<T, true, true, false, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL,
                                                RowLoadOnce<T>{in + r0 * a.in_pitch, (unsigned)a.in_pitch}, NoStore{});
        r2c_epilogue<T>(ctx, s, out, a.out_pitch, r0, logL, a.logC, NT, RS, a.tw, a.logTw, a.scale, false, a.wcols);
    } else if constexpr (MODE == ROW_WIN) {
        // inverse packed transform with its result left in LDS (natural order, swapped: the inverse runs as a forward transform of
        // the swapped data), window multiply in place, forward packed transform from LDS, untangle of the kept columns.  Same
        // arithmetic as ROW_C2R (with `mul`) followed by ROW_R2C; the real rows never leave the CU.
        constexpr int L = 1 << logL;
        const cx<T>* win = reinterpret_cast<const cx<T>*>(a.mul) + r0 * (long)L;      // real plane: L packed pairs per row, no padding
        c2r_prologue<T>(ctx, s, in, a.in_pitch, r0, logL, a.logC, NT, RS, a.tw, a.logTw, 0x7fffffff);
        ctx.sync();
        fft_pipeline<T, true, false, false, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL, NoLoad{}, NoStore{});
        for (int i = tid; i < (C << logL); i += NT) {
            const int c = i >> logL, n = i & (L - 1);
            const cx<T> v = swp(s[lds_addr<true>(n, c, 0, RS)]) * a.scale;
            const cx<T> m = win[(unsigned)c * (unsigned)L + (unsigned)n];
            s[lds_addr<true>(n, c, 0, RS)] = mk<T>(v.x * m.x, v.y * m.y);
        }
        ctx.sync();
        fft_pipeline<T, true, false, false, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL, NoLoad{}, NoStore{});
        r2c_epilogue<T>(ctx, s, out, a.out_pitch, r0, logL, a.logC, NT, RS, a.tw, a.logTw, (T)1, false, a.wcols);
    } else {
        c2r_prologue<T>(ctx, s, in, a.in_pitch, r0, logL, a.logC, NT, RS, a.tw, a.logTw, a.wcols, reinterpret_cast<const T*>(a.dlx), apow);
        ctx.sync();
        fft_pipeline<T, true, false, true, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL, NoLoad{},
                                                RowStore<T, true>{out + r0 * a.out_pitch, (unsigned)a.out_pitch, a.scale,
                                                                  a.mul ? reinterpret_cast<const cx<T>*>(a.mul) + r0 * a.out_pitch : nullptr});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// R2C row pass with the first radix-R butterfly of the column transform on top (see RowArgs::lr).  Only kept columns
// k < wcols <= L / 2 are produced; a thread owns the columns k0 + i * (threads per row), i < RS_MAXS, of one row slot.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RS_MAXS = 2;     // (4 slots spill the f32 build at 4 waves/SIMD; 2 cover every band-limited geometry: wcols <= L / 8)

// first-stage taps of this thread (what stage_in<.., SRC_G = true> gathers), loads only
template <typename T, int R0, class Ld>
OA_HD void rsplit_first_taps(cx<T>* v, int tid, int NT, int logL, const Ld& ld) {
    constexpr int LR0 = Log2c<R0>::v, NB = EPT / R0;
    const int logLR = logL - LR0;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT, j = b & ((1 << logLR) - 1), c = b >> logLR;
#pragma unroll
        for (int t = 0; t < R0; ++t) v[u * R0 + t] = ld.template get<T>(j + (t << logLR), c);
    }
}

// PF (prefetch): the workgroup is persistent over its groups (bid, bid + grid, ...) and issues the global loads of the NEXT row
// right after the first butterfly stage of the current one has left its registers for LDS: they stay in flight across the
// remaining stages and the untangle (GpuCtx::sync does not drain vmcnt), so a row costs max(arithmetic, HBM) instead of their
// sum.  !PF: loads at the top of each row (the round-2 order; A/B: OA_RSPLIT_NOPF=1).
template <typename T, class SEQ, int LR, bool PF = true, class Ctx>
OA_HD void row_r2c_rsplit_body(Ctx& ctx, const RowArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    constexpr int logL = seq_total_log<SEQ>();
    constexpr int L = 1 << logL, R = 1 << LR, R0 = SEQ::r0;
    const int C = 1 << a.logC, RS = a.rowStride;
    const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in);
    cx<T>* out = reinterpret_cast<cx<T>*>(a.out);
    cx<T>* twl = s + C * RS;
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    ctx.sync();
    const int tpr = (NT >> a.logC) > 0 ? (NT >> a.logC) : 1;
    const int c = tid / tpr, k0 = tid - c * tpr;
    const int sh = a.logTw - (logL + 1);
    static_assert(R == 4, "the butterfly factors below are W_4^e = (-i)^e");
    const unsigned nym = ((unsigned)a.my << LR) - 1u;
    const int ngroups = a.my >> a.logC, gstep = ctx.grid_x();
    cx<T> v[EPT];
    auto taps = [&](long grp, int n) {
        rsplit_first_taps<T, R0>(v, tid, NT, logL, RowLoadOnce<T>{in + (grp * C + (long)n * a.my) * a.in_pitch, (unsigned)a.in_pitch});
    };
    // untangle factors W_2L^kk of this thread's columns: row-invariant, loaded once (inside the loop they would queue behind the
    // prefetch in vmcnt order and the untangle would wait for the whole next row)
    cx<T> twk[RS_MAXS];
#pragma unroll
    for (int i = 0; i < RS_MAXS; ++i) { const int kk = k0 + i * tpr; twk[i] = a.tw[(kk < a.wcols ? kk : 0) << sh]; }
    long grp = ctx.bid_x();
    if (PF && grp < ngroups) taps(grp, 0);
    for (; grp < ngroups; grp += gstep) {
        const long r0 = grp * C;                               // first row slot of this group
        cx<T> acc[R][RS_MAXS];
        // W_ny^g: issued before the next prefetch (a later load would have to wait for it); its square and cube are formed at the
        // flush (1 ulp; 12 fewer float64 registers held across the four rows than with three table entries)
        const cx<T> wy1 = a.twy[(unsigned)(r0 + (c < C ? c : 0)) & nym];
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1)
#pragma unroll
            for (int i = 0; i < RS_MAXS; ++i) acc[k1][i] = mk<T>((T)0, (T)0);
#pragma unroll 1
        for (int n = 0; n < R; ++n) {
            if (!PF) taps(grp, n);
            {   // first stage on the taps in v (no twiddles: logNs = 0), results to LDS
                constexpr int NB = EPT / R0;
#pragma unroll
                for (int u = 0; u < NB; ++u) Dft<T, R0>::run(v + u * R0);
                stage_out<T, R0, true, false>(s, v, tid, NT, logL, a.logC, RS, 0, NoStore{});
            }
            if (PF) {
                if (n + 1 < R) taps(grp, n + 1);
                else if (grp + gstep < ngroups) taps(grp + gstep, 0);
            }
            ctx.sync();
            fft_pipeline_rest<T, true, SEQ>(ctx, s, tid, NT, logL, a.logC, RS, twl, logL);
            if (c < C) {
                const cx<T>* sr = s + c * RS;
#pragma unroll
                for (int i = 0; i < RS_MAXS; ++i) {
                    const int kk = k0 + i * tpr;
                    if (kk < a.wcols) {
                        const int km = (L - kk) & (L - 1);
                        const cx<T> Zk = sr[kk + (kk >> 4)];
                        const cx<T> Zm = sr[km + (km >> 4)];
                        const cx<T> E = (Zk + conj(Zm)) * (T)0.5;
                        const cx<T> O = mul_mi(Zk - conj(Zm)) * (T)0.5;
                        const cx<T> X = (E + twk[i] * O) * a.scale;
#pragma unroll
                        for (int k1 = 0; k1 < R; ++k1) {
                            const int e = (n * k1) & 3;                                                  // W_4^(n k1) = (-i)^e, uniform: no table load
                            const cx<T> w = mk<T>((T)((e == 0) - (e == 2)), (T)((e == 3) - (e == 1)));
                            acc[k1][i] = acc[k1][i] + X * w;
                        }
                    }
                }
            }
            ctx.sync();                                        // these LDS reads precede the next row's first-stage writes
        }
        if (c < C) {
            const long g = r0 + c;
            const cx<T> wy2 = wy1 * wy1;
            const cx<T> wy[R] = {mk<T>((T)1, (T)0), wy1, wy2, wy2 * wy1};
#pragma unroll
            for (int k1 = 0; k1 < R; ++k1) {
                cx<T>* row = out + (long)k1 * a.kplane + g * a.out_pitch;
#pragma unroll
                for (int i = 0; i < RS_MAXS; ++i) {
                    const int kk = k0 + i * tpr;
                    if (kk < a.wcols) row[kk] = k1 ? acc[k1][i] * wy[k1] : acc[k1][i];
                }
            }
        }
    }
}

// ===========================================================================
// Fused QE row stage (TT and every other estimator term): for each row
//   h = C2R(H),  for leg in (Gx, Gy):  P_leg = R2C( C2R(leg) * h )
// 3 half-complex planes in, 2 out; the real-space planes never touch HBM.
// LDS: one padded work row set + one unpadded real row set (h).
// ===========================================================================
// per-map operands of a multi-map row-stage launch whose planes are not evenly spaced (oa_qe_mv: the k-th separable piece of
// every estimator in one launch; the leg planes of a piece are shared, arbitrary slots of the pool) -- a DEVICE array
template <typename T>
struct RowQeMap {
    const cx<T>* gx; const cx<T>* gy; const cx<T>* h;
    cx<T>* px; cx<T>* py;
    T scale;
};

template <typename T>
struct RowQeArgs {
    const cx<T>* gx; const cx<T>* gy; const cx<T>* h;
    cx<T>* px; cx<T>* py;
    long pitch;   // row pitch of gx, gy, h (complex elements)
    long opitch;  // row pitch of px, py
    int logL, logC, NT, rowStride;
    const cx<T>* tw;
    int logTw;
    T scale;      // product scale: (1/Npix)^2 for two normalised inverse transforms
    int accumulate;  // != 0: add the (scaled) result to the existing contents of px, py
    int win, wout;   // leg columns >= win are zero (not read); only product columns < wout are written
    // SEVERAL MAPS PER LAUNCH (pair row stage only): workgroups [m npairs, (m + 1) npairs) work on map m, whose gx / gy planes
    // sit m in_moff, whose h plane m h_moff and whose product planes m out_moff elements behind the first map's.
    // npairs = 0: one map.
    int npairs; long in_moff, out_moff, h_moff;
    const RowQeMap<T>* tab;   // != nullptr: map m takes its planes and its scale from tab[m] instead
    int lr, nrows;            // pair row stage: lr = 2 -> the leg planes are in the R-LAYOUT of col_fband_body (nrows = rows of the grid)
    // CHAINS (row_qe_pair_body<.., CHAIN = true>, oa_qe_mv): map m is an ESTIMATOR whose separable pieces are the table entries
    // tab[chain[2 m] .. chain[2 m] + chain[2 m + 1]): per piece three inverse transforms and the real-space product, summed
    // over the pieces in registers, then ONE forward pair per estimator (the forward transform is linear) into the px / py
    // of the chain's first entry -- 3 n + 2 transforms per row pair instead of 5 n, no read-modify-write of the product planes
    const int* chain;
    const cx<T>* rq8c;        // row_qe8_body: the per-thread constants of its grid (rq8_make_consts, fft_rowqe8.hpp)
};

// LDS -> LDS stage I of the reversed (inverse) / forward sequence
template <typename T, class SEQ, int I, bool REV, class Ctx>
OA_HD void lds_stage(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw, int logTw) {
    constexpr int R = REV ? SEQ::rget(I) : SEQ::get(I);
    constexpr int lns = REV ? rev_logns<SEQ>(I) : fwd_logns<SEQ>(I);
    stage<T, R, true, false, false>(ctx, s, tid, NT, logL, logC, RS, lns, tw, logTw, NoLoad{}, NoStore{});
    ctx.sync();
}

// inverse stages 0..n-2 of the reversed sequence (the last one is left to the caller)
template <typename T, class SEQ, class Ctx>
OA_HD void inverse_head(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw, int logTw) {
    if constexpr (SEQ::n >= 2) lds_stage<T, SEQ, 0, true>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
    if constexpr (SEQ::n >= 3) lds_stage<T, SEQ, 1, true>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
    if constexpr (SEQ::n >= 4) lds_stage<T, SEQ, 2, true>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
}
// forward stages 1..n-1 (stage 0 is done in registers by the caller)
template <typename T, class SEQ, class Ctx>
OA_HD void forward_tail(Ctx& ctx, cx<T>* s, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw, int logTw) {
    if constexpr (SEQ::n >= 2) lds_stage<T, SEQ, 1, false>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
    if constexpr (SEQ::n >= 3) lds_stage<T, SEQ, 2, false>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
    if constexpr (SEQ::n >= 4) lds_stage<T, SEQ, 3, false>(ctx, s, tid, NT, logL, logC, RS, tw, logTw);
}

// ---- active-column first stage of the inverse row transforms ------------------------------------------------
// With win active columns only the taps t < NZ and t >= R-NZ of the first (radix R, stride S = L/R) inverse stage
// can carry data (n = j + t*S < win, or its mirror partner L-n < win).  Those few operands are untangled straight
// from the half-complex rows,  Z'[n] = (X[n] + conj X[L-n]) + i conj(W_N^n) (X[n] - conj X[L-n]),  the others are
// literal (negative) zeros that the compiler folds out of the butterfly: no prologue pass through LDS, no LDS reads
// in this stage, one barrier less per transform.  NZ = 0 selects the general prologue path.
template <typename T>
OA_HD cx<T> c2r_tap(const cx<T>* row, int n, int L, int win, const cx<T>* tw, int sh) {
    const int m = L - n;
    cx<T> A = mk<T>((T)0, (T)0), B = A;
    if (n >= win && m >= win) return A;
    if (n < win) A = row[n];
    if (m < win) B = row[m];
    const cx<T> d = A - conj(B);
    return swp((A + conj(B)) + mul_pi(conj(tw[n << sh]) * d));
}
// Number of live taps per side (1 or 2) of a radix-16 first inverse stage on L points with `win` active columns,
// or 0 = general prologue path.
OA_HD int qe_first_stage_nz(int logL, int R, int win) {
    // only radix-16 first stages with <= 2 live taps per side pay (rows of 8192 points and up at the reference's
    // band limits): with more live taps the guarded per-tap loads serialise and the variants spill -- measured
    // 1.2-2x slower than the prologue pass on 2048..4096-point rows (profiles/r02c_rowqe_variants.txt)
    if (R != 16 || win <= 0 || win > (1 << logL)) return 0;
    const int S = (1 << logL) / R;
    const int need = (win + S - 1) / S;
    return need <= 1 ? 1 : (need <= 2 ? 2 : 0);
}
// host-side: call f(std::integral_constant<int, NZ>) for the instantiated variant (NZ in {0, 1, 2})
template <class SEQ, class F>
inline void dispatch_nz(int nz, F&& f) {
    if constexpr (SEQ::n >= 2 && SEQ::rget(0) == 16) {
        if (nz == 1) { f(std::integral_constant<int, 1>{}); return; }
        if (nz == 2) { f(std::integral_constant<int, 2>{}); return; }
    }
    f(std::integral_constant<int, 0>{});
}

template <typename T, class SEQ, int NZ, class Ctx>
OA_HD void inverse_to_regs_pruned(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw,
                                  const cx<T>* in, long pitch, long r0, const cx<T>* gtw, int logTw, int win) {
    constexpr int n = SEQ::n;
    constexpr int R = SEQ::rget(0), LR = Log2c<R>::v, NB = EPT / R;
    static_assert(n >= 2, "pruned first stage needs at least two stages");
    const int logLR = logL - LR, L = 1 << logL, sh = logTw - (logL + 1);
    {
        cx<T> w[EPT];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int b = tid + u * NT;
            const int j = b & ((1 << logLR) - 1), c = b >> logLR;
            const cx<T>* row = in + (r0 + c) * pitch;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                if (t < NZ || t >= R - NZ) w[u * R + t] = c2r_tap<T>(row, j + (t << logLR), L, win, gtw, sh);
                else w[u * R + t] = mk<T>((T)-0.0, (T)-0.0);
            }
            Dft<T, R>::run(w + u * R);
        }
        stage_out<T, R, true, false>(s, w, tid, NT, logL, logC, RS, 0, NoStore{});
    }
    ctx.sync();
    if constexpr (n >= 3) lds_stage<T, SEQ, 1, true>(ctx, s, tid, NT, logL, logC, RS, tw, logL);
    if constexpr (n >= 4) lds_stage<T, SEQ, 2, true>(ctx, s, tid, NT, logL, logC, RS, tw, logL);
    stage_in<T, SEQ::get(0), true, false>(s, v, tid, NT, logL, logC, RS, rev_logns<SEQ>(n - 1), tw, logL, NoLoad{});
}

// The inverse transforms run the REVERSED radix sequence, so their last stage has radix R0 and leaves
// point n = j + t*(L/R0) in register t of thread j -- exactly the operand layout of the forward
// transform's first (twiddle-free) stage.  h therefore stays in 16 registers per thread, the
// real-space product is a register multiply, and the only LDS is one padded work row set.
template <typename T, class SEQ, int NZ = 0, class Ctx>
OA_HD void row_qe_body(Ctx& ctx, const RowQeArgs<T>& a) {
    cx<T>* work = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    constexpr int logL = seq_total_log<SEQ>();
    const int logC = a.logC, C = 1 << logC, RS = a.rowStride;
    const long r0 = (long)ctx.bid_x() * C;
    constexpr int R0 = SEQ::get(0);
    constexpr int lastns = rev_logns<SEQ>(SEQ::n - 1);
    cx<T> hreg[EPT], v[EPT];
    cx<T>* twl = work + C * RS;                   // two-level stage-twiddle table (LDS)
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);

    if constexpr (NZ > 0 && SEQ::n >= 2) {
        ctx.sync();   // twiddle table complete before the first LDS-reading stage
        inverse_to_regs_pruned<T, SEQ, NZ>(ctx, work, hreg, tid, NT, logL, logC, RS, twl, a.h, a.pitch, r0, a.tw, a.logTw, a.win);
    } else {
        c2r_prologue<T>(ctx, work, a.h, a.pitch, r0, logL, logC, NT, RS, a.tw, a.logTw, a.win);
        ctx.sync();
        inverse_head<T, SEQ>(ctx, work, tid, NT, logL, logC, RS, twl, logL);
        stage_in<T, R0, true, false>(work, hreg, tid, NT, logL, logC, RS, lastns, twl, logL, NoLoad{});
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t) hreg[t] = mk<T>(hreg[t].y * a.scale, hreg[t].x * a.scale);  // unswap -> (h[2n], h[2n+1])
    ctx.sync();
    for (int leg = 0; leg < 2; ++leg) {
        const cx<T>* src = leg ? a.gy : a.gx;
        cx<T>* dst = leg ? a.py : a.px;
        if constexpr (NZ > 0 && SEQ::n >= 2) {
            inverse_to_regs_pruned<T, SEQ, NZ>(ctx, work, v, tid, NT, logL, logC, RS, twl, src, a.pitch, r0, a.tw, a.logTw, a.win);
        } else {
            c2r_prologue<T>(ctx, work, src, a.pitch, r0, logL, logC, NT, RS, a.tw, a.logTw, a.win);
            ctx.sync();
            inverse_head<T, SEQ>(ctx, work, tid, NT, logL, logC, RS, twl, logL);
            stage_in<T, R0, true, false>(work, v, tid, NT, logL, logC, RS, lastns, twl, logL, NoLoad{});
        }
        // v holds the swapped C2R result: (im, re) = (x[2n+1], x[2n]); product with h, repacked for R2C
#pragma unroll
        for (int t = 0; t < EPT; ++t) v[t] = mk<T>(v[t].y * hreg[t].x, v[t].x * hreg[t].y);
        // forward stage 0 (no twiddles) straight from registers
#pragma unroll
        for (int u = 0; u < EPT / R0; ++u) Dft<T, R0>::run(v + u * R0);
        ctx.sync();
        stage_out<T, R0, true, false>(work, v, tid, NT, logL, logC, RS, 0, NoStore{});
        ctx.sync();
        forward_tail<T, SEQ>(ctx, work, tid, NT, logL, logC, RS, twl, logL);
        r2c_epilogue<T>(ctx, work, dst, a.opitch, r0, logL, logC, NT, RS, a.tw, a.logTw, (T)1, a.accumulate != 0, a.wout);
        ctx.sync();
    }
}

// ===========================================================================
// Fused QE row stage on an alias-free row grid, TWO ROWS PER TRANSFORM.
// Band-limited legs (columns >= win vanish, 2 win + wout <= M) let the row stage run on M < nx points (ROW GRID,
// include/orphics_amd.h).  On that grid two real rows are carried by ONE complex transform of length M instead of
// two packed M/2-point transforms with their (un)tangle passes:
//   Z[k] = X0[k] + i X1[k] (k < win),  Z[M-k] = conj X0[k] + i conj X1[k],  0 elsewhere
//   IDFT_M(Z) = x0 + i x1;  the product with h0 + i h1 is taken per component: p = x0 h0 + i x1 h1;
//   P = DFT_M(p):  P0[k] = (P[k] + conj P[M-k]) / 2,  P1[k] = (P[k] - conj P[M-k]) / (2i),  k < wout.
// The packing costs one add per live tap of the first inverse stage (taken straight from the two half-complex rows,
// dead taps are literal zeros), the unpacking one LDS pass over the wout kept columns -- no twiddles, no tangle.
// One workgroup of M/16 threads per row pair; h0 + i h1 stays in 16 registers per thread as in row_qe_body.
// ===========================================================================
// the two rows of a pair at column idx.  LR = 0: two adjacent rows of a natural-order plane.  LR = 2 (R-LAYOUT, the output
// of col_fband_body): the plane holds B[k1][y_lo] at row y_lo R + k1 and row y_lo + Mq y_hi of the field is
// sum_k1 W_R^(-k1 y_hi) B[k1][y_lo] -- the last radix-R butterfly of the inverse column transform, taken here at the load;
// the pair (y_hi = 2 p, 2 p + 1) of group y_lo: with s02 = B0 + B2, d02 = B0 - B2, s13 = B1 + B3, d13 = B1 - B3 and
// sg = +1 (p = 0) / -1 (p = 1):  a0 = s02 + sg s13,  a1 = d02 + sg i d13.
// LAY = 1 (R = 2, the wide band of 8192^2 maps on the 4096-row column grid; 8-point row stage only): rows 2 y_lo, 2 y_lo + 1 of the
// plane hold B0, B1 and the one workgroup of group y_lo forms the rows y_lo, y_lo + Mq:  a0 = B0 + B1,  a1 = B0 - B1.
// LAY = 3 (R = 8: 16384-row maps on the 2048-row column grid): the plane holds B[k1][y_lo] at row 8 y_lo + k1; workgroup p < 4 of
// group y_lo forms the rows y_hi = 2 p and 2 p + 1:  a0 = sum_k W_8^(-2 p k) b_k,  a1 = sum_k W_8^(-(2 p + 1) k) b_k.  With
// s_k = b_k + b_(k+4), d_k = b_k - b_(k+4) (k < 4):  a0 = (s0 + i^(2p) s2) + i^p (s1 + i^(2p) s3),
// a1 = (d0 + i^(2p+1) d2) + th (d1 + i^(2p+1) d3),  th = exp(+i pi (2 p + 1) / 4): one complex product per tap and row pair.
template <typename T, int LAY>
OA_HD void pair_rows_at(const cx<T>* row0, const cx<T>* row1, long pitch, T sg, int idx, cx<T>& a0, cx<T>& a1, int p = 0) {
    static_assert(LAY >= 0 && LAY <= 3, "pair_rows_at: natural layout, R = 2, R = 4 or R = 8");
    auto rd = [](const cx<T>* q) { return ldg(q); };
    if (LAY == 0) { a0 = rd(row0 + idx); a1 = rd(row1 + idx); return; }
    if (LAY == 1) {                                          // R = 2: rows y_lo and y_lo + Mq of the field are B0 + B1 and B0 - B1
        const cx<T> b0 = rd(row0 + idx), b1 = rd(row1 + idx);
        a0 = b0 + b1; a1 = b0 - b1;
        return;
    }
    if (LAY == 3) {
        cx<T> b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = rd(row0 + k * pitch + idx);
        const cx<T> s0 = b[0] + b[4], s1 = b[1] + b[5], s2 = b[2] + b[6], s3 = b[3] + b[7];
        const cx<T> d0 = b[0] - b[4], d1 = b[1] - b[5], d2 = b[2] - b[6], d3 = b[3] - b[7];
        a0 = (s0 + rot_i(s2, 2 * p)) + rot_i(s1 + rot_i(s3, 2 * p), p);
        const T h = (T)0.70710678118654752440L;
        const cx<T> th = mk<T>((p == 0 || p == 3) ? h : -h, (p < 2) ? h : -h);
        a1 = (d0 + rot_i(d2, 2 * p + 1)) + th * (d1 + rot_i(d3, 2 * p + 1));
        return;
    }
    const cx<T> b0 = rd(row0 + idx), b1 = rd(row0 + pitch + idx), b2 = rd(row0 + 2 * pitch + idx), b3 = rd(row0 + 3 * pitch + idx);
    const cx<T> s02 = b0 + b2, d02 = b0 - b2, s13 = (b1 + b3) * sg, d13 = (b1 - b3) * sg;
    a0 = s02 + s13;
    a1 = add_pi(d02, d13);
}

template <typename T, class SEQ, int NZ, int LAY = 0, class Ctx>
OA_HD void pair_inverse_to_regs(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int RS, const cx<T>* tw,
                                const cx<T>* row0, const cx<T>* row1, int win, long pitch = 0, T sg = (T)1, int p = 0) {
    constexpr int n = SEQ::n;
    constexpr int logM = seq_total_log<SEQ>();
    constexpr int R = SEQ::rget(0), LR = Log2c<R>::v, NB = EPT / R;
    static_assert(n >= 2, "pair kernel needs at least two stages");
    constexpr int logS = logM - LR, M = 1 << logM;
    {
        cx<T> w[EPT];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int j = (tid + u * NT) & ((1 << logS) - 1);
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const int nn = j + (t << logS);
                if (t < NZ) {                       // low side: Z[n] = X0[n] + i X1[n], n < win
                    const bool ok = nn < win;
                    const int idx = ok ? nn : 0;    // unconditional loads from a valid address, masked afterwards
                    cx<T> a0, a1;
                    pair_rows_at<T, LAY>(row0, row1, pitch, sg, idx, a0, a1, p);
                    const cx<T> z = add_pi(a0, a1);
                    w[u * R + t] = ok ? swp(z) : mk<T>((T)0, (T)0);
                } else if (t >= R - NZ) {            // high side: Z[n] = conj X0[M-n] + i conj X1[M-n], M - n < win
                    const int m = M - nn;
                    const bool ok = m < win;
                    const int idx = ok ? m : 0;
                    cx<T> a0, a1;
                    pair_rows_at<T, LAY>(row0, row1, pitch, sg, idx, a0, a1, p);
                    const cx<T> z = mk<T>(a0.x + a1.y, a1.x - a0.y);
                    w[u * R + t] = ok ? swp(z) : mk<T>((T)0, (T)0);
                } else {
                    w[u * R + t] = mk<T>((T)-0.0, (T)-0.0);
                }
            }
            Dft<T, R>::run(w + u * R);
        }
        stage_out<T, R, true, false>(s, w, tid, NT, logM, 0, RS, 0, NoStore{});
    }
    ctx.sync();
    if constexpr (n >= 3) lds_stage<T, SEQ, 1, true>(ctx, s, tid, NT, logM, 0, RS, tw, logM);
    if constexpr (n >= 4) lds_stage<T, SEQ, 2, true>(ctx, s, tid, NT, logM, 0, RS, tw, logM);
    stage_in<T, SEQ::get(0), true, false>(s, v, tid, NT, logM, 0, RS, rev_logns<SEQ>(n - 1), tw, logM, NoLoad{});
}

// live taps per side of the pair kernel's first inverse stage: ceil(win / (M/R)) rounded up to a power of two (<= R/2)
OA_HD int pair_first_stage_nz(int logM, int R, int win) {
    const int S = (1 << logM) / R;
    const int need = (win + S - 1) / S;
    int nz = 1;
    while (nz < need && nz < R / 2) nz <<= 1;
    return nz;
}
template <class SEQ, int NZ = 1, class F>
inline void dispatch_pair_nz(int nz, F&& f) {
    constexpr int R = SEQ::rget(0);
    if constexpr (2 * NZ >= R) {
        f(std::integral_constant<int, NZ>{});
    } else {
        if (nz <= NZ) f(std::integral_constant<int, NZ>{});
        else dispatch_pair_nz<SEQ, 2 * NZ>(nz, f);
    }
}

template <typename T, class SEQ, int NZ, int LAY = 0, bool CHAIN = false, class Ctx>
OA_HD void row_qe_pair_body(Ctx& ctx, const RowQeArgs<T>& a) {
    cx<T>* work = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    constexpr int logM = seq_total_log<SEQ>();
    constexpr int M = 1 << logM;
    const int RS = a.rowStride;
    long wg = ctx.bid_x();
    long imo = 0, omo = 0, hmo = 0;
    int m = 0;
    while (a.npairs && wg >= a.npairs) { wg -= a.npairs; imo += a.in_moff; omo += a.out_moff; hmo += a.h_moff; ++m; }
    const cx<T>* gxp = a.gx + imo; const cx<T>* gyp = a.gy + imo; const cx<T>* hp = a.h + hmo;
    cx<T>* pxp = a.px + omo; cx<T>* pyp = a.py + omo;
    T scale = a.scale;
    if (a.tab) {                       // uniform index: scalar loads
        const RowQeMap<T> e = a.tab[m];
        gxp = e.gx; gyp = e.gy; hp = e.h; pxp = e.px; pyp = e.py; scale = e.scale;
    }
    // natural layout: the pair is rows 2 wg, 2 wg + 1 of the leg planes and of the product planes.  R-LAYOUT (LR = 2: leg planes
    // from col_fband_body): workgroup wg = 2 y_lo + p reads the four rows 4 y_lo .. 4 y_lo + 3 (B[k1][y_lo]) and its pair is
    // rows y_lo + Mq (2 p), y_lo + Mq (2 p + 1) of the field, Mq = rows / 4, which is where its products go.
    long r0 = wg * 2, ra = wg * 2, rb = wg * 2 + 1;
    T sg = (T)1;
    int pp = 0;
    if (LAY == 3) {
        // R = 8: the FOUR workgroups of a group read the same eight rows: workgroups b, b + 8, b + 16, b + 24 of a block of 32 (same XCD)
        const long blk = wg & ~31L;
        const int r = (int)(wg & 31);
        pp = r >> 3;
        const long ylo = (blk >> 2) + (r & 7), mq = a.nrows >> 3;
        r0 = ylo << 3;
        ra = ylo + mq * (2 * pp);
        rb = ra + mq;
    } else if (LAY > 0) {
        // the two workgroups of a group read the same four rows: they are workgroups b and b + 8 of a block of 16 -- the same
        // XCD under the round-robin dispatch, a few slots apart -- so the second read is an L2 hit, not a second trip over the fabric
        const long blk = wg & ~15L;
        const int r = (int)(wg & 15), p = r >> 3;
        const long ylo = (blk >> 1) + (r & 7), mq = a.nrows >> LAY;
        sg = p ? (T)-1 : (T)1;
        r0 = ylo << LAY;
        ra = ylo + mq * (2 * p);
        rb = ra + mq;
    }
    constexpr int R0 = SEQ::get(0);
    cx<T> hreg[EPT], v[EPT];
    cx<T>* twl = work + RS;
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logM, NT);
    ctx.sync();
    if constexpr (CHAIN) {
        static_assert(LAY == 0, "chains read natural-order leg planes");
        const int first = a.chain[2 * m], count = a.chain[2 * m + 1];        // uniform: scalar loads
        // the running products of both legs live in REGISTERS: the workgroup then needs the transform tile only (35 KB at float64: two
        // workgroups per CU at the one-wave launch bound, whose unified file has 512 registers per lane; 18 KB and four per CU at
        // float32).  Behind the twiddle table in LDS they made a float64 workgroup 100 KB -- ONE 128-thread workgroup per CU, half the
        // SIMDs idle: 541 of the 911 us of an 8192^2 MV reconstruction -- and a float32 one 50 KB (three per CU): MV 1207 -> 1565
        // reconstructions/s at float64, 2790 -> 3065 at float32
        cx<T> acc[2][EPT];
#pragma unroll 1
        for (int i = 0; i < count; ++i) {
            const RowQeMap<T> e = a.tab[first + i];
            pair_inverse_to_regs<T, SEQ, NZ, 0>(ctx, work, hreg, tid, NT, RS, twl, e.h + r0 * a.pitch, e.h + (r0 + 1) * a.pitch, a.win);
#pragma unroll
            for (int t = 0; t < EPT; ++t) hreg[t] = hreg[t] * e.scale;
            ctx.sync();
#pragma unroll
            for (int leg = 0; leg < 2; ++leg) {
                const cx<T>* src = leg ? e.gy : e.gx;
                pair_inverse_to_regs<T, SEQ, NZ, 0>(ctx, work, v, tid, NT, RS, twl, src + r0 * a.pitch, src + (r0 + 1) * a.pitch, a.win);
                if (i == 0) {
#pragma unroll
                    for (int t = 0; t < EPT; ++t) acc[leg][t] = mk<T>(v[t].y * hreg[t].y, v[t].x * hreg[t].x);
                } else {
#pragma unroll
                    for (int t = 0; t < EPT; ++t) acc[leg][t] = acc[leg][t] + mk<T>(v[t].y * hreg[t].y, v[t].x * hreg[t].x);
                }
                ctx.sync();
            }
        }
        const RowQeMap<T> e0 = a.tab[first];
#pragma unroll
        for (int leg = 0; leg < 2; ++leg) {
            cx<T>* dst = leg ? e0.py : e0.px;
#pragma unroll
            for (int t = 0; t < EPT; ++t) v[t] = acc[leg][t];
#pragma unroll
            for (int u = 0; u < EPT / R0; ++u) Dft<T, R0>::run(v + u * R0);
            stage_out<T, R0, true, false>(work, v, tid, NT, logM, 0, RS, 0, NoStore{});
            ctx.sync();
            forward_tail<T, SEQ>(ctx, work, tid, NT, logM, 0, RS, twl, logM);
            cx<T>* o0 = dst + ra * a.opitch;
            cx<T>* o1 = dst + rb * a.opitch;
            for (int k = tid; k < a.wout; k += NT) {
                const int km = (M - k) & (M - 1);
                const cx<T> Pk = work[k + (k >> 4)];
                const cx<T> Pm = conj(work[km + (km >> 4)]);
                o0[k] = (Pk + Pm) * (T)0.5;
                o1[k] = mul_mi(Pk - Pm) * (T)0.5;
            }
            ctx.sync();
        }
        return;
    }
    pair_inverse_to_regs<T, SEQ, NZ, LAY>(ctx, work, hreg, tid, NT, RS, twl, hp + r0 * a.pitch, hp + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp);
    // hreg holds the swapped inverse: (h1, h0); the product scale rides on it
#pragma unroll
    for (int t = 0; t < EPT; ++t) hreg[t] = hreg[t] * scale;
    ctx.sync();
    for (int leg = 0; leg < 2; ++leg) {
        const cx<T>* src = leg ? gyp : gxp;
        cx<T>* dst = leg ? pyp : pxp;
        pair_inverse_to_regs<T, SEQ, NZ, LAY>(ctx, work, v, tid, NT, RS, twl, src + r0 * a.pitch, src + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp);
        // v = (g1, g0) swapped; p = g0 h0 + i g1 h1
#pragma unroll
        for (int t = 0; t < EPT; ++t) v[t] = mk<T>(v[t].y * hreg[t].y, v[t].x * hreg[t].x);
#pragma unroll
        for (int u = 0; u < EPT / R0; ++u) Dft<T, R0>::run(v + u * R0);
        ctx.sync();
        stage_out<T, R0, true, false>(work, v, tid, NT, logM, 0, RS, 0, NoStore{});
        ctx.sync();
        forward_tail<T, SEQ>(ctx, work, tid, NT, logM, 0, RS, twl, logM);
        // unpack the kept columns of both rows
        cx<T>* o0 = dst + ra * a.opitch;
        cx<T>* o1 = dst + rb * a.opitch;
        for (int k = tid; k < a.wout; k += NT) {
            const int km = (M - k) & (M - 1);
            const cx<T> Pk = work[k + (k >> 4)];
            const cx<T> Pm = conj(work[km + (km >> 4)]);
            cx<T> p0 = (Pk + Pm) * (T)0.5;
            cx<T> p1 = mul_mi(Pk - Pm) * (T)0.5;
            if (a.accumulate) { p0 = p0 + o0[k]; p1 = p1 + o1[k]; }
            o0[k] = p0;
            o1[k] = p1;
        }
        ctx.sync();
    }
}

// ---------------------------------------------------------------------------
// In-place variant of the fused row stage.  The Stockham stages above read one index set and write another, so
// every LDS->LDS stage costs two barriers (all reads before any write, all writes before the next reads).  Here the
// inverse transforms are decimation-in-frequency IN PLACE (natural order in, digit-reversed order out) and the
// forward transforms decimation-in-time IN PLACE (digit-reversed in, natural out): a thread rewrites exactly the
// points it read, so a stage needs ONE barrier, and the real-space product -- elementwise, hence indifferent to the
// common permutation of both factors -- sits between the last DIF stage and the first DIT stage, which own the same
// 16 contiguous points per thread: h, the product and both of those stages stay in registers without any barrier.
// 18 barriers per row instead of 30.
//   stage of block size B = 2^logB, radix R, sub-stride S = B/R: sets { q*B + j + t*S : t < R },  j < S
//   DIF: y = DFT_R(x) ; y_t *= W_B^(j t)        DIT: x_t *= W_B^(j t) ; y = DFT_R(x)
// ---------------------------------------------------------------------------
template <typename T, int R, bool DIT, bool TO_LDS, bool FROM_LDS>
OA_HD void ip_stage(cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int RS, int logB, const cx<T>* tw) {
    constexpr int LR = Log2c<R>::v;
    constexpr int NB = EPT / R;
    const int logS = logB - LR, logSets = logL - LR;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b >> logSets, sg = b & ((1 << logSets) - 1);
        const int j = sg & ((1 << logS) - 1), q = sg >> logS;
        const int base = (q << logB) + j;
        cx<T>* w = v + u * R;
        if (FROM_LDS) {
#pragma unroll
            for (int t = 0; t < R; ++t) w[t] = s[lds_addr<true>(base + (t << logS), c, logC, RS)];
        }
        if (DIT && logS > 0) apply_twiddles<T, R>(w, tw, j, logL - logB, tw_lds_h(logL));
        Dft<T, R>::run(w);
        if (!DIT && logS > 0) apply_twiddles<T, R>(w, tw, j, logL - logB, tw_lds_h(logL));
        if (TO_LDS) {
#pragma unroll
            for (int t = 0; t < R; ++t) s[lds_addr<true>(base + (t << logS), c, logC, RS)] = w[t];
        }
    }
}

// log2 of the block size of DIF stage i (radix order SEQ::get(0), get(1), ...): L / (r0 ... r_{i-1})
template <class SEQ> constexpr int dif_logb(int i) { return seq_total_log<SEQ>() - fwd_logns<SEQ>(i); }

// natural-order spectrum in LDS (caller synced) -> digit-reversed transform; the LAST stage stays in registers v
template <typename T, class SEQ, class Ctx>
OA_HD void dif_to_regs(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw) {
    constexpr int n = SEQ::n;
    cx<T> w[EPT];
    if constexpr (n >= 2) { ip_stage<T, SEQ::get(0), false, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(0), tw); ctx.sync(); }
    if constexpr (n >= 3) { ip_stage<T, SEQ::get(1), false, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(1), tw); ctx.sync(); }
    if constexpr (n >= 4) { ip_stage<T, SEQ::get(2), false, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(2), tw); ctx.sync(); }
    ip_stage<T, SEQ::get(n - 1), false, false, true>(s, v, tid, NT, logL, logC, RS, dif_logb<SEQ>(n - 1), tw);
}
// registers v (digit-reversed data, the layout dif_to_regs leaves) -> natural-order transform in LDS (synced)
template <typename T, class SEQ, class Ctx>
OA_HD void dit_from_regs(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int logL, int logC, int RS, const cx<T>* tw) {
    constexpr int n = SEQ::n;
    cx<T> w[EPT];
    ip_stage<T, SEQ::get(n - 1), true, true, false>(s, v, tid, NT, logL, logC, RS, dif_logb<SEQ>(n - 1), tw);
    ctx.sync();
    if constexpr (n >= 4) { ip_stage<T, SEQ::get(2), true, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(2), tw); ctx.sync(); }
    if constexpr (n >= 3) { ip_stage<T, SEQ::get(1), true, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(1), tw); ctx.sync(); }
    if constexpr (n >= 2) { ip_stage<T, SEQ::get(0), true, true, true>(s, w, tid, NT, logL, logC, RS, dif_logb<SEQ>(0), tw); ctx.sync(); }
}

template <typename T, class SEQ, class Ctx>
OA_HD void row_qe_body_inplace(Ctx& ctx, const RowQeArgs<T>& a) {
    cx<T>* work = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid(), NT = a.NT;
    constexpr int logL = seq_total_log<SEQ>();
    const int logC = a.logC, C = 1 << logC, RS = a.rowStride;
    const long r0 = (long)ctx.bid_x() * C;
    cx<T> hreg[EPT], v[EPT];
    cx<T>* twl = work + C * RS;
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);

    c2r_prologue<T>(ctx, work, a.h, a.pitch, r0, logL, logC, NT, RS, a.tw, a.logTw, a.win);
    ctx.sync();
    dif_to_regs<T, SEQ>(ctx, work, hreg, tid, NT, logL, logC, RS, twl);
#pragma unroll
    for (int t = 0; t < EPT; ++t) hreg[t] = mk<T>(hreg[t].y * a.scale, hreg[t].x * a.scale);  // unswap -> (h[2n], h[2n+1])
    ctx.sync();
    for (int leg = 0; leg < 2; ++leg) {
        const cx<T>* src = leg ? a.gy : a.gx;
        cx<T>* dst = leg ? a.py : a.px;
        c2r_prologue<T>(ctx, work, src, a.pitch, r0, logL, logC, NT, RS, a.tw, a.logTw, a.win);
        ctx.sync();
        dif_to_regs<T, SEQ>(ctx, work, v, tid, NT, logL, logC, RS, twl);
        // swapped C2R result (im, re) = (x[2n+1], x[2n]) at the same permuted positions as h: product, repacked for R2C
#pragma unroll
        for (int t = 0; t < EPT; ++t) v[t] = mk<T>(v[t].y * hreg[t].x, v[t].x * hreg[t].y);
        dit_from_regs<T, SEQ>(ctx, work, v, tid, NT, logL, logC, RS, twl);   // this thread rewrites the points it just read
        r2c_epilogue<T>(ctx, work, dst, a.opitch, r0, logL, logC, NT, RS, a.tw, a.logTw, (T)1, a.accumulate != 0, a.wout);
        ctx.sync();
    }
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
    // batched launch (grid z = plane 0..2): element offsets of planes 1 and 2 from plane 0.  Offsets + arithmetic
    // select on purpose: a ?: chain over pointer members is turned into an indexed read of the by-value
    // argument struct, which drags the whole struct into scratch memory (2x slower passes)
    long in_off1, in_off2, out_off1, out_off2;
    int nbz; long in_moff, out_moff;   // two maps per launch: z = map * nbz + plane, the second map's planes sit *_moff behind (nbz = 0: one map)
    int rband, ny;             // rband > 0: output rows outside the band (rband <= y <= ny - rband) are not stored
    long in_pitch, out_pitch;  // complex elements
    int width;                 // valid columns
    int logL, logC, NT;
    const cx<T>* tw;           // master table of length 2^logTw (= Ny)
    int logTw;
    long in_gs, in_ns, out_gs, out_ks;
    int twiddle;               // multiply bin k of group g by W_{Ny}^{g k}
    int inverse;
    T scale;
};

template <typename T>
struct ColLoad {
    static constexpr bool reads_lds = false;
    const cx<T>* base;  // already offset to (group row origin, first column)
    unsigned nstride;   // elements between consecutive points n (32-bit: planes hold < 2^31 elements)
    int ncols;          // valid columns in this tile
    bool inv;
    template <typename U> OA_HD cx<U> get(int n, int c) const {
        cx<U> x = mk<U>((U)0, (U)0);
        if (c < ncols) x = base[(unsigned)n * nstride + (unsigned)c];
        return inv ? swp(x) : x;
    }
};
template <typename T>
struct ColStore {
    cx<T>* base;
    unsigned kstride;
    int ncols;
    bool inv;
    const cx<T>* tw;  // LDS table of the inter-pass twiddles W_N^(g k), k < L, or nullptr
    unsigned g;
    T scale;
    int rb, row0, rowstep, ny;   // rb > 0: rows row0 + k*rowstep inside [rb, ny - rb] are not stored
    template <typename U> OA_HD void put(int k, int c, cx<U> v) const {
        if (c >= ncols) return;
        if (rb) { const int y = row0 + k * rowstep; if (y >= rb && y <= ny - rb) return; }
        if (tw) v = v * tw[k];
        if (inv) v = swp(v);
        base[(unsigned)k * kstride + (unsigned)c] = v * scale;
    }
};

// ---- column pipelines that start from / end in registers (fused estimator passes) ----------------
// v holds this thread's first-stage inputs (element u*R0+t <-> point j_u + t*L/R0); results go to `st`.
template <typename T, class SEQ, class Ctx, class St>
OA_HD void col_pipeline_from_regs(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int logC, const cx<T>* tw, int logTw,
                                  const St& st) {
    constexpr int logL = Log2x<SEQ::r0>::v + Log2x<SEQ::r1>::v + Log2x<SEQ::r2>::v + Log2x<SEQ::r3>::v;
    constexpr int R0 = SEQ::get(0);
#pragma unroll
    for (int u = 0; u < EPT / R0; ++u) Dft<T, R0>::run(v + u * R0);
    if constexpr (SEQ::n == 1) {
        stage_out<T, R0, false, true>(s, v, tid, NT, logL, logC, 0, 0, st);
    } else {
        stage_out<T, R0, false, false>(s, v, tid, NT, logL, logC, 0, 0, NoStore{});
        ctx.sync();
        constexpr int l0 = Log2x<SEQ::r0>::v, l1 = l0 + Log2x<SEQ::r1>::v, l2 = l1 + Log2x<SEQ::r2>::v;
        if constexpr (SEQ::n == 2) {
            stage<T, SEQ::r1, false, false, true>(ctx, s, tid, NT, logL, logC, 0, l0, tw, logTw, NoLoad{}, st);
        } else {
            stage<T, SEQ::r1, false, false, false>(ctx, s, tid, NT, logL, logC, 0, l0, tw, logTw, NoLoad{}, NoStore{});
            ctx.sync();
            if constexpr (SEQ::n == 3) {
                stage<T, SEQ::r2, false, false, true>(ctx, s, tid, NT, logL, logC, 0, l1, tw, logTw, NoLoad{}, st);
            } else {
                stage<T, SEQ::r2, false, false, false>(ctx, s, tid, NT, logL, logC, 0, l1, tw, logTw, NoLoad{}, NoStore{});
                ctx.sync();
                stage<T, SEQ::r3, false, false, true>(ctx, s, tid, NT, logL, logC, 0, l2, tw, logTw, NoLoad{}, st);
            }
        }
    }
}

// full pipeline from a global load functor whose LAST stage stays in registers:
// v[u*RL + t] = bin (base_u + t*Ns) of the transform, RL = last radix, Ns = L/RL.
template <typename T, class SEQ, class Ctx, class Ld>
OA_HD void col_pipeline_to_regs(Ctx& ctx, cx<T>* s, cx<T>* v, int tid, int NT, int logC, const cx<T>* tw, int logTw,
                                const Ld& ld) {
    constexpr int logL = Log2x<SEQ::r0>::v + Log2x<SEQ::r1>::v + Log2x<SEQ::r2>::v + Log2x<SEQ::r3>::v;
    constexpr int n = SEQ::n;
    constexpr int RL = SEQ::get(n - 1);
    if constexpr (n == 1) {
        stage_in<T, RL, false, true>(s, v, tid, NT, logL, logC, 0, 0, tw, logTw, ld);
    } else {
        stage<T, SEQ::r0, false, true, false>(ctx, s, tid, NT, logL, logC, 0, 0, tw, logTw, ld, NoStore{});
        ctx.sync();
        constexpr int l0 = Log2x<SEQ::r0>::v, l1 = l0 + Log2x<SEQ::r1>::v, l2 = l1 + Log2x<SEQ::r2>::v;
        if constexpr (n >= 3) {
            stage<T, SEQ::r1, false, false, false>(ctx, s, tid, NT, logL, logC, 0, l0, tw, logTw, NoLoad{}, NoStore{});
            ctx.sync();
        }
        if constexpr (n >= 4) {
            stage<T, SEQ::r2, false, false, false>(ctx, s, tid, NT, logL, logC, 0, l1, tw, logTw, NoLoad{}, NoStore{});
            ctx.sync();
        }
        constexpr int ll = (n == 2) ? l0 : ((n == 3) ? l1 : l2);
        stage_in<T, RL, false, false>(s, v, tid, NT, logL, logC, 0, ll, tw, logTw, NoLoad{});
    }
}

// ===========================================================================
// (A) leg filters fused into the inverse column pass 1: kX, kY are read ONCE per tile and the three
//     leg planes  Gx = i lx FG kX,  Gy = i ly FG kX,  H = FH kY  leave as pass-1 outputs.
// ===========================================================================
template <typename T>
struct ColLegsArgs {
    const cx<T>* kX; const cx<T>* kY;
    const T* FG; const T* FH;
    const T* lxd; const T* lyd;
    cx<T>* gx; cx<T>* gy; cx<T>* h;
    long pitch;      // row pitch of kX, kY
    long fpitch;     // row pitch of the filter planes FG, FH
    long opitch;     // row pitch of the three output planes (compact work planes: see Fft2dPlan::work_pitch)
    int width, logC, NT;
    const cx<T>* tw;
    int logTw;
    long in_gs, in_ns, out_gs, out_ks;
    int twiddle;
    int rband, ny;   // rband > 0: the filters vanish on input rows rband <= y <= ny - rband, which are not read
    // COLUMN GRID (include/orphics_amd.h): this transform runs on ny = My < ny_full rows; its input row y stands for
    // row y + (y >= ny/2 ? yshift : 0) of the full-resolution grid, where the filters, the ly axis and -- when
    // xfull -- kX / kY live.  yshift = ny_full - My; 0 = the plan's own grid.
    int yshift, xfull;
    // split != 0: launched with grid z = 3, workgroup z computes ONE leg plane (0 = H, 1 = Gx, 2 = Gy): three times the
    // workgroups for the small latency-bound launches of the column grid (the tile's inputs are re-read per leg)
    int split;
    // with split: legs zbase .. zbase + zcount - 1 only (zcount = 0: all three).  H alone (0, 1) or the gradient pair alone
    // (1, 2): estimators that share a filtered field transform it once (oa_qe_mv)
    int zbase, zcount;
    // batch > 0 (oa_qe_mv): grid z = batch leg planes in ONE launch.  Plane z < 2 ngrad: Gx (z even) / Gy (z odd) of gradient
    // field z / 2; plane z >= 2 ngrad: H of field z - ngrad.  Field f is filter plane ftab[f] (device table of the caller's
    // h + z * ostride.  (Offsets and bit fields, not by-value pointer tables: see ColArgs; the filter pointers sit in a small
    // DEVICE table, read with a uniform index.)
    int batch, ngrad;
    int selbits;     // 2: source of field f = {0, src_off1, src_off2}[(srcsel >> 2f) & 3]; 4: source (srcsel >> 4f) & 15 of an evenly
                     // spaced family, src_off1 apart (oa_mc_run: one realisation each)
    unsigned long long srcsel;
    long src_off1, src_off2, ostride;
    const T* const* ftab;
};

// LOGC: log2 of the tile width.  The default (32 columns) is the two-pass layout (inverse pass 1; the caller runs pass 2); with a
// WHOLE column in the tile (SEQ = the full coarse column length, 8 or 16 columns -- f64: 4 or 8 --, in_ns = out_ks = 1, one group,
// no inter-pass twiddle) the same body is a SINGLE-PASS filtered inverse column transform: the leg plane is written once, in
// natural order, and no pass-2 launch follows (batch mode: oa_qe_mv, oa_mc_run).
template <typename T, class SEQ, class Ctx, int LOGC = COL_LOGC>
OA_HD void col_legs_body(Ctx& ctx, const ColLegsArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    constexpr int logL = Log2x<SEQ::r0>::v + Log2x<SEQ::r1>::v + Log2x<SEQ::r2>::v + Log2x<SEQ::r3>::v;
    constexpr int R0 = SEQ::get(0), LR = Log2x<R0>::v, NB = EPT / R0;
    constexpr int logC = LOGC;
    constexpr int NT = ((1 << (seq_total_log<SEQ>() + LOGC)) / EPT) > 0 ? ((1 << (seq_total_log<SEQ>() + LOGC)) / EPT) : 1;
    const int tid = ctx.tid();
    const int c0 = ctx.bid_x() << logC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << logC)) ncols = 1 << logC;
    // register budget: gv (kept across the three pipelines) + v (pipeline operand) only
    cx<T> gv[EPT], v[EPT];
    cx<T>* twl = s + (1 << (logL + logC));        // stage twiddles, then the inter-pass twiddles W_N^(g k)
    cx<T>* ti = twl + tw_lds_size(logL);
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    if (a.twiddle)
        for (int i = tid; i < (1 << logL); i += NT) ti[i] = a.tw[(unsigned)g * (unsigned)i];
    ctx.sync();
    const long org = g * a.in_gs * a.pitch + c0;   // scalar tile origin; per-element offsets are 32-bit
    const long forg = g * a.in_gs * a.fpitch + c0;
    const cx<T>* kXb = a.kX + org; const cx<T>* kYb = a.kY + org;
    const T* FGb = a.FG + forg; const T* FHb = a.FH + forg;
    int only = a.split ? ctx.bid_z() + a.zbase : -1;       // uniform per workgroup
    bool ldx = a.zcount == 0 || a.zbase, ldy = a.zbase == 0;
    cx<T>* ogx = a.gx; cx<T>* ogy = a.gy; cx<T>* oh = a.h;
    if (a.batch) {
        const int z = ctx.bid_z();
        const bool grad = z < 2 * a.ngrad;
        const int f = grad ? (z >> 1) : z - a.ngrad;
        if (a.selbits == 4) kXb += (long)((unsigned)(a.srcsel >> (4 * f)) & 15u) * a.src_off1;
        else {
            const unsigned sel = (unsigned)(a.srcsel >> (2 * f)) & 3u;
            kXb += (long)(sel == 1u) * a.src_off1 + (long)(sel == 2u) * a.src_off2;
        }
        kYb = kXb;
        FGb = a.ftab[f] + forg;
        FHb = FGb;
        only = grad ? 1 + (z & 1) : 0;
        ldx = grad; ldy = !grad;
        ogx = ogy = oh = a.h + (long)z * a.ostride;
    }
    const unsigned nstr = (unsigned)(a.in_ns * a.pitch), fstr = (unsigned)(a.in_ns * a.fpitch);
    const unsigned fsh = (unsigned)a.yshift * (unsigned)a.fpitch, xsh = a.xfull ? (unsigned)a.yshift * (unsigned)a.pitch : 0u;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & ((1 << logC) - 1), j = b >> logC;
        const bool ok = c < ncols;
#pragma unroll
        for (int t = 0; t < R0; ++t) {
            const int n = j + (t << (logL - LR));
            const unsigned i = (unsigned)n * nstr + (unsigned)c, fi = (unsigned)n * fstr + (unsigned)c;
            cx<T> kx = mk<T>((T)0, (T)0), ky = kx;
            T fg = 0, fh = 0;
            bool live = ok;
            const int y = (int)(g * a.in_gs) + n * (int)a.in_ns;
            if (a.rband) live = ok && !(y >= a.rband && y <= a.ny - a.rband);
            const unsigned up = (a.yshift && y >= (a.ny >> 1)) ? 1u : 0u;     // upper half: rows of negative ky
            const unsigned ix = i + up * xsh, ifl = fi + up * fsh;
            if (live) {
                if (ldx) { kx = kXb[ix]; fg = FGb[ifl]; }      // the gradient leg's operands
                if (ldy) { ky = kYb[ix]; fh = FHb[ifl]; }
            }
            gv[u * R0 + t] = kx * fg;
            v[u * R0 + t] = swp(ky * fh);  // inverse transform = forward transform of the swapped data
        }
    }
    if (only < 0 || only == 0) {   // H = FH kY
        const ColStore<T> st{oh + g * a.out_gs * a.opitch + c0, (unsigned)(a.out_ks * a.opitch), ncols, true,
                             a.twiddle ? ti : nullptr, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQ>(ctx, s, v, tid, NT, logC, twl, logL, st);
        if (only < 0) ctx.sync();
    }
    if (only < 0 || only == 1) {   // Gx = i lx FG kX   (lx is constant along a column)
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int c = (tid + u * NT) & ((1 << logC) - 1);
            const T lx = (c < ncols) ? a.lxd[c0 + c] : (T)0;
#pragma unroll
            for (int t = 0; t < R0; ++t) v[u * R0 + t] = swp(mul_pi(gv[u * R0 + t]) * lx);
        }
        const ColStore<T> st{ogx + g * a.out_gs * a.opitch + c0, (unsigned)(a.out_ks * a.opitch), ncols, true,
                             a.twiddle ? ti : nullptr, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQ>(ctx, s, v, tid, NT, logC, twl, logL, st);
        if (only < 0) ctx.sync();
    }
    if (only < 0 || only == 2) {   // Gy = i ly FG kX   (ly follows the input row of each tap)
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int j = (tid + u * NT) >> logC;
#pragma unroll
            for (int t = 0; t < R0; ++t) {
                unsigned y = (unsigned)(g * a.in_gs) + (unsigned)(j + (t << (logL - LR))) * (unsigned)a.in_ns;
                if (a.yshift && y >= (unsigned)(a.ny >> 1)) y += (unsigned)a.yshift;
                v[u * R0 + t] = swp(mul_pi(gv[u * R0 + t]) * a.lyd[y]);
            }
        }
        const ColStore<T> st{ogy + g * a.out_gs * a.opitch + c0, (unsigned)(a.out_ks * a.opitch), ncols, true,
                             a.twiddle ? ti : nullptr, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQ>(ctx, s, v, tid, NT, logC, twl, logL, st);
    }
}

// ===========================================================================
// (A') the same with the FORWARD column pass 2 of the input map in front: when both legs come from one real map,
//      kT itself never exists in HBM.  The forward column transform is split Ny = N1f * L (pass 1: N1f points,
//      pass 2: L points, L = 2^floor(log2(Ny)/2)); its pass-2 tile of group g holds exactly the rows g + N1f*n that
//      an inverse transform split the other way round (pass 1: L points at stride N1f, pass 2: N1f points) needs
//      for ITS pass 1.  The inverse pipelines run the REVERSED radix sequence, so the forward result (last radix
//      RL, bins j + t*L/RL in register t) is already the operand layout of their first stage (same trick as the
//      fused row stage).  in = output of the forward column pass 1 (block-transposed, twiddled).
// ===========================================================================
template <class SEQ> struct RevSeq {
    using type = Seq<SEQ::rget(0), (SEQ::n > 1 ? SEQ::rget(1) : 1), (SEQ::n > 2 ? SEQ::rget(2) : 1), (SEQ::n > 3 ? SEQ::rget(3) : 1)>;
};

template <typename T>
struct ColFwdLegsArgs {
    const cx<T>* in;            // forward pass-1 output of the map's row transform
    const T* FG; const T* FH;
    const T* lxd; const T* lyd;
    cx<T>* gx; cx<T>* gy; cx<T>* h;
    long pitch;                 // row pitch of `in`
    long fpitch;                // row pitch of FG, FH
    long opitch;                // row pitch of gx, gy, h
    int width;
    const cx<T>* tw;            // W_Ny^k
    int logTw;
    long n1f;                   // row stride of the tile: rows g + n1f * n, n < L
    int rband, ny;
};

template <typename T, class SEQF, class Ctx>
OA_HD void col_fwdlegs_body(Ctx& ctx, const ColFwdLegsArgs<T>& a) {
    using SEQI = typename RevSeq<SEQF>::type;
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    constexpr int logL = seq_total_log<SEQF>();
    constexpr int RL = SEQF::get(SEQF::n - 1), LRL = Log2x<RL>::v, NB = EPT / RL;
    constexpr int logNs = logL - LRL;
    constexpr int logC = COL_LOGC;
    constexpr int NT = ((1 << (logL + COL_LOGC)) / EPT) > 0 ? ((1 << (logL + COL_LOGC)) / EPT) : 1;
    static_assert(SEQI::get(0) == RL, "inverse pipelines must start with the forward pipeline's last radix");
    const int tid = ctx.tid();
    const int c0 = ctx.bid_x() << logC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << logC)) ncols = 1 << logC;
    cx<T> gv[EPT], v[EPT];
    cx<T>* twl = s + (1 << (logL + logC));
    cx<T>* ti = twl + tw_lds_size(logL);
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    for (int i = tid; i < (1 << logL); i += NT) ti[i] = a.tw[(unsigned)g * (unsigned)i];   // inverse inter-pass W_Ny^(g k)
    ctx.sync();
    const long org = g * a.pitch + c0;
    const unsigned rstr = (unsigned)(a.n1f * a.pitch);
    // forward pass 2 of this tile, result left in registers: gv[u*RL+t] = kT[row g + n1f*(j_u + t*Ns)][c]
    const ColLoad<T> ld{a.in + org, rstr, ncols, false};
    col_pipeline_to_regs<T, SEQF>(ctx, s, gv, tid, NT, logC, twl, logL, ld);
    ctx.sync();
    const T* FGb = a.FG + (g * a.fpitch + c0); const T* FHb = a.FH + (g * a.fpitch + c0);
    const unsigned fstr = (unsigned)(a.n1f * a.fpitch);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & ((1 << logC) - 1), j = b >> logC;
        const bool ok = c < ncols;
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            const int k = j + (t << logNs);
            const unsigned i = (unsigned)k * fstr + (unsigned)c;
            T fg = 0, fh = 0;
            bool live = ok;
            if (a.rband) { const int y = (int)g + k * (int)a.n1f; live = ok && !(y >= a.rband && y <= a.ny - a.rband); }
            if (live) { fg = FGb[i]; fh = FHb[i]; }
            const cx<T> kx = gv[u * RL + t];
            gv[u * RL + t] = kx * fg;
            v[u * RL + t] = swp(kx * fh);
        }
    }
    // inverse pass 1 (length L, input stride n1f): outputs block-transposed at rows g*L + k, twiddled by W_Ny^(g k)
    const long oorg = g * ((long)1 << logL) * a.opitch + c0;
    {   // H = FH kT
        const ColStore<T> st{a.h + oorg, (unsigned)a.opitch, ncols, true, ti, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQI>(ctx, s, v, tid, NT, logC, twl, logL, st);
        ctx.sync();
    }
    {   // Gx = i lx FG kT
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int c = (tid + u * NT) & ((1 << logC) - 1);
            const T lx = (c < ncols) ? a.lxd[c0 + c] : (T)0;
#pragma unroll
            for (int t = 0; t < RL; ++t) v[u * RL + t] = swp(mul_pi(gv[u * RL + t]) * lx);
        }
        const ColStore<T> st{a.gx + oorg, (unsigned)a.opitch, ncols, true, ti, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQI>(ctx, s, v, tid, NT, logC, twl, logL, st);
        ctx.sync();
    }
    {   // Gy = i ly FG kT
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int j = (tid + u * NT) >> logC;
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                const unsigned y = (unsigned)g + (unsigned)(j + (t << logNs)) * (unsigned)a.n1f;
                v[u * RL + t] = swp(mul_pi(gv[u * RL + t]) * a.lyd[y]);
            }
        }
        const ColStore<T> st{a.gy + oorg, (unsigned)a.opitch, ncols, true, ti, (unsigned)g, (T)1, 0, 0, 0, 0};
        col_pipeline_from_regs<T, SEQI>(ctx, s, v, tid, NT, logC, twl, logL, st);
    }
}

// ===========================================================================
// (A'') COLUMN GRID version of (A'): forward column pass 2 of the map's transform (length L, all ny rows) + leg filter +
//       inverse pass 1 of the My-row transform in ONE kernel.
//       ny = N1f * L and My = N1f * Lq: the tile of group g = k1 holds X[k1 + N1f k2], k2 < L; the My-row spectrum is
//       X'[k1 + N1f k2'] with k2' = k2 (k2 < Lq/2) or k2 - (L - Lq) (k2 >= L - Lq/2), the rest of the band being empty.
//       The inverse splits My = Lq (pass 1, over k2', here) x N1f (pass 2, over k1, a plain column pass afterwards):
//         x[y_lo + Lq y_hi] = sum_k1 W_N1f^(-k1 y_hi) [ W_My^(-k1 y_lo) sum_k2' X'[k1 + N1f k2'] W_Lq^(-k2' y_lo) ].
//       Built for Lq = 16 = L / (last forward radix): after the forward pass every butterfly of a thread holds exactly one
//       kept bin, k2' = j (j = its position, j < 16): a thread owns j = q + 4 u (q = tid / 32, u = 0..3) of one column,
//       so the 16-point inverse is a DFT-4 over u in registers, a W16 twiddle, and a DFT-4 over q through 4 KB of LDS.
// ===========================================================================
template <typename T>
struct ColFwdLegsCgArgs {
    const cx<T>* in;            // forward pass-1 output of the map's row transform (full resolution, ny rows)
    const T* FG; const T* FH;   // full-resolution filter planes
    const T* lxd; const T* lyd;
    cx<T>* gx; cx<T>* gy; cx<T>* h;   // My-row planes, block-transposed pass-1 layout: row k1 * Lq + y_lo
    long pitch, fpitch, opitch;
    int width;
    const cx<T>* tw;            // W_ny^k   (forward stage twiddles)
    int logTw;
    const cx<T>* twc;           // W_My^k   (inverse inter-pass twiddle)
    long n1f;                   // row stride of the tile: rows g + n1f * n, n < L
    long in_moff, out_moff;     // two maps per launch (grid z = map): offsets of the second map's `in` and leg planes
};

template <typename T, class SEQF, class Ctx>
OA_HD void col_fwdlegs_cg_body(Ctx& ctx, const ColFwdLegsCgArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    constexpr int logL = seq_total_log<SEQF>();
    constexpr int L = 1 << logL;
    constexpr int RL = SEQF::get(SEQF::n - 1), LRL = Log2x<RL>::v, NB = EPT / RL;
    constexpr int Ns = L >> LRL;
    constexpr int LQ = 16;
    static_assert(Ns == LQ && NB == 4, "col_fwdlegs_cg: needs 16 positions per butterfly and 4 butterflies per thread");
    constexpr int logC = COL_LOGC;
    constexpr int NT = (1 << (logL + COL_LOGC)) / EPT;
    static_assert(NT == 4 << COL_LOGC, "col_fwdlegs_cg: a thread owns positions q + 4 u of one column");
    const int tid = ctx.tid();
    const int c0 = ctx.bid_x() << logC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << logC)) ncols = 1 << logC;
    cx<T> gv[EPT];
    cx<T>* twl = s + (1 << (logL + logC));
    cx<T>* ex = twl + tw_lds_size(logL);                  // [4 q][4 a][C] exchange of the 16-point inverse
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    ctx.sync();
    const long imo = ctx.bid_z() ? a.in_moff : 0, omo = ctx.bid_z() ? a.out_moff : 0;
    const ColLoad<T> ld{a.in + imo + g * a.pitch + c0, (unsigned)(a.n1f * a.pitch), ncols, false};
    col_pipeline_to_regs<T, SEQF>(ctx, s, gv, tid, NT, logC, twl, logL, ld);
    const int c = tid & ((1 << logC) - 1), q = tid >> logC;
    const bool ok = c < ncols;
    const T lx = ok ? a.lxd[c0 + c] : (T)0;
    // the forward pass is done ONCE per tile; the three legs follow from its registers (tripling the workgroups so
    // that each redoes the forward pass measured slower: 24.8 us vs 19.1 us for the two launches this kernel replaces)
    for (int leg = 0; leg < 3; ++leg) {                   // 0 = H, 1 = Gx, 2 = Gy
        cx<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = q + 4 * u;                      // kept bin k2' of this butterfly
            const bool low = j < LQ / 2;
            const cx<T> x = low ? gv[u * RL] : gv[u * RL + RL - 1];
            const int k2 = low ? j : (L - LQ) + j;
            const unsigned y = (unsigned)g + (unsigned)k2 * (unsigned)a.n1f;      // full-resolution row of this mode
            cx<T> val = mk<T>((T)0, (T)0);
            if (ok) {
                const unsigned fi = y * (unsigned)a.fpitch + (unsigned)(c0 + c);
                if (leg == 0) val = x * a.FH[fi];
                else if (leg == 1) val = mul_pi(x * a.FG[fi]) * lx;
                else val = mul_pi(x * a.FG[fi]) * a.lyd[y];
            }
            v[u] = swp(val);                              // inverse transform = forward transform of the swapped data
        }
        Dft<T, 4>::run(v);                                // over u: B[a] = sum_u v[q + 4 u] W4^(u a)
#pragma unroll
        for (int aa = 1; aa < 4; ++aa) {                  // W16^(q a), q a in {0,1,2,3,4,6,9}; W16^(8 + k) = -W16^k
            const int e = q * aa;
            const cx<T> w = w16<T>(e & 7);
            v[aa] = v[aa] * (e >= 8 ? mk<T>(-w.x, -w.y) : w);
        }
        if (leg) ctx.sync();                              // the previous leg's exchange reads are complete
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) ex[((q * 4 + aa) << logC) + c] = v[aa];
        ctx.sync();
        cx<T>* out = (leg == 0 ? a.h : (leg == 1 ? a.gx : a.gy)) + omo + (g * LQ) * a.opitch + c0;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            cx<T> t[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) t[qq] = ex[((qq * 4 + aa) << logC) + c];
            Dft<T, 4>::run(t);                            // over q: X[a + 4 b] = sum_q B'[q][a] W4^(q b); this thread keeps b = q
            const int y = aa + 4 * q;
            cx<T> r = (q == 0 ? t[0] : (q == 1 ? t[1] : (q == 2 ? t[2] : t[3])));
            r = swp(r * a.twc[(unsigned)g * (unsigned)y]);    // inter-pass twiddle W_My^(g y_lo), then back from the swapped domain
            if (ok) out[(unsigned)y * (unsigned)a.opitch + (unsigned)c] = r;
        }
    }
}

// ===========================================================================
// (B) divergence * normalisation fused into the forward column pass 2:
//     out = Fn * (i lx FFTcol[A] + i ly FFTcol[B])   (+ out if accumulate)
// ===========================================================================
template <typename T>
struct ColDivArgs {
    const cx<T>* A; const cx<T>* B;
    const T* Fn;
    const T* lxd; const T* lyd;
    cx<T>* out;
    long pitch;      // row pitch of A, B
    long opitch;     // row pitch of Fn and out
    int width, logC, NT;
    const cx<T>* tw;
    int logTw;
    long in_gs, in_ns, out_gs, out_ks;
    int accumulate;
    int rband, ny;   // rband > 0: Fn vanishes on output rows rband <= y <= ny - rband, which are not written
    int yshift;      // COLUMN GRID: output row y of this My-row transform is row y + (y >= ny/2 ? yshift : 0) of Fn, ly, out
    long in_moff, out_moff;   // several maps per launch (grid z = map): map z's A / B planes and its `out` sit z * {in,out}_moff behind
    long fn_moff;             // ... and its Fn plane z * fn_moff behind (0: one Fn for all -- two Monte-Carlo maps; oa_qe_mv: one per estimator)
    // SINGLE-PASS launches with ONE Fn for all maps: TILE-MAJOR copy of Fn on the coarse grid, [tile][coarse row][C] (made when the
    // filters are bound: pipeline.hip) -- read instead of the 16- / 32-byte row segments of the full-pitch plane, of which whole 64- /
    // 128-byte lines travelled (profiles/r04_overfetch.txt: 10.7 / 14.6 MB fetched for 3.5 / 7 MB).  nullptr: the plane itself.
    const T* Fn_t;
};

// LOGC: log2 of the tile width.  The default (32 columns) is the two-pass layout; with a WHOLE column in the tile (SEQ = the
// full column length, 8 or 16 columns, 1024 threads, in_ns = out_ks = 1, one group) the same body is a SINGLE-PASS forward
// column transform + divergence: the product planes are read once, no pass-1 plane is written and read back.
// Tail: what else happens to each kappa value while it is in registers.  NoDivTail: nothing.  DivBinTail (fft_divbin.hpp, GPU
// only): the radial histogram of |kappa|^2 and the moment update of the one-call entries, so that the kappa plane is never
// re-read (and need not be written at all: a.out == nullptr).
struct NoDivTail {
    static constexpr bool active = false;
    template <class Ctx> OA_HD void begin(Ctx&, void*) {}
    OA_HD int id_at(unsigned, int) const { return -1; }
    OA_HD bool packed_ids() const { return false; }
    OA_HD int id_at_t(long) const { return -1; }
    template <class Ctx, typename T> OA_HD void add(Ctx&, int, T, int) {}
    template <class Ctx> OA_HD void finish(Ctx&) {}
};

template <typename T, class SEQ, class Ctx, int LOGC = COL_LOGC, class Tail = NoDivTail>
OA_HD void col_div_body(Ctx& ctx, const ColDivArgs<T>& a, Tail tail = Tail{}) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    constexpr int logL = Log2x<SEQ::r0>::v + Log2x<SEQ::r1>::v + Log2x<SEQ::r2>::v + Log2x<SEQ::r3>::v;
    constexpr int n = SEQ::n;
    constexpr int RL = SEQ::get(n - 1), LRL = Log2x<RL>::v, NB = EPT / RL;
    constexpr int logNs = logL - LRL;
    constexpr int logC = LOGC;
    constexpr int NT = ((1 << (seq_total_log<SEQ>() + LOGC)) / EPT) > 0 ? ((1 << (seq_total_log<SEQ>() + LOGC)) / EPT) : 1;
    const int tid = ctx.tid();
    int tile = ctx.bid_x();
    if ((sizeof(cx<T>) << LOGC) < 128) {
        // 8-column f32 / 4-column f64 tiles read 64-byte row segments: two adjacent tiles share every 128-byte line.  Workgroups are dealt
        // round-robin over the 8 XCDs, so give tiles 2m and 2m+1 to workgroups b and b + 8 of a group of 16: the same XCD,
        // dispatched together -- the second tile's lines are L2 hits instead of a second trip over the fabric.
        const int nt = ctx.grid_x(), base = tile & ~15, r = tile & 15;
        if (base + 16 <= nt) tile = base + 2 * (r & 7) + (r >> 3);
    }
    const int c0 = tile << logC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << logC)) ncols = 1 << logC;
    cx<T> va[EPT], vb[EPT];
    cx<T>* twl = s + (1 << (logL + logC));
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, NT);
    ctx.sync();
    const long zmap = ctx.bid_z();
    const long imo = zmap * a.in_moff, omo = zmap * a.out_moff;
    const ColLoad<T> la{a.A + imo + g * a.in_gs * a.pitch + c0, (unsigned)(a.in_ns * a.pitch), ncols, false};
    const ColLoad<T> lb{a.B + imo + g * a.in_gs * a.pitch + c0, (unsigned)(a.in_ns * a.pitch), ncols, false};
    if constexpr (n == 2) {
        // both tiles' global loads are issued back to back (twice the bytes in flight per workgroup) before either
        // plane goes through LDS
        constexpr int R0 = SEQ::get(0);
        constexpr int l0 = Log2x<R0>::v;
        stage_in<T, R0, false, true>(s, va, tid, NT, logL, logC, 0, 0, twl, logL, la);
        stage_in<T, R0, false, true>(s, vb, tid, NT, logL, logC, 0, 0, twl, logL, lb);
        stage_out<T, R0, false, false>(s, va, tid, NT, logL, logC, 0, 0, NoStore{});
        ctx.sync();
        stage_in<T, RL, false, false>(s, va, tid, NT, logL, logC, 0, l0, twl, logL, NoLoad{});
        ctx.sync();
        stage_out<T, R0, false, false>(s, vb, tid, NT, logL, logC, 0, 0, NoStore{});
        ctx.sync();
        stage_in<T, RL, false, false>(s, vb, tid, NT, logL, logC, 0, l0, twl, logL, NoLoad{});
    } else {
        col_pipeline_to_regs<T, SEQ>(ctx, s, va, tid, NT, logC, twl, logL, la);
        ctx.sync();
        col_pipeline_to_regs<T, SEQ>(ctx, s, vb, tid, NT, logC, twl, logL, lb);
    }
    const long oorg = g * a.out_gs * a.opitch + c0;
    const T* Fnb = a.Fn + zmap * a.fn_moff + oorg;
    cx<T>* outb = a.out + omo + oorg;
    const unsigned ostr = (unsigned)(a.out_ks * a.opitch);
    if constexpr (Tail::active) {
        // every lane takes part in every step of the tail (wave-level reductions): no early exits; the tile is free by now
        ctx.sync();
        tail.begin(ctx, s);
        ctx.sync();
        // phase A, straight-line: every kappa value of this thread, its squared modulus and its bin id (loads batched by the
        // compiler: inside the wave-level steps of phase B each would be a dependent trip to memory)
        T pw[EPT];
        int idv[EPT];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int b = tid + u * NT;
            const int c = b & ((1 << logC) - 1), j = b >> logC;
            const bool cok = c < ncols;
            const int kq = j & ((1 << logNs) - 1);
            const int base = ((j - kq) << LRL) + kq;
            const T lx = cok ? a.lxd[c0 + c] : (T)0;
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                const int k = base + (t << logNs);
                const unsigned y = (unsigned)(g * a.out_gs) + (unsigned)k * (unsigned)a.out_ks;
                const bool ok = cok && !(a.rband && (int)y >= a.rband && (int)y <= a.ny - a.rband);
                const unsigned up = (a.yshift && y >= (unsigned)(a.ny >> 1)) ? (unsigned)a.yshift : 0u;
                const unsigned i = (unsigned)k * ostr + (unsigned)c + up * (unsigned)a.opitch;
                pw[u * RL + t] = (T)0;
                idv[u * RL + t] = -1;
                if (ok) {
                    // (tile-major tables: entry [tile][k][c]; single pass: k is the coarse row)
                    const long ti_ = (((long)tile << logL) + k) * (1 << logC) + c;
                    const T fn = a.Fn_t ? a.Fn_t[ti_] : Fnb[i];
                    cx<T> d = mul_pi(va[u * RL + t] * lx + vb[u * RL + t] * a.lyd[y + up]) * fn;
                    if (a.out) {
                        if (a.accumulate) d = d + outb[i];
                        outb[i] = d;
                    }
                    pw[u * RL + t] = d.x * d.x + d.y * d.y;
                    idv[u * RL + t] = tail.packed_ids() ? tail.id_at_t((((long)tile << logL) + k) * (1 << logC) + c) : tail.id_at(y + up, c0 + c);
                }
            }
        }
        // phase B: wave-level accumulation, every lane in every step
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int c = (tid + u * NT) & ((1 << logC) - 1);
#pragma unroll
            for (int t = 0; t < RL; ++t) tail.add(ctx, idv[u * RL + t], pw[u * RL + t], c0 + c);
        }
        tail.finish(ctx);
    } else {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & ((1 << logC) - 1), j = b >> logC;
        if (c >= ncols) continue;
        const int kq = j & ((1 << logNs) - 1);
        const int base = ((j - kq) << LRL) + kq;
        const T lx = a.lxd[c0 + c];
#pragma unroll
        for (int t = 0; t < RL; ++t) {
            const int k = base + (t << logNs);
            const unsigned y = (unsigned)(g * a.out_gs) + (unsigned)k * (unsigned)a.out_ks;
            if (a.rband && (int)y >= a.rband && (int)y <= a.ny - a.rband) continue;
            const unsigned up = (a.yshift && y >= (unsigned)(a.ny >> 1)) ? (unsigned)a.yshift : 0u;
            const unsigned i = (unsigned)k * ostr + (unsigned)c + up * (unsigned)a.opitch;
            cx<T> d = mul_pi(va[u * RL + t] * lx + vb[u * RL + t] * a.lyd[y + up]) * Fnb[i];
            if (a.accumulate) d = d + outb[i];
            outb[i] = d;
        }
    }
    }
}

// ===========================================================================
// Inverse column pass 1 of the Fourier-space DERIVATIVES of a map (flat-sky Taylor lensing, lensing.py:395-440): the factor
// (i lx)^a (i ly)^b is applied at the load, so the derivative spectra never exist in HBM.  grid z = map * nd + plane; plane
// idx(a, b) = n (n + 1) / 2 - 1 + b, n = a + b = 1 .. order - 1 (the order of hc_derivs_kernel / lens_taylor_kernel).  Same
// four-step layout as col_fft_body's pass 1 (the caller runs the in-place pass 2 over all planes in one launch).
// ===========================================================================
template <typename T>
struct ColDerivArgs {
    const cx<T>* in;            // nmaps source transforms, in_mstride elements apart, row pitch `pitch`
    cx<T>* out;                 // the planes of THIS launch (grid z of them), out_pstride elements apart, row pitch `pitch`
    long in_mstride, out_pstride, pitch;
    int width, nd, logL;
    int zbase;                  // first plane of this launch (chunked launches: plane index = zbase + grid z)
    int bonly;                  // != 0: plane d of a map carries (i ly)^d only (d < nd = order): the x-derivatives ride on the row pass
    const cx<T>* tw;            // W_ny^k
    int logTw;
    long in_ns, out_gs;         // pass-1 strides (rows): point n of group g is row g + n in_ns; bin k goes to row g out_gs + k
    const T* lxd; const T* lyd;
};

// BONLY (what oa_lens_maps runs: the x-derivatives ride on the row pass): the factor is i^b ly^b -- one LDS read and two multiplies per
// point, the power of i an exact swap / sign (same values as the general form, whose products with cr, ci in {0, +-1} are exact).
// BRANCH-FREE: the column index is clamped and the value zeroed by a select, the power of i is a select + signed factor -- with
// the load inside an `if (c < ncols)` block next to its arithmetic the compiler waited for every load before issuing the next
// (16 round trips per thread: 86 us per 4096^2 float64 plane against 59 us for the plain pass 1)
template <typename T, bool BONLY = false>
struct ColDerivLoad {
    static constexpr bool reads_lds = false;    // (the factor tables sit behind the tile, filled and synced before the pipeline: no hazard with its writes)
    const cx<T>* base;
    unsigned nstride;
    int ncols;
    const T* fx;                // LDS: lx^a of the tile's columns
    const T* fy;                // LDS: ly^b of the tile's rows (point n of this group)
    T cr, ci;                   // i^(a + b)
    int q;                      // (a + b) & 3
    template <typename U> OA_HD cx<U> get(int n, int c) const {
        const int cc = c < ncols ? c : ncols - 1;
        const cx<U> v = base[(unsigned)n * nstride + (unsigned)cc];
        cx<U> x;
        if constexpr (BONLY) {
            const U f = c < ncols ? fy[n] : (U)0;
            const bool odd = (q & 1) != 0;
            const U sr = (q == 1 || q == 2) ? -f : f, si = (q >= 2) ? -f : f;     // i^q (x + i y): (x, y), (-y, x), (-x, -y), (y, -x)
            x = mk<U>((odd ? v.y : v.x) * sr, (odd ? v.x : v.y) * si);
        } else {
            const U f = c < ncols ? fx[cc] * fy[n] : (U)0;
            x = mk<U>((v.x * cr - v.y * ci) * f, (v.x * ci + v.y * cr) * f);
        }
        return swp(x);          // inverse transform = forward transform of the swapped data
    }
};

template <typename T, class SEQ, class Ctx>
OA_HD void col_deriv_body(Ctx& ctx, const ColDerivArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid();
    constexpr int CLC = COL_LOGC;
    constexpr int logL = seq_total_log<SEQ>();
    constexpr int CNT = ((1 << (logL + COL_LOGC)) / EPT) > 0 ? ((1 << (logL + COL_LOGC)) / EPT) : 1;
    const int c0 = ctx.bid_x() << CLC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << CLC)) ncols = 1 << CLC;
    cx<T>* twl = s + (1 << (logL + CLC));
    cx<T>* ti = twl + tw_lds_size(logL);
    T* fy = reinterpret_cast<T*>(ti + (1 << logL));         // [L]: ly^b at the rows of this group's points
    T* fx = fy + (1 << logL);                               // [C]: lx^a of the tile's columns
    const int z = a.zbase + ctx.bid_z(), m = z / a.nd, d = z - m * a.nd;
    // plane d -> (n, b): n (n + 1) / 2 - 1 <= d < (n + 1)(n + 2) / 2 - 1
    int n = 1;
    while ((n + 1) * (n + 2) / 2 - 1 <= d) ++n;
    int b = d - (n * (n + 1) / 2 - 1), aa = n - b;
    if (a.bonly) { b = d; aa = 0; n = d; }
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, CNT);
    for (int i = tid; i < (1 << logL); i += CNT) {
        ti[i] = a.tw[(unsigned)g * (unsigned)i];
        const T ly = a.lyd[(unsigned)g + (unsigned)i * (unsigned)a.in_ns];
        T f = (T)1;
        for (int k = 0; k < b; ++k) f *= ly;
        fy[i] = f;
    }
    for (int i = tid; i < (1 << CLC); i += CNT) {
        const T lx = (c0 + i < a.width) ? a.lxd[c0 + i] : (T)0;
        T f = (T)1;
        for (int k = 0; k < aa; ++k) f *= lx;
        fx[i] = f;
    }
    ctx.sync();
    const int q = n & 3;                                     // i^n
    const T cr = (T)((q == 0) - (q == 2)), ci = (T)((q == 1) - (q == 3));
    const ColStore<T> st{a.out + (long)ctx.bid_z() * a.out_pstride + g * a.out_gs * a.pitch + c0, (unsigned)a.pitch, ncols, true, ti, (unsigned)g, (T)1,
                         0, 0, 0, 0};
    if (a.bonly) {
        const ColDerivLoad<T, true> ld{a.in + (long)m * a.in_mstride + g * a.pitch + c0, (unsigned)(a.in_ns * a.pitch), ncols, fx, fy, cr, ci, q};
        fft_pipeline<T, false, true, true, SEQ>(ctx, s, tid, CNT, logL, CLC, 0, twl, logL, ld, st);
    } else {
        const ColDerivLoad<T, false> ld{a.in + (long)m * a.in_mstride + g * a.pitch + c0, (unsigned)(a.in_ns * a.pitch), ncols, fx, fy, cr, ci, q};
        fft_pipeline<T, false, true, true, SEQ>(ctx, s, tid, CNT, logL, CLC, 0, twl, logL, ld, st);
    }
}

template <typename T, class SEQ, class Ctx>
OA_HD void col_fft_body(Ctx& ctx, const ColArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid();
    constexpr int CLC = COL_LOGC;
    constexpr int CNT = ((1 << (seq_total_log<SEQ>() + COL_LOGC)) / EPT) > 0 ? ((1 << (seq_total_log<SEQ>() + COL_LOGC)) / EPT) : 1;
    const int c0 = ctx.bid_x() << CLC;
    const long g = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > (1 << CLC)) ncols = 1 << CLC;
    constexpr int logL = seq_total_log<SEQ>();
    cx<T>* twl = s + (1 << (logL + CLC));
    cx<T>* ti = twl + tw_lds_size(logL);
    tw_lds_fill<T>(ctx, twl, a.tw, a.logTw, logL, CNT);
    if (a.twiddle)
        for (int i = tid; i < (1 << logL); i += CNT) ti[i] = a.tw[(unsigned)g * (unsigned)i];
    ctx.sync();
    int z = ctx.bid_z();
    long imo = 0, omo = 0;
    while (a.nbz && z >= a.nbz) { z -= a.nbz; imo += a.in_moff; omo += a.out_moff; }    // map index = z / nbz (2 maps: one step)
    const cx<T>* in = a.in + imo + (long)(z & 1) * a.in_off1 + (long)(z >> 1) * a.in_off2;
    cx<T>* out = a.out + omo + (long)(z & 1) * a.out_off1 + (long)(z >> 1) * a.out_off2;
    const ColLoad<T> ld{in + g * a.in_gs * a.in_pitch + c0, (unsigned)(a.in_ns * a.in_pitch), ncols, a.inverse != 0};
    const ColStore<T> st{out + g * a.out_gs * a.out_pitch + c0, (unsigned)(a.out_ks * a.out_pitch), ncols, a.inverse != 0,
                         a.twiddle ? ti : nullptr, (unsigned)g, a.scale, a.rband, (int)(g * a.out_gs), (int)a.out_ks, a.ny};
    fft_pipeline<T, false, true, true, SEQ>(ctx, s, tid, CNT, logL, CLC, 0, twl, logL, ld, st);
}

}  // namespace oa
