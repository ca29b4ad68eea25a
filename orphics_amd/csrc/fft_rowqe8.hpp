// Fused QE row stage on the alias-free row grid, EIGHT POINTS PER THREAD (round 5).
//
// Same contract as row_qe_pair_body (fft_kernels.hpp): per row pair three inverse and two forward complex transforms of length M
// (two real rows per transform), the real-space products in registers; reference contract /root/reference/orphics/lensing.py:973-976
// (kappa_from_map: what the estimator object returns for a pair of filtered maps).
//
// What bounds this stage (profiles/r05b_pmc_rowqe.txt): every kernel of the family -- 16 or 8 points per thread, either precision --
// spends the same ~5.7 ns per vector instruction per SIMD: the SIMDs are issue-bound, each wave-instruction (vector, LDS, scalar)
// holds its SIMD's issue for a quad of cycles.  What shortens the stage is therefore FEWER INSTRUCTIONS PER ROW PAIR:
//   * row grids of 3 x 512 points: M >= 2 w_leg + w_kappa is all the alias-free argument needs (include/orphics_amd.h, ROW GRID), and
//     1536 covers the reference's TT band limits at 8192^2 (2 x 380 + 664 = 1424) where the power-of-two grid is 2048: a quarter
//     of every transform's points, LDS entries and instructions gone;
//   * M = A x 512: A waves per row pair (A = 2, 3, 4, 8), 8 points per thread, ONE cross-wave stage per transform (radix A over the
//     waves' 512-point blocks); a wave's 512-point sub-transform is three radix-8 stages whose two exchanges stay inside that wave's
//     own region of LDS -- no s_barrier, the wave only waits for its own LDS operations (Ctx::wsync);
//   * inverse = decimation in frequency (natural in, digit-reversed out), forward = decimation in time (digit-reversed in, natural
//     out): the product is elementwise, so both run in place and the real-space samples never leave the registers;
//   * every exchange position is  base(lane) + constant(register):  padded layouts (72 entries per row of 64, rows of 8 at stride
//     9) that are conflict-free for ds_read/write_b64 and _b128 in both directions (tools/lds_swizzle_search.py), so an LDS access
//     costs no address arithmetic -- three base registers per thread serve all twelve access patterns;
//   * stage factors are per-thread constants (the same thread sees the same (lane, register) -> exponent map in all five
//     transforms of every row pair): float32 keeps all of them in registers (14 + <= 7), float64 keeps W^1, W^2, W^4 and forms
//     the rest with one product each, and reads the 64-point stage's factors from a 56-entry LDS table;
//   * 576 A complex entries of LDS per row pair (float64, M = 2048: 36.9 KB -- four workgroups, sixteen waves per CU).
//
// Index maps (S = 512, wave k0 < A, lane l = 8 c1 + lo, register c < 8):
//   inverse  x[n] -> X[k]:  n = j + S t;  stage A (cross-wave): y[k0][j] = W_M^(j k0) sum_t x[j + S t] W_A^(t k0)  -> region k0, entry j
//            wave k0: j = l + 64 t1 | radix 8 over t1, x W_512^(l c) | exchange E1 | radix 8 over t2 (j = lo' + 8 t2), x W_64^(lo' c)
//            | exchange E2 | radix 8 over lo'  ->  register c3 of lane (c1, lo = c2) holds X[k0 + A (c1 + 8 c2 + 64 c3)]
//   forward  the mirror image: radix 8 over c3 | E2^T | x W_64^(c2 q3), radix 8 over c2 | E1^T | x W_512^(c1 l), radix 8 over c1
//            -> U_k0[l + 64 q1] in region k0; cross-wave: P[q + S r] = sum_k0 W_A^(k0 r) W_M^(k0 q) U_k0[q]
//   E1: element (c, l) of a wave's region at 72 c + l;  E2: element (c1, c2, l0) at 72 c1 + 9 c2 + l0
#pragma once
#include <vector>
#include "fft_kernels.hpp"
#ifndef RQ8_STAMP
#define RQ8_STAMP(i)          // (tools/probes/rq8_probe.hip records s_memtime at the phase boundaries)
#endif

namespace oa {

template <int A_> struct Rq8Geom {
    static constexpr int A = A_, NT = 64 * A, S = 512, RS = 576, M = A * S, JPT = (S + NT - 1) / NT;
    static constexpr bool FULLJ = JPT * NT == S;            // A = 3: the third position of a thread exists for tid < 128 only
    // A = 16 (8192-point grids): TWO threads per cross-stage position j = tid mod 512 -- thread half p = tid / 512 takes the outputs
    // k = 2 k0 + p of the radix-16 butterfly (k even: DFT_8 of x_t + x_(t+8); k odd: DFT_8 of (x_t - x_(t+8)) W_16^t)
    static constexpr bool HALVES = A == 16;
    static constexpr int AR = HALVES ? 8 : A;               // radix of a thread's cross-stage butterfly
    static constexpr int NTW = HALVES ? 8 : A - 1;          // cross-stage factors per position of a thread
    static_assert(A == 2 || A == 3 || A == 4 || A == 8 || A == 16, "rowqe8: row grids of 1024, 1536, 2048, 4096 or 8192 points");
};
// the grids a plan carries constants for (slot order).  (3072 = 6 x 512 -- the wide band's 2 x 1138 + 665 points -- was built and
// measured, radix-6 cross stage, two 6-wave workgroups per CU: 169 us float64 and 92 us float32 against 169 / 77 us on the 4096-point
// grid, whose eight waves fill the SIMDs evenly; not kept.  profiles/r05_wideband.txt)
constexpr int RQ8_NGRIDS = 5;
constexpr int RQ8_WAVES[RQ8_NGRIDS] = {2, 3, 4, 8, 16};
OA_HD int rq8_slot(int m) {
    for (int i = 0; i < RQ8_NGRIDS; ++i) if (m == 512 * RQ8_WAVES[i]) return i;
    return -1;
}

// LDS entries behind the A regions: float64 keeps the 64-point stage's factors W_64^(lo c), [c - 1][lo], there
constexpr int RQ8_TAB = 7 * 8;
template <typename T> constexpr bool rq8_tw_in_regs() { return sizeof(T) == 4; }

// in-register DFT over the waves' blocks
template <typename T, int A> OA_HD void rq8_dft(cx<T>* z) {
    if constexpr (A == 3) {
        // y0 = x0 + x1 + x2,  y1 = x0 + w x1 + w^2 x2,  y2 = x0 + w^2 x1 + w x2,  w = exp(-2 pi i / 3) = -1/2 - i sqrt(3)/2
        const T hs = (T)0.86602540378443864676L;
        const cx<T> s = z[1] + z[2], d = (z[1] - z[2]) * hs;
        const cx<T> m = z[0] - s * (T)0.5;
        z[0] = z[0] + s;
        z[1] = add_mi(m, d);                                // m - i d
        z[2] = add_pi(m, d);                                // m + i d
    } else {
        Dft<T, A>::run(z);
    }
}

// ---- per-thread stage factors ------------------------------------------------------------------------------------------------------
// FULL (float32): W_512^(l c), W_64^(lo c), c = 1 .. 7, and W_M^(j_u k), k = 1 .. A - 1, in registers.
// COMPACT (float64, 128 registers for four waves per SIMD): W_512^(l c) for c = 1, 2, 4 in registers, the other powers by one product
// each; W_64^(lo c) from a 56-entry LDS table; W_M^(j_u) re-read from the (cache-resident) global table next to the transform's own
// loads, its powers by products.
template <typename T, int A, bool FULL>
struct Rq8Tw {
    static constexpr int JPT = Rq8Geom<A>::JPT, NT = Rq8Geom<A>::NT;
    cx<T> r1[FULL ? 7 : 3], r2[FULL ? 7 : 1];
    const cx<T>* s2;                                        // COMPACT: LDS table + lo
    cx<T> ra[FULL ? JPT : 1][FULL ? Rq8Geom<A>::NTW : 1];
    const cx<T>* ga;                                        // COMPACT: this thread's W_M^(j_0) in the constants table
    int gs;                                                 // ... and the distance to W_M^(j_(u+1)) (A = 16: one table row)
    // v[c] *= W_512^(l c)
    OA_HD void mul1(cx<T>* v) const {
        if constexpr (FULL) {
#pragma unroll
            for (int c = 1; c < 8; ++c) v[c] = v[c] * r1[c - 1];
        } else {
            const cx<T> w1 = r1[0], w2 = r1[1], w4 = r1[2], w3 = w1 * w2;
            v[1] = v[1] * w1; v[2] = v[2] * w2; v[3] = v[3] * w3; v[4] = v[4] * w4;
            v[5] = v[5] * (w4 * w1); v[6] = v[6] * (w4 * w2); v[7] = v[7] * (w4 * w3);
        }
    }
    // v[c] *= W_64^(lo c)
    OA_HD void mul2(cx<T>* v) const {
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            if constexpr (FULL) v[c] = v[c] * r2[c - 1];
            else v[c] = v[c] * s2[8 * (c - 1)];
        }
    }
    // z[k] *= W_M^(j_u k), k = 1 .. A - 1   (A = 16, thread half p: z[k0] *= W_M^(j (2 k0 + p)), k0 = 0 .. 7)
    OA_HD void cross(cx<T>* z, int u) const {
        constexpr int NTW = Rq8Geom<A>::NTW;
        if constexpr (FULL) {
#pragma unroll
            for (int k = 0; k < NTW; ++k) z[k + (A == 16 ? 0 : 1)] = z[k + (A == 16 ? 0 : 1)] * ra[u][k];
        } else if constexpr (A == 16) {
            cx<T> c = ldg(ga);                              // W_M^(j p)
            const cx<T> st = ldg(ga + 8 * gs);              // W_M^(2 j)
#pragma unroll
            for (int k = 0; k < 8; ++k) { z[k] = z[k] * c; if (k < 7) c = c * st; }
        } else {
            cx<T> w[A];
            w[1] = ldg(ga + u * gs);
#pragma unroll
            for (int k = 2; k < A; ++k) w[k] = w[k / 2] * w[k - k / 2];
#pragma unroll
            for (int k = 1; k < A; ++k) z[k] = z[k] * w[k];
        }
    }
};

// Per-thread constants of the grid M = 512 A in the order the threads read them (consecutive lanes -> consecutive entries: the
// per-thread gathers from the W_nx table were 21 scattered 8-byte reads per thread, a quarter of the float32 kernel's lifetime):
//   [c - 1][l]            W_512^(l c),  c = 1 .. 7, l < 64
//   [c - 1][lo]           W_64^(lo c),  c = 1 .. 7, lo < 8
//   [u][k - 1][tid]       W_M^(j k),    j = (tid + NT u) mod 512, k = 1 .. A - 1
//   A = 16:  [k0][tid]    W_M^(j (2 k0 + p)), k0 < 8, p = tid / 512, j = tid mod 512;  then  [8][tid] = W_M^(2 j)
// Built on the host (plan creation / the emulator's Holder), one table per grid.
template <typename T>
inline std::vector<cx<T>> rq8_make_consts(int A) {
    const int NT = 64 * A, M = 512 * A, JPT = (512 + NT - 1) / NT;
    std::vector<cx<T>> t((size_t)(7 * 64 + 7 * 8 + (A == 16 ? 9 : JPT * (A - 1)) * NT));
    const long double tau = 6.283185307179586476925286766559005768L;
    auto w = [&](long e, long n) { const long double x = tau * (long double)(e % n) / (long double)n; return mk<T>((T)cosl(x), (T)(-sinl(x))); };
    size_t o = 0;
    for (int c = 1; c < 8; ++c) for (int l = 0; l < 64; ++l) t[o++] = w((long)l * c, 512);
    for (int c = 1; c < 8; ++c) for (int lo = 0; lo < 8; ++lo) t[o++] = w((long)lo * c, 64);
    if (A == 16) {
        for (int k0 = 0; k0 < 8; ++k0) for (int tid = 0; tid < NT; ++tid) t[o++] = w((long)(tid & 511) * (2 * k0 + (tid >> 9)), M);
        for (int tid = 0; tid < NT; ++tid) t[o++] = w((long)(tid & 511) * 2, M);
        return t;
    }
    for (int u = 0; u < JPT; ++u) for (int k = 1; k < A; ++k) for (int tid = 0; tid < NT; ++tid) t[o++] = w((long)((tid + NT * u) & 511) * k, M);
    return t;
}

// tc: the table above
template <typename T, int A, bool FULL, class Ctx>
OA_HD void rq8_tw_init(Ctx& ctx, Rq8Tw<T, A, FULL>& tw, cx<T>* tab, const cx<T>* tc, int tid) {
    using G = Rq8Geom<A>;
    const int l = tid & 63, lo = l & 7;
    const cx<T>* t2 = tc + 7 * 64;
    const cx<T>* ta = t2 + 7 * 8;
    if constexpr (FULL) {
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            tw.r1[c - 1] = tc[64 * (c - 1) + l];
            tw.r2[c - 1] = t2[8 * (c - 1) + lo];
        }
        tw.s2 = nullptr; tw.ga = nullptr; tw.gs = 0;
#pragma unroll
        for (int u = 0; u < G::JPT; ++u)
#pragma unroll
            for (int k = 0; k < G::NTW; ++k) tw.ra[u][k] = ta[(u * G::NTW + k) * G::NT + tid];
    } else {
        tw.r1[0] = tc[l];
        tw.r1[1] = tc[64 + l];
        tw.r1[2] = tc[3 * 64 + l];
        for (int e = tid; e < RQ8_TAB; e += G::NT) tab[e] = t2[e];
        tw.s2 = tab + lo;
        tw.ga = ta + tid;                                   // W_M^(j_u): entry (u, k = 1); A = 16: entry k0 = 0, the step W_M^(2 j) 8 rows on
        tw.gs = A == 16 ? G::NT : (A - 1) * G::NT;
    }
}

// ---- the 512-point sub-transform of one wave, in place in its region Dk --------------------------------------------------------
// three bases per thread: bn = l (natural positions and E1 element (c, l): + 72 c), b1 = 72 c1 + lo (E1 element (c1, lo + 8 t): + 8 t;
// E2 element (c1, c, lo): + 9 c), b2 = 72 c1 + 9 lo (E2 element (c1, lo, t): + t)
// decimation in frequency: Dk[j] natural (caller synced) -> v[c3] = Z[c1 + 8 lo + 64 c3]
template <typename T, class TW, class Ctx>
OA_HD void rq8_sub_dif(Ctx& ctx, cx<T>* Dk, int l, const TW& tw, cx<T>* v) {
    cx<T>* const Bn = Dk + l;
    cx<T>* const B1 = Dk + 72 * (l >> 3) + (l & 7);
    cx<T>* const B2 = Dk + 72 * (l >> 3) + 9 * (l & 7);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = Bn[64 * t];
    Dft<T, 8>::run(v);
    tw.mul1(v);
    ctx.wsync();                                            // every lane's reads precede the in-place writes
#pragma unroll
    for (int c = 0; c < 8; ++c) Bn[72 * c] = v[c];          // E1: element (c, l)
    ctx.wsync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = B1[8 * t];           // element (c1, lo + 8 t)
    Dft<T, 8>::run(v);
    tw.mul2(v);
    ctx.wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) B1[9 * c] = v[c];           // E2: element (c1, c2 = c, l0 = lo)
    ctx.wsync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = B2[t];               // element (c1, c2 = lo, l0 = t)
    Dft<T, 8>::run(v);
}
// decimation in time: v[c3] = p[c1 + 8 lo + 64 c3] -> Dk[q] = U[q] natural (NOT synced on exit)
template <typename T, class TW, class Ctx>
OA_HD void rq8_sub_dit(Ctx& ctx, cx<T>* Dk, int l, const TW& tw, cx<T>* v) {
    cx<T>* const Bn = Dk + l;
    cx<T>* const B1 = Dk + 72 * (l >> 3) + (l & 7);
    cx<T>* const B2 = Dk + 72 * (l >> 3) + 9 * (l & 7);
    Dft<T, 8>::run(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) B2[q] = v[q];               // E2^T: element (c1, c2 = lo, q3 = q)
    ctx.wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = B1[9 * c];           // element (c1, c2 = c, q3 = lo)
    tw.mul2(v);
    Dft<T, 8>::run(v);
    ctx.wsync();
#pragma unroll
    for (int q = 0; q < 8; ++q) B1[8 * q] = v[q];           // E1^T: element (c1, lo + 8 q)
    ctx.wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = Bn[72 * c];          // element (c, l)
    tw.mul1(v);
    Dft<T, 8>::run(v);
    ctx.wsync();
#pragma unroll
    for (int q = 0; q < 8; ++q) Bn[64 * q] = v[q];
}

// ---- inverse transform of a row pair's packed spectrum, pruned input: Z[n] = X0[n] + i X1[n] (n < win), Z[M - n] = conj X0[n] + i conj X1[n],
// zero elsewhere (2 win <= M).  NZ = live taps per side of the cross-wave butterfly (win <= 512 NZ).  The inverse runs as the forward
// transform of the swapped data: v = swapped result, (x1, x0).
// SEQ: one position at a time, the barrier that frees D first (float64: 16 to 32 operands of 16 bytes per position do not fit next to
// h and the butterfly in 128 registers); otherwise every load of the thread is in flight before that barrier.
// (Measured and dropped, tools/probes/rq8_probe.hip + profiles/r05_rowqe_probe.txt: the gradient legs' rows requested a transform ahead by
// LDS-DMA into a staging buffer, two row pairs of an R-layout group per workgroup sharing it -- the load phases shrink, every other
// phase grows by as much: 26.3 vs 24.7 us float32, 43.7 vs 36.2 us float64.)
template <typename T, int A, int NZ, int LAY, bool SEQ, class TW, class Ctx>
OA_HD void rq8_inverse(Ctx& ctx, cx<T>* D, cx<T>* v, int tid, const TW& tw, const cx<T>* row0, const cx<T>* row1, int win,
                       long pitch = 0, T sg = (T)1, int p = 0, int sb = 0) {
    using G = Rq8Geom<A>;
    constexpr int S = G::S, M = G::M, JPT = G::JPT;
    static_assert(NZ >= 1 && 2 * NZ <= A, "rowqe8: live taps per side");
    cx<T> z[SEQ ? 1 : JPT][A];
    if (SEQ) ctx.sync();                                    // whoever still reads D (previous transform) is done
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
        cx<T>* zu = z[SEQ ? 0 : u];
        const int j = tid + G::NT * u;
        const bool own = G::FULLJ || j < S;
#pragma unroll
        for (int t = 0; t < A; ++t) {
            const int n = j + S * t;
            if (t < NZ) {
                const bool ok = own && n < win;
                cx<T> a0, a1;
                pair_rows_at<T, LAY>(row0, row1, pitch, sg, ok ? n : 0, a0, a1, p);      // unconditional load from a valid address
                zu[t] = ok ? swp(add_pi(a0, a1)) : mk<T>((T)0, (T)0);
            } else if (t >= A - NZ) {
                const int m = M - n;
                const bool ok = own && m < win;
                cx<T> a0, a1;
                pair_rows_at<T, LAY>(row0, row1, pitch, sg, ok ? m : 0, a0, a1, p);
                zu[t] = ok ? mk<T>(a1.x - a0.y, a0.x + a1.y) : mk<T>((T)0, (T)0);        // swp(conj a0 + i conj a1)
            } else {
                zu[t] = mk<T>((T)-0.0, (T)-0.0);                                          // literal zero: folded out of the butterfly
            }
        }
        rq8_dft<T, A>(zu);
        tw.cross(zu, u);
        if (SEQ) {
            if (own) {
#pragma unroll
                for (int k = 0; k < A; ++k) D[G::RS * k + j] = zu[k];
            }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::: "memory");                  // (keeps the next position's loads behind this one's)
#endif
        }
    }
    RQ8_STAMP(sb);
    if (!SEQ) {
        ctx.sync();                                         // whoever still reads D (previous transform) is done
#pragma unroll
        for (int u = 0; u < JPT; ++u) {
            const int j = tid + G::NT * u;
            if (G::FULLJ || j < S) {
#pragma unroll
                for (int k = 0; k < A; ++k) D[G::RS * k + j] = z[u][k];
            }
        }
    }
    ctx.sync();
    RQ8_STAMP(sb + 1);
}

// ---- forward transform of the product in registers, kept columns k < wout of both rows unpacked and stored:
// P0[k] = (P[k] + conj P[M - k]) / 2,  P1[k] = (P[k] - conj P[M - k]) / 2i
template <typename T, int A, class TW, class Ctx>
OA_HD void rq8_forward(Ctx& ctx, cx<T>* D, cx<T>* v, int tid, const TW& tw, cx<T>* o0, cx<T>* o1, int wout, int accumulate, int sb = 0) {
    using G = Rq8Geom<A>;
    constexpr int S = G::S, M = G::M, JPT = G::JPT;
    rq8_sub_dit<T>(ctx, D + G::RS * (tid >> 6), tid & 63, tw, v);
    RQ8_STAMP(sb);
    ctx.sync();
    cx<T> a[JPT][A];
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
        const int q = (tid + G::NT * u) & (S - 1);          // (A = 3: the unused slot reads a valid entry)
#pragma unroll
        for (int k = 0; k < A; ++k) a[u][k] = D[G::RS * k + q];
        tw.cross(a[u], u);
        rq8_dft<T, A>(a[u]);
    }
    RQ8_STAMP(sb + 1);
    ctx.sync();                                             // all cross-stage reads precede the natural-order writes
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
        const int q = tid + G::NT * u;
        if (G::FULLJ || q < S) {
#pragma unroll
            for (int r = 0; r < A; ++r) {
                const int k = q + S * r;
                if (k < wout || k > M - wout) D[G::RS * r + q] = a[u][r];  // only what the unpack reads: P[k] at region k / 512, entry k % 512
            }
        }
    }
    ctx.sync();
    RQ8_STAMP(sb + 2);
    for (int k = tid; k < wout; k += G::NT) {
        const int km = k ? M - k : 0;
        const cx<T> Pk = D[G::RS * (k >> 9) + (k & 511)];
        const cx<T> Pm = conj(D[G::RS * (km >> 9) + (km & 511)]);
        cx<T> p0 = (Pk + Pm) * (T)0.5;
        cx<T> p1 = mul_mi(Pk - Pm) * (T)0.5;
        if (accumulate) { p0 = p0 + o0[k]; p1 = p1 + o1[k]; }
        o0[k] = p0;
        o1[k] = p1;
    }
    RQ8_STAMP(sb + 3);
}

// ---- A = 16 (8192-point grids: the map's own row length at 8192^2): the radix-16 cross-wave stage on two threads per position ------
// inverse, as rq8_inverse: thread (p, j) loads the 16 taps x[j + 512 tt] (pruned: tt < NZ low, tt >= 16 - NZ high), forms
// u_t = x_t + x_(t+8) (p = 0) or (x_t - x_(t+8)) W_16^t (p = 1), a radix-8 butterfly, the factors W_M^(j (2 k0 + p)), and stores
// region 2 k0 + p.  M == nx is allowed: the Nyquist column n = M / 2 is its own mirror image (taken once, on the high side).
template <typename T, int NZ, int LAY, class TW, class Ctx>
OA_HD void rq8_inverse16(Ctx& ctx, cx<T>* D, int tid, const TW& tw, const cx<T>* row0, const cx<T>* row1, int win, long pitch, T sg, int pp) {
    using G = Rq8Geom<16>;
    constexpr int S = G::S, M = G::M;
    static_assert(NZ >= 1 && NZ <= 8, "rowqe8: live taps per side");
    const int p = tid >> 9, j = tid & 511;
    auto tap = [&](int tt) -> cx<T> {                       // swapped packed operand Z[j + 512 tt], zero outside the band
        const int n = j + S * tt;
        if (tt < NZ) {
            const bool ok = n < win && 2 * n < M;           // (n = M / 2 belongs to the high side)
            cx<T> a0, a1;
            pair_rows_at<T, LAY>(row0, row1, pitch, sg, ok ? n : 0, a0, a1, pp);
            return ok ? swp(add_pi(a0, a1)) : mk<T>((T)0, (T)0);
        } else if (tt >= 16 - NZ) {
            const int m = M - n;
            const bool ok = m < win;
            cx<T> a0, a1;
            pair_rows_at<T, LAY>(row0, row1, pitch, sg, ok ? m : 0, a0, a1, pp);
            return ok ? mk<T>(a1.x - a0.y, a0.x + a1.y) : mk<T>((T)0, (T)0);
        }
        return mk<T>((T)-0.0, (T)-0.0);
    };
    cx<T> z[8];
    ctx.sync();                                             // whoever still reads D (previous transform) is done
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const cx<T> lo = tap(t), hi = tap(t + 8);
        z[t] = p ? (lo - hi) : (lo + hi);
    }
    if (p) {
#pragma unroll
        for (int t = 1; t < 8; ++t) {
            if (t == 4) z[t] = mul_mi(z[t]);                // W_16^4 = -i
            else z[t] = z[t] * w16<T>(t);
        }
    }
    Dft<T, 8>::run(z);
    tw.cross(z, 0);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) D[G::RS * (2 * k0) + G::RS * p + j] = z[k0];
    ctx.sync();
}

// forward: thread (p, q): E_p[r'] = DFT_8 over k0 of W_M^(q (2 k0 + p)) U_(2 k0 + p)[q], times W_16^r' for p = 1;
// P[q + 512 r'] = E_0 + E_1', P[q + 512 (r' + 8)] = E_0 - E_1' are formed by the unpack from the two stored halves
template <typename T, class TW, class Ctx>
OA_HD void rq8_forward16(Ctx& ctx, cx<T>* D, cx<T>* v, int tid, const TW& tw, cx<T>* o0, cx<T>* o1, int wout, int accumulate) {
    using G = Rq8Geom<16>;
    constexpr int S = G::S, M = G::M;
    rq8_sub_dit<T>(ctx, D + G::RS * (tid >> 6), tid & 63, tw, v);
    ctx.sync();
    const int p = tid >> 9, q = tid & 511;
    cx<T> a[8];
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) a[k0] = D[G::RS * (2 * k0) + G::RS * p + q];
    tw.cross(a, 0);
    Dft<T, 8>::run(a);
    if (p) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            if (r == 4) a[r] = mul_mi(a[r]);
            else a[r] = a[r] * w16<T>(r);
        }
    }
    ctx.sync();                                             // all cross-stage reads precede the writes
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = q + S * r;                             // the bins this half contributes to: k and k + M / 2
        if (k < wout || k > M - wout || k + M / 2 < wout || k + M / 2 > M - wout) D[G::RS * (r + 8 * p) + q] = a[r];
    }
    ctx.sync();
    auto bin = [&](int k) -> cx<T> {                        // P[k] from the two halves
        const int r = k >> 9, qq = k & 511;
        const cx<T> e0 = D[G::RS * (r & 7) + qq], e1 = D[G::RS * ((r & 7) + 8) + qq];
        return (r & 8) ? e0 - e1 : e0 + e1;
    };
    for (int k = tid; k < wout; k += G::NT) {
        const cx<T> Pk = bin(k);
        const cx<T> Pm = conj(bin(k ? M - k : 0));
        cx<T> p0 = (Pk + Pm) * (T)0.5;
        cx<T> p1 = mul_mi(Pk - Pm) * (T)0.5;
        if (accumulate) { p0 = p0 + o0[k]; p1 = p1 + o1[k]; }
        o0[k] = p0;
        o1[k] = p1;
    }
}

// ---- which grids this body runs -------------------------------------------------------------------------------------------------
OA_HD bool rq8_is_m3(int m) { return m == 1536; }
OA_HD int rq8_waves(int m) { return rq8_slot(m) >= 0 ? m / 512 : 0; }
// live taps per side of the cross-wave butterfly for `win` active columns: ceil(win / 512) rounded up to a power of two (<= A / 2)
OA_HD int rq8_nz(int m, int win) {
    const int A = rq8_waves(m);
    const int need = (win + 511) / 512;
    int nz = 1;
    while (nz < need && 2 * nz < A) nz <<= 1;
    return nz;
}
OA_HD bool rq8_covers(int m, int win, int wout) {
    const int A = rq8_waves(m);
    if (!A || 2L * win + wout > m || wout > m) return false;
    return A == 3 ? win <= 512 : 2 * win <= m;
}
// the map's OWN row length (no alias argument needed: the products are then formed on the grid the reference forms them on): any band
OA_HD bool rq8_covers_full(int m, int win, int wout) { return rq8_waves(m) == 16 && win <= m / 2 + 1 && wout <= m / 2 + 1; }

// LDS of one workgroup: the transform regions, the float64 factor table, the park area
// float64 at four waves per SIMD has 128 registers for h (32), the transform in flight (32), its factors and the butterfly: the
// compiler spilled half of h to scratch and reloaded it one dword pair at a time behind vmcnt(0) -- each reload also waiting for the
// previous transform's stores (stamps: 8 k of a 3.5 k-cycle phase, twice per row pair).  Where the LDS of four workgroups per CU has
// room (M = 1024 and 1536) the upper half of h is PARKED in LDS instead: 4 entries per thread, [t - 4][tid]
template <typename T, int A, bool CHAIN> constexpr bool rq8_park() { return sizeof(T) == 8 && !CHAIN && (A == 2 || A == 3); }
template <typename T, int A, bool CHAIN> constexpr size_t rq8_lds_bytes() {
    return ((size_t)A * Rq8Geom<A>::RS + (rq8_tw_in_regs<T>() ? 0 : RQ8_TAB) + (rq8_park<T, A, CHAIN>() ? 4 * Rq8Geom<A>::NT : 0)) * sizeof(cx<T>);
}

template <typename T, int A, int NZ, int LAY = 0, bool CHAIN = false, class Ctx>
OA_HD void row_qe8_body(Ctx& ctx, const RowQeArgs<T>& a) {
    using G = Rq8Geom<A>;
    constexpr bool FULL = rq8_tw_in_regs<T>();
    cx<T>* D = reinterpret_cast<cx<T>*>(ctx.smem());
    const int tid = ctx.tid();
    cx<T>* TAB = D + A * G::RS;
    cx<T>* PKB = TAB + (FULL ? 0 : RQ8_TAB);
    long wg = ctx.bid_x();
    long imo = 0, omo = 0, hmo = 0;
    int m = 0;
    while (a.npairs && wg >= a.npairs) { wg -= a.npairs; imo += a.in_moff; omo += a.out_moff; hmo += a.h_moff; ++m; }
    const cx<T>* gxp = a.gx + imo; const cx<T>* gyp = a.gy + imo; const cx<T>* hp = a.h + hmo;
    cx<T>* pxp = a.px + omo; cx<T>* pyp = a.py + omo;
    T scale = a.scale;
    if (a.tab && !CHAIN) {
        const RowQeMap<T> e = a.tab[m];
        gxp = e.gx; gyp = e.gy; hp = e.h; pxp = e.px; pyp = e.py; scale = e.scale;
    }
    // row addressing as row_qe_pair_body: natural pairs, or the R-layouts of col_fband_body (LAY = 1: R = 2, LAY = 2: R = 4, LAY = 3: R = 8)
    long r0 = wg * 2, ra = wg * 2, rb = wg * 2 + 1;
    T sg = (T)1;
    int pp = 0;
    if (LAY == 3) {
        const long blk = wg & ~31L;
        const int r = (int)(wg & 31);
        pp = r >> 3;
        const long ylo = (blk >> 2) + (r & 7), mq = a.nrows >> 3;
        r0 = ylo << 3;
        ra = ylo + mq * (2 * pp);
        rb = ra + mq;
    } else if (LAY == 1) {
        const long mq = a.nrows >> 1;                        // group y_lo = wg: plane rows 2 wg, 2 wg + 1 -> field rows wg, wg + Mq
        ra = wg;
        rb = wg + mq;
    } else if (LAY > 0) {
        // the two workgroups of a group read the same four rows: workgroups b and b + 8 of a block of 16 (same XCD under the round-robin dispatch)
        const long blk = wg & ~15L;
        const int r = (int)(wg & 15), p = r >> 3;
        const long ylo = (blk >> 1) + (r & 7), mq = a.nrows >> LAY;
        sg = p ? (T)-1 : (T)1;
        r0 = ylo << LAY;
        ra = ylo + mq * (2 * p);
        rb = ra + mq;
    }
    RQ8_STAMP(0);
    Rq8Tw<T, A, FULL> tw;
    rq8_tw_init<T, A, FULL>(ctx, tw, TAB, a.rq8c, tid);
    RQ8_STAMP(1);
    cx<T> hreg[8], v[8];
    cx<T>* Dk = D + G::RS * (tid >> 6);
    if constexpr (CHAIN) {
        static_assert(LAY == 0 && A != 16, "chains read natural-order leg planes on grids of up to 4096 points");
        const int first = a.chain[2 * m], count = a.chain[2 * m + 1];
        cx<T> acc[2][8];
#pragma unroll 1
        for (int i = 0; i < count; ++i) {
            const RowQeMap<T> e = a.tab[first + i];
            rq8_inverse<T, A, NZ, 0, !FULL>(ctx, D, hreg, tid, tw, e.h + r0 * a.pitch, e.h + (r0 + 1) * a.pitch, a.win);
            rq8_sub_dif<T>(ctx, Dk, tid & 63, tw, hreg);
#pragma unroll
            for (int t = 0; t < 8; ++t) hreg[t] = hreg[t] * e.scale;
#pragma unroll
            for (int leg = 0; leg < 2; ++leg) {
                const cx<T>* src = leg ? e.gy : e.gx;
                rq8_inverse<T, A, NZ, 0, !FULL>(ctx, D, v, tid, tw, src + r0 * a.pitch, src + (r0 + 1) * a.pitch, a.win);
                rq8_sub_dif<T>(ctx, Dk, tid & 63, tw, v);
                if (i == 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[leg][t] = mk<T>(v[t].y * hreg[t].y, v[t].x * hreg[t].x);
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[leg][t] = acc[leg][t] + mk<T>(v[t].y * hreg[t].y, v[t].x * hreg[t].x);
                }
            }
        }
        const RowQeMap<T> e0 = a.tab[first];
#pragma unroll
        for (int leg = 0; leg < 2; ++leg) {
            cx<T>* dst = leg ? e0.py : e0.px;
            ctx.sync();                                     // the unpack reads of the previous leg precede this leg's in-place exchanges
            rq8_forward<T, A>(ctx, D, acc[leg], tid, tw, dst + ra * a.opitch, dst + rb * a.opitch, a.wout, 0);
        }
        return;
    }
    if constexpr (A == 16) rq8_inverse16<T, NZ, LAY>(ctx, D, tid, tw, hp + r0 * a.pitch, hp + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp);
    else rq8_inverse<T, A, NZ, LAY, !FULL>(ctx, D, hreg, tid, tw, hp + r0 * a.pitch, hp + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp, 2);
    rq8_sub_dif<T>(ctx, Dk, tid & 63, tw, hreg);
    RQ8_STAMP(4);
    // hreg holds the swapped inverse: (h1, h0); the product scale rides on it
#pragma unroll
    for (int t = 0; t < 8; ++t) hreg[t] = hreg[t] * scale;
    constexpr bool PARK = rq8_park<T, A, CHAIN>();
    cx<T>* const PK = PKB + tid;
    if constexpr (PARK) {
#pragma unroll
        for (int t = 4; t < 8; ++t) PK[(t - 4) * G::NT] = hreg[t];
    }
    for (int leg = 0; leg < 2; ++leg) {
        const cx<T>* src = leg ? gyp : gxp;
        cx<T>* dst = leg ? pyp : pxp;
        if constexpr (A == 16) rq8_inverse16<T, NZ, LAY>(ctx, D, tid, tw, src + r0 * a.pitch, src + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp);
        else rq8_inverse<T, A, NZ, LAY, !FULL>(ctx, D, v, tid, tw, src + r0 * a.pitch, src + (r0 + 1) * a.pitch, a.win, a.pitch, sg, pp, 5 + 7 * leg);
        rq8_sub_dif<T>(ctx, Dk, tid & 63, tw, v);
        RQ8_STAMP(7 + 7 * leg);
        // v = (g1, g0) swapped; p = g0 h0 + i g1 h1
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const cx<T> hh = (PARK && t >= 4) ? PK[(t - 4) * G::NT] : hreg[t];
            v[t] = mk<T>(v[t].y * hh.y, v[t].x * hh.x);
        }
        if constexpr (A == 16) rq8_forward16<T>(ctx, D, v, tid, tw, dst + ra * a.opitch, dst + rb * a.opitch, a.wout, a.accumulate);
        else rq8_forward<T, A>(ctx, D, v, tid, tw, dst + ra * a.opitch, dst + rb * a.opitch, a.wout, a.accumulate, 8 + 7 * leg);
    }
}

// host-side dispatch: f(A, NZ, LAY, CHAIN as integral constants) for the instantiated variant; false: not built
template <class F>
inline bool dispatch_rq8(int M, int win, int lr, bool chain, F&& f) {
    const int nz = rq8_nz(M, win);
    using std::integral_constant;
    auto with_lay = [&](auto ac, auto nzc) -> bool {
        if (chain) {
            if (lr != 0) return false;
            f(ac, nzc, integral_constant<int, 0>{}, std::true_type{});
            return true;
        }
        switch (lr) {
            case 0: f(ac, nzc, integral_constant<int, 0>{}, std::false_type{}); return true;
            case 1:                                         // R = 2: the wide band's 4096-point grid only
                if constexpr (decltype(ac)::value == 8) { f(ac, nzc, integral_constant<int, 1>{}, std::false_type{}); return true; }
                else return false;
            case 2: f(ac, nzc, integral_constant<int, 2>{}, std::false_type{}); return true;
            case 3: f(ac, nzc, integral_constant<int, 3>{}, std::false_type{}); return true;
            default: return false;
        }
    };
    switch (M) {
        case 1024: return with_lay(integral_constant<int, 2>{}, integral_constant<int, 1>{});
        case 1536: return with_lay(integral_constant<int, 3>{}, integral_constant<int, 1>{});
        case 2048:
            if (nz == 1) return with_lay(integral_constant<int, 4>{}, integral_constant<int, 1>{});
            return with_lay(integral_constant<int, 4>{}, integral_constant<int, 2>{});
        case 4096:
            if (nz == 1) return with_lay(integral_constant<int, 8>{}, integral_constant<int, 1>{});
            if (nz == 2) return with_lay(integral_constant<int, 8>{}, integral_constant<int, 2>{});
            return with_lay(integral_constant<int, 8>{}, integral_constant<int, 4>{});
        case 8192:                                          // natural layout, no chains: the map's own rows at 8192^2 (fullres / dense)
            if (chain || lr != 0) return false;
            if (nz == 1) f(integral_constant<int, 16>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, std::false_type{});
            else if (nz <= 4) f(integral_constant<int, 16>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, std::false_type{});
            else return false;                              // every column live: the packed full-row kernel (row_qe_body) is faster
            return true;
        default: return false;
    }
}

}  // namespace oa
