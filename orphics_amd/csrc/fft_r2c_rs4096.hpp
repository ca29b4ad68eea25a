// R-split row R2C of 8192-point real rows (4096 packed complex points), band-limited output (<= 512 columns), either precision.
//
// The general R-split pass (row_r2c_rsplit_body, fft_kernels.hpp) is not HBM-bound: with its global loads removed a float64
// 8192^2 map still takes 116 of its 131 us (profiles/r03w_r2c_f64.txt) -- three workgroup-wide LDS exchanges and seven
// barriers per row at two waves per SIMD leave the vector unit 46 % busy.  Here the 4096-point transform is cut
// 4096 = 16 x 256 so that only ONE exchange crosses waves:
//   stage 0 (thread j0 = tid): radix-16 butterfly over x[j0 + 256 t], times W_4096^(j0 k0)  ->  Y[k0][j0], k0 < 16
//   -- workgroup barrier --
//   the 16 sub-transforms (256 points over j0, one per k0) are independent: wave w owns k0 = 4 w + s, s < 4 (its lanes
//   16 s + i), as two radix-16 stages with an exchange INSIDE the wave's own quarter of the buffer -- no s_barrier: the
//   wave only waits for its own LDS operations (Ctx::wsync = s_waitcnt lgkmcnt(0); no measurable cost);
//   the last stage computes only the bins the consumers keep: columns < 512 and their mirror images (the untangle's
//   partners), 4 of its 16 outputs;
//   -- workgroup barrier --   untangle + radix-4 column butterfly accumulation (as the general pass)   -- barrier --
// Three barriers per row instead of seven, 64 + 64 + 16 KB of LDS stores per float64 row instead of 192, no padding
// (every access below is conflict-free by layout or by an XOR swizzle), and the next row's global loads are issued as
// soon as stage 0 has left its registers (prefetch, as row_r2c_rsplit_body<.., PF = true>).
#pragma once
#include "fft_kernels.hpp"

namespace oa {

// Row lengths: LOGL = 12 (8192-point rows, 256 threads = 4 waves per row, sub-transforms of 256 = 16 x 16 points, <= 512
// columns kept), LOGL = 11 (4096-point rows, 128 threads = 2 waves, sub-transforms of 128 = 16 x 8 points: a lane finishes
// two of the 16 eight-point butterflies; <= 256 columns kept) and LOGL = 13 (16384-point rows -- BASELINE config 5 --, 512
// threads = 8 waves, sub-transforms of 512 = 16 x 32 points: a PAIR of lanes finishes one 32-point butterfly, each lane the
// 16-point transform of its even / odd samples, pruned to the 4 kept bins; the two halves meet as a sum of two exchange entries
// at the untangle; <= 512 columns kept, one per thread).  S = L / 16 = threads per row = points per sub-transform.
// Column butterfly on top: R = 4 (LR = 2: two radix-2 levels, three live values per column) or R = 8 (LR = 3, 16384^2 maps on the
// 2048-row column grid: eight running sums per column -- one column per thread there).
// WIDE BAND (RKW = 5, LOGL = 12, R = 2: SURVEY section 8(d)'s T filter up to ell = 6000 -- 1138 kept columns of an 8192^2 map on the
// 4096-row column grid): the last stage keeps 5 of its 16 bins per side (columns < 1280 and their mirror images), a thread
// untangles five columns tid + 256 r, one running value per column (A = X0, then Y0 = A + X1, Y1 = (A - X1) W_ny^g).
constexpr int RS4096_NT = 256;
template <int LOGL> constexpr int rs_nt() { return (1 << LOGL) / 16; }
template <int LOGL> constexpr int rs_ncol() { return LOGL == 13 ? 1 : 2; }          // kept columns per thread
template <int LOGL> constexpr int rs_twn() { return LOGL == 13 ? 192 : 128; }       // entries reserved for the two-level W_L table
#ifndef OA_RS4096_PFH
#define OA_RS4096_PFH 8      // (float64, 8192-point rows: 8 taps right after stage 0, 8 behind the first sub-transform stage -- round 5, profiles/r05_r2c_prefetch.txt)
#endif
#ifndef OA_RS2048_PFH
#define OA_RS2048_PFH 16
#endif
#ifndef OA_RS8192_PFH
#define OA_RS8192_PFH 8      // (float64, 16384-point rows: all 16 taps across the first sub-transform stage spill 14 registers)
#endif
// sub-transform pitch in LDS: S points plus a pad chosen so that the sub-transforms one LDS instruction's lane group spans
// start on different banks (ds_read_b64: 32 lanes, ds_read_b128: 16 lanes; checked case by case in DESIGN section 3).  S = 512: a
// lane group never spans two sub-transforms (32 lanes each): no pad
template <typename T, int LOGL> constexpr int rs_sub() { return LOGL == 13 ? 512 : (LOGL == 12 ? (sizeof(T) == 4 ? 272 : 256) : 136); }
template <typename T> constexpr int rs4096_sub() { return rs_sub<T, 12>(); }
template <typename T, int LOGL, int RKW = 0> constexpr size_t rs_lds_bytes() {
    // data + two-level W_L table + W_S table [m][i] + the untangle factors of the kept columns (wide band: of the first NT, the
    // others are those times W_32^r -- five columns' worth would cost the second workgroup of a CU its LDS)
    return (size_t)(16 * rs_sub<T, LOGL>() + rs_twn<LOGL>() + rs_nt<LOGL>() + (RKW ? 1 : rs_ncol<LOGL>()) * rs_nt<LOGL>()) * sizeof(cx<T>);
}
template <typename T> constexpr size_t rs4096_lds_bytes() { return rs_lds_bytes<T, 12>(); }

// V[m][i] of a sub-transform (m < 16 first-stage bins, i < LPS lanes): place of lane i inside row m, chosen so that both the
// writes (lanes i, fixed m) and the second-stage reads (lanes m, fixed i; LPS = 32: lanes (m, parity), fixed t: i = 2 t + parity)
// hit distinct banks
template <int LPS> OA_HD int rs_swz(int i, int m) { return LPS == 32 ? ((i ^ (2 * m)) & 31) : (LPS == 16 ? ((i ^ m) & 15) : ((i + (m >> 1)) & 7)); }

// position of kept bin (rr, k0, m) in the exchange area -- rr < 2 RK: the RK lowest and the RK highest second-stage bins (LPS =
// 32: rr = 2 x that + the lane's parity: the two partial sums of a bin) -- inside the owner wave's part of the buffer (NSW
// sub-transforms per wave), the low nibble swizzled by k0 (writes: lanes m consecutive; reads: lanes k0 consecutive)
template <int SUB, int NSW>
OA_HD int rs_epos(int rr, int k0, int m) { return (k0 / NSW) * (NSW * SUB) + rr * (NSW * 16) + (k0 % NSW) * 16 + ((m ^ k0) & 15); }

// W_8^e = exp(-2 pi i e / 8)
template <typename T>
OA_HD cx<T> w8(int e) {
    const T h = (T)0.70710678118654752440L;
    switch (e & 7) {
        case 0: return mk<T>((T)1, (T)0);
        case 1: return mk<T>(h, -h);
        case 2: return mk<T>((T)0, (T)-1);
        case 3: return mk<T>(-h, -h);
        case 4: return mk<T>((T)-1, (T)0);
        case 5: return mk<T>(-h, h);
        case 6: return mk<T>((T)0, (T)1);
        default: return mk<T>(h, h);
    }
}

// W_32^r = exp(-2 pi i r / 32), r < 5
template <typename T>
OA_HD cx<T> w32(int r) {
    switch (r) {
        case 0: return mk<T>((T)1, (T)0);
        case 1: return mk<T>((T)0.98078528040323044913L, (T)-0.19509032201612826785L);
        case 2: return mk<T>((T)0.92387953251128675613L, (T)-0.38268343236508977173L);
        case 3: return mk<T>((T)0.83146961230254523708L, (T)-0.55557023301960222474L);
        default: return mk<T>((T)0.70710678118654752440L, (T)-0.70710678118654752440L);
    }
}

template <typename T, int LOGL, int LR, bool PF = true, int RKW = 0, class Ctx>
OA_HD void row_r2c_rs_body(Ctx& ctx, const RowArgs<T>& a) {
    constexpr int L = 1 << LOGL, R = 1 << LR, NT = rs_nt<LOGL>(), S = NT, SUB = rs_sub<T, LOGL>();
    constexpr int LPS = S / 16, NSW = 64 / LPS, RK = RKW ? RKW : (LOGL >= 12 ? 2 : 1);      // lanes per sub-transform, sub-transforms per wave, kept bins per side
    constexpr int NCOL = RKW ? RKW : rs_ncol<LOGL>();
    static_assert(LOGL == 13 || LOGL == 12 || LOGL == 11, "row_r2c_rs_body: 16384-, 8192- or 4096-point rows");
    static_assert(R == 4 || (R == 8 && NCOL == 1) || (R == 2 && RKW == 5 && LOGL == 12),
                  "column butterfly: R = 4, R = 8 with one column per thread, or R = 2 on the wide band of 8192-point rows");
    static_assert(RKW == 0 || (LPS == 16 && RKW <= 5), "wide band: 16 x 16-point sub-transforms (the exchange area holds 2 RK x 64 entries per wave)");
    cx<T>* D = reinterpret_cast<cx<T>*>(ctx.smem());          // [k0][j0]: 16 x S
    cx<T>* TW = D + 16 * SUB;                                 // two-level W_L table
    cx<T>* TS = TW + rs_twn<LOGL>();                          // [m][i] = W_S^(i m), m < 16, i < LPS
    const int tid = ctx.tid();
    tw_lds_fill<T>(ctx, TW, a.tw, a.logTw, LOGL, NT);
    TS[tid] = a.tw[((unsigned)((tid / LPS) * (tid % LPS)) & (unsigned)(S - 1)) << (a.logTw - (LOGL - 4))];
    ctx.sync();
    const int w = tid >> 6, l = tid & 63, s = l / LPS, k0 = NSW * w + s;
    int i = l % LPS;
    cx<T>* Dk = D + SUB * k0;                                 // this thread's sub-transform
    const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in);
    cx<T>* out = reinterpret_cast<cx<T>*>(a.out);
    const unsigned nym = ((unsigned)a.my << LR) - 1u;
    const int ngroups = a.my, gstep = ctx.grid_x();
    // untangle: this thread's columns kk = tid + NT r, r < NCOL (coalesced stores); factors W_2L^kk in LDS (in registers they cost
    // up to 8 VGPRs across the whole row; loaded from global inside the loop they would queue behind the prefetch)
    cx<T>* TWK = TS + NT;
#pragma unroll
    for (int r = 0; r < (RKW ? 1 : NCOL); ++r) TWK[tid + NT * r] = a.tw[(unsigned)(tid + NT * r) << (a.logTw - (LOGL + 1))];
    cx<T> v[16];
    auto taps = [&](long grp, int n, int t0 = 0, int t1 = 16) {
        const cx<T>* src = in + (grp + (long)n * a.my) * a.in_pitch + tid;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t >= t0 && t < t1) v[t] = RowLoadOnce<T>{src, 0u}.template get<T>(S * t, 0);
    };
    // row of the group taken at step `step`: R = 4: n = 0, 2, 1, 3 (two radix-2 levels, below); R = 8: natural order
    auto row_of = [](int step) { return R == 4 ? (((step & 1) << 1) | (step >> 1)) : step; };
    // prefetch of the next row in two halves (PFH = taps issued right after stage 0; the rest after the first sub-transform
    // stage, whose butterfly + 15 factors are the register peak): float64 has no room for all 16 taps across that stage
    constexpr int PFH = sizeof(T) == 8 ? (LOGL == 13 ? OA_RS8192_PFH : (LOGL == 12 ? OA_RS4096_PFH : OA_RS2048_PFH)) : 16;
    auto next_taps = [&](long grp, int step, int t0, int t1) {
        if (step + 1 < R) taps(grp, row_of(step + 1), t0, t1);
        else if (grp + gstep < ngroups) taps(grp + gstep, 0, t0, t1);
    };
    // kept bin (rr, k0, m) of the exchange area: one entry, or the sum of the two lanes' partial sums (LPS = 32)
    auto kept = [&](int rr, int kk0, int m) {
        if constexpr (LPS == 32) return D[rs_epos<SUB, NSW>(2 * rr, kk0, m)] + D[rs_epos<SUB, NSW>(2 * rr + 1, kk0, m)];
        else return D[rs_epos<SUB, NSW>(rr, kk0, m)];
    };
    long grp = ctx.bid_x();
    if (PF && grp < ngroups) taps(grp, 0);
    for (; grp < ngroups; grp += gstep) {
        // R = 4: rows in the order n = 0, 2, 1, 3 -- two radix-2 levels: A = X0 + X2, B = X0 - X2, then Y0 = A + c, Y2 = A - c,
        // Y1 = B - i d, Y3 = B + i d with c = X1 + X3, d = X1 - X3: three live values per column and no products.
        // R = 8: acc[k1] = sum_n X_n W_8^(n k1), eight running sums of this thread's one column
        cx<T> A[NCOL], B[NCOL], Cc[NCOL];
        cx<T> acc[R == 8 ? 8 : 1];
        const cx<T> wy1 = a.twy[(unsigned)grp & nym];          // W_ny^g (issued before the next prefetch)
#pragma unroll 1
        for (int step = 0; step < R; ++step) {
            const int n = row_of(step);
            if (!PF) taps(grp, n);
            // ---- stage 0: residues k0 = t of the 16-point butterfly over x[tid + S t], twiddled, to D[k0][tid]
            Dft<T, 16>::run(v);
            apply_twiddles<T, 16>(v, TW, tid, 0, tw_lds_h(LOGL));
#pragma unroll
            for (int t = 0; t < 16; ++t) D[SUB * t + tid] = v[t];
            if (PF && PFH > 0) next_taps(grp, step, 0, PFH);
            ctx.sync();
            // ---- sub-transform k0 (S points over j0 = i + LPS t), inside this wave's part of D
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OA_RS4096_NO_OPAQUE)
            // float64: the lane index is made opaque per row, so that the ~40 swizzled LDS addresses below are recomputed (two integer
            // instructions each) instead of being kept in registers across the rows: 220 -> 200 VGPRs, 123 -> 120 us at 8192^2,
            // 36.6 -> 34.3 us at 4096^2.  float: measured equal (8192^2) or slower (4096^2: 19.9 -> 21.3 us): left to the compiler.
            if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(i));
#endif
            cx<T> u[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) u[t] = Dk[i + LPS * t];
            Dft<T, 16>::run(u);                                // u[m] = sum_t Y[i + LPS t] W_16^(t m)
#pragma unroll
            for (int m = 1; m < 16; ++m) {
                if ((m & 3) == 0) ctx.wsync();                 // (keeps the compiler from fetching all 15 factors at once: registers)
                u[m] = u[m] * TS[LPS * m + i];
            }
            ctx.wsync();                                       // every lane's reads precede the in-place writes
#pragma unroll
            for (int m = 0; m < 16; ++m) Dk[LPS * m + rs_swz<LPS>(i, m)] = u[m];
            if (PF && PFH < 16) next_taps(grp, step, PFH, 16);
            ctx.wsync();
            if constexpr (LPS == 32) {
                // lanes (m2, par) of this sub-transform: the 16-point transform of the even (par = 0) / odd (par = 1) samples of row
                // m2, pruned to the bins q = 0, 1, 14, 15; Z[k0 + 16 m2 + 256 r] = E[r mod 16] + W_32^r O[r mod 16], r = 0, 1, 30, 31:
                // the odd lane carries the factor, the two halves are added at the untangle (two exchange entries per bin)
                const int m2 = i >> 1, par = i & 1;
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t] = Dk[32 * m2 + rs_swz<32>(2 * t + par, m2)];
                Dft<T, 16>::run(u);
                const T c1 = (T)0.98078528040323044913L, s1 = (T)0.19509032201612826785L;      // cos, sin(pi / 16)
                const T c2 = (T)0.92387953251128675613L, s2 = (T)0.38268343236508977173L;      // cos, sin(pi / 8)
                if (par) {
                    u[1] = u[1] * mk<T>(c1, -s1);              // W_32^1
                    u[14] = u[14] * mk<T>(c2, s2);             // W_32^30 = conj W_32^2
                    u[15] = u[15] * mk<T>(c1, s1);             // W_32^31 = conj W_32^1
                }
                ctx.wsync();                                   // the reads above precede the exchange writes (same part of D)
                D[rs_epos<SUB, NSW>(0 + par, k0, m2)] = u[0];
                D[rs_epos<SUB, NSW>(2 + par, k0, m2)] = u[1];
                D[rs_epos<SUB, NSW>(4 + par, k0, m2)] = u[14];
                D[rs_epos<SUB, NSW>(6 + par, k0, m2)] = u[15];
            } else if constexpr (LPS == 16) {
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t] = Dk[16 * i + rs_swz<16>(t, i)];   // V[m = i][t]
                Dft<T, 16>::run(u);                            // u[r] = Z[k0 + 16 i + 256 r]; r = 0, 1, 14, 15 used (the rest is dead code)
                ctx.wsync();                                   // the reads above precede the exchange writes (same part of D)
#pragma unroll
                for (int r = 0; r < RK; ++r) {                 // RK = 2: r = 0, 1, 14, 15;  wide band: 0 .. 4 and 11 .. 15
                    D[rs_epos<SUB, NSW>(r, k0, i)] = u[r];
                    D[rs_epos<SUB, NSW>(RK + r, k0, i)] = u[16 - RK + r];
                }
            } else {
                // lane i finishes m = 2 i and 2 i + 1: two 8-point butterflies over the lanes' values V[m][t], t < 8
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 8; ++t) u[8 * h + t] = Dk[8 * (2 * i + h) + rs_swz<8>(t, 2 * i + h)];
                Dft<T, 8>::run(u);                             // u[r] = Z[k0 + 16 (2 i) + 256 r]; r = 0, 7 used
                Dft<T, 8>::run(u + 8);
                ctx.wsync();
                D[rs_epos<SUB, NSW>(0, k0, 2 * i)] = u[0];
                D[rs_epos<SUB, NSW>(1, k0, 2 * i)] = u[7];
                D[rs_epos<SUB, NSW>(0, k0, 2 * i + 1)] = u[8];
                D[rs_epos<SUB, NSW>(1, k0, 2 * i + 1)] = u[15];
            }
            ctx.sync();
            // ---- untangle columns kk = tid + NT r and accumulate the radix-R column butterfly
#pragma unroll
            for (int r = 0; r < NCOL; ++r) {
                const int kk = tid + NT * r;
                if (kk < a.wcols) {
                    const int P = (L - kk) & (L - 1);          // partner bin: 0 (kk = 0) or among the RK highest blocks of 256
                    const cx<T> Zk = kept(kk >> 8, kk & 15, (kk >> 4) & 15);
                    const cx<T> Zm = kept(P ? (P >> 8) - (LPS - 2 * RK) : 0, P & 15, (P >> 4) & 15);
                    const cx<T> E = (Zk + conj(Zm)) * (T)0.5;
                    const cx<T> O = mul_mi(Zk - conj(Zm)) * (T)0.5;
                    const cx<T> wk = RKW ? TWK[tid] * w32<T>(r) : TWK[kk];      // W_2L^kk  (wide band: W_2L^tid W_32^r)
                    const cx<T> X = (E + wk * O) * a.scale;
                    if constexpr (R == 2) {
                        if (step == 0) A[r] = X;
                        else {
                            cx<T>* dst = out + grp * a.out_pitch + kk;
                            dst[0] = A[r] + X;
                            dst[a.kplane] = (A[r] - X) * wy1;
                        }
                    } else if constexpr (R == 8) {
                        if (step == 0) {
#pragma unroll
                            for (int k1 = 0; k1 < 8; ++k1) acc[k1] = X;
                        } else {
#pragma unroll
                            for (int k1 = 1; k1 < 8; ++k1) acc[k1] = acc[k1] + X * w8<T>(n * k1);       // (n uniform: scalar selects)
                            acc[0] = acc[0] + X;
                        }
                        if (step == R - 1) {
                            cx<T>* dst = out + grp * a.out_pitch + kk;
                            cx<T> wp = wy1;
                            dst[0] = acc[0];
#pragma unroll
                            for (int k1 = 1; k1 < 8; ++k1) {
                                dst[(long)k1 * a.kplane] = acc[k1] * wp;
                                wp = wp * wy1;
                            }
                        }
                    } else {
                        if (step == 0) A[r] = X;
                        else if (step == 1) { B[r] = A[r] - X; A[r] = A[r] + X; }
                        else if (step == 2) Cc[r] = X;
                        else {
                            const cx<T> c = Cc[r] + X, d = Cc[r] - X;
                            const cx<T> wy2 = wy1 * wy1;
                            cx<T>* dst = out + grp * a.out_pitch + kk;
                            dst[0] = A[r] + c;
                            dst[a.kplane] = add_mi(B[r], d) * wy1;
                            dst[2 * a.kplane] = (A[r] - c) * wy2;
                            dst[3 * a.kplane] = add_pi(B[r], d) * (wy2 * wy1);
                        }
                    }
                }
            }
            ctx.sync();                                        // the exchange reads precede the next row's stage-0 writes
        }
    }
}
template <typename T, int LR, bool PF = true, class Ctx>
OA_HD void row_r2c_rs4096_body(Ctx& ctx, const RowArgs<T>& a) { row_r2c_rs_body<T, 12, LR, PF>(ctx, a); }

}  // namespace oa
