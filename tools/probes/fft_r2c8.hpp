// R-split row R2C of 8192-point real rows, band-limited output (<= 512 columns), EIGHT POINTS PER THREAD (round 5): the engine of the
// fused row stage (fft_rowqe8.hpp) under the contract of row_r2c_rs_body<T, 12, 2> (fft_r2c_rs4096.hpp) -- reference: the forward
// transform of FourierCalc.power2d / lensing.qest(...).kappa_from_map on a real map (maps.py:1613).
//
// Why: the 16-points-per-thread kernel holds 220 registers and 68 KB of LDS per 256-thread workgroup at float64 -- two workgroups, two
// waves per SIMD -- and cannot prefetch the next row there (16 taps = 64 more registers); its arithmetic + LDS alone take 77 us, its
// loads alone 81 us, together 115 us (profiles/r03x_r2c_variants.txt): at two waves per SIMD the two do not overlap.  Here a 4096-point
// packed row is 8 waves x 512 points, 8 points per thread: 512 threads and 74 KB per workgroup (the same LDS), FOUR waves per SIMD at
// <= 128 registers, and the next row's 8 taps (32 registers) in flight across the whole transform of the current one.
//   stage A   thread j: radix-8 butterfly over z[j + 512 t], x W_4096^(j k0)  -> region k0, entry j            -- barrier --
//   wave k0:  512-point sub-transform (radix 8 x 8 x 8, exchanges inside the wave's own region: rq8_sub_dif), its LAST stage pruned to
//             the two outputs the band keeps: bins k0 + 8 m (m < 64: columns < 512) and their mirror images k0 + 8 (m + 448)
//   exchange: the two bins of a lane go to entries (k0 + 8 m) and ((k0 + 1) mod 8 + 8 m) of the wave's OWN region (no barrier in
//             front; reads by consecutive columns are conflict-free)                                             -- barrier --
//   untangle + radix-4 column butterfly over the group's rows (as row_r2c_rs_body), one column per thread        -- barrier --
#pragma once
#include "fft_rowqe8.hpp"

namespace oa {

template <typename T> constexpr size_t r2c8_lds_bytes() { return rq8_lds_bytes<T, 8, false>(); }

// the sub-transform of rq8_sub_dif with its last radix-8 stage pruned to outputs 0 and 7: lo = Z[c1 + 8 l0], hi = Z[c1 + 8 l0 + 448]
template <typename T, class TW, class Ctx>
OA_HD void r2c8_sub(Ctx& ctx, cx<T>* Dk, int l, const TW& tw, cx<T>& lo, cx<T>& hi) {
    cx<T>* const Bn = Dk + l;
    cx<T>* const B1 = Dk + 72 * (l >> 3) + (l & 7);
    cx<T>* const B2 = Dk + 72 * (l >> 3) + 9 * (l & 7);
    cx<T> v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = Bn[64 * t];
    Dft<T, 8>::run(v);
    tw.mul1(v);
    ctx.wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) Bn[72 * c] = v[c];
    ctx.wsync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = B1[8 * t];
    Dft<T, 8>::run(v);
    tw.mul2(v);
    ctx.wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) B1[9 * c] = v[c];
    ctx.wsync();
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = B2[t];
    // X_0 = sum_t v_t;  X_7 = sum_t v_t W_8^(7 t) = (b0 + i b2) + W_8^7 (b1 + i b3),  b_t = v_t - v_(t+4)
    const cx<T> a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
    const cx<T> b0 = v[0] - v[4], b1 = v[1] - v[5], b2 = v[2] - v[6], b3 = v[3] - v[7];
    lo = (a0 + a2) + (a1 + a3);
    const T h = (T)0.70710678118654752440L;
    const cx<T> e = add_pi(b1, b3);                         // b1 + i b3
    hi = add_pi(b0, b2) + mk<T>((e.x - e.y) * h, (e.x + e.y) * h);      // W_8^7 = (1 + i) / sqrt 2
}

template <typename T, int LR, class Ctx>
OA_HD void row_r2c8_body(Ctx& ctx, const RowArgs<T>& a, const cx<T>* consts) {
    using G = Rq8Geom<8>;
    constexpr bool FULL = rq8_tw_in_regs<T>();
    constexpr int L = 4096, R = 1 << LR, NT = G::NT;
    static_assert(R == 4, "row_r2c8: radix-4 column butterfly (8192^2 maps on the 2048-row column grid, 4096^2 on 1024)");
    cx<T>* D = reinterpret_cast<cx<T>*>(ctx.smem());
    cx<T>* TAB = D + 8 * G::RS;
    const int tid = ctx.tid();
    Rq8Tw<T, 8, FULL> tw;
    rq8_tw_init<T, 8, FULL>(ctx, tw, TAB, consts, tid);
    const int w = tid >> 6, l = tid & 63;
    cx<T>* Dk = D + G::RS * w;
    const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in);
    cx<T>* out = reinterpret_cast<cx<T>*>(a.out);
    const unsigned nym = ((unsigned)a.my << LR) - 1u;
    const int ngroups = a.my, gstep = ctx.grid_x();
    auto row_of = [](int step) { return ((step & 1) << 1) | (step >> 1); };       // rows in the order n = 0, 2, 1, 3 (two radix-2 levels)
    const int kk = tid;                                     // this thread's kept column
    const cx<T>* const twkp = a.tw + ((unsigned)kk << (a.logTw - 13));        // W_8192^kk: the untangle factor (re-read per row: L1-resident)
    // where the untangle finds Z[kk] and Z[L - kk]: low bin (r, m) at entry r + 8 m of region r; high bin (r, m) at entry ((r + 1) & 7) + 8 m
    const int P = (L - kk) & (L - 1);
    const int zk_at = G::RS * (kk & 7) + kk;
    const int zm_at = P ? G::RS * (P & 7) + (((P & 7) + 1) & 7) + 8 * ((P >> 3) - 448) : 0;
    const int m = (l >> 3) + 8 * (l & 7);                   // this lane's kept bins: w + 8 m and w + 8 (m + 448)
    const int lo_at = w + 8 * m, hi_at = ((w + 1) & 7) + 8 * m;
    cx<T> v[8];
    // float32: the next row's 8 taps are requested right after stage A.  float64: 4 there and 4 behind the sub-transform -- all 8 next to
    // the sub-transform's own 8 points and its factors do not fit 128 registers
    constexpr int PFH = sizeof(T) == 8 ? 4 : 8;
    auto taps = [&](long grp, int n, int t0, int t1) {
        const cx<T>* src = in + (grp + (long)n * a.my) * a.in_pitch + tid;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t >= t0 && t < t1) v[t] = RowLoadOnce<T>{src, 0u}.template get<T>(512 * t, 0);
    };
    auto next_taps = [&](long grp, int step, int t0, int t1) {
        if (step + 1 < R) taps(grp, row_of(step + 1), t0, t1);
        else if (grp + gstep < ngroups) taps(grp + gstep, 0, t0, t1);
    };
    long grp = ctx.bid_x();
    if (grp < ngroups) taps(grp, 0, 0, 8);
    ctx.sync();                                             // the LDS factor table (float64) is complete
    for (; grp < ngroups; grp += gstep) {
        cx<T> A = mk<T>((T)0, (T)0), B = A, Cc = A;
#pragma unroll 1
        for (int step = 0; step < R; ++step) {
            // ---- stage A on the taps in v, then the NEXT row's taps are requested: they land across the rest of this row
            Dft<T, 8>::run(v);
            tw.cross(v, 0);
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) D[G::RS * k0 + tid] = v[k0];
            next_taps(grp, step, 0, PFH);
            ctx.sync();
            cx<T> zl, zh;
            r2c8_sub<T>(ctx, Dk, l, tw, zl, zh);
            ctx.wsync();                                    // this wave's last exchange reads precede the writes into its region
            Dk[lo_at] = zl;
            Dk[hi_at] = zh;
            if (PFH < 8) next_taps(grp, step, PFH, 8);
            ctx.sync();
            if (kk < a.wcols) {
                const cx<T> Zk = D[zk_at];
                const cx<T> Zm = D[zm_at];
                const cx<T> E = (Zk + conj(Zm)) * (T)0.5;
                const cx<T> O = mul_mi(Zk - conj(Zm)) * (T)0.5;
                const cx<T> X = (E + ldg(twkp) * O) * a.scale;
                if (step == 0) A = X;
                else if (step == 1) { B = A - X; A = A + X; }
                else if (step == 2) Cc = X;
                else {
                    const cx<T> c = Cc + X, d = Cc - X;
                    const cx<T> wy1 = ldg(a.twy + ((unsigned)grp & nym));      // W_ny^g
                    const cx<T> wy2 = wy1 * wy1;
                    cx<T>* dst = out + grp * a.out_pitch + kk;
                    dst[0] = A + c;
                    dst[a.kplane] = add_mi(B, d) * wy1;
                    dst[2 * a.kplane] = (A - c) * wy2;
                    dst[3 * a.kplane] = add_pi(B, d) * (wy2 * wy1);
                }
            }
            ctx.sync();                                     // the untangle's reads precede the next row's stage-A writes
        }
    }
}

}  // namespace oa
