// Single-pass column stage between the R-split row R2C (row_r2c_rsplit_body / row_r2c_w64r_body) and the fused row stage:
// ONE kernel takes the row pass's output to the three leg planes the row stage reads.
//
// The column transform of length ny = R My (My = the alias-free column grid, include/orphics_amd.h COLUMN GRID) is split
// y = g + My n, k = k1 + R k2.  The row pass already did the radix-R butterfly over n and the twiddle W_ny^(g k1): its
// output plane k1 holds Y[k1][g], g < My, and
//     X[k1 + R k2] = sum_g Y[k1][g] W_My^(g k2)                     (forward, My points, THIS kernel, one k1 per workgroup)
// Only the band |ky| < leg_rows survives the filters; on the My-row grid it sits at k' = k1 + R k2' with
//     k2' = k2 (k2 < Mq / 2),   k2' = k2 - (My - Mq) (k2 >= My - Mq / 2),   Mq = My / R.
// The last forward radix is 2 R, so a thread's butterfly u holds bins j + t Mq / 2 (t < 2 R): exactly one low-band bin
// (t = 0 -> k2' = j) and one high-band bin (t = 2 R - 1 -> k2' = j + Mq / 2) -- every k2' of the coarse spectrum exactly once,
// no zero-fill.  (Round 5: any last radix RL that is a multiple of 2 R -- the RL / 2R lowest and highest outputs of a butterfly are kept;
// the wide band runs its 4096-point forward as 16 x 16 x 16 with 8 of the last stage's 16 outputs.)  The inverse on the My-row grid splits y' = y_lo + Mq y_hi:
//     x[y_lo + Mq y_hi] = sum_k1 W_R^(-k1 y_hi) { W_My^(-k1 y_lo) sum_k2' X'[k1 + R k2'] W_Mq^(-k2' y_lo) }
// The braces -- an Mq-point inverse transform over k2' and a twiddle, for the three filtered legs -- are computed here
// (B[k1][y_lo], stored at row y_lo R + k1: the R-LAYOUT); the radix-R butterfly over k1 is taken by the row stage at its
// loads (pair_rows_at).  What used to be forward pass 1, [forward pass 2 + filters + 16-point inverse pass 1] and inverse
// pass 2 x 3 -- three launches, two round trips of the row pass's output and one of the leg planes -- is one launch that
// reads the row pass's output once and writes the leg planes once.
// LDS: the [My][C] forward tile (128 KB: C = 8 f32 / 4 f64 columns at My = 2048) is reused as R buffers of [Mq][C]: three
// legs + idle groups that only keep the barriers company.
// R = 2 (the wide band: 8192^2 maps on the 4096-row column grid, tiles of 4 f32 / 2 f64 columns): two buffers for three legs -- H and Gx
// are transformed first, Gy waits in registers (two values per butterfly) and takes buffer 0 in a second round; W_My^(k1 y_lo) is 1
// (k1 = 0) or taken from the forward transform's two-level table at the store (k1 = 1: 2048 entries would not fit next to the tile).
#pragma once
#include "fft_kernels.hpp"
#ifndef FB_STAMP
#define FB_STAMP(i)           // (tools/probes/fband_probe.hip records the cycle counter at the phase boundaries)
#endif

namespace oa {

template <typename T>
struct ColFBandArgs {
    const cx<T>* in;            // R planes Y[k1] (kplane elements apart) of My rows, row pitch `pitch`
    long kplane, pitch;
    const T* FG; const T* FH;   // full-resolution filter planes (row pitch fpitch)
    long fpitch;
    // fgh != nullptr: (FG, FH) of every kept bin in the order the threads read them -- col_fband_pack_body, made once per binding:
    // [k1][tile][KQ u + q][tid], dead bins (beyond the band or the width) zero.  From the planes a wave's read of 64 bins touches
    // 32 rows x 16 bytes; the probe (tools/probes/fband_probe.hip) put a fifth of the R = 2 kernel's time there
    const cx<T>* fgh;
    const T* lxd; const T* lyd;
    cx<T>* gx; cx<T>* gy; cx<T>* h;   // My-row leg planes in the R-LAYOUT, row pitch opitch
    long opitch;
    int width;
    const cx<T>* tw;            // W_My^k
    int ny_full, rband;         // rband > 0: the filters vanish on rows rband <= y <= ny_full - rband (not read)
    long in_moff, out_moff;     // several maps per launch (grid z = map)
};

// R = 2: ColStore with the factor W_My^(k1 y_lo) taken from the forward transform's two-level LDS table (a global table read between the
// stores of ColStore::put would wait for each of them in turn)
template <typename T>
struct FBandStore2 {
    cx<T>* base;
    unsigned kstride;
    int ncols;
    const cx<T>* twl;
    int h, k1;
    template <typename U> OA_HD void put(int k, int c, cx<U> v) const {
        if (c >= ncols) return;
        if (k1) v = v * tw_lds(twl, h, k);
        base[(unsigned)k * kstride + (unsigned)c] = swp(v);
    }
};

template <typename T, class SEQF, int LR, int LOGC, class Ctx>
OA_HD void col_fband_body(Ctx& ctx, const ColFBandArgs<T>& a) {
    cx<T>* s = reinterpret_cast<cx<T>*>(ctx.smem());
    constexpr int logL = seq_total_log<SEQF>();
    constexpr int L = 1 << logL, R = 1 << LR, logMq = logL - LR, Mq = 1 << logMq;
    constexpr int RL = SEQF::get(SEQF::n - 1), NB = EPT / RL, Ns = L / RL;
    // a last-stage butterfly holds bins j + t Ns, t < RL: the coarse spectrum keeps the KQ / 2 lowest (t < KQ / 2: k2' = k2) and the KQ / 2
    // highest (t >= RL - KQ / 2: k2' = k2 - (L - Mq)) -- KQ = RL / R; RL = 2 R: one of each (t = 0, t = RL - 1)
    constexpr int KQ = RL / R, NK = KQ * NB;                 // kept bins per butterfly / per thread
    static_assert(RL % (2 * R) == 0 && KQ >= 2, "col_fband: the last forward radix must be a multiple of 2 R");
    static_assert(Ns * KQ == Mq, "col_fband: band layout");
    constexpr int C = 1 << LOGC;
    constexpr int NT = (1 << (logL + LOGC)) / EPT;          // forward: all threads
    constexpr int NTQ = NT / R;                              // inverse: threads per leg buffer
    static_assert(NTQ * EPT == Mq * C, "col_fband: inverse thread groups");
    const int tid = ctx.tid();
    int tile = ctx.bid_x();
    if ((sizeof(cx<T>) << LOGC) < 128) {
        // row segments of 64 (32) bytes: G = 2 (4) neighbouring tiles share every 128-byte line -> give them to workgroups b, b + 8, ..
        // of a group of 8 G (same XCD under the round-robin dispatch, resident together; see col_div_body)
        constexpr int G = 128 / (int)(sizeof(cx<T>) << LOGC), GW = 8 * G;
        const int nt = ctx.grid_x(), base = tile & ~(GW - 1), r = tile & (GW - 1);
        if (base + GW <= nt) tile = base + G * (r & 7) + (r >> 3);
    }
    const int c0 = tile << LOGC;
    const int k1 = ctx.bid_y();
    int ncols = a.width - c0;
    if (ncols > C) ncols = C;
    cx<T> gv[EPT];
    cx<T>* twl = s + (1 << (logL + LOGC));                   // forward stage twiddles (W_My)
    cx<T>* twq = twl + tw_lds_size(logL);                    // inverse stage twiddles (W_Mq)
    cx<T>* ti = twq + tw_lds_size(logMq);                    // W_My^(k1 y_lo), y_lo < Mq
    FB_STAMP(0);
    tw_lds_fill<T>(ctx, twl, a.tw, logL, logL, NT);
    tw_lds_fill<T>(ctx, twq, a.tw, logL, logMq, NT);
    if constexpr (LR != 1)
        for (int i = tid; i < Mq; i += NT) ti[i] = a.tw[((unsigned)k1 * (unsigned)i) & (unsigned)(L - 1)];
    ctx.sync();
    FB_STAMP(1);
    const long zmap = ctx.bid_z();
    // the filter values of this thread's 2 NB kept bins, requested together, ahead of the forward transform, from valid addresses (dead bins: element 0, value unused) --
    // inside per-bin `if` blocks the compiler issued one read, waited, used it, issued the next (tools/isa_loads.py)
    T fgv[NK], fhv[NK], lyv[NK], lxv[NB];
    bool lv[NK];
    const cx<T>* fq = a.fgh ? a.fgh + (((long)k1 * ctx.grid_x() + tile) * NK) * NT + tid : nullptr;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & (C - 1), j = b >> LOGC;            // j < Ns = Mq / 2
        const bool ok = c < ncols;
        lxv[u] = ldg(a.lxd + (ok ? c0 + c : 0));
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int t = q < KQ / 2 ? q : RL - KQ + q;
            const int k2 = j + t * Ns;                       // bin of the My-point forward transform
            const int yf = k1 + R * k2;                      // row of the full-resolution grid
            bool live = ok;
            if (a.rband) live = ok && !(yf >= a.rband && yf <= a.ny_full - a.rband);
            if (fq) {
                const cx<T> f = ldg(fq + (KQ * u + q) * NT);
                fgv[KQ * u + q] = f.x; fhv[KQ * u + q] = f.y;
            } else {
                const long fi = live ? (long)yf * a.fpitch + (c0 + c) : 0;
                fgv[KQ * u + q] = ldg(a.FG + fi);
                fhv[KQ * u + q] = ldg(a.FH + fi);
            }
            lyv[KQ * u + q] = ldg(a.lyd + (live ? yf : 0));
            lv[KQ * u + q] = live;
        }
    }
    FB_STAMP(2);
    const ColLoad<T> ld{a.in + zmap * a.in_moff + (long)k1 * a.kplane + c0, (unsigned)a.pitch, ncols, false};
    col_pipeline_to_regs<T, SEQF>(ctx, s, gv, tid, NT, LOGC, twl, logL, ld);
    ctx.sync();
    FB_STAMP(3);                                              // every LDS read of the forward precedes the leg buffers' writes
    cx<T>* bh = s;
    cx<T>* bx = s + (Mq << LOGC);
    cx<T>* by = s + 2 * (Mq << LOGC);
    cx<T> gyv[LR == 1 ? NK : 1];                             // R = 2: Gy's spectrum until buffer 0 is free
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & (C - 1), j = b >> LOGC;
        const T lx = (c < ncols) ? lxv[u] : (T)0;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int t = q < KQ / 2 ? q : RL - KQ + q;
            const cx<T> x = gv[u * RL + t];
            const int k2p = j + q * Ns;                      // the bin's place in the Mq-point coarse spectrum of this k1
            const bool live = lv[KQ * u + q];
            const T fg = live ? fgv[KQ * u + q] : (T)0, fh = live ? fhv[KQ * u + q] : (T)0, ly = live ? lyv[KQ * u + q] : (T)0;
            const cx<T> g = mul_pi(x * fg);
            const int at = (k2p << LOGC) + c;
            bh[at] = swp(x * fh);                            // inverse transform = forward transform of the swapped data
            bx[at] = swp(g * lx);
            if constexpr (LR == 1) gyv[KQ * u + q] = swp(g * ly);
            else by[at] = swp(g * ly);
        }
    }
    ctx.sync();
    FB_STAMP(4);
    const int grp = tid / NTQ, tq = tid - grp * NTQ;         // groups 0..2: H, Gx, Gy; the others idle along
    cx<T>* outp = grp == 0 ? a.h : (grp == 1 ? a.gx : a.gy);
    using SI = typename SeqOf<logMq>::type;
    if constexpr (LR != 1) {
        const ColStore<T> st{outp + zmap * a.out_moff + (long)k1 * a.opitch + c0, (unsigned)(R * a.opitch), grp < 3 ? ncols : 0, true, ti, 0u, (T)1,
                             0, 0, 0, 0};
        fft_pipeline<T, false, false, true, SI>(ctx, s + grp * (Mq << LOGC), tq, NTQ, logMq, LOGC, 0, twq, logMq, NoLoad{}, st);
    } else {
        const FBandStore2<T> st{outp + zmap * a.out_moff + (long)k1 * a.opitch + c0, (unsigned)(R * a.opitch), ncols, twl, tw_lds_h(logL), k1};
        fft_pipeline<T, false, false, true, SI>(ctx, s + grp * (Mq << LOGC), tq, NTQ, logMq, LOGC, 0, twq, logMq, NoLoad{}, st);
    }
    FB_STAMP(5);
    if constexpr (LR == 1) {
        ctx.sync();                                          // round 1 has left buffer 0
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int b = tid + u * NT;
            const int c = b & (C - 1), j = b >> LOGC;
#pragma unroll
            for (int q = 0; q < KQ; ++q) bh[((j + q * Ns) << LOGC) + c] = gyv[KQ * u + q];
        }
        ctx.sync();
        FB_STAMP(6);
        const FBandStore2<T> st2{a.gy + zmap * a.out_moff + (long)k1 * a.opitch + c0, (unsigned)(R * a.opitch), grp == 0 ? ncols : 0, twl, tw_lds_h(logL), k1};
        fft_pipeline<T, false, false, true, SI>(ctx, s + grp * (Mq << LOGC), tq, NTQ, logMq, LOGC, 0, twq, logMq, NoLoad{}, st2);      // (group 1 keeps the barriers company)
        FB_STAMP(7);
    }
}

// the packed filter table of ColFBandArgs::fgh: same grid and workgroup size as col_fband_body (grid z unused)
template <typename T, class SEQF, int LR, int LOGC, class Ctx>
OA_HD void col_fband_pack_body(Ctx& ctx, const ColFBandArgs<T>& a, cx<T>* out) {
    constexpr int logL = seq_total_log<SEQF>();
    constexpr int L = 1 << logL, R = 1 << LR;
    constexpr int RL = SEQF::get(SEQF::n - 1), NB = EPT / RL, Ns = L / RL, C = 1 << LOGC, KQ = RL / R, NK = KQ * NB;
    constexpr int NT = (1 << (logL + LOGC)) / EPT;
    const int tid = ctx.tid(), tile = ctx.bid_x(), k1 = ctx.bid_y(), c0 = tile << LOGC;
    cx<T>* o = out + (((long)k1 * ctx.grid_x() + tile) * NK) * NT + tid;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int b = tid + u * NT;
        const int c = b & (C - 1), j = b >> LOGC;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int t = q < KQ / 2 ? q : RL - KQ + q;
            const int k2 = j + t * Ns;
            const int yf = k1 + R * k2;
            bool live = c0 + c < a.width;
            if (a.rband) live = live && !(yf >= a.rband && yf <= a.ny_full - a.rband);
            cx<T> f = mk<T>((T)0, (T)0);
            if (live) { const long fi = (long)yf * a.fpitch + (c0 + c); f = mk<T>(a.FG[fi], a.FH[fi]); }
            o[(KQ * u + q) * NT] = f;
        }
    }
}

}  // namespace oa
