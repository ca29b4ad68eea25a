// Host-side decomposition of a 2-D power-of-two FFT into row / column passes.
// Shared by the HIP launcher (fft.hip) and the CPU emulator (tests/emul).
#pragma once
#include <cmath>
#include <cstdlib>
#include <vector>
#include "fft_kernels.hpp"
#include "fft_fband.hpp"
#include "fft_rowqe8.hpp"

namespace oa {

template <typename T>
inline std::vector<cx<T>> make_twiddles(int M) {
    std::vector<cx<T>> t((size_t)M);
    const long double tau = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < M; ++k) {
        long double a = tau * (long double)k / (long double)M;
        t[(size_t)k].x = (T)cosl(a);
        t[(size_t)k].y = (T)(-sinl(a));
    }
    return t;
}

inline long kpitch_for(int nx) { return nx / 2 + 16; }

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Geometry + device tables of one (ny, nx) transform.
template <typename T>
struct Fft2dPlan {
    int ny = 0, nx = 0, logNy = 0, logNx = 0;
    long kp = 0;                 // half-complex pitch (complex elements)
    const cx<T>* tw_x = nullptr; // W_nx^k, k < nx
    const cx<T>* tw_y = nullptr; // W_ny^k, k < ny
    const cx<T>* rq8c[RQ8_NGRIDS] = {};   // constants of the 8-point row stage's grids (slot = rq8_slot(M): 1024, 1536, 2048, 4096, 8192; nullptr: not offered)
    bool rq8_ready(int m) const { const int i = rq8_slot(m); return rowqe8_on() && i >= 0 && rq8c[i] != nullptr; }
    // COLUMN GRID view (fft.hip coarse_view): this plan describes ny = My rows of a map with ny_full rows; filters, ly
    // axis and caller-owned planes are addressed at the full-resolution rows (ColLegsArgs::yshift).  0 = own grid.
    int ny_full = 0;
    int yshift() const { return ny_full > ny ? ny_full - ny : 0; }
    static constexpr int COLC = COL_LOGC;  // log2 columns per column tile (kernels assume it at compile time)

    // COMPACT WORK PLANES.  A plane with `w` active columns stored at the full pitch kp leaves each row's w*8 bytes
    // 8 kp bytes apart: a column tile then touches 128 rows in 128 different DRAM pages / TLB entries for 256 bytes
    // each, and the narrow column passes ran at 3.3-3.9 TB/s where the dense ones reach 5.2-6.1 (profiles/r02g_*).
    // Plan-owned intermediates therefore use the smallest pitch that holds whole 32-column tiles (measured: the same
    // passes 26-32 % faster, tools/pitch_probe.py); planes that belong to the caller keep kp.
    long work_pitch(int w) const {
        const long c = 1L << COLC;
        const long p = ((long)clampw(w) + c - 1) / c * c;
        return p < kp ? p : kp;
    }
    // active-column count of an hc plane: <= 0 or too large means all nx/2+1 columns
    int clampw(int w) const { return (w <= 0 || w > nx / 2 + 1) ? nx / 2 + 1 : w; }
    // row band: rows y < rb or y > ny - rb are active; 0 (or a band covering every row) = all rows
    int clampr(int rb) const { return (rb <= 0 || 2L * rb - 1 >= ny) ? 0 : rb; }

    // ---- row passes -------------------------------------------------------
    template <class Launcher>
    void rows(Launcher& q, int mode, const void* in, long in_pitch, void* out, long out_pitch, T scale,
              int wcols = 0x7fffffff, const void* mul = nullptr, int nz = 0, long in_zoff = 0, long out_zoff = 0,
              const T* dlx = nullptr, int dpow0 = 0, int dcol_b = 0) const {
        RowArgs<T> a{};
        a.mul = mul;
        a.nz = nz; a.in_zoff = in_zoff; a.out_zoff = out_zoff;     // nz > 0: that many planes in one launch (grid y)
        a.dlx = dlx; a.dpow0 = dpow0; a.dcol_b = dcol_b;           // x-derivative C2R (RowArgs::dlx)
        const bool real_mode = (mode == ROW_R2C || mode == ROW_C2R || mode == ROW_WIN);
        a.logL = real_mode ? logNx - 1 : logNx;
        const int L = 1 << a.logL;
        int C = 4096 / L;
        if (C < 1) C = 1;
        if (C > ny) C = ny;
        a.logC = ilog2(C);
        a.NT = (L * C) / EPT;
        if (a.NT < 1) a.NT = 1;
        a.rowStride = L + (L >> 4) + 2;
        a.tw = tw_x;
        a.logTw = logNx;
        a.scale = scale;
        a.mode = mode;
        a.wcols = wcols;
        a.in = in; a.out = out; a.in_pitch = in_pitch; a.out_pitch = out_pitch;
        q.row(ny / C, a.NT, ((size_t)C * a.rowStride + tw_lds_size(a.logL)) * sizeof(cx<T>), a);
    }

    // ---- R-SPLIT: row R2C with the first radix-R butterfly of the column transform on top (RowArgs::lr), then ONE
    //      single-pass column kernel to the leg planes (fft_fband.hpp).  THIS plan is the full-resolution one; my = the
    //      column grid.  Available for R = ny / my = 4, my = 1024 or 2048, leg widths up to a quarter of the packed row; R = 8 (16384^2,
    //      float64) and R = 2 (8192^2 on 4096 rows, <= 1280 kept columns).
    static bool has_rsplit(int logNy, int logNx, int my, int wl) {
        if (my <= 0) return false;
        static const bool off = exp_env("OA_NO_RSPLIT") != nullptr;        // A/B switch
        const int logMy = ilog2(my), L = 1 << (logNx - 1);
        if (off || !is_pow2(my)) return false;
        // R = 8: 16384^2 maps on the 2048-row column grid (row_r2c_rs_body<T, 13, 3>: 16384-point rows, <= 512 kept columns)
        // (float64: the float build measured slower than the two-waves-per-row kernel + multi-pass columns, which float keeps)
        // (round 5 re-measured it at two workgroups per CU -- 128 registers, 42 of them spilled, no prefetch: R2C 364 us against 263 us, 2288
        //  against 2932 reconstructions/s, profiles/r05_16384_f32_rs8.txt)
        if (logNy - logMy == 3) return sizeof(T) == 8 && logMy == 11 && logNx == 14 && wl <= 512;
        // R = 2: the wide band of 8192^2 maps on the 4096-row column grid (row_r2c_rs_body<T, 12, 1, .., 5>: <= 1280 kept columns)
        if (logNy - logMy == 1) return logMy == 12 && logNx == 13 && wl <= 1280;
        return logNy - logMy == 2 && (logMy == 10 || logMy == 11) && logNx >= 11 && logNx <= 14 && wl <= L / 4 && wl <= RS_MAXS * (L / EPT);
    }
    template <class Launcher>
    void rows_rsplit(Launcher& q, const void* in, void* out, long out_pitch, long kplane, int wcols, int my) const {
        RowArgs<T> a{};
        a.logL = logNx - 1;
        const int L = 1 << a.logL;
        int C = 4096 / L;
        if (C < 1) C = 1;
        if (C > my) C = my;
        a.logC = ilog2(C);
        a.NT = (L * C) / EPT;
        a.rowStride = L + (L >> 4) + 2;
        a.tw = tw_x; a.logTw = logNx; a.scale = (T)1; a.mode = ROW_R2C; a.wcols = wcols;
        a.in = in; a.out = out; a.in_pitch = nx / 2; a.out_pitch = out_pitch;
        a.lr = logNy - ilog2(my); a.my = my; a.kplane = kplane; a.twy = tw_y;
        q.row_rsplit(my / C, a.NT, ((size_t)C * a.rowStride + tw_lds_size(a.logL)) * sizeof(cx<T>), a);
    }
    // log2 of the points of a col_fband tile: 128 KB of LDS (8 float / 4 double columns of 2048 rows); OA_FBAND_NARROW=1: half
    // (twice the workgroups, two per CU: A/B)
    static int fband_lt() {
        static const int narrow = [] { const char* e = exp_env("OA_FBAND_NARROW"); return e ? atoi(e) : 0; }();
        return (sizeof(T) == 4 ? 14 : 13) - (narrow > 0 ? 1 : 0);
    }
    // cv: the coarse view (ny = my rows, tw_y = W_my); Y: the row pass's R planes; leg planes in the R-LAYOUT (row y_lo R + k1)
    template <class Launcher>
    void legs_fband(Launcher& q, const Fft2dPlan<T>& cv, const cx<T>* Y, long kplane, long pin, const T* FG, const T* FH, const T* lxd,
                    const T* lyd, cx<T>* gx, cx<T>* gy, cx<T>* h, int wmax, int rband, long pout, int nmaps = 1, long in_moff = 0,
                    long out_moff = 0, const cx<T>* fgh = nullptr, cx<T>* pack_out = nullptr) const {
        // fgh: the packed filter table of this (FG, FH, wmax, rband, grid) -- pack_out != nullptr: MAKE it (nothing else runs)
        ColFBandArgs<T> a{};
        a.in = Y; a.kplane = kplane; a.pitch = pin; a.FG = FG; a.FH = FH; a.fpitch = kp; a.lxd = lxd; a.lyd = lyd; a.fgh = fgh;
        a.gx = gx; a.gy = gy; a.h = h; a.opitch = pout; a.width = clampw(wmax); a.tw = cv.tw_y; a.ny_full = ny; a.rband = clampr(rband);
        a.in_moff = in_moff; a.out_moff = out_moff;
        const int lt = fband_lt(), lc = lt - cv.logNy, Cs = 1 << lc;
        const int logMq = cv.logNy - (logNy - cv.logNy);
        // (R = 2: no LDS table of W_My^(k1 y_lo) -- col_fband_body)
        const size_t smem = ((size_t)(1 << lt) + tw_lds_size(cv.logNy) + tw_lds_size(logMq) + (logNy - cv.logNy == 1 ? 0 : (1 << logMq))) * sizeof(cx<T>);
        if (pack_out) q.col_fband_pack((a.width + Cs - 1) / Cs, 1 << (logNy - cv.logNy), cv.logNy, a, pack_out);
        else q.col_fband((a.width + Cs - 1) / Cs, 1 << (logNy - cv.logNy), nmaps, smem, cv.logNy, a);
    }
    // entries (cx<T>) of the packed filter table for `wmax` leg columns on the column grid `cv`
    long fband_table_entries(const Fft2dPlan<T>& cv, int wmax) const {
        const int lc = fband_lt() - cv.logNy, Cs = 1 << lc;
        return (long)((clampw(wmax) + Cs - 1) / Cs) * Cs * cv.ny;          // tiles x C x My
    }

    // ---- fused QE row stage: 3 hc planes (column-transformed legs) -> 2 hc planes -------------
    // mrow: length of the real row transforms (power of two, <= nx; 0 = nx).  Legs that vanish beyond column `win`
    // have products band-limited to 2 (win - 1), so the row stage evaluated on ANY grid of mrow >= 2 win + wout
    // points yields the same product columns k < wout (no aliasing reaches them) times mrow / nx -- folded into the
    // scale here.  The real-space planes exist only in LDS, so the sampling grid is not observable.
    // factor rows_qe puts on the caller's scale when the products are formed on an mrow-point grid
    double row_grid_scale(int mrow) const { return (mrow > 0 && mrow < nx) ? (double)nx / (double)mrow : 1.0; }
    // does rows_qe run a two-rows-per-transform kernel (the only row stage that takes two maps per launch)?
    // mrow: a power of two in [1024, 8192], or 1536 (= 3 x 512, eight-points-per-thread body only)
    bool rows_qe_is_pair(int win, int wout, int mrow) const {
        if (mrow <= 0 || ny % 2) return false;
        const int M = mrow < nx ? mrow : nx;
        if (rq8_is_m3(M)) return rq8_ready(M) && rq8_covers(M, win, wout);
        return is_pow2(M) && M >= 1024 && M <= 8192 && 2L * win + wout <= M;
    }
    // can the row stage read R = 2 planes (LAY = 1) of this band on grid mrow?
    bool rows_qe_lr1(int win, int wout, int mrow) const { return mrow == 4096 && mrow <= nx && rq8_ready(mrow) && rq8_covers(mrow, win, wout); }
    // which body runs the two-rows-per-transform row stage on 1024- to 4096-point grids: 8 points per thread (default) or 16
    // (experiment builds: OA_NO_ROWQE8=1; the emulator tests run both)
    static bool& rowqe8_on() {
        static bool on = exp_env("OA_NO_ROWQE8") == nullptr;
        return on;
    }
    // smallest row grid the two-rows-per-transform kernels are built for that holds 2 win + wout points without aliasing
    static int row_grid_min(int nx, int win, int wout) {
        long need = 2L * win + wout;
        if (rowqe8_on() && need > 1024 && need <= 1536 && win <= 512 && nx >= 2048) return 1536;
        int m = 1024;
        while (m < need && m < nx) m <<= 1;
        return m >= nx ? nx : m;
    }
    // rows per workgroup of the fused row stage (tuning hook: OA_QE_ROWS_PER_WG in the environment)
    static int qe_rows_per_wg(int L) {
        static const int forced = [] { const char* e = exp_env("OA_QE_ROWS_PER_WG"); return e ? atoi(e) : 0; }();
        if (forced > 0) return forced;
        return 4096 / L;
    }
    template <class Launcher>
    void rows_qe(Launcher& q, const cx<T>* gx, const cx<T>* gy, const cx<T>* h, cx<T>* px, cx<T>* py, T scale,
                 int accumulate = 0, int win = 0x7fffffff, int wout = 0x7fffffff, int mrow = 0, long pin = 0,
                 long pout = 0, int nmaps = 1, long in_moff = 0, long out_moff = 0, long h_moff = -1,
                 const RowQeMap<T>* tab = nullptr, int lr = 0, const int* chain = nullptr) const {
        // chain != nullptr (with tab): map m = estimator with pieces tab[chain[2m] ..+ chain[2m+1]) (RowQeArgs::chain)
        // lr = 2: the leg planes are in the R-LAYOUT of legs_fband (pair kernel only)
        // tab (device array of nmaps entries, pair kernel only): per-map planes and FINAL scales (see row_grid_scale)
        RowQeArgs<T> a{};
        // mrow > 0 ("grid mode", mrow <= nx): band-limited legs declared by the caller; mrow == 0: legacy full-length
        // transforms with no assumption beyond win / wout
        const int Mg = (mrow > 0 && mrow < nx) ? mrow : nx;          // the row grid
        const int logM = ilog2(Mg);
        if (Mg < nx) scale = scale * (T)((double)nx / (double)Mg);
        a.tw = tw_x; a.logTw = logNx; a.scale = scale;
        a.pitch = pin > 0 ? pin : kp; a.opitch = pout > 0 ? pout : kp;
        a.gx = gx; a.gy = gy; a.h = h; a.px = px; a.py = py; a.accumulate = accumulate;
        a.win = win; a.wout = wout;
        // the map's own 8192-point rows (mrow = 0 / mrow = nx) with band-limited legs: the two-rows-per-transform stage on the radix-16
        // cross stage of row_qe8_body, one map per launch, natural layout (1327 against 1101 reconstructions/s with both grids off;
        // with every column live the packed kernel below stays faster: 340 against 290 /s -- profiles/r05_fullrows_a16.txt)
        const bool full8 = Mg == nx && nx == 8192 && ny % 2 == 0 && nmaps <= 1 && !tab && !chain && lr == 0 && rq8_ready(Mg) && win <= 2048 &&
                           rq8_covers_full(Mg, win, wout < nx ? wout : nx / 2 + 1);
        if (full8 || (mrow > 0 && rows_qe_is_pair(win, wout, mrow))) {
            // alias-free row grid: two rows per complex transform of length M
            const int M = Mg;
            if (full8) {
                a.wout = wout < nx ? wout : nx / 2 + 1;
                a.logL = logM; a.logC = 0; a.NT = M / 8; a.rowStride = M;
                a.rq8c = rq8c[rq8_slot(M)];
                a.tab = nullptr; a.lr = 0; a.nrows = ny; a.chain = nullptr;
                q.row_qe_pair8(ny / 2, M, a);
                return;
            }
            if (rq8_ready(M) && rq8_covers(M, win, wout) && (lr == 0 || lr == 2 || lr == 3 || (lr == 1 && M == 4096))) {
                // eight points per thread, M / 512 waves per row pair (row_qe8_body, fft_rowqe8.hpp)
                a.logL = logM; a.logC = 0; a.NT = M / 8; a.rowStride = M;
                a.rq8c = rq8c[rq8_slot(M)];
                if (nmaps > 1 || tab) { a.npairs = ny / 2; a.in_moff = in_moff; a.out_moff = out_moff; a.h_moff = h_moff < 0 ? in_moff : h_moff; }
                a.tab = tab;
                a.lr = lr; a.nrows = ny; a.chain = chain;
                q.row_qe_pair8(ny / 2 * (nmaps > 1 ? nmaps : 1), M, a);
                return;
            }
            // 16 points per thread, M / 16 threads per row pair (row_qe_pair_body): 8192-point grids
            if (lr == 1) { q.fail_rlayout(); return; }        // (R = 2 exists in the 8-point body only; callers check rows_qe_lr1)
            a.logL = logM; a.logC = 0; a.NT = M / EPT; a.rowStride = M + (M >> 4) + 2;
            if (nmaps > 1 || tab) { a.npairs = ny / 2; a.in_moff = in_moff; a.out_moff = out_moff; a.h_moff = h_moff < 0 ? in_moff : h_moff; }
            a.tab = tab;
            a.lr = lr; a.nrows = ny; a.chain = chain;
            q.row_qe_pair(ny / 2 * (nmaps > 1 ? nmaps : 1), a.NT, ((size_t)a.rowStride + tw_lds_size(logM)) * sizeof(cx<T>), a);
            return;
        }
        if (lr) { q.fail_rlayout(); return; }                 // (callers check rows_qe_is_pair first)
        a.logL = logM - 1;
        const int L = 1 << a.logL;
        int C = qe_rows_per_wg(L);
        if (C < 1) C = 1;
        if (C > ny) C = ny;
        a.logC = ilog2(C);
        a.NT = (L * C) / EPT;
        if (a.NT < 1) a.NT = 1;
        a.rowStride = L + (L >> 4) + 2;
        q.row_qe(ny / C, a.NT, ((size_t)C * a.rowStride + tw_lds_size(a.logL)) * sizeof(cx<T>), a);
    }

    // ---- inverse column transform of the nd Fourier-space derivative fields (i lx)^a (i ly)^b k of each of nmaps transforms
    //      (col_deriv_body: the factor rides on pass 1's load), all planes per launch: pass 1, then the in-place pass 2
    // z0, nz: the planes [z0, z0 + nz) of the nmaps * nd only, written to out[0 .. nz) (chunked launches: a chunk small enough for the 256 MB infinity cache
    // goes through pass 1, pass 2 and the row pass back to back and its intermediates never leave the cache)
    template <class Launcher>
    void cols_derivs(Launcher& q, const cx<T>* in, long in_mstride, cx<T>* out, long out_pstride, int nmaps, int nd, const T* lxd,
                     const T* lyd, int z0 = 0, int nz = -1, int bonly = 0) const {
        if (nz < 0) nz = nmaps * nd;
        const int logN1 = (logNy + 1) / 2, logN2 = logNy - logN1;
        const long N1 = 1L << logN1, N2 = 1L << logN2;
        const int C = 1 << COLC, width = nx / 2 + 1, tiles = (width + C - 1) / C;
        ColDerivArgs<T> a{};
        a.in = in; a.out = out; a.in_mstride = in_mstride; a.out_pstride = out_pstride; a.pitch = kp; a.width = width; a.nd = nd;
        a.logL = logN1; a.tw = tw_y; a.logTw = logNy; a.in_ns = N2; a.out_gs = N1; a.lxd = lxd; a.lyd = lyd; a.zbase = z0; a.bonly = bonly;
        q.col_deriv(tiles, (int)N2, (int)((N1 * C) / EPT), ((size_t)N1 * C + tw_lds_size(logN1) + N1) * sizeof(cx<T>) + (size_t)(N1 + C) * sizeof(T), a, nz);
        cols(q, out, kp, out, kp, width, true, (T)1, 2, 1, nullptr, nullptr, 0, false, -1, nz, out_pstride, out_pstride);
    }

    // ---- full column transform of `width` columns (two passes) -------------
    // in -> out (out != in), result in natural order in `out`.
    // which: 0 = both passes, 1 = pass 1 only, 2 = pass 2 only (microbenchmarks)
    // nb > 1: the same pass over up to 3 planes in ONE launch (grid z = plane; in place only, which = 1 or 2
    // with in == out for pass 2 / distinct planes for pass 1) -- small active-column launches fill the chip better
    template <class Launcher>
    void cols(Launcher& q, const cx<T>* in, long in_pitch, cx<T>* out, long out_pitch, int width, bool inverse,
              T scale, int which = 0, int nb = 1, const cx<T>* const* ins = nullptr, cx<T>* const* outs = nullptr,
              int rband = 0, bool swap = false, int logn1 = -1, int nmaps = 1, long in_moff = 0, long out_moff = 0) const {
        // rband: the natural-order result (pass 2) is stored on the band rows only
        // nmaps = 2: the same pass over a second set of planes *_moff elements behind the first (two maps per launch)
        // swap: split Ny the other way round (pass 1 the SHORTER length) -- the inverse after legs_cols_from_pass1
        // logn1 >= 0: explicit pass-1 length (the inverse after legs_cols_from_pass1_cg: 16 x Ny/16)
        const int logN1 = logn1 >= 0 ? logn1 : (swap ? logNy / 2 : (logNy + 1) / 2), logN2 = logNy - logN1;
        const long N1 = 1L << logN1, N2 = 1L << logN2;
        const int C = 1 << COLC;
        const int tiles = (width + C - 1) / C;
        ColArgs<T> a{};
        a.width = width; a.logC = COLC; a.tw = tw_y; a.logTw = logNy; a.inverse = inverse ? 1 : 0;
        if (which != 2) {
            // pass 1: length N1 over y1 (stride N2), twiddle, write block-transposed
            a.in = in; a.in_pitch = in_pitch; a.out = out; a.out_pitch = out_pitch;
            if (nb > 1) {
                a.in_off1 = ins[1] - in; a.out_off1 = outs[1] - out;
                if (nb > 2) { a.in_off2 = ins[2] - in; a.out_off2 = outs[2] - out; }
            }
            a.logL = logN1; a.NT = (int)((N1 * C) / EPT);
            a.in_gs = 1; a.in_ns = N2; a.out_gs = N1; a.out_ks = 1;
            a.twiddle = (logN2 > 0) ? 1 : 0;
            a.scale = (logN2 > 0) ? (T)1 : scale;
            a.nbz = nmaps > 1 ? nb : 0; a.in_moff = in_moff; a.out_moff = out_moff;
            q.col(tiles, (int)N2, a.NT, ((size_t)N1 * C + tw_lds_size(logN1) + N1) * sizeof(cx<T>), a, nb * nmaps);
        }
        if (logN2 == 0 || which == 1) return;
        // pass 2: length N2 over y2 (stride N1), in place, natural order out
        a.in = out; a.in_pitch = out_pitch; a.out = out; a.out_pitch = out_pitch;
        if (nb > 1) {
            a.in_off1 = a.out_off1 = outs[1] - out;
            if (nb > 2) a.in_off2 = a.out_off2 = outs[2] - out;
        }
        a.logL = logN2; a.NT = (int)((N2 * C) / EPT);
        if (a.NT < 1) a.NT = 1;
        a.in_gs = 1; a.in_ns = N1; a.out_gs = 1; a.out_ks = N1;
        a.twiddle = 0; a.scale = scale; a.rband = clampr(rband); a.ny = ny;
        a.nbz = nmaps > 1 ? nb : 0; a.in_moff = out_moff; a.out_moff = out_moff;       // in place
        q.col(tiles, (int)N1, a.NT, ((size_t)N2 * C + tw_lds_size(logN2) + N2) * sizeof(cx<T>), a, nb * nmaps);
    }

    // (A) legs + inverse column transform of the three leg planes (outputs ready for rows_qe)
    template <class Launcher>
    void legs_cols(Launcher& q, const cx<T>* kX, const cx<T>* kY, const T* FG, const T* FH, const T* lxd, const T* lyd,
                   cx<T>* gx, cx<T>* gy, cx<T>* h, int wmax = 0x7fffffff, int rband = 0, long pin = 0, long pout = 0,
                   bool in_full = true, int subset = 0) const {   // in_full: kX, kY are stored on the full-resolution rows (column grid views)
        // subset: 1 = H only (kY, FH -> h), 2 = the gradient pair only (kX, FG -> gx, gy); pass 1 only -- the caller runs
        // the inverse pass 2 over all its planes in one launch (oa_qe_mv)
        const long pi = pin > 0 ? pin : kp, po = pout > 0 ? pout : kp;
        const int logN1 = (logNy + 1) / 2, logN2 = logNy - logN1;
        const long N1 = 1L << logN1, N2 = 1L << logN2;
        const int C = 1 << COLC;
        const int width = clampw(wmax);
        const int tiles = (width + C - 1) / C;
        ColLegsArgs<T> a{};
        a.kX = kX; a.kY = kY; a.FG = FG; a.FH = FH; a.lxd = lxd; a.lyd = lyd; a.gx = gx; a.gy = gy; a.h = h;
        a.pitch = pi; a.fpitch = kp; a.opitch = po;
        a.width = width; a.logC = COLC; a.NT = (int)((N1 * C) / EPT); a.tw = tw_y; a.logTw = logNy;
        a.in_gs = 1; a.in_ns = N2; a.out_gs = N1; a.out_ks = 1; a.twiddle = 1;
        a.rband = clampr(rband); a.ny = ny; a.yshift = yshift(); a.xfull = in_full ? 1 : 0;
        a.split = (yshift() && (long)tiles * N2 < 1024) ? 1 : 0;     // small launches of the column grid: one leg per workgroup
        if (subset) { a.split = 1; a.zbase = subset == 1 ? 0 : 1; a.zcount = subset == 1 ? 1 : 2; }
        q.col_legs(tiles, (int)N2, a.NT, ((size_t)N1 * C + tw_lds_size(logN1) + N1) * sizeof(cx<T>), logN1, a);
        if (subset) return;
        cx<T>* outs[3] = {gx, gy, h};
        const cx<T>* ins[3] = {gx, gy, h};
        cols(q, gx, po, gx, po, width, true, (T)1, 2, 3, ins, outs);      // pass 2 of the three planes, one launch
    }

    // (A-batch) oa_qe_mv: ngrad gradient fields (2 planes each) and nh H fields in ONE inverse pass-1 launch (grid z = plane);
    //      field f = filter plane ftab[f] (device table) applied to source src0 + {0, off1, off2}[(srcsel >> 2f) & 3]; plane z is
    //      stored at pool + z * ostride.  The caller runs the inverse pass 2 over the pool.
    //      Returns true when the leg planes are FINISHED (single pass: a coarse grid of 1024 rows holds a whole column of a
    //      16-column (f64: 8-column) tile in LDS -- col_legs_sp; OA_NO_LEGS_SP=1: off), false when the caller has to run the
    //      inverse pass 2.
    static bool legs_single_pass() {
        static const bool off = exp_env("OA_NO_LEGS_SP") != nullptr;
        return !off;
    }
    template <class Launcher>
    bool legs_cols_batch(Launcher& q, const cx<T>* src0, long off1, long off2, unsigned long long srcsel, const T* const* ftab, int ngrad, int nh, const T* lxd, const T* lyd, cx<T>* pool, long ostride, int wmax,
                         int rband, long pin, long pout, int selbits = 2) const {
        const long pi = pin > 0 ? pin : kp, po = pout > 0 ? pout : kp;
        // (2048-row grids -- 8192^2 maps: float32 is no faster in one pass than in two -- 17 planes x 48 tiles of 128 KB are 3.2 rounds of
        //  workgroups, 3156 vs 3309 MV reconstructions/s --, float64 is: 1848 vs 1775 (round 5, tools/r05_mv_f64.sh); experiment builds:
        //  OA_LEGS_SP_2048=1 / 0 forces it on / off.  1024-row grids: +7 % in oa_mc_run)
        static const int sp2048 = [] { const char* e = exp_env("OA_LEGS_SP_2048"); return e ? atoi(e) : -1; }();
        if (legs_single_pass() && (logNy == 10 || (logNy == 11 && (sp2048 >= 0 ? sp2048 != 0 : sizeof(T) == 8)))) {
            const int lt = sizeof(T) == 4 ? 14 : 13, lc = lt - logNy, Cs = 1 << lc;
            ColLegsArgs<T> a{};
            a.kX = src0; a.kY = src0; a.FG = nullptr; a.FH = nullptr; a.ftab = ftab; a.lxd = lxd; a.lyd = lyd; a.gx = pool; a.gy = pool; a.h = pool;
            a.pitch = pi; a.fpitch = kp; a.opitch = po;
            a.width = clampw(wmax); a.logC = lc; a.NT = (1 << lt) / EPT; a.tw = tw_y; a.logTw = logNy;
            a.in_gs = 1; a.in_ns = 1; a.out_gs = 1; a.out_ks = 1; a.twiddle = 0;
            a.rband = clampr(rband); a.ny = ny; a.yshift = yshift(); a.xfull = 1;
            a.split = 1; a.batch = 2 * ngrad + nh; a.ngrad = ngrad; a.selbits = selbits; a.srcsel = srcsel; a.src_off1 = off1; a.src_off2 = off2;
            a.ostride = ostride;
            q.col_legs_sp((a.width + Cs - 1) / Cs, a.NT, ((size_t)(1 << lt) + tw_lds_size(logNy) + 1) * sizeof(cx<T>), logNy, a);
            return true;
        }
        const int logN1 = (logNy + 1) / 2, logN2 = logNy - logN1;
        const long N1 = 1L << logN1, N2 = 1L << logN2;
        const int C = 1 << COLC;
        const int width = clampw(wmax);
        const int tiles = (width + C - 1) / C;
        ColLegsArgs<T> a{};
        a.kX = src0; a.kY = src0; a.FG = nullptr; a.FH = nullptr; a.ftab = ftab; a.lxd = lxd; a.lyd = lyd; a.gx = pool; a.gy = pool; a.h = pool;
        a.pitch = pi; a.fpitch = kp; a.opitch = po;
        a.width = width; a.logC = COLC; a.NT = (int)((N1 * C) / EPT); a.tw = tw_y; a.logTw = logNy;
        a.in_gs = 1; a.in_ns = N2; a.out_gs = N1; a.out_ks = 1; a.twiddle = 1;
        a.rband = clampr(rband); a.ny = ny; a.yshift = yshift(); a.xfull = 1;
        a.split = 1; a.batch = 2 * ngrad + nh; a.ngrad = ngrad; a.selbits = selbits; a.srcsel = srcsel; a.src_off1 = off1; a.src_off2 = off2;
        a.ostride = ostride;
        q.col_legs(tiles, (int)N2, a.NT, ((size_t)N1 * C + tw_lds_size(logN1) + N1) * sizeof(cx<T>), logN1, a);
        return false;
    }

    // (A') legs straight from the forward column pass 1 of the map's row transform (both legs from ONE map):
    //      forward pass 2 + filters + inverse pass 1 in one kernel, then the 3-plane inverse pass 2.
    //      Returns false when this geometry has no fused kernel (caller falls back to cols + legs_cols).
    static bool has_fwdlegs(int logNy) { return logNy / 2 >= 5 && logNy / 2 <= 7; }
    template <class Launcher>
    bool legs_cols_from_pass1(Launcher& q, const cx<T>* p1, const T* FG, const T* FH, const T* lxd, const T* lyd,
                              cx<T>* gx, cx<T>* gy, cx<T>* h, int wmax = 0x7fffffff, int rband = 0, long pin = 0,
                              long pout = 0) const {
        if (!has_fwdlegs(logNy)) return false;
        const long pi = pin > 0 ? pin : kp, po = pout > 0 ? pout : kp;
        const int logL = logNy / 2, logN1f = logNy - logL;
        const long L = 1L << logL, N1f = 1L << logN1f;
        const int C = 1 << COLC;
        const int width = clampw(wmax);
        const int tiles = (width + C - 1) / C;
        ColFwdLegsArgs<T> a{};
        a.in = p1; a.FG = FG; a.FH = FH; a.lxd = lxd; a.lyd = lyd; a.gx = gx; a.gy = gy; a.h = h;
        a.pitch = pi; a.fpitch = kp; a.opitch = po;
        a.width = width; a.tw = tw_y; a.logTw = logNy; a.n1f = N1f; a.rband = clampr(rband); a.ny = ny;
        q.col_fwdlegs(tiles, (int)N1f, (int)((L * C) / EPT), ((size_t)L * C + tw_lds_size(logL) + L) * sizeof(cx<T>), logL, a);
        cx<T>* outs[3] = {gx, gy, h};
        const cx<T>* ins[3] = {gx, gy, h};
        cols(q, gx, po, gx, po, width, true, (T)1, 2, 3, ins, outs, 0, true);   // inverse pass 2, split swapped
        return true;
    }

    // (A'') COLUMN GRID: legs from the forward column pass 1 of the map's transform on THIS (full-resolution) plan, outputs
    //       on `my` rows: forward pass 2 + filter + 16-point inverse pass 1 in one kernel (one leg per workgroup), then the
    //       3-plane inverse pass 2 of length ny / L on the coarse view `cv`.  false: geometry without this kernel.
    static bool has_fwdlegs_cg(int logNy, int my) {
        const int logL = logNy / 2;
        return logL == 6 && my > 0 && (my >> (logNy - logL)) == 16;
    }
    template <class Launcher>
    bool legs_cols_from_pass1_cg(Launcher& q, const Fft2dPlan<T>& cv, const cx<T>* p1, const T* FG, const T* FH, const T* lxd,
                                 const T* lyd, cx<T>* gx, cx<T>* gy, cx<T>* h, int wmax, long pin, long pout, int nmaps = 1,
                                 long in_moff = 0, long out_moff = 0) const {
        if (!has_fwdlegs_cg(logNy, cv.ny)) return false;
        const long pi = pin > 0 ? pin : kp, po = pout > 0 ? pout : kp;
        const int logL = logNy / 2, logN1f = logNy - logL;
        const long L = 1L << logL, N1f = 1L << logN1f;
        const int C = 1 << COLC;
        const int width = clampw(wmax);
        const int tiles = (width + C - 1) / C;
        ColFwdLegsCgArgs<T> a{};
        a.in = p1; a.FG = FG; a.FH = FH; a.lxd = lxd; a.lyd = lyd; a.gx = gx; a.gy = gy; a.h = h;
        a.pitch = pi; a.fpitch = kp; a.opitch = po; a.width = width; a.tw = tw_y; a.logTw = logNy; a.twc = cv.tw_y; a.n1f = N1f;
        a.in_moff = in_moff; a.out_moff = out_moff;
        q.col_fwdlegs_cg(tiles, (int)N1f, (int)((L * C) / EPT), ((size_t)L * C + tw_lds_size(logL) + 16 * C) * sizeof(cx<T>), logL, a,
                         nmaps > 1 ? 2 : 1);
        cx<T>* outs[3] = {gx, gy, h};
        const cx<T>* ins[3] = {gx, gy, h};
        // inverse pass 2: length My / 16 at row stride 16
        cv.cols(q, gx, po, gx, po, width, true, (T)1, 2, 3, ins, outs, 0, false, 4, nmaps, out_moff, out_moff);
        return true;
    }

    // log2 of the points of a single-pass divergence tile (as fband_lt); OA_DIV_NARROW=1: half
    static int div_lt() {
        static const int narrow = [] { const char* e = exp_env("OA_DIV_NARROW"); return e ? atoi(e) : 0; }();
        return (sizeof(T) == 4 ? 14 : 13) - (narrow > 0 ? 1 : 0);
    }
    static bool single_pass_div() {
        static const bool on = [] { const char* e = exp_env("OA_SINGLE_PASS_DIV"); return e ? atoi(e) != 0 : true; }();
        return on;
    }
    // (B) forward column transforms of two row-transformed planes + divergence * Fnorm
    //     tmpA, tmpB: two hc scratch planes
    template <class Launcher>
    void cols_div(Launcher& q, const cx<T>* pa, const cx<T>* pb, const T* Fn, const T* lxd, const T* lyd, cx<T>* out,
                  cx<T>* tmpA, cx<T>* tmpB, int accumulate, int wmax = 0x7fffffff, int rband = 0, long pin = 0, int nmaps = 1,
                  long in_moff = 0, long tmp_moff = 0, long out_moff = 0, long fn_moff = 0) const {
        const long pi = pin > 0 ? pin : kp;     // pitch of pa, pb AND of the two scratch planes
        if (single_pass_div() && (logNy == 10 || logNy == 11 || logNy == 12)) {
            // SINGLE PASS (short coarse-grid columns): a whole column of an 8- / 16-column (f64: 4- / 8-column; 4096 rows: 4 / 2) tile in LDS --
            // 128 KB --, the product planes are read once and nothing is written back but kappa's band rows
            const int lt = div_lt();
            const int lc = lt - logNy, Cs = 1 << lc;
            ColDivArgs<T> a{};
            a.A = pa; a.B = pb; a.Fn = Fn; a.lxd = lxd; a.lyd = lyd; a.out = out; a.pitch = pi; a.opitch = kp; a.width = clampw(wmax);
            a.logC = lc; a.NT = (1 << lt) / EPT; a.tw = tw_y; a.logTw = logNy; a.in_gs = 1; a.in_ns = 1; a.out_gs = 1; a.out_ks = 1;
            a.accumulate = accumulate; a.rband = clampr(rband); a.ny = ny; a.yshift = yshift();
            a.in_moff = in_moff; a.out_moff = out_moff; a.fn_moff = fn_moff;
            const int tl = (a.width + Cs - 1) / Cs;
            if (q.col_div_sp(tl, ((size_t)(1 << lt) + tw_lds_size(logNy)) * sizeof(cx<T>), logNy, a, nmaps)) return;
        }
        const int logN1 = (logNy + 1) / 2, logN2 = logNy - logN1;
        const long N1 = 1L << logN1, N2 = 1L << logN2;
        const int C = 1 << COLC;
        const int width = clampw(wmax);
        const int tiles = (width + C - 1) / C;
        const cx<T>* ins[2] = {pa, pb};
        cx<T>* outs[2] = {tmpA, tmpB};
        cols(q, pa, pi, tmpA, pi, width, false, (T)1, 1, 2, ins, outs, 0, false, -1, nmaps, in_moff, tmp_moff);   // pass 1 of both planes, one launch
        ColDivArgs<T> a{};
        a.A = tmpA; a.B = tmpB; a.Fn = Fn; a.lxd = lxd; a.lyd = lyd; a.out = out; a.pitch = pi; a.opitch = kp; a.width = width;
        a.logC = COLC; a.NT = (int)((N2 * C) / EPT); if (a.NT < 1) a.NT = 1;
        a.tw = tw_y; a.logTw = logNy; a.in_gs = 1; a.in_ns = N1; a.out_gs = 1; a.out_ks = N1; a.accumulate = accumulate;
        a.rband = clampr(rband); a.ny = ny; a.yshift = yshift();
        a.in_moff = tmp_moff; a.out_moff = out_moff; a.fn_moff = fn_moff;
        q.col_div(tiles, (int)N1, a.NT, ((size_t)N2 * C + tw_lds_size(logN2) + N2) * sizeof(cx<T>), logN2, a, nmaps);
    }

    // real (ny,nx) -> half-complex (ny, kp); tmp: one hc plane
    template <class Launcher>
    // only the first `wmax` columns of `out` are produced when wmax is given
    void r2c(Launcher& q, const T* in, cx<T>* out, cx<T>* tmp, T scale, int wmax = 0x7fffffff, int rband = 0) const {
        const int w = clampw(wmax);
        rows(q, ROW_R2C, in, nx / 2, tmp, kp, (T)1, w);
        cols(q, tmp, kp, out, kp, w, false, scale, 0, 1, nullptr, nullptr, rband);
    }
    // half-complex -> real; input preserved; tmp: two hc planes (tmp, tmp2)
    template <class Launcher>
    // columns >= wmax of `in` are taken as zero (never read) when wmax is given
    // mul != nullptr: real (ny, nx) plane multiplied into the result at the row pass's store (a real-space window)
    void c2r(Launcher& q, const cx<T>* in, T* out, cx<T>* tmp, T scale, int wmax = 0x7fffffff, const T* mul = nullptr) const {
        const int w = clampw(wmax);
        cols(q, in, kp, tmp, kp, w, true, (T)1);
        rows(q, ROW_C2R, tmp, kp, out, nx / 2, scale, w, mul);
    }
    // full complex (ny,nx) contiguous; tmp: one full plane; out != in
    template <class Launcher>
    void c2c(Launcher& q, const cx<T>* in, cx<T>* out, cx<T>* tmp, bool inverse, T scale) const {
        rows(q, inverse ? ROW_C2C_I : ROW_C2C_F, in, nx, tmp, nx, (T)1);
        cols(q, tmp, nx, out, nx, nx, inverse, scale);
    }
};

}  // namespace oa
