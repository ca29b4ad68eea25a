// CPU thread emulator for the FFT pass kernels (TEST INFRASTRUCTURE).
// Runs the exact kernel bodies of orphics_amd/csrc/fft_kernels.hpp with one
// std::thread per GPU thread and std::barrier for __syncthreads(), so the
// index math / pass decomposition is validated in the GPU-less container.
#include <barrier>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include "../../orphics_amd/csrc/fft_plan.hpp"
#include "../../orphics_amd/csrc/fft_r2c_w64.hpp"
#include "../../orphics_amd/csrc/fft_r2c_rs4096.hpp"
#include "../../orphics_amd/csrc/fft_mixed.hpp"

using namespace oa;

struct EmuCtx {
    int tid_, bx_, by_, bz_, gx_ = 1, nt_ = 0;
    int nthreads() const { return nt_; }
    std::barrier<>* bar;
    char* sm;
    int tid() const { return tid_; }
    int bid_x() const { return bx_; }
    int bid_y() const { return by_; }
    int bid_z() const { return bz_; }
    int grid_x() const { return gx_; }
    void sync() const { bar->arrive_and_wait(); }
    void wsync() const { bar->arrive_and_wait(); }   // (lanes are threads here: a wave-level exchange needs the real barrier)
    void* smem() const { return sm; }
};

static bool stockham_qe = false;
static bool emu_fband_packed = false;     // do_rsplit_legs: the filters through the packed table (emu_set_fband_packed)
static bool rsplit_pf = false;     // general R-split row pass: persistent workgroups with prefetch order (emu_set_rsplit_pf)   // which fused-row-stage body the emulator runs (both are tested)

struct EmuLauncher {
    template <class F>
    void run(int gx, int gy, int nt, size_t smem, F body, int gz = 1) {
        std::vector<char> sm(smem + 64);
        std::barrier<> bar(nt);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                for (int bz = 0; bz < gz; ++bz)
                  for (int by = 0; by < gy; ++by)
                    for (int bx = 0; bx < gx; ++bx) {
                        EmuCtx c{t, bx, by, bz, gx, nt, &bar, sm.data()};
                        body(c);
                        bar.arrive_and_wait();
                    }
            });
        for (auto& x : th) x.join();
    }
    template <typename T> void row(int grid, int nt, size_t smem, const RowArgs<T>& a) {
        dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            switch (a.mode) {
                case ROW_R2C: run(grid, 1, nt, smem, [&](EmuCtx& c) { row_fft_body<T, ROW_R2C, S>(c, a); }); break;
                case ROW_C2R: run(grid, a.nz > 0 ? a.nz : 1, nt, smem, [&](EmuCtx& c) { row_fft_body<T, ROW_C2R, S>(c, a); }); break;
                case ROW_C2C_F: run(grid, 1, nt, smem, [&](EmuCtx& c) { row_fft_body<T, ROW_C2C_F, S>(c, a); }); break;
                case ROW_WIN: run(grid, 1, nt, smem, [&](EmuCtx& c) { row_fft_body<T, ROW_WIN, S>(c, a); }); break;
                default: run(grid, 1, nt, smem, [&](EmuCtx& c) { row_fft_body<T, ROW_C2C_I, S>(c, a); }); break;
            }
        });
    }
    template <typename T> void row_qe(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        dispatch_seq_qe(a.logL, [&](auto seq) {
            using S = decltype(seq);
            int nz = 0;
            if constexpr (S::n >= 2) nz = qe_first_stage_nz(a.logL, S::rget(0), a.win);
            if (!stockham_qe) { run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_body_inplace<T, S>(c, a); }); return; }
            dispatch_nz<S>(nz, [&](auto nzc) {
                run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_body<T, S, decltype(nzc)::value>(c, a); });
            });
        });
    }
    template <typename T> void row_qe_pair(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_total_log<S>() >= 10 && seq_total_log<S>() <= 13)
                dispatch_pair_nz<S>(pair_first_stage_nz(a.logL, S::rget(0), a.win), [&](auto nzc) {
                    if (a.chain) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_pair_body<T, S, decltype(nzc)::value, 0, true>(c, a); });
                    else
                    if (a.lr == 2) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_pair_body<T, S, decltype(nzc)::value, 2>(c, a); });
                    else if (a.lr == 3) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_pair_body<T, S, decltype(nzc)::value, 3>(c, a); });
                    else run(grid, 1, nt, smem, [&](EmuCtx& c) { row_qe_pair_body<T, S, decltype(nzc)::value, 0>(c, a); });
                });
        });
    }
    template <typename T> void row_qe_pair8(int pairs, int M, const RowQeArgs<T>& a) {
        dispatch_rq8(M, a.win, a.lr, a.chain != nullptr, [&](auto ac, auto nzc, auto lay, auto ch) {
            constexpr int A = decltype(ac)::value;
            run(pairs, 1, 64 * A, rq8_lds_bytes<T, A, decltype(ch)::value>(),
                [&](EmuCtx& c) { row_qe8_body<T, A, decltype(nzc)::value, decltype(lay)::value, decltype(ch)::value>(c, a); });
        });
    }
    void fail_rlayout() {}
    template <typename T> void row_rsplit(int grid, int nt, size_t smem, const RowArgs<T>& a) {
        if (a.lr == 1 && a.logL == 12 && a.wcols <= 1280) {     // the wide band (fft.hip row_rs4096: persistent workgroups; here 3 walk the groups)
            run(a.my < 3 ? a.my : 3, 1, RS4096_NT, rs_lds_bytes<T, 12, 5>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 12, 1, false, 5>(c, a); });
            return;
        }
        if (a.lr != 2) return;
        if (rsplit_pf) {
            grid = grid > 2 ? (grid + 2) / 3 : grid;     // persistent workgroups: each walks ~3 groups (prefetch across group boundaries)
            if (a.logL == 10) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 4>, 2, true>(c, a); });
            else if (a.logL == 11) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 8>, 2, true>(c, a); });
            else if (a.logL == 12) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 16>, 2, true>(c, a); });
            return;
        }
        if (a.logL == 10) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 4>, 2, false>(c, a); });
        else if (a.logL == 11) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 8>, 2, false>(c, a); });
        else if (a.logL == 12) run(grid, 1, nt, smem, [&](EmuCtx& c) { row_r2c_rsplit_body<T, Seq<16, 16, 16>, 2, false>(c, a); });
    }
    template <typename T> void col_fband(int gx, int gy, int gz, size_t smem, int logMy, const ColFBandArgs<T>& a) {
        constexpr int nt = sizeof(T) == 4 ? 1024 : 512;
        constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
        if (gy == 4 && logMy == 11) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fband_body<T, Seq<16, 16, 8>, 2, lc11>(c, a); }, gz);
        else if (gy == 4 && logMy == 10) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fband_body<T, Seq<16, 8, 8>, 2, lc10>(c, a); }, gz);
        else if (gy == 8 && logMy == 11) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fband_body<T, Seq<16, 8, 16>, 3, lc11>(c, a); }, gz);
        else if (gy == 2 && logMy == 12) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fband_body<T, Seq<16, 16, 16>, 1, lc11 - 1>(c, a); }, gz);
    }
    template <typename T> void col_fband_pack(int gx, int gy, int logMy, const ColFBandArgs<T>& a, cx<T>* out) {
        constexpr int nt = sizeof(T) == 4 ? 1024 : 512;
        constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
        if (gy == 4 && logMy == 11) run(gx, gy, nt, 0, [&](EmuCtx& c) { col_fband_pack_body<T, Seq<16, 16, 8>, 2, lc11>(c, a, out); });
        else if (gy == 4 && logMy == 10) run(gx, gy, nt, 0, [&](EmuCtx& c) { col_fband_pack_body<T, Seq<16, 8, 8>, 2, lc10>(c, a, out); });
        else if (gy == 8 && logMy == 11) run(gx, gy, nt, 0, [&](EmuCtx& c) { col_fband_pack_body<T, Seq<16, 8, 16>, 3, lc11>(c, a, out); });
        else if (gy == 2 && logMy == 12) run(gx, gy, nt, 0, [&](EmuCtx& c) { col_fband_pack_body<T, Seq<16, 16, 16>, 1, lc11 - 1>(c, a, out); });
    }
    template <typename T> void col_deriv(int gx, int gy, int nt, size_t smem, const ColDerivArgs<T>& a, int nz) {
        dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_total_log<S>() <= 8) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_deriv_body<T, S>(c, a); }, nz);
        });
    }
    template <typename T> void col_legs(int gx, int gy, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
        dispatch_seq(logL, [&](auto seq) {
            using S = decltype(seq);
            run(gx, gy, nt, smem, [&](EmuCtx& c) { col_legs_body<T, S>(c, a); }, a.batch ? a.batch : (a.split ? (a.zcount ? a.zcount : 3) : 1));
        });
    }
    template <typename T> void col_legs_sp(int gx, int nt, size_t smem, int logL, const ColLegsArgs<T>& a) {
        constexpr int lc11 = sizeof(T) == 4 ? 3 : 2, lc10 = lc11 + 1;
        if (logL == 11) run(gx, 1, nt, smem, [&](EmuCtx& c) { col_legs_body<T, Seq<16, 16, 8>, EmuCtx, lc11>(c, a); }, a.batch);
        else if (logL == 10) run(gx, 1, nt, smem, [&](EmuCtx& c) { col_legs_body<T, Seq<16, 16, 4>, EmuCtx, lc10>(c, a); }, a.batch);
    }
    template <typename T> void col_fwdlegs(int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsArgs<T>& a) {
        dispatch_seq(logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_total_log<S>() >= 5 && seq_total_log<S>() <= 7)
                run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fwdlegs_body<T, S>(c, a); });
        });
    }
    template <typename T> void col_fwdlegs_cg(int gx, int gy, int nt, size_t smem, int logL, const ColFwdLegsCgArgs<T>& a, int gz = 1) {
        if (logL == 6) run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fwdlegs_cg_body<T, Seq<16, 4>>(c, a); }, gz);
    }
    template <typename T> void col_div(int gx, int gy, int nt, size_t smem, int logL, const ColDivArgs<T>& a, int gz = 1) {
        dispatch_seq(logL, [&](auto seq) {
            using S = decltype(seq);
            run(gx, gy, nt, smem, [&](EmuCtx& c) { col_div_body<T, S>(c, a); }, gz);
        });
    }
    template <typename T> bool col_div_sp(int gx, size_t smem, int logL, const ColDivArgs<T>& a, int gz = 1) {
        if constexpr (sizeof(T) == 4) {
            if (logL == 11) { run(gx, 1, 1024, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 8>, EmuCtx, 3>(c, a); }, gz); return true; }
            if (logL == 10) { run(gx, 1, 1024, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 4>, EmuCtx, 4>(c, a); }, gz); return true; }
            if (logL == 12) { run(gx, 1, 1024, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 16>, EmuCtx, 2>(c, a); }, gz); return true; }
        } else {
            if (logL == 11) { run(gx, 1, 512, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 8>, EmuCtx, 2>(c, a); }, gz); return true; }
            if (logL == 10) { run(gx, 1, 512, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 4>, EmuCtx, 3>(c, a); }, gz); return true; }
            if (logL == 12) { run(gx, 1, 512, smem, [&](EmuCtx& c) { col_div_body<T, Seq<16, 16, 16>, EmuCtx, 1>(c, a); }, gz); return true; }
        }
        return false;
    }
    template <typename T> void col(int gx, int gy, int nt, size_t smem, const ColArgs<T>& a, int nz = 1) {
        dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            run(gx, gy, nt, smem, [&](EmuCtx& c) { col_fft_body<T, S>(c, a); }, nz);
        });
    }
};

template <typename T>
struct Holder {
    std::vector<cx<T>> twx, twy, rq8t[RQ8_NGRIDS];
    Fft2dPlan<T> p;
    Holder(int ny, int nx) {
        twx = make_twiddles<T>(nx);
        twy = make_twiddles<T>(ny);
        p.ny = ny; p.nx = nx; p.logNy = ilog2(ny); p.logNx = ilog2(nx);
        p.kp = kpitch_for(nx); p.tw_x = twx.data(); p.tw_y = twy.data();
        for (int i = 0; i < RQ8_NGRIDS; ++i) if (512 * RQ8_WAVES[i] <= nx) { rq8t[i] = rq8_make_consts<T>(RQ8_WAVES[i]); p.rq8c[i] = rq8t[i].data(); }
    }
};

template <typename T>
static int do_r2c(int ny, int nx, const T* in, cx<T>* out, double scale, int width = 0, int rband = 0) {
    Holder<T> h(ny, nx);
    std::vector<cx<T>> tmp((size_t)ny * h.p.kp);
    EmuLauncher q;
    h.p.r2c(q, in, out, tmp.data(), (T)scale, width, rband);
    return 0;
}
// fused windowed row pass (ROW_WIN): hc rows (pitch kp) -> C2R -> x window -> R2C -> hc rows (pitch opitch), wcols kept
template <typename T>
static int do_rows_win(int ny, int nx, const cx<T>* in, const T* window, cx<T>* out, long opitch, double scale, int wcols) {
    Holder<T> h(ny, nx);
    EmuLauncher q;
    h.p.rows(q, ROW_WIN, in, h.p.kp, out, opitch, (T)scale, wcols, window);
    return 0;
}
template <typename T>
static int do_c2r(int ny, int nx, const cx<T>* in, T* out, double scale, int width = 0) {
    Holder<T> h(ny, nx);
    std::vector<cx<T>> tmp((size_t)ny * h.p.kp);
    EmuLauncher q;
    h.p.c2r(q, in, out, tmp.data(), (T)scale, width);
    return 0;
}
template <typename T>
static int do_c2c(int ny, int nx, const cx<T>* in, cx<T>* out, int inverse, double scale) {
    Holder<T> h(ny, nx);
    std::vector<cx<T>> tmp((size_t)ny * nx);
    EmuLauncher q;
    h.p.c2c(q, in, out, tmp.data(), inverse != 0, (T)scale);
    return 0;
}

template <typename T>
static int do_qe_rows(int ny, int nx, const cx<T>* gx, const cx<T>* gy, const cx<T>* h, cx<T>* px, cx<T>* py, double s,
                      int win = 0, int wout = 0, int mrow = 0) {
    Holder<T> hd(ny, nx);
    EmuLauncher q;
    if (mrow < 0) {
        mrow = Fft2dPlan<T>::row_grid_min(nx, hd.p.clampw(win), hd.p.clampw(wout));
        if (2L * hd.p.clampw(win) + hd.p.clampw(wout) > mrow) mrow = 0;
    }
    hd.p.rows_qe(q, gx, gy, h, px, py, (T)s, 0, hd.p.clampw(win), hd.p.clampw(wout), mrow);
    return mrow == 0 ? nx : mrow;
}

template <typename T>
static int do_legs_cols(int ny, int nx, const cx<T>* kX, const cx<T>* kY, const T* FG, const T* FH, const T* lxd, const T* lyd,
                        cx<T>* gx, cx<T>* gy, cx<T>* h, int width = 0, int rband = 0) {
    Holder<T> hd(ny, nx);
    EmuLauncher q;
    hd.p.legs_cols(q, kX, kY, FG, FH, lxd, lyd, gx, gy, h, width, rband);
    return 0;
}
template <typename T>
static int do_cols_div(int ny, int nx, const cx<T>* pa, const cx<T>* pb, const T* Fn, const T* lxd, const T* lyd, cx<T>* out,
                       int width = 0, int rband = 0) {
    Holder<T> hd(ny, nx);
    std::vector<cx<T>> tA((size_t)ny * hd.p.kp), tB((size_t)ny * hd.p.kp);
    EmuLauncher q;
    hd.p.cols_div(q, pa, pb, Fn, lxd, lyd, out, tA.data(), tB.data(), 0, width, rband);
    return 0;
}

// real map -> leg planes through the fused forward-pass-2 + legs kernel (geometry must support it)
template <typename T>
static int do_map_legs_cols(int ny, int nx, const T* map, const T* FG, const T* FH, const T* lxd, const T* lyd, cx<T>* gx,
                            cx<T>* gy, cx<T>* h, int width, int rband) {
    Holder<T> hd(ny, nx);
    if (!Fft2dPlan<T>::has_fwdlegs(hd.p.logNy)) return 1;
    std::vector<cx<T>> tA((size_t)ny * hd.p.kp), tB((size_t)ny * hd.p.kp);
    EmuLauncher q;
    const int w = hd.p.clampw(width);
    hd.p.rows(q, ROW_R2C, map, nx / 2, tA.data(), hd.p.kp, (T)1, w);
    hd.p.cols(q, tA.data(), hd.p.kp, tB.data(), hd.p.kp, w, false, (T)1, 1);
    return hd.p.legs_cols_from_pass1(q, tB.data(), FG, FH, lxd, lyd, gx, gy, h, width, rband) ? 0 : 1;
}

// COLUMN GRID view of a (ny_full, nx) map: the column passes run on my rows, filters / ly / kX / out stay full-resolution
template <typename T>
struct CoarseHolder {
    std::vector<cx<T>> twx, twy;
    Fft2dPlan<T> p;
    CoarseHolder(int ny_full, int my, int nx) {
        twx = make_twiddles<T>(nx);
        twy = make_twiddles<T>(my);
        p.ny = my; p.nx = nx; p.logNy = ilog2(my); p.logNx = ilog2(nx);
        p.kp = kpitch_for(nx); p.tw_x = twx.data(); p.tw_y = twy.data(); p.ny_full = ny_full;
    }
};

template <typename T>
static int do_cols_div_batch(int ny_full, int my, int nx, const void* prod, const T* Fn, const T* lxd, const T* lyd, void* out, int nmaps,
                             int width, int rband) {
    CoarseHolder<T> hd(ny_full, my, nx);
    const long plane = (long)my * hd.p.kp;
    std::vector<cx<T>> tmp((size_t)plane * 2 * nmaps);
    EmuLauncher q;
    hd.p.cols_div(q, (const cx<T>*)prod, (const cx<T>*)prod + plane, Fn, lxd, lyd, (cx<T>*)out, tmp.data(), tmp.data() + plane, 0, width, rband, 0,
                  nmaps, 2 * plane, 2 * plane, (long)ny_full * hd.p.kp, (long)ny_full * hd.p.kp);
    return 0;
}

// R-SPLIT from-map leg path (rows_rsplit -> legs_fband) and the row stage on its R-LAYOUT planes
template <typename T>
static int do_rsplit_rows(int ny, int my, int nx, const T* map, cx<T>* Y, long pitch, int width) {
    Holder<T> hd(ny, nx);
    EmuLauncher q;
    if (!Fft2dPlan<T>::has_rsplit(hd.p.logNy, hd.p.logNx, my, hd.p.clampw(width))) return 1;
    hd.p.rows_rsplit(q, map, Y, pitch, (long)my * pitch, hd.p.clampw(width), my);
    return 0;
}
template <typename T>
static int do_rsplit_legs(int ny, int my, int nx, const cx<T>* Y, long pitch, const T* FG, const T* FH, const T* lxd, const T* lyd, cx<T>* gx,
                          cx<T>* gy, cx<T>* h, long opitch, int width, int rband, int nmaps, long in_moff, long out_moff) {
    Holder<T> hd(ny, nx);
    CoarseHolder<T> cv(ny, my, nx);
    EmuLauncher q;
    if (emu_fband_packed) {
        // the filters through the packed table (what the one-call entries of pipeline.hip do); the planes themselves are then not read
        std::vector<cx<T>> tab((size_t)hd.p.fband_table_entries(cv.p, width));
        hd.p.legs_fband(q, cv.p, (const cx<T>*)nullptr, 0, 0, FG, FH, lxd, lyd, (cx<T>*)nullptr, (cx<T>*)nullptr, (cx<T>*)nullptr, width, rband, 0, 1, 0, 0, (const cx<T>*)nullptr, tab.data());
        hd.p.legs_fband(q, cv.p, Y, (long)my * pitch, pitch, (const T*)nullptr, (const T*)nullptr, lxd, lyd, gx, gy, h, width, rband, opitch, nmaps, in_moff, out_moff, tab.data());
        return 0;
    }
    hd.p.legs_fband(q, cv.p, Y, (long)my * pitch, pitch, FG, FH, lxd, lyd, gx, gy, h, width, rband, opitch, nmaps, in_moff, out_moff);
    return 0;
}
template <typename T>
static int do_qe_rows_rlayout(int my, int nx, const cx<T>* gx, const cx<T>* gy, const cx<T>* h, cx<T>* px, cx<T>* py, double s, int win, int wout,
                              int mrow, int lr) {
    Holder<T> hd(my, nx);
    EmuLauncher q;
    if (!hd.p.rows_qe_is_pair(hd.p.clampw(win), hd.p.clampw(wout), mrow)) return 1;
    hd.p.rows_qe(q, gx, gy, h, px, py, (T)s, 0, hd.p.clampw(win), hd.p.clampw(wout), mrow, 0, 0, 1, 0, 0, -1, nullptr, lr);
    return 0;
}

// R-split R2C with one workgroup-wide exchange (fft_r2c_rs4096.hpp): nx = 8192 / 4096 with ny = 4 my (R = 4), nx = 16384 with ny = 8 my
// (R = 8), either precision
template <typename T>
static int do_rs4096(int ny, int nx, const T* in, void* out, long pitch, int width, int nwg, int pf) {
    const bool wide = nx == 8192 && width > 512 && width <= 1280;      // the wide band: R = 2, five kept bins per side
    const int R = nx == 16384 ? 8 : (wide ? 2 : 4);
    if (!((nx == 16384 && width <= 512) || (nx == 8192 && (width <= 512 || wide)) || (nx == 4096 && width <= 256)) || (ny % R)) return 1;
    auto tw = make_twiddles<T>(nx);
    auto twy = make_twiddles<T>(ny);
    RowArgs<T> a{};
    a.in = in; a.out = out; a.in_pitch = nx / 2; a.out_pitch = pitch; a.logL = ilog2(nx) - 1; a.logC = 0; a.NT = nx / 32;
    a.tw = tw.data(); a.logTw = ilog2(nx); a.scale = (T)1; a.mode = ROW_R2C; a.wcols = width; a.lr = R == 8 ? 3 : (R == 2 ? 1 : 2); a.my = ny / R;
    a.kplane = (long)(ny / R) * pitch; a.twy = twy.data();
    EmuLauncher q;
    if (nx == 16384) {
        if (pf) q.run(nwg, 1, 512, rs_lds_bytes<T, 13>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 13, 3, true>(c, a); });
        else q.run(nwg, 1, 512, rs_lds_bytes<T, 13>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 13, 3, false>(c, a); });
    } else if (wide) {
        if (pf) q.run(nwg, 1, RS4096_NT, rs_lds_bytes<T, 12, 5>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 12, 1, true, 5>(c, a); });
        else q.run(nwg, 1, RS4096_NT, rs_lds_bytes<T, 12, 5>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 12, 1, false, 5>(c, a); });
    } else if (nx == 8192) {
        if (pf) q.run(nwg, 1, RS4096_NT, rs_lds_bytes<T, 12>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 12, 2, true>(c, a); });
        else q.run(nwg, 1, RS4096_NT, rs_lds_bytes<T, 12>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 12, 2, false>(c, a); });
    } else {
        if (pf) q.run(nwg, 1, 128, rs_lds_bytes<T, 11>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 11, 2, true>(c, a); });
        else q.run(nwg, 1, 128, rs_lds_bytes<T, 11>(), [&](EmuCtx& c) { row_r2c_rs_body<T, 11, 2, false>(c, a); });
    }
    return 0;
}
// all nd derivative fields of ONE transform, inverse-transformed, the way fft.hip lens_derivs_impl does it: per y-derivative order b
// one column transform of (i ly)^b k0 (col_deriv_body, b-only) and ONE row launch for the x-derivative orders a (RowArgs::dlx);
// separable = 0: the first implementation (one column transform per (a, b), plain row C2Rs).  out = nd real planes
template <typename T>
static int do_lens_derivs(int ny, int nx, const cx<T>* k0, const T* lxd, const T* lyd, T* out, int nd, int separable) {
    Holder<T> h(ny, nx);
    const long hcp = (long)ny * h.p.kp, rp = (long)ny * (nx / 2);
    EmuLauncher q;
    if (!separable) {
        std::vector<cx<T>> pool((size_t)nd * hcp);
        h.p.cols_derivs(q, k0, hcp, pool.data(), hcp, 1, nd, lxd, lyd);
        h.p.rows(q, ROW_C2R, pool.data(), h.p.kp, out, nx / 2, (T)(1.0 / ((double)ny * nx)), 0x7fffffff, nullptr, nd, hcp, rp);
        return 0;
    }
    int order = 1;
    while (order * (order + 1) / 2 - 1 < nd) ++order;
    std::vector<cx<T>> plane((size_t)hcp);
    // separable = 2 (oa_lens_maps_hc): the undisplaced field (a, b) = (0, 0) is one more plane of the b = 0 launch -- out = nd + 1
    // planes, D_00 first
    const int d00 = separable == 2 ? 1 : 0;
    for (int b = 0; b < order; ++b) {
        const int a0 = (b == 0 && !d00) ? 1 : 0, na = order - b - a0;
        if (na <= 0) continue;
        h.p.cols_derivs(q, k0, hcp, plane.data(), hcp, 1, order, lxd, lyd, b, 1, 1);
        h.p.rows(q, ROW_C2R, plane.data(), h.p.kp, out + (size_t)d00 * ny * nx, nx / 2, (T)(1.0 / ((double)ny * nx)), 0x7fffffff, nullptr, na, 0, rp, lxd, a0, b);
    }
    return 0;
}
// ---- mixed-radix passes (fft_mixed.hpp): sides 2^a 3^b 5^c.  what = 0: r2c (real (ny, nx) -> hc (ny, nx/2 + 1), contiguous),
// 1: c2r (the inverse, unnormalised), 2 / 3: c2c forward / inverse on a full (ny, nx) complex plane
template <typename T>
static int do_mixed(int ny, int nx, int what, const void* in, void* out) {
    if (!mixed_ok(ny) || !mixed_ok(nx) || !mixed_ok(nx / 2) || (nx & 1)) return 1;
    auto tab = [](int N, int extra) {
        std::vector<cx<T>> t((size_t)N + extra);
        const long double tau = 6.283185307179586476925286766559005768L;
        for (int k = 0; k < N + extra; ++k) { const long double x = tau * k / (long double)N; t[(size_t)k] = mk<T>((T)cosl(x), (T)(-sinl(x))); }
        return t;
    };
    const auto twx = tab(nx, 1), twxh = tab(nx / 2, 0), twy = tab(ny, 0);
    const long kp = nx / 2 + 1;
    EmuLauncher q;
    auto rows = [&](int mode, const void* i, long ip, void* o, long op, int N, const cx<T>* tw) {
        MrRowArgs<T> a{};
        a.in = i; a.out = o; a.in_pitch = ip; a.out_pitch = op; a.N = N; a.f = mixed_factor(N); a.tw = tw; a.tw2 = twx.data(); a.scale = (T)1; a.mode = mode;
        q.run(ny, 1, 64, 2 * ((size_t)N + 1) * sizeof(cx<T>), [&](EmuCtx& c) { mr_row_body<T>(c, a); });
    };
    auto cols = [&](const cx<T>* i, long ip, cx<T>* o, long op, int width, bool inv) {
        MrColArgs<T> a{};
        a.in = i; a.out = o; a.in_pitch = ip; a.out_pitch = op; a.N = ny; a.width = width; a.logC = 2; a.f = mixed_factor(ny); a.tw = twy.data(); a.scale = (T)1;
        a.inverse = inv ? 1 : 0;
        q.run((width + 3) / 4, 1, 64, 2 * ((size_t)ny << 2) * sizeof(cx<T>), [&](EmuCtx& c) { mr_col_body<T>(c, a); });
    };
    if (what == 0) { rows(MR_R2C, in, nx, out, kp, nx / 2, twxh.data()); cols((const cx<T>*)out, kp, (cx<T>*)out, kp, (int)kp, false); }
    else if (what == 1) {
        std::vector<cx<T>> tmp((size_t)ny * kp);
        cols((const cx<T>*)in, kp, tmp.data(), kp, (int)kp, true);
        rows(MR_C2R, tmp.data(), kp, out, nx, nx / 2, twxh.data());
    } else { rows(what == 3 ? MR_C2C_I : MR_C2C_F, in, nx, out, nx, nx, twx.data()); cols((const cx<T>*)out, nx, (cx<T>*)out, nx, nx, what == 3); }
    return 0;
}

extern "C" {
int emu_mixed_f64(int ny, int nx, int what, const void* in, void* out) { return do_mixed<double>(ny, nx, what, in, out); }
int emu_mixed_f32(int ny, int nx, int what, const void* in, void* out) { return do_mixed<float>(ny, nx, what, in, out); }
int emu_lens_derivs_f64(int ny, int nx, const void* k0, const double* lxd, const double* lyd, double* out, int nd, int separable) { return do_lens_derivs<double>(ny, nx, (const cx<double>*)k0, lxd, lyd, out, nd, separable); }
int emu_rows_win_f64(int ny, int nx, const void* in, const double* w, void* out, long opitch, double s, int wcols) { return do_rows_win<double>(ny, nx, (const cx<double>*)in, w, (cx<double>*)out, opitch, s, wcols); }
int emu_rows_win_f32(int ny, int nx, const void* in, const float* w, void* out, long opitch, double s, int wcols) { return do_rows_win<float>(ny, nx, (const cx<float>*)in, w, (cx<float>*)out, opitch, s, wcols); }
void emu_set_rsplit_pf(int on) { rsplit_pf = on != 0; }
void emu_set_fband_packed(int on) { emu_fband_packed = on != 0; }
int emu_rsplit_rows_f32(int ny, int my, int nx, const float* map, void* Y, long pitch, int width) { return do_rsplit_rows<float>(ny, my, nx, map, (cx<float>*)Y, pitch, width); }
int emu_rsplit_rows_f64(int ny, int my, int nx, const double* map, void* Y, long pitch, int width) { return do_rsplit_rows<double>(ny, my, nx, map, (cx<double>*)Y, pitch, width); }
int emu_rsplit_legs_f32(int ny, int my, int nx, const void* Y, long pitch, const float* FG, const float* FH, const float* lxd, const float* lyd, void* gx,
                        void* gy, void* h, long opitch, int width, int rband, int nmaps, long in_moff, long out_moff) {
    return do_rsplit_legs<float>(ny, my, nx, (const cx<float>*)Y, pitch, FG, FH, lxd, lyd, (cx<float>*)gx, (cx<float>*)gy, (cx<float>*)h, opitch, width, rband,
                                 nmaps, in_moff, out_moff);
}
int emu_rsplit_legs_f64(int ny, int my, int nx, const void* Y, long pitch, const double* FG, const double* FH, const double* lxd, const double* lyd, void* gx,
                        void* gy, void* h, long opitch, int width, int rband, int nmaps, long in_moff, long out_moff) {
    return do_rsplit_legs<double>(ny, my, nx, (const cx<double>*)Y, pitch, FG, FH, lxd, lyd, (cx<double>*)gx, (cx<double>*)gy, (cx<double>*)h, opitch, width,
                                  rband, nmaps, in_moff, out_moff);
}
int emu_qe_rows_rlayout_f64(int my, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s, int win, int wout, int mrow, int lr) {
    return do_qe_rows_rlayout<double>(my, nx, (const cx<double>*)gx, (const cx<double>*)gy, (const cx<double>*)h, (cx<double>*)px, (cx<double>*)py, s, win, wout, mrow, lr);
}
int emu_qe_rows_rlayout_f32(int my, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s, int win, int wout, int mrow, int lr) {
    return do_qe_rows_rlayout<float>(my, nx, (const cx<float>*)gx, (const cx<float>*)gy, (const cx<float>*)h, (cx<float>*)px, (cx<float>*)py, s, win, wout, mrow, lr);
}
int emu_rsplit_rows_rs4096_f32(int ny, int nx, const float* in, void* out, long pitch, int width, int nwg, int pf) { return do_rs4096<float>(ny, nx, in, out, pitch, width, nwg, pf); }
int emu_rsplit_rows_rs4096_f64(int ny, int nx, const double* in, void* out, long pitch, int width, int nwg, int pf) { return do_rs4096<double>(ny, nx, in, out, pitch, width, nwg, pf); }
// one-wave-per-row R2C pass (fft_r2c_w64.hpp): nx must be 8192; out has pitch nx/2+16
int emu_r2c_rows_w64_f32(int ny, int nx, const float* in, void* out, double scale, int width, int nwg) {
    if (nx != 8192 || width > 512) return 1;
    auto tw = make_twiddles<float>(nx);
    RowW64Args a{};
    a.in = (const cx<float>*)in; a.out = (cx<float>*)out; a.in_pitch = nx / 2; a.out_pitch = kpitch_for(nx);
    a.tw = tw.data(); a.logTw = ilog2(nx); a.scale = (float)scale; a.wcols = width; a.ny = ny; a.nwg = nwg;
    EmuLauncher q;
    q.run(nwg, 1, 64, W64_LDS_BYTES, [&](EmuCtx& c) { row_r2c_w64_body(c, a); });
    return 0;
}
// two-waves-per-row R2C pass for 16384-point rows
int emu_r2c_rows_w64x2_f32(int ny, int nx, const float* in, void* out, double scale, int width, int nwg) {
    if (nx != 16384 || width > 64 * W64X2_KEEP) return 1;
    auto tw = make_twiddles<float>(nx);
    RowW64Args a{};
    a.in = (const cx<float>*)in; a.out = (cx<float>*)out; a.in_pitch = nx / 2; a.out_pitch = kpitch_for(nx);
    a.tw = tw.data(); a.logTw = ilog2(nx); a.scale = (float)scale; a.wcols = width; a.ny = ny; a.nwg = nwg;
    EmuLauncher q;
    q.run(nwg, 1, 128, W64X2_LDS_BYTES, [&](EmuCtx& c) { row_r2c_w64x2_body(c, a); });
    return 0;
}
// real map -> leg planes on the my-row column grid: fused (fwd pass 2 + legs + 16-point inverse pass 1) vs the three-launch path
int emu_map_legs_cols_cg_f64(int ny, int my, int nx, const double* map, const double* FG, const double* FH, const double* lxd,
                             const double* lyd, void* gx, void* gy, void* h, int width, int rband, int fused) {
    Holder<double> hd(ny, nx);
    CoarseHolder<double> cv(ny, my, nx);
    std::vector<cx<double>> tA((size_t)ny * hd.p.kp), tB((size_t)ny * hd.p.kp);
    EmuLauncher q;
    const int w = hd.p.clampw(width);
    hd.p.rows(q, ROW_R2C, map, nx / 2, tA.data(), hd.p.kp, 1.0, w);
    hd.p.cols(q, tA.data(), hd.p.kp, tB.data(), hd.p.kp, w, false, 1.0, 1);
    if (fused)
        return hd.p.legs_cols_from_pass1_cg(q, cv.p, tB.data(), FG, FH, lxd, lyd, (cx<double>*)gx, (cx<double>*)gy, (cx<double>*)h, width, 0, 0) ? 0 : 1;
    hd.p.cols(q, tA.data(), hd.p.kp, tB.data(), hd.p.kp, w, false, 1.0, 2, 1, nullptr, nullptr, rband);
    cv.p.legs_cols(q, tB.data(), tB.data(), FG, FH, lxd, lyd, (cx<double>*)gx, (cx<double>*)gy, (cx<double>*)h, width, rband, 0, 0, true);
    return 0;
}
int emu_legs_cols_cg_f64(int ny_full, int my, int nx, const void* kX, const void* kY, const double* FG, const double* FH,
                         const double* lxd, const double* lyd, void* gx, void* gy, void* h, int width, int rband) {
    CoarseHolder<double> hd(ny_full, my, nx);
    EmuLauncher q;
    hd.p.legs_cols(q, (const cx<double>*)kX, (const cx<double>*)kY, FG, FH, lxd, lyd, (cx<double>*)gx, (cx<double>*)gy,
                   (cx<double>*)h, width, rband, 0, 0, true);
    return 0;
}
int emu_cols_div_cg_f64(int ny_full, int my, int nx, const void* pa, const void* pb, const double* Fn, const double* lxd,
                        const double* lyd, void* out, int width, int rband) {
    CoarseHolder<double> hd(ny_full, my, nx);
    std::vector<cx<double>> tA((size_t)my * hd.p.kp), tB((size_t)my * hd.p.kp);
    EmuLauncher q;
    hd.p.cols_div(q, (const cx<double>*)pa, (const cx<double>*)pb, Fn, lxd, lyd, (cx<double>*)out, tA.data(), tB.data(), 0, width, rband);
    return 0;
}
int emu_cols_div_cg_f32(int ny_full, int my, int nx, const void* pa, const void* pb, const float* Fn, const float* lxd,
                        const float* lyd, void* out, int width, int rband) {
    CoarseHolder<float> hd(ny_full, my, nx);
    std::vector<cx<float>> tA((size_t)my * hd.p.kp), tB((size_t)my * hd.p.kp);
    EmuLauncher q;
    hd.p.cols_div(q, (const cx<float>*)pa, (const cx<float>*)pb, Fn, lxd, lyd, (cx<float>*)out, tA.data(), tB.data(), 0, width, rband);
    return 0;
}
// oa_qe_mv's batched launches: ngrad gradient fields + nh H fields of up to three sources in ONE inverse pass-1 launch and
// one pass-2 launch over the pool; the divergence of nmaps estimators in one launch
int emu_legs_batch_cg_f64(int ny_full, int my, int nx, const void* src0, long off1, long off2, unsigned long long srcsel,
                          const double* const* ftab, int ngrad, int nh, const double* lxd, const double* lyd, void* pool, long ostride,
                          int width, int rband) {
    CoarseHolder<double> hd(ny_full, my, nx);
    EmuLauncher q;
    // (1024- / 2048-row column grids: single pass, the planes are finished; otherwise the inverse pass 2 over the pool)
    if (!hd.p.legs_cols_batch(q, (const cx<double>*)src0, off1, off2, srcsel, ftab, ngrad, nh, lxd, lyd, (cx<double>*)pool, ostride, width, rband,
                              0, 0))
        hd.p.cols(q, (const cx<double>*)pool, hd.p.kp, (cx<double>*)pool, hd.p.kp, hd.p.clampw(width), true, 1.0, 2, 1, nullptr, nullptr, 0, false,
                  -1, 2 * ngrad + nh, ostride, ostride);
    return 0;
}
int emu_cols_div_batch_cg_f64(int ny_full, int my, int nx, const void* prod, const double* Fn, const double* lxd, const double* lyd, void* out,
                              int nmaps, int width, int rband) {
    return do_cols_div_batch<double>(ny_full, my, nx, prod, Fn, lxd, lyd, out, nmaps, width, rband);
}
int emu_cols_div_batch_cg_f32(int ny_full, int my, int nx, const void* prod, const float* Fn, const float* lxd, const float* lyd, void* out,
                              int nmaps, int width, int rband) {
    return do_cols_div_batch<float>(ny_full, my, nx, prod, Fn, lxd, lyd, out, nmaps, width, rband);
}
int emu_map_legs_cols_f64(int ny, int nx, const double* map, const double* FG, const double* FH, const double* lxd,
                          const double* lyd, void* gx, void* gy, void* h, int width, int rband) {
    return do_map_legs_cols<double>(ny, nx, map, FG, FH, lxd, lyd, (cx<double>*)gx, (cx<double>*)gy, (cx<double>*)h, width, rband);
}
int emu_legs_cols_f64(int ny, int nx, const void* kX, const void* kY, const double* FG, const double* FH, const double* lxd,
                      const double* lyd, void* gx, void* gy, void* h) {
    return do_legs_cols<double>(ny, nx, (const cx<double>*)kX, (const cx<double>*)kY, FG, FH, lxd, lyd, (cx<double>*)gx,
                                (cx<double>*)gy, (cx<double>*)h);
}
int emu_cols_div_f64(int ny, int nx, const void* pa, const void* pb, const double* Fn, const double* lxd, const double* lyd,
                     void* out) {
    return do_cols_div<double>(ny, nx, (const cx<double>*)pa, (const cx<double>*)pb, Fn, lxd, lyd, (cx<double>*)out);
}
int emu_qe_rows_f32(int ny, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s) {
    do_qe_rows<float>(ny, nx, (const cx<float>*)gx, (const cx<float>*)gy, (const cx<float>*)h, (cx<float>*)px, (cx<float>*)py, s);
    return 0;
}
int emu_qe_rows_f64(int ny, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s) {
    do_qe_rows<double>(ny, nx, (const cx<double>*)gx, (const cx<double>*)gy, (const cx<double>*)h, (cx<double>*)px, (cx<double>*)py, s);
    return 0;
}
// active-column variants (width / win / wout as in include/orphics_amd.h)
int emu_r2c_w_f64(int ny, int nx, const double* in, void* out, double s, int width, int rband) { return do_r2c<double>(ny, nx, in, (cx<double>*)out, s, width, rband); }
int emu_c2r_w_f64(int ny, int nx, const void* in, double* out, double s, int width) { return do_c2r<double>(ny, nx, (const cx<double>*)in, out, s, width); }
int emu_qe_rows_w_f64(int ny, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s, int win, int wout) {
    do_qe_rows<double>(ny, nx, (const cx<double>*)gx, (const cx<double>*)gy, (const cx<double>*)h, (cx<double>*)px, (cx<double>*)py, s, win, wout);
    return 0;
}
// row-grid variant (mrow as in oa_qe_rows; < 0 = auto); returns the grid used
int emu_qe_rows_wm_f64(int ny, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s, int win, int wout, int mrow) {
    return do_qe_rows<double>(ny, nx, (const cx<double>*)gx, (const cx<double>*)gy, (const cx<double>*)h, (cx<double>*)px, (cx<double>*)py, s, win, wout, mrow);
}
int emu_qe_rows_wm_f32(int ny, int nx, const void* gx, const void* gy, const void* h, void* px, void* py, double s, int win, int wout, int mrow) {
    return do_qe_rows<float>(ny, nx, (const cx<float>*)gx, (const cx<float>*)gy, (const cx<float>*)h, (cx<float>*)px, (cx<float>*)py, s, win, wout, mrow);
}
// several maps in one row-stage launch (two-rows-per-transform kernel): planes[m] = {gx, gy, h, px, py} of map m, scales[m];
// table != 0: per-map operands through a RowQeMap table, else evenly spaced planes (the maps' planes must then be evenly
// spaced in memory: gx / gy / px / py by the same offsets the caller passes).  Returns 0, or 1 if this geometry is another kernel.
int emu_qe_rows_multi_f64(int ny, int nx, int nmaps, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                          void* const* py, const double* scales, int accumulate, int win, int wout, int mrow, int table) {
    Holder<double> hd(ny, nx);
    EmuLauncher q;
    const int wi = hd.p.clampw(win), wo = hd.p.clampw(wout);
    if (!hd.p.rows_qe_is_pair(wi, wo, mrow)) return 1;
    typedef cx<double> C;
    if (table) {
        std::vector<RowQeMap<double>> tab(nmaps);
        for (int m = 0; m < nmaps; ++m)
            tab[m] = RowQeMap<double>{(const C*)gx[m], (const C*)gy[m], (const C*)h[m], (C*)px[m], (C*)py[m], scales[m] * hd.p.row_grid_scale(mrow)};
        hd.p.rows_qe(q, (const C*)gx[0], (const C*)gy[0], (const C*)h[0], (C*)px[0], (C*)py[0], scales[0], accumulate, wi, wo, mrow, 0, 0, nmaps, 0, 0, 0,
                     tab.data());
    } else {
        const long io = nmaps > 1 ? (const C*)gx[1] - (const C*)gx[0] : 0, ho = nmaps > 1 ? (const C*)h[1] - (const C*)h[0] : 0;
        const long oo = nmaps > 1 ? (C*)px[1] - (C*)px[0] : 0;
        hd.p.rows_qe(q, (const C*)gx[0], (const C*)gy[0], (const C*)h[0], (C*)px[0], (C*)py[0], scales[0], accumulate, wi, wo, mrow, 0, 0, nmaps, io, oo, ho);
    }
    return 0;
}
// estimator chains (RowQeArgs::chain): `total` pieces grouped into nest estimators (first[e], count[e]); products into the px / py of each
// chain's first piece
int emu_qe_rows_chain_f64(int ny, int nx, int nest, int total, const void* const* gx, const void* const* gy, const void* const* h, void* const* px,
                          void* const* py, const double* scales, const int* first, const int* count, int win, int wout, int mrow) {
    Holder<double> hd(ny, nx);
    EmuLauncher q;
    const int wi = hd.p.clampw(win), wo = hd.p.clampw(wout);
    if (!hd.p.rows_qe_is_pair(wi, wo, mrow)) return 1;
    typedef cx<double> C;
    std::vector<RowQeMap<double>> tab(total);
    for (int i = 0; i < total; ++i)
        tab[i] = RowQeMap<double>{(const C*)gx[i], (const C*)gy[i], (const C*)h[i], (C*)px[i], (C*)py[i], scales[i] * hd.p.row_grid_scale(mrow)};
    std::vector<int> ch(2 * nest);
    for (int e = 0; e < nest; ++e) { ch[2 * e] = first[e]; ch[2 * e + 1] = count[e]; }
    hd.p.rows_qe(q, (const C*)gx[0], (const C*)gy[0], (const C*)h[0], (C*)px[0], (C*)py[0], scales[0], 0, wi, wo, mrow, 0, 0, nest, 0, 0, 0, tab.data(), 0,
                 ch.data());
    return 0;
}
int emu_legs_cols_w_f64(int ny, int nx, const void* kX, const void* kY, const double* FG, const double* FH, const double* lxd,
                        const double* lyd, void* gx, void* gy, void* h, int width, int rband) {
    return do_legs_cols<double>(ny, nx, (const cx<double>*)kX, (const cx<double>*)kY, FG, FH, lxd, lyd, (cx<double>*)gx,
                                (cx<double>*)gy, (cx<double>*)h, width, rband);
}
int emu_cols_div_w_f64(int ny, int nx, const void* pa, const void* pb, const double* Fn, const double* lxd, const double* lyd,
                       void* out, int width, int rband) {
    return do_cols_div<double>(ny, nx, (const cx<double>*)pa, (const cx<double>*)pb, Fn, lxd, lyd, (cx<double>*)out, width, rband);
}
void emu_set_stockham_qe(int on) { stockham_qe = on != 0; }
// which two-rows-per-transform row stage the plan launches: 8 points per thread (fft_rowqe8.hpp, default) or 16
void emu_set_rowqe8(int on) { Fft2dPlan<float>::rowqe8_on() = on != 0; Fft2dPlan<double>::rowqe8_on() = on != 0; }
long emu_kpitch(int nx) { return kpitch_for(nx); }
int emu_r2c_f32(int ny, int nx, const float* in, void* out, double s) { return do_r2c<float>(ny, nx, in, (cx<float>*)out, s); }
int emu_r2c_f64(int ny, int nx, const double* in, void* out, double s) { return do_r2c<double>(ny, nx, in, (cx<double>*)out, s); }
int emu_c2r_f32(int ny, int nx, const void* in, float* out, double s) { return do_c2r<float>(ny, nx, (const cx<float>*)in, out, s); }
int emu_c2r_f64(int ny, int nx, const void* in, double* out, double s) { return do_c2r<double>(ny, nx, (const cx<double>*)in, out, s); }
int emu_c2c_f32(int ny, int nx, const void* in, void* out, int inv, double s) { return do_c2c<float>(ny, nx, (const cx<float>*)in, (cx<float>*)out, inv, s); }
int emu_c2c_f64(int ny, int nx, const void* in, void* out, int inv, double s) { return do_c2c<double>(ny, nx, (const cx<double>*)in, (cx<double>*)out, inv, s); }
}
