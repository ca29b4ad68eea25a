// Executed arithmetic of the fused QE row stage (row_qe_kernel), counted by running the KERNEL BODY ITSELF
// (orphics_amd/csrc/fft_kernels.hpp, the same templates the HIP kernel instantiates) on the host over a
// counting scalar type.  Used by bench.py for the `valu` roofline of that kernel: achieved TFLOP/s =
// flops_per_row x rows / kernel time.
//
// What is counted (per lane, f32 device build):
//   * scalar +, -, * with both operands non-constant                      1 flop each
//   * the packed-asm complex forms of cx.hpp (opaque to the compiler):    a*b -> 6, a +- i b -> 2, always
//   * operations the compiler folds are NOT counted: const (op) const, x + (-0.0), x - (+0.0), x * 1.0,
//     x * (-1.0) (a sign modifier).  x * 0.0 and x + (+0.0) are not IEEE-foldable and are counted.
// Pruned taps of the active-column first stage enter as literal (-0.0) constants, exactly as in the kernel
// source, so their folded additions are not counted: the figure is the arithmetic actually issued, not the
// nominal 5 N log2 N.
//
//   usage: count_flops N win wout [mrow]   -> one JSON line {"n":..,"win":..,"wout":..,"flops_per_row":..,...}
//          mrow: row grid of oa_qe_rows (0 / absent = N, -1 = smallest alias-free power of two)
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static thread_local unsigned long long t_flops = 0;

struct CF {
    float v;
    bool c;   // compile-time constant (literal) -> foldable
    CF() : v(0.f), c(false) {}
    CF(float x) : v(x), c(true) {}
    CF(double x) : v((float)x), c(true) {}
    CF(long double x) : v((float)x), c(true) {}
    CF(int x) : v((float)x), c(true) {}
    static CF run(float x) { CF r; r.v = x; r.c = false; return r; }
};
static inline bool is_negzero(const CF& a) { return a.c && a.v == 0.f && std::signbit(a.v); }
static inline bool is_poszero(const CF& a) { return a.c && a.v == 0.f && !std::signbit(a.v); }
static inline CF mkc(float v, bool c) { CF r; r.v = v; r.c = c; return r; }
inline CF operator-(CF a) { return mkc(-a.v, a.c); }
inline CF operator+(CF a, CF b) {
    if (a.c && b.c) return mkc(a.v + b.v, true);
    if (is_negzero(b)) return a;
    if (is_negzero(a)) return b;
    ++t_flops;
    return mkc(a.v + b.v, false);
}
inline CF operator-(CF a, CF b) {
    if (a.c && b.c) return mkc(a.v - b.v, true);
    if (is_poszero(b)) return a;
    if (is_negzero(a)) return mkc(-b.v, false);   // (-0.0) - x = -x: sign modifier
    ++t_flops;
    return mkc(a.v - b.v, false);
}
inline CF operator*(CF a, CF b) {
    if (a.c && b.c) return mkc(a.v * b.v, true);
    if ((a.c && a.v == 1.f) || (b.c && b.v == 1.f)) return mkc(a.v * b.v, false);
    if ((a.c && a.v == -1.f) || (b.c && b.v == -1.f)) return mkc(a.v * b.v, false);
    ++t_flops;
    return mkc(a.v * b.v, false);
}

#include "../orphics_amd/csrc/cx.hpp"
namespace oa {
// the device build replaces these three by packed-f32 inline asm for float (cx.hpp): opaque, never folded
inline cx<CF> add_mi(cx<CF> a, cx<CF> b) { t_flops += 2; return mk<CF>(mkc(a.x.v + b.y.v, false), mkc(a.y.v - b.x.v, false)); }
inline cx<CF> add_pi(cx<CF> a, cx<CF> b) { t_flops += 2; return mk<CF>(mkc(a.x.v - b.y.v, false), mkc(a.y.v + b.x.v, false)); }
inline cx<CF> operator*(cx<CF> a, cx<CF> b) {
    t_flops += 6;
    return mk<CF>(mkc(a.x.v * b.x.v - a.y.v * b.y.v, false), mkc(a.x.v * b.y.v + a.y.v * b.x.v, false));
}
}  // namespace oa
#include "../orphics_amd/csrc/fft_plan.hpp"

using namespace oa;

struct EmuCtx {
    int tid_, bx_;
    std::barrier<>* bar;
    char* sm;
    int tid() const { return tid_; }
    int bid_x() const { return bx_; }
    int bid_y() const { return 0; }
    int bid_z() const { return 0; }
    void sync() const { bar->arrive_and_wait(); }
    void wsync() const { bar->arrive_and_wait(); }
    void* smem() const { return sm; }
};

static std::atomic<unsigned long long> g_flops{0};

struct CountLauncher {
    int nz_used = -1;
    int rows_per_wg = 1;
    void fail_rlayout() {}
    // eight points per thread (fft_rowqe8.hpp): ONE workgroup = one row pair
    template <typename T> void row_qe_pair8(int pairs, int M, const RowQeArgs<T>& a) {
        rows_per_wg = 2;
        dispatch_rq8(M, a.win, a.lr, a.chain != nullptr, [&](auto ac, auto nzc, auto lay, auto ch) {
            constexpr int A = decltype(ac)::value;
            nz_used = decltype(nzc)::value;
            const int nt = 64 * A;
            std::vector<char> sm(rq8_lds_bytes<T, A, decltype(ch)::value>() + 64);
            std::barrier<> bar(nt);
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t]() {
                    t_flops = 0;
                    EmuCtx c{t, 0, &bar, sm.data()};
                    row_qe8_body<T, A, decltype(nzc)::value, decltype(lay)::value, decltype(ch)::value>(c, a);
                    g_flops += t_flops;
                });
            for (auto& x : th) x.join();
        });
    }
    template <typename T> void row_qe_pair(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        rows_per_wg = 2;
        dispatch_seq(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_total_log<S>() >= 10 && seq_total_log<S>() <= 13) {
                const int nz = pair_first_stage_nz(a.logL, S::rget(0), a.win);
                nz_used = nz;
                std::vector<char> sm(smem + 64);
                std::barrier<> bar(nt);
                std::vector<std::thread> th;
                dispatch_pair_nz<S>(nz, [&](auto nzc) {
                    for (int t = 0; t < nt; ++t)
                        th.emplace_back([&, t]() {
                            t_flops = 0;
                            EmuCtx c{t, 0, &bar, sm.data()};     // ONE workgroup (block 0): one row pair
                            row_qe_pair_body<T, S, decltype(nzc)::value>(c, a);
                            g_flops += t_flops;
                        });
                });
                for (auto& x : th) x.join();
            }
        });
    }
    template <typename T> void row_qe(int grid, int nt, size_t smem, const RowQeArgs<T>& a) {
        dispatch_seq_qe(a.logL, [&](auto seq) {
            using S = decltype(seq);
            if constexpr (seq_total_log<S>() >= 4) {
                int nz = 0;
                if constexpr (S::n >= 2) nz = qe_first_stage_nz(a.logL, S::rget(0), a.win);
                nz_used = nz;
                rows_per_wg = 1 << a.logC;
                std::vector<char> sm(smem + 64);
                std::barrier<> bar(nt);
                std::vector<std::thread> th;
                dispatch_nz<S>(nz, [&](auto nzc) {
                    for (int t = 0; t < nt; ++t)
                        th.emplace_back([&, t]() {
                            t_flops = 0;
                            EmuCtx c{t, 0, &bar, sm.data()};     // ONE workgroup (block 0): C rows
                            row_qe_body<T, S, decltype(nzc)::value>(c, a);
                            g_flops += t_flops;
                        });
                });
                for (auto& x : th) x.join();
            }
        });
    }
};

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: count_flops N win wout [mrow]\n"); return 2; }
    const int N = atoi(argv[1]);
    int win = atoi(argv[2]), wout = atoi(argv[3]);
    int mrow = argc > 4 ? atoi(argv[4]) : 0;
    if (N < 64 || (N & (N - 1))) { fprintf(stderr, "N must be a power of two >= 64\n"); return 2; }
    Fft2dPlan<CF> p;
    auto tw = make_twiddles<float>(N);
    std::vector<cx<CF>> twx((size_t)N);
    for (int i = 0; i < N; ++i) twx[i] = mk<CF>(CF::run(tw[i].x), CF::run(tw[i].y));
    p.ny = N; p.nx = N; p.logNy = ilog2(N); p.logNx = ilog2(N); p.kp = kpitch_for(N); p.tw_x = twx.data(); p.tw_y = twx.data();
    // per-thread constants of the 8-point row stage's grids, as run-time (non-foldable) values
    std::vector<cx<CF>> rq8t[RQ8_NGRIDS];
    const int* waves = RQ8_WAVES;
    for (int i = 0; i < RQ8_NGRIDS; ++i)
        if (512 * waves[i] <= N) {
            const auto tf = rq8_make_consts<float>(waves[i]);
            rq8t[i].resize(tf.size());
            for (size_t k = 0; k < tf.size(); ++k) rq8t[i][k] = mk<CF>(CF::run(tf[k].x), CF::run(tf[k].y));
            p.rq8c[i] = rq8t[i].data();
        }
    if (mrow < 0) {
        mrow = Fft2dPlan<CF>::row_grid_min(N, p.clampw(win), p.clampw(wout));
        if (2L * p.clampw(win) + p.clampw(wout) > mrow) mrow = 0;
    }
    if (mrow > N) mrow = N;
    const int grid = mrow == 0 ? N : mrow;
    const int L = grid / 2;         // (nominal count below: packed transforms of L points; the 1536 grid has no power-of-two L -- the ratio is informative only)
    int C = 4096 / L; if (C < 2) C = 2;
    const size_t rows = (size_t)C;
    std::vector<cx<CF>> gx(rows * p.kp), gy(rows * p.kp), h(rows * p.kp), px(rows * p.kp), py(rows * p.kp);
    for (size_t i = 0; i < gx.size(); ++i) {
        const float a = 0.001f * (float)(i % 977), b = 0.002f * (float)(i % 751);
        gx[i] = mk<CF>(CF::run(a), CF::run(b)); gy[i] = mk<CF>(CF::run(b), CF::run(a)); h[i] = mk<CF>(CF::run(a + b), CF::run(a - b));
    }
    CountLauncher q;
    p.rows_qe(q, gx.data(), gy.data(), h.data(), px.data(), py.data(), CF::run(1.0f), 0, p.clampw(win), p.clampw(wout), mrow);
    C = q.rows_per_wg;
    const double per_row = (double)g_flops.load() / (double)C;
    const double nominal = 5.0 * 5.0 * L * std::log2((double)L);
    printf("{\"n\": %d, \"win\": %d, \"wout\": %d, \"mrow\": %d, \"rows_per_workgroup\": %d, \"first_stage_nz\": %d, \"flops_per_row\": %.1f, "
           "\"nominal_5NlogN_flops_per_row\": %.1f, \"ratio_to_nominal\": %.4f}\n",
           N, p.clampw(win), p.clampw(wout), grid, C, q.nz_used, per_row, nominal, per_row / nominal);
    return 0;
}
