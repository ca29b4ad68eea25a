// One-call entry points of the C-ABI: a plan that knows its estimator filters and its
// radial bins runs a whole reconstruction -- or a whole Monte-Carlo shard -- per call, stream-ordered, without the
// host touching intermediate planes.  They are thin sequencers over the fused passes of fft.hip / bin.hip /
// rng.hip (the same kernels the fine-grained calls launch), plus an RCCL all-reduce for hosts that do not bring
// torch.distributed (librccl is dlopen'ed on first use, so the library has no link-time dependency on it).
#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "common.hpp"
#include "fft_plan.hpp"

namespace oa {

constexpr int MC_BATCH_MAX = 6;      // realisations per launch in oa_mc_run (kappa planes: the six plan-owned work planes in front of kk)

struct Pipeline {
    // filters (caller-owned device planes) + active region of the TT estimator
    const void* FG = nullptr; const void* FH = nullptr; const void* Fn = nullptr;
    int wl = 0, wk = 0, rl = 0, rk = 0, mrow = -1;
    int mcol = -1;   // requested column grid (-1 auto, 0 = the map's own ny rows, > 0 explicit)
    int my = 0;      // resolved: rows the legs / row stage / divergence run on (0 = ny)
    // plan-owned work planes (hc): legs x3, products x2, input transform, kappa
    void* work = nullptr;
    void* c[3] = {nullptr, nullptr, nullptr};
    void* g[2] = {nullptr, nullptr};
    void* kT = nullptr;
    void* kk = nullptr;
    // bins
    const int32_t* ids = nullptr;
    int nids = 0;
    double norm = 1.0;
    void* bin_scratch = nullptr;
    double* sums = nullptr;
    int64_t* counts_full = nullptr;   // data-independent mode counts over the whole plane
    int64_t* counts_tmp = nullptr;
    unsigned* ticket = nullptr;       // last-workgroup ticket of the fused bin + moments launch (bin.hip, BinTail)
    void* split_legs = nullptr;       // oa_qe_tt_splits / oa_qe_mv: pool of compact leg planes
    size_t split_bytes = 0;
    void* mv_rtab = nullptr;          // oa_qe_mv: device table of per-piece row-stage operands (RowQeMap), 64 entries
    std::vector<unsigned long long> mv_rkey;
    void* mc_src = nullptr;           // oa_mc_run: hc planes of a batch of realisations
    int mc_cap = 0;
    void* lens_pool = nullptr;        // oa_lens_maps: transforms + derivative planes (hc and real) of the maps of one call
    size_t lens_bytes = 0;
    // tile-major copies of Fnorm and of the bin ids on the coarse grid of the fused divergence + binning launch (what they were
    // made from: rebuilt when the filters / bins / column grid change)
    void* fn_t = nullptr; int32_t* ids_t = nullptr;
    size_t fn_t_bytes = 0, ids_t_bytes = 0;
    // keyed on a GENERATION bumped by every oa_plan_set_filters / oa_plan_set_bins call, not on the planes' addresses: a caching
    // allocator hands a new estimator the addresses of a freed one, and a caller may refill Fnorm or the ids in place
    unsigned long bind_gen = 1, tab_gen = 0;
    int tab_rows = 0, tab_logc = 0, tab_wk = 0;
    // the packed (FG, FH) table of the R-split column stage (ColFBandArgs::fgh), same generation key
    void* fb_t = nullptr;
    size_t fb_t_bytes = 0;
    unsigned long fb_gen = 0;
    int fb_my = 0, fb_wl = 0, fb_rl = 0;
    void** mv_ftab = nullptr;         // oa_qe_mv: device table of the distinct filter planes (gradient fields, then H fields)
    std::vector<const void*> mv_fkey; // what the table holds
    // oa_plan_set_option (include/orphics_amd.h): which of the equivalent launch sequences the one-call entries run
    int opt_mc_batch = MC_BATCH_MAX;  // realisations per launch in oa_mc_run
    bool opt_mv_batch = true;         // oa_qe_mv / oa_qe_tt_splits: all leg planes / all divergences in one launch each
    bool opt_mv_rowbatch = true;      // oa_qe_mv: the row stage of several pieces per launch
    bool opt_mv_chain = true;         // oa_qe_mv: estimator chains (pieces summed in real space inside one row-stage launch)
    bool opt_divbin = true;           // moment entries: radial binning + moments in the tail of the single-pass divergence launch
    bool opt_win_fused = true;        // oa_mc_run_windowed: C2R x window -> R2C as one row pass (the real map stays in LDS)
};

static size_t plane_bytes(const oa_plan* p) { return (size_t)p->ny * p->kp * 2 * (p->dtype == OA_F32 ? 4 : 8); }

static Pipeline* pipe_of(oa_plan* p) {
    if (!p->pipe) p->pipe = new Pipeline();
    return (Pipeline*)p->pipe;
}

void pipeline_release(oa_plan* p) {
    if (!p || !p->pipe) return;
    Pipeline* q = (Pipeline*)p->pipe;
    if (q->work) (void)hipFree(q->work);
    if (q->bin_scratch) (void)hipFree(q->bin_scratch);
    if (q->sums) (void)hipFree(q->sums);
    if (q->counts_full) (void)hipFree(q->counts_full);
    if (q->counts_tmp) (void)hipFree(q->counts_tmp);
    if (q->ticket) (void)hipFree(q->ticket);
    if (q->split_legs) (void)hipFree(q->split_legs);
    if (q->mv_ftab) (void)hipFree(q->mv_ftab);
    if (q->mc_src) (void)hipFree(q->mc_src);
    if (q->mv_rtab) (void)hipFree(q->mv_rtab);
    if (q->lens_pool) (void)hipFree(q->lens_pool);
    if (q->fn_t) (void)hipFree(q->fn_t);
    if (q->fb_t) (void)hipFree(q->fb_t);
    if (q->ids_t) (void)hipFree(q->ids_t);
    delete q;
    p->pipe = nullptr;
}

static int ensure_work(oa_plan* p, Pipeline* q) {
    if (q->work) return 0;
    const size_t pb = plane_bytes(p);
    OA_HIP(hipMalloc(&q->work, 7 * pb));
    OA_HIP(hipMemset(q->work, 0, 7 * pb));          // kappa plane zero outside its active region from the start
    char* b = (char*)q->work;
    for (int i = 0; i < 3; ++i) q->c[i] = b + i * pb;
    for (int i = 0; i < 2; ++i) q->g[i] = b + (3 + i) * pb;
    q->kT = b + 5 * pb;
    q->kk = b + 6 * pb;
    return plan_ensure_scratch(p, 2 * pb);          // never reallocated inside a stream-ordered call afterwards
}

// pool of compact planes of the multi-map entries (grown on demand; growing synchronises the device once)
static int ensure_pool(Pipeline* q, size_t bytes) {
    if (q->split_bytes >= bytes) return 0;
    if (q->split_legs) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->split_legs); q->split_legs = nullptr; q->split_bytes = 0; }
    OA_HIP(hipMalloc(&q->split_legs, bytes));
    q->split_bytes = bytes;
    return 0;
}

// zero the part of a caller-supplied output plane that the pruned divergence kernel never writes: every 16-byte unit of
// the plane outside (columns < wk) x (band rows), in one streaming launch (two hipMemset2DAsync calls ran at 1.2 TB/s:
// 365 us per 8192^2 plane, a third of an MV reconstruction into a caller-owned plane)
__global__ __launch_bounds__(256) void zero_complement_kernel(uint4* __restrict__ out, int ny, int units_per_row, int wk_units,
                                                              int rk, int odd_col) {
    const int y = blockIdx.y;
    const bool band = (rk <= 0) || y < rk || y > ny - rk;      // rows that hold kappa's active columns
    const int x0 = band ? wk_units : 0;
    uint4* row = out + (size_t)y * units_per_row;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int x = x0 + blockIdx.x * blockDim.x + threadIdx.x; x < units_per_row; x += gridDim.x * blockDim.x) row[x] = z;
    // f32 planes with an odd number of active columns: column wk shares its 16-byte unit with the last active column
    if (band && odd_col >= 0 && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<uint2*>(row)[odd_col] = make_uint2(0u, 0u);
}

static int zero_complement(oa_plan* p, void* out, int wk, int rk, hipStream_t st) {
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8);
    const int per16 = (int)(16 / es);                           // complex elements per 16-byte unit (2 in f32, 1 in f64)
    if (!(wk > 0 && wk < p->kp)) wk = (int)p->kp;
    if (!(rk > 0 && 2L * rk - 1 < p->ny)) rk = 0;
    if (wk >= p->kp && rk == 0) return 0;
    const int units = (int)(p->kp / per16);                     // kp is a multiple of 16 elements
    const int wk_units = (wk + per16 - 1) / per16;              // first unit wholly outside the active columns
    const int odd_col = (wk < p->kp && wk % per16) ? wk : -1;
    hipLaunchKernelGGL(zero_complement_kernel, dim3(4, p->ny), dim3(256), 0, st, (uint4*)out, p->ny, units, wk_units, rk, odd_col);
    OA_LAUNCH_CHECK();
    return 0;
}

// COLUMN GRID.  Legs confined to rows |ky| < rl have real-space products confined to |ky| <= 2 (rl - 1); sampled on
// my >= 2 rl + rk rows no aliased product frequency reaches the kept kappa rows |ky| < rk (same argument as the row
// grid, include/orphics_amd.h), so the inverse column transforms, the row stage and the forward column transforms run
// on my instead of ny rows and return the same kappa_hat rows.
static int resolve_my(oa_plan* p, int mcol, int rl, int rk, int* my_out) {
    *my_out = 0;
    if (mcol == 0 || rl <= 0 || rk <= 0) return 0;
    const long need = std::max(2L * rl + rk, 2L * rk);
    int my = mcol;
    if (my < 0) { my = 64; while (my < need && my < p->ny) my <<= 1; }
    else if (my < need) return fail("column grid < max(2*leg_rows + kappa_rows, 2*kappa_rows) would alias the leg products into the kept rows");
    if (my >= p->ny) return 0;
    if (int rc = plan_ensure_col_grid(p, my)) return rc;
    *my_out = my;
    return 0;
}
static int resolve_col_grid(oa_plan* p, Pipeline* q) { return resolve_my(p, q->mcol, q->rl, q->rk, &q->my); }

}  // namespace oa

using namespace oa;

static int ensure_div_tables(oa_plan* p, oa::Pipeline* q, hipStream_t st);

extern "C" {

int oa_plan_set_col_grid(oa_plan* p, int mcol) {
    OA_REQUIRE(p, "oa_plan_set_col_grid: NULL plan");
    OA_NEED_POW2(p, "oa_plan_set_col_grid");
    OA_REQUIRE(mcol <= 0 || is_pow2(mcol), "oa_plan_set_col_grid: mcol must be -1 (auto), 0 (off) or a power of two");
    Pipeline* q = pipe_of(p);
    q->mcol = mcol;
    return q->FG ? resolve_col_grid(p, q) : 0;     // oa_qe_pol resolves it per call from its own row bands
}

int oa_plan_col_grid(const oa_plan* p) { return (p && p->pipe) ? ((Pipeline*)p->pipe)->my : 0; }

int oa_plan_set_option(oa_plan* p, int option, int value) {
    OA_REQUIRE(p, "oa_plan_set_option: NULL plan");
    Pipeline* q = pipe_of(p);
    switch (option) {
        case OA_OPT_MC_BATCH:
            OA_REQUIRE(value >= 0 && value <= MC_BATCH_MAX, "oa_plan_set_option: OA_OPT_MC_BATCH takes 0 (default) .. 6");
            q->opt_mc_batch = value ? value : MC_BATCH_MAX; return 0;
        case OA_OPT_MV_BATCH: q->opt_mv_batch = value != 0; return 0;
        case OA_OPT_MV_ROWBATCH: q->opt_mv_rowbatch = value != 0; return 0;
        case OA_OPT_MV_CHAIN: q->opt_mv_chain = value != 0; return 0;
        case OA_OPT_DIV_BIN: q->opt_divbin = value != 0; return 0;
        case OA_OPT_WIN_FUSED: q->opt_win_fused = value != 0; return 0;
        default: return fail("oa_plan_set_option: unknown option");
    }
}
int oa_plan_rsplit(const oa_plan* p) {
    if (!p || !p->pipe) return 0;
    const Pipeline* q = (const Pipeline*)p->pipe;
    return q->FG ? (1 << qe_rsplit_lr(p, q->my, q->wl, q->wk, q->mrow)) & ~1 : 0;
}

int oa_plan_div_fused(const oa_plan* p) {
    if (!p || !p->pipe) return 0;
    const Pipeline* q = (const Pipeline*)p->pipe;
    if (!q->FG || !q->ids || !q->opt_divbin) return 0;
    const int rows = q->my ? q->my : p->ny;            // rows of the grid the divergence runs on
    const bool sp = p->dtype == OA_F32 ? Fft2dPlan<float>::single_pass_div() : Fft2dPlan<double>::single_pass_div();
    if (!(sp && (rows == 1024 || rows == 2048 || rows == 4096))) return 0;
    const int logc = div_tile_logc(p, rows);           // columns per 128 KB tile of this grid and precision
    const long tiles = ((long)(q->wk > 0 ? q->wk : p->nx / 2 + 1) + (1 << logc) - 1) >> logc;
    return (tiles * MC_BATCH_MAX * q->nids <= (long)(oa_bin_scratch_bytes(q->nids) / 8) * MC_BATCH_MAX) ? 1 : 0;
}

int oa_plan_set_filters(oa_plan* p, const void* FG, const void* FH, const void* Fnorm, int leg_cols, int kappa_cols,
                        int leg_rows, int kappa_rows, int mrow) {
    OA_REQUIRE(p && FG && FH && Fnorm, "oa_plan_set_filters: NULL argument");
    OA_NEED_POW2(p, "oa_plan_set_filters");
    OA_REQUIRE(p->have_laxes, "oa_plan_set_filters: call oa_plan_set_laxes first");
    Pipeline* q = pipe_of(p);
    q->FG = FG; q->FH = FH; q->Fn = Fnorm;
    ++q->bind_gen;                             // the tile-major copy of Fnorm is repacked by the next call that uses it
    q->wl = leg_cols; q->wk = kappa_cols; q->rl = leg_rows; q->rk = kappa_rows; q->mrow = mrow;
    q->mcol = mrow == 0 ? 0 : -1;             // the map's own grid in x means the map's own grid in y too
    if (int rc = ensure_work(p, q)) return rc;
    return resolve_col_grid(p, q);
}

int oa_plan_set_bins(oa_plan* p, const int32_t* ids_hc, int nids, double norm, void* stream) {
    OA_REQUIRE(p && ids_hc && nids >= 3, "oa_plan_set_bins: bad argument");
    Pipeline* q = pipe_of(p);
    if (int rc = ensure_work(p, q)) return rc;
    if (q->nids != nids) {
        if (q->bin_scratch) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->bin_scratch); (void)hipFree(q->sums); (void)hipFree(q->counts_full); (void)hipFree(q->counts_tmp); }
        const long sb = oa_bin_scratch_bytes(nids);
        OA_REQUIRE(sb > 0, "oa_plan_set_bins: bad nids");
        // (x MC_BATCH_MAX: oa_mc_run bins a batch of realisations per launch)
        OA_HIP(hipMalloc(&q->bin_scratch, (size_t)sb * MC_BATCH_MAX));
        OA_HIP(hipMalloc((void**)&q->sums, MC_BATCH_MAX * nids * sizeof(double)));
        OA_HIP(hipMalloc((void**)&q->counts_full, nids * sizeof(int64_t)));
        OA_HIP(hipMalloc((void**)&q->counts_tmp, MC_BATCH_MAX * nids * sizeof(int64_t)));
    }
    if (!q->ticket) {
        OA_HIP(hipMalloc((void**)&q->ticket, sizeof(unsigned)));
        OA_HIP(hipMemset(q->ticket, 0, sizeof(unsigned)));
    }
    q->ids = ids_hc; q->nids = nids; q->norm = norm;
    ++q->bind_gen;
    // mode counts per bin over the WHOLE plane (the per-call binning visits only kappa's active region)
    if (int rc = oa_bin_power(p->dtype, q->c[0], q->c[0], norm, ids_hc, nullptr, (long)p->ny * p->kp, nids, p->kp, p->nx / 2, q->sums,
                              q->counts_full, nullptr, q->bin_scratch, 0, 0, stream)) return rc;
    return ensure_div_tables(p, q, (hipStream_t)stream);
}

void* oa_plan_kappa(oa_plan* p) { return (p && p->pipe) ? ((Pipeline*)p->pipe)->kk : nullptr; }
const int64_t* oa_plan_bin_counts(oa_plan* p) { return (p && p->pipe) ? ((Pipeline*)p->pipe)->counts_full : nullptr; }

// Binning + moments in the tail of the single-pass divergence launch (fft_divbin.hpp): the request the one-call entries hand to
// the divergence wrappers.  OA_OPT_DIV_BIN = 0: always the separate histogram launches (the path of the other geometries).
static bool divbin_enabled(const Pipeline* q) { return q->opt_divbin; }
// (re)build the tile-major copies of Fnorm / ids for the single-pass divergence launch on this plan's column grid.  Called where the
// filters, the bins and the grid are known (oa_plan_set_bins, and again by make_fuse_tabs if any of them changed since): the first
// build of a size allocates (one device synchronisation), later rebuilds are two small stream-ordered launches.
static int ensure_div_tables(oa_plan* p, Pipeline* q, hipStream_t st) {
    const int rows = q->my;
    if (!(q->Fn && q->ids && (rows == 1024 || rows == 2048 || rows == 4096) && q->wk > 0)) { q->tab_rows = 0; return 0; }
    const int logc = div_tile_logc(p, rows);
    if (q->tab_gen == q->bind_gen && q->tab_rows == rows && q->tab_logc == logc && q->tab_wk == q->wk) return 0;
    const size_t rs = p->dtype == OA_F32 ? 4 : 8;
    const long tiles = ((long)q->wk + (1 << logc) - 1) >> logc, total = (tiles * rows) << logc;
    if (q->fn_t_bytes < (size_t)total * rs) {
        if (q->fn_t) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->fn_t); q->fn_t = nullptr; }
        OA_HIP(hipMalloc(&q->fn_t, (size_t)total * rs));
        q->fn_t_bytes = (size_t)total * rs;
    }
    if (q->ids_t_bytes < (size_t)total * 4) {
        if (q->ids_t) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->ids_t); q->ids_t = nullptr; }
        OA_HIP(hipMalloc((void**)&q->ids_t, (size_t)total * 4));
        q->ids_t_bytes = (size_t)total * 4;
    }
    if (int rc = pack_tiles(p, q->Fn, q->fn_t, rows, logc, q->wk, (int)rs, st)) return rc;
    if (int rc = pack_tiles(p, q->ids, q->ids_t, rows, logc, q->wk, 4, st)) return rc;
    q->tab_gen = q->bind_gen; q->tab_rows = rows; q->tab_logc = logc; q->tab_wk = q->wk;
    return 0;
}
// the packed filter table of the R-split column stage for the bound filters on this plan's column grid, (re)made when the filters, the
// band or the grid changed: nullptr when this geometry does not take the R-split path (or on failure: the kernel then reads the planes)
static const void* fband_table(oa_plan* p, Pipeline* q, hipStream_t st) {
    static const bool off = exp_env("OA_NO_FBAND_TABLE") != nullptr;        // A/B switch
    if (off || !q->FG || q->my <= 0 || !qe_rsplit_lr(p, q->my, q->wl, q->wk, q->mrow)) return nullptr;
    if (q->fb_t && q->fb_gen == q->bind_gen && q->fb_my == q->my && q->fb_wl == q->wl && q->fb_rl == q->rl) return q->fb_t;
    const size_t need = (size_t)qe_fband_table_entries(p, q->wl, q->my) * 2 * (p->dtype == OA_F32 ? 4 : 8);
    if (q->fb_t_bytes < need) {
        if (q->fb_t) { if (hipDeviceSynchronize() != hipSuccess) return nullptr; (void)hipFree(q->fb_t); q->fb_t = nullptr; q->fb_t_bytes = 0; }
        if (hipMalloc(&q->fb_t, need) != hipSuccess) { q->fb_t = nullptr; (void)hipGetLastError(); return nullptr; }
        q->fb_t_bytes = need;
    }
    if (qe_fband_pack_w(p, q->FG, q->FH, q->fb_t, q->wl, q->rl, q->my, st)) return nullptr;
    q->fb_gen = q->bind_gen; q->fb_my = q->my; q->fb_wl = q->wl; q->fb_rl = q->rl;
    return q->fb_t;
}
static DivBinFuse make_fuse(const oa_plan* p, const Pipeline* q, int64_t* n, double* S, double* C, int store) {
    DivBinFuse f{};
    if (q->tab_rows && q->tab_gen == q->bind_gen && q->tab_rows == q->my && q->tab_wk == q->wk) {
        f.ids_t = q->ids_t; f.fn_t = q->fn_t; f.tab_logc = q->tab_logc; f.tab_rows = q->tab_rows;
    }
    f.ids = q->ids; f.ipitch = p->kp; f.pnorm = q->norm; f.nids = q->nids; f.nxh = p->nx / 2;
    f.part = (double*)q->bin_scratch; f.part_cap = (long)(oa_bin_scratch_bytes(q->nids) / (long)sizeof(double)) * MC_BATCH_MAX;
    f.sums = q->sums; f.ticket = q->ticket; f.mcounts = q->counts_full; f.n = n; f.S = S; f.C = C; f.store = store; f.done = false;
    return f;
}

static int qe_tt_impl(oa_plan* p, const void* real_map, const void* kX, const void* kY, void* out_kappa_hc, int zero_outside,
                      void* stream, DivBinFuse* fuse, int rows_done = 0);
int oa_qe_tt(oa_plan* p, const void* real_map, const void* kX, const void* kY, void* out_kappa_hc, int zero_outside,
             void* stream) {
    return qe_tt_impl(p, real_map, kX, kY, out_kappa_hc, zero_outside, stream, nullptr);
}
// rows_done: the row-transformed map already sits on the plan's scratch plane (qe_windowed_rows_w): column stages only, multi-pass
static int qe_tt_impl(oa_plan* p, const void* real_map, const void* kX, const void* kY, void* out_kappa_hc, int zero_outside,
                      void* stream, DivBinFuse* fuse, int rows_done) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG, "oa_qe_tt: call oa_plan_set_filters first");
    OA_REQUIRE((real_map != nullptr) != (kX != nullptr), "oa_qe_tt: pass either a real map or the Fourier-space leg(s)");
    Pipeline* q = (Pipeline*)p->pipe;
    void* out = out_kappa_hc ? out_kappa_hc : q->kk;
    if (out_kappa_hc && zero_outside)
        if (int rc = zero_complement(p, out, q->wk, q->rk, (hipStream_t)stream)) return rc;
    // intermediates live on the plan's compact work planes (Fft2dPlan::work_pitch): only `out` has the caller's pitch
    const long pl = work_pitch(p, q->wl), pk = work_pitch(p, q->wk);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int my = q->my;
    const int lr = (real_map && !rows_done) ? qe_rsplit_lr(p, my, q->wl, q->wk, q->mrow) : 0;     // from a map: R-split row pass + one column kernel
    if (real_map) rc = qe_map_legs_cols_w(p, real_map, q->FG, q->FH, q->c[0], q->c[1], q->c[2], q->wl, q->rl, pl, st, rows_done ? 6 : 7, my, lr,
                                          lr ? fband_table(p, q, st) : nullptr);
    else rc = qe_legs_cols_w(p, kX, kY ? kY : kX, q->FG, q->FH, q->c[0], q->c[1], q->c[2], q->wl, q->rl, pl, st, my);
    if (rc) return rc;
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;   // DFT on my rows = my/ny x the full one
    if ((rc = qe_rows_w(p, q->c[0], q->c[1], q->c[2], q->g[0], q->g[1], s * s * sy, 0, q->wl, q->wk, q->mrow, pl, pk, st, my, lr))) return rc;
    return qe_cols_div_w(p, q->g[0], q->g[1], q->Fn, out, 0, q->wk, q->rk, pk, st, my, fuse);
}

int oa_qe_pol(oa_plan* p, int npieces, const double* host_signs, const void* const* host_FG, const void* const* host_FH,
              const int* host_swap, const void* kX, const void* kY, const void* Fnorm, void* out, int accumulate,
              int leg_cols, int kappa_cols, int leg_rows, int kappa_rows, int mrow, int zero_outside, void* stream) {
    OA_REQUIRE(p && npieces >= 1 && host_signs && host_FG && host_FH && kX && kY && Fnorm && out, "oa_qe_pol: bad argument");
    OA_NEED_POW2(p, "oa_qe_pol");
    Pipeline* q = pipe_of(p);
    if (int rc = ensure_work(p, q)) return rc;
    if (!accumulate && zero_outside && out != q->kk)
        if (int rc = zero_complement(p, out, kappa_cols, kappa_rows, (hipStream_t)stream)) return rc;
    const long pl = work_pitch(p, leg_cols), pk = work_pitch(p, kappa_cols);
    hipStream_t st = (hipStream_t)stream;
    int my = 0;                                     // column grid from this call's row bands (policy: oa_plan_set_col_grid)
    if (int rc = resolve_my(p, mrow == 0 ? 0 : q->mcol, leg_rows, kappa_rows, &my)) return rc;
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;
    // products accumulate in g[0], g[1] over the separable pieces (compact work planes)
    for (int i = 0; i < npieces; ++i) {
        const bool sw = host_swap && host_swap[i];
        int rc = qe_legs_cols_w(p, sw ? kY : kX, sw ? kX : kY, host_FG[i], host_FH[i], q->c[0], q->c[1], q->c[2], leg_cols, leg_rows, pl, st, my);
        if (rc) return rc;
        if ((rc = qe_rows_w(p, q->c[0], q->c[1], q->c[2], q->g[0], q->g[1], host_signs[i] * s * s * sy, i > 0, leg_cols, kappa_cols, mrow, pl, pk, st, my))) return rc;
    }
    return qe_cols_div_w(p, q->g[0], q->g[1], Fnorm, out, accumulate, kappa_cols, kappa_rows, pk, st, my);
}

/* flat_taylens (lensing.py:395-440) of nmaps real maps by ONE deflection field, given as its nearest-pixel shifts and sub-pixel
 * remainders (oa_lens_split): out_m(x) = sum_{a + b < order} dx^a dy^b / (a! b!) D_ab[m](x + shift).  Per call: nmaps R2Cs, then the
 * inverse transforms of all nmaps * nd derivative fields (nd = order (order + 1) / 2 - 1), separably (lens_derivs_impl, fft.hip): per
 * (map, y-derivative order b) the column transform of (i ly)^b k on ONE hc plane and a row launch that takes every x-derivative at
 * its load -- and one gather pass per map.  The planes live in a plan-owned pool (allocated / grown on first use: that call
 * synchronises the device once; oa_plan_release_pools frees it). */
static int lens_maps_impl(oa_plan* p, int nmaps, const void* real_in, long in_stride, const void* hc_in, long hc_stride, double hc_scale, int order,
                          const int32_t* shift_x, const int32_t* shift_y, const void* dx, const void* dy, void* real_out, long out_stride, void* stream) {
    const long rplane = (long)p->ny * p->nx;
    Pipeline* q = pipe_of(p);
    const size_t rs = p->dtype == OA_F32 ? 4 : 8;
    const int nd = order * (order + 1) / 2 - 1;
    const int d00 = hc_in ? 1 : 0;               // transforms in: the undisplaced map is a pool plane too (the first of each map's nd + 1)
    hipStream_t st = (hipStream_t)stream;
    char* realp = nullptr;
    if (nd + d00 > 0) {
        const size_t hcb = plane_bytes(p), rb = (size_t)rplane * rs;
        const int chunk = 1;                      // ONE hc plane: the column-transformed field of the current (map, y-derivative order)
        const size_t nk0 = hc_in ? 0 : (size_t)nmaps;
        const size_t need = nk0 * hcb + (size_t)chunk * hcb + (size_t)nmaps * (nd + d00) * rb;
        if (q->lens_bytes < need) {
            if (q->lens_pool) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->lens_pool); q->lens_pool = nullptr; q->lens_bytes = 0; }
            OA_HIP(hipMalloc(&q->lens_pool, need));
            q->lens_bytes = need;
        }
        char* k0 = (char*)q->lens_pool;
        char* hcp = k0 + nk0 * hcb;
        realp = hcp + (size_t)chunk * hcb;
        if (int rc = qe_lens_derivs_w(p, nmaps, real_in, in_stride, k0, hcp, realp, nd, st, hc_in, hc_stride, hc_scale)) return rc;
    }
    for (int m = 0; m < nmaps; ++m) {
        const char* pm = realp + (size_t)m * (nd + d00) * rplane * rs;
        const void* src = hc_in ? (const void*)pm : (const void*)((const char*)real_in + (size_t)m * in_stride * rs);
        void* dst = (char*)real_out + (size_t)m * out_stride * rs;
        if (int rc = oa_lens_taylor(p, src, nd > 0 ? pm + (size_t)d00 * rplane * rs : nullptr, rplane, order, shift_x, shift_y, dx, dy, dst, stream)) return rc;
    }
    return 0;
}

int oa_lens_maps(oa_plan* p, int nmaps, const void* real_in, long in_stride, int order, const int32_t* shift_x, const int32_t* shift_y,
                 const void* dx, const void* dy, void* real_out, long out_stride, void* stream) {
    OA_REQUIRE(p && real_in && shift_x && shift_y && dx && dy && real_out && nmaps >= 1, "oa_lens_maps: bad argument");
    OA_REQUIRE(p->pow2 || p->mixed, "oa_lens_maps: needs map sides of the form 2^a 3^b 5^c");
    OA_REQUIRE(p->have_laxes, "oa_lens_maps: call oa_plan_set_laxes first");
    OA_REQUIRE(order >= 1 && order <= 8, "oa_lens_maps: order must be 1..8");
    OA_REQUIRE(real_in != real_out, "oa_lens_maps: in-place not supported");
    const long rplane = (long)p->ny * p->nx;
    OA_REQUIRE(in_stride >= rplane && out_stride >= rplane, "oa_lens_maps: plane stride smaller than a plane");
    return lens_maps_impl(p, nmaps, real_in, in_stride, nullptr, 0, 1.0, order, shift_x, shift_y, dx, dy, real_out, out_stride, stream);
}

int oa_lens_maps_hc(oa_plan* p, int nmaps, const void* hc_in, long hc_stride, double scale, int order, const int32_t* shift_x,
                    const int32_t* shift_y, const void* dx, const void* dy, void* real_out, long out_stride, void* stream) {
    OA_REQUIRE(p && hc_in && shift_x && shift_y && dx && dy && real_out && nmaps >= 1, "oa_lens_maps_hc: bad argument");
    OA_REQUIRE(p->pow2 || p->mixed, "oa_lens_maps_hc: needs map sides of the form 2^a 3^b 5^c");
    OA_REQUIRE(p->have_laxes, "oa_lens_maps_hc: call oa_plan_set_laxes first");
    OA_REQUIRE(order >= 1 && order <= 8, "oa_lens_maps_hc: order must be 1..8");
    OA_REQUIRE(hc_stride >= (long)p->ny * p->kp && out_stride >= (long)p->ny * p->nx, "oa_lens_maps_hc: plane stride smaller than a plane");
    return lens_maps_impl(p, nmaps, nullptr, 0, hc_in, hc_stride, scale, order, shift_x, shift_y, dx, dy, real_out, out_stride, stream);
}

/* frees the plan-owned pools that the multi-map entries grow on demand (oa_lens_maps, oa_qe_mv / oa_qe_tt_splits, oa_mc_run): they
 * are reallocated by the next call that needs them.  Synchronises the device. */
int oa_plan_release_pools(oa_plan* p) {
    OA_REQUIRE(p, "oa_plan_release_pools: NULL plan");
    if (!p->pipe) return 0;
    Pipeline* q = (Pipeline*)p->pipe;
    OA_HIP(hipDeviceSynchronize());
    if (q->lens_pool) { (void)hipFree(q->lens_pool); q->lens_pool = nullptr; q->lens_bytes = 0; }
    if (q->split_legs) { (void)hipFree(q->split_legs); q->split_legs = nullptr; q->split_bytes = 0; }
    if (q->mc_src) { (void)hipFree(q->mc_src); q->mc_src = nullptr; q->mc_cap = 0; }
    return 0;
}

int oa_filter_map(oa_plan* p, const void* real_in, const void* filt_hcreal, void* real_out, void* stream) {
    OA_REQUIRE(p && real_in && filt_hcreal && real_out, "oa_filter_map: NULL argument");
    Pipeline* q = pipe_of(p);
    if (int rc = ensure_work(p, q)) return rc;
    int rc = oa_fft_r2c(p, real_in, q->kT, 1.0, 0, 0, stream);
    if (rc) return rc;
    if ((rc = oa_cmul_real(p->dtype, q->kT, filt_hcreal, q->kT, (long)p->ny * p->kp, stream))) return rc;
    return oa_fft_c2r(p, q->kT, real_out, 1.0 / ((double)p->ny * p->nx), 0, stream);
}

// kappa_hat (plan-owned plane) -> bandpower sums over its active region -> n += 1, S += b, C += b b^T (b = bin means)
static int bandpower_moments(oa_plan* p, Pipeline* q, int64_t* n, double* S, double* C, void* stream, const void* kappa = nullptr) {
    // one launch: the last workgroup of the histogram reduces the partials and adds the bandpower vector to n, S, C
    return bin_power_moments(p->dtype, kappa ? kappa : q->kk, q->norm, q->ids, (long)p->ny * p->kp, q->nids, p->kp, p->nx / 2, q->sums, q->counts_tmp,
                             q->bin_scratch, q->wk, q->rk, q->ticket, q->counts_full, n, S, C, (hipStream_t)stream);
}

int oa_qe_tt_moments(oa_plan* p, const void* real_map, int64_t* n, double* S, double* C, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG && ((Pipeline*)p->pipe)->ids, "oa_qe_tt_moments: call oa_plan_set_filters and oa_plan_set_bins first");
    OA_REQUIRE(real_map && n && S && C, "oa_qe_tt_moments: NULL argument");
    Pipeline* q = (Pipeline*)p->pipe;
    if (int rc = ensure_div_tables(p, q, (hipStream_t)stream)) return rc;
    DivBinFuse f = make_fuse(p, q, n, S, C, 0);
    if (int rc = qe_tt_impl(p, real_map, nullptr, nullptr, nullptr, 0, stream, divbin_enabled(q) ? &f : nullptr)) return rc;
    if (f.done) return 0;                      // binned and accumulated in the divergence launch
    return bandpower_moments(p, q, n, S, C, stream);
}

/* Several estimators accumulated into one kappa plane (minimum-variance combination) with every distinct filtered field
 * transformed once: a leg plane is identified by (source transform, filter plane) pointers, so estimators that share a
 * filter plane object share the transform (TE / TB: the gradient of W^TE T cos, sin; EE / EB; TE / EE: E / C^EE cos, sin ...). */
int oa_qe_mv(oa_plan* p, int nest, const int* host_npieces, const double* host_signs, const void* const* host_FG,
             const void* const* host_FH, const int* host_swap, const void* const* host_kX, const void* const* host_kY,
             const void* const* host_Fnorm, void* out, int accumulate, int leg_cols, int kappa_cols, int leg_rows, int kappa_rows,
             int mrow, int zero_outside, void* stream) {
    OA_REQUIRE(p && nest >= 1 && host_npieces && host_signs && host_FG && host_FH && host_kX && host_kY && host_Fnorm && out,
               "oa_qe_mv: bad argument");
    OA_NEED_POW2(p, "oa_qe_mv");
    Pipeline* q = pipe_of(p);
    if (int rc = ensure_work(p, q)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && zero_outside && out != q->kk)
        if (int rc = zero_complement(p, out, kappa_cols, kappa_rows, st)) return rc;
    const long pl = work_pitch(p, leg_cols), pk = work_pitch(p, kappa_cols);
    int my = 0;
    if (int rc = resolve_my(p, mrow == 0 ? 0 : q->mcol, leg_rows, kappa_rows, &my)) return rc;
    // distinct leg planes: gradient pairs (source, FG) and H planes (source, FH)
    struct Key { const void* src; const void* f; };
    std::vector<Key> grad, hpl;
    std::vector<int> gslot, hslot;
    auto slot_of = [](std::vector<Key>& v, const void* src, const void* f) {
        for (size_t i = 0; i < v.size(); ++i) if (v[i].src == src && v[i].f == f) return (int)i;
        v.push_back(Key{src, f});
        return (int)v.size() - 1;
    };
    int total = 0;
    for (int e = 0; e < nest; ++e) {
        OA_REQUIRE(host_npieces[e] >= 1 && host_kX[e] && host_kY[e] && host_Fnorm[e], "oa_qe_mv: bad estimator entry");
        for (int i = 0; i < host_npieces[e]; ++i, ++total) {
            OA_REQUIRE(host_FG[total] && host_FH[total], "oa_qe_mv: NULL filter plane");
            const bool sw = host_swap && host_swap[total];
            gslot.push_back(slot_of(grad, sw ? host_kY[e] : host_kX[e], host_FG[total]));
            hslot.push_back(slot_of(hpl, sw ? host_kX[e] : host_kY[e], host_FH[total]));
        }
    }
    const int ng = (int)grad.size(), nh = (int)hpl.size(), nplanes = 2 * ng + nh;
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8);
    const size_t lb = (size_t)pl * p->ny * es;                      // one compact leg plane
    // pool: leg planes | 2 product planes per estimator | 2 pass-1 planes per estimator (two-pass divergence)
    const size_t lbk = (size_t)pk * p->ny * es;                     // one compact product plane
    if (int rc = ensure_pool(q, nplanes * lb + 4 * (size_t)nest * lbk)) return rc;
    char* const prod = (char*)q->split_legs + nplanes * lb;
    char* const tmp = prod + 2 * (size_t)nest * lbk;
    auto plane = [&](int k) { return (void*)((char*)q->split_legs + (size_t)k * lb); };      // gradient pair g: 2g, 2g + 1; H plane h: 2 ng + h
    // all leg planes in ONE inverse pass-1 launch when the fields come from at most three sources (T, E, B)
    std::vector<const void*> srcs;
    unsigned long long srcsel = 0;
    bool batch = ng + nh <= 32 && q->opt_mv_batch;                  // (OA_OPT_MV_BATCH = 0: one launch per field)
    int legs_done = 0;
    for (int f = 0; f < ng + nh && batch; ++f) {
        const void* sp = f < ng ? grad[f].src : hpl[f - ng].src;
        size_t k = 0;
        while (k < srcs.size() && srcs[k] != sp) ++k;
        if (k == srcs.size()) srcs.push_back(sp);
        if (srcs.size() > 3) batch = false;
        srcsel |= (unsigned long long)k << (2 * f);
    }
    if (batch) {
        std::vector<const void*> key;
        for (int g = 0; g < ng; ++g) key.push_back(grad[g].f);
        for (int h = 0; h < nh; ++h) key.push_back(hpl[h].f);
        if (!q->mv_ftab) OA_HIP(hipMalloc((void**)&q->mv_ftab, 32 * sizeof(void*)));
        if (key != q->mv_fkey) {       // (pageable source: staged before the call returns; ordered on this stream)
            OA_HIP(hipMemcpyAsync(q->mv_ftab, key.data(), key.size() * sizeof(void*), hipMemcpyHostToDevice, st));
            q->mv_fkey = key;
        }
        const long off1 = srcs.size() > 1 ? (long)(((const char*)srcs[1] - (const char*)srcs[0]) / (long)es) : 0;
        const long off2 = srcs.size() > 2 ? (long)(((const char*)srcs[2] - (const char*)srcs[0]) / (long)es) : 0;
        bool aligned = true;
        for (size_t k = 1; k < srcs.size(); ++k) aligned = aligned && (((const char*)srcs[k] - (const char*)srcs[0]) % (long)es == 0);
        if (aligned) {
            if (int rc = qe_legs_batch_w(p, srcs[0], off1, off2, srcsel, (const void* const*)q->mv_ftab, ng, nh, q->split_legs, (long)(lb / es),
                                         leg_cols, leg_rows, pl, st, my, 2, &legs_done)) return rc;
        } else batch = false;
    }
    if (!batch) {
        for (int g = 0; g < ng; ++g)
            if (int rc = qe_legs_subset_w(p, grad[g].src, grad[g].f, plane(2 * g), plane(2 * g + 1), 2, leg_cols, leg_rows, pl, st, my)) return rc;
        for (int h = 0; h < nh; ++h)
            if (int rc = qe_legs_subset_w(p, hpl[h].src, hpl[h].f, plane(2 * ng + h), nullptr, 1, leg_cols, leg_rows, pl, st, my)) return rc;
    }
    if (!legs_done)                  // (single-pass leg kernel on 1024- / 2048-row column grids: nothing left to do)
        if (int rc = qe_legs_pass2_w(p, q->split_legs, nplanes, (long)(lb / es), leg_cols, pl, st, my)) return rc;
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;
    // divergence of all estimators in ONE launch when their Fnorm planes are evenly spaced (one stacked allocation): each
    // estimator's weighted kappa goes to its own plan-owned plane (c[0..2], g[0..1], kT are contiguous and unused here),
    // then one pass sums them in estimator order
    const size_t rs = es / 2, pb = plane_bytes(p);
    long fn_moff = 0;
    bool dbatch = nest >= 2 && nest <= 6 && q->opt_mv_batch;
    if (dbatch) {
        const long d = (long)((const char*)host_Fnorm[1] - (const char*)host_Fnorm[0]);
        dbatch = d > 0 && d % (long)rs == 0;
        for (int e = 2; e < nest && dbatch; ++e) dbatch = ((const char*)host_Fnorm[e] - (const char*)host_Fnorm[0]) == e * d;
        fn_moff = d / (long)rs;
    }
    // ROW STAGE: the k-th separable piece of every estimator in ONE launch (they write different product planes; a launch
    // of one piece is 1024 workgroups of two waves and leaves most of the chip's wave slots empty), per-piece planes and
    // scales through a device table; pieces k > 0 accumulate.  Same arithmetic per piece, same order per estimator.
    bool rbatch = dbatch && total <= 64 && q->opt_mv_rowbatch;
    if (rbatch) {
        int maxp = 0;
        for (int e = 0; e < nest; ++e) maxp = std::max(maxp, host_npieces[e]);
        std::vector<const void*> tgx, tgy, th;
        std::vector<void*> tpx, tpy;
        std::vector<double> tsc;
        std::vector<int> count(maxp, 0);
        for (int k = 0; k < maxp; ++k) {
            int base = 0;
            for (int e = 0; e < nest; base += host_npieces[e], ++e) {
                if (host_npieces[e] <= k) continue;
                const int idx = base + k;
                tgx.push_back(plane(2 * gslot[idx])); tgy.push_back(plane(2 * gslot[idx] + 1)); th.push_back(plane(2 * ng + hslot[idx]));
                tpx.push_back(prod + 2 * (size_t)e * lbk); tpy.push_back(prod + (2 * (size_t)e + 1) * lbk);
                tsc.push_back(host_signs[idx] * s * s * sy);
                ++count[k];
            }
        }
        std::vector<unsigned long long> key;
        for (size_t i = 0; i < tgx.size(); ++i) {
            unsigned long long bits;
            memcpy(&bits, &tsc[i], sizeof bits);
            key.insert(key.end(), {(unsigned long long)(uintptr_t)tgx[i], (unsigned long long)(uintptr_t)tgy[i], (unsigned long long)(uintptr_t)th[i],
                                   (unsigned long long)(uintptr_t)tpx[i], (unsigned long long)(uintptr_t)tpy[i], bits});
        }
        key.push_back((unsigned long long)my); key.push_back((unsigned long long)(unsigned)mrow); key.push_back((unsigned long long)p->dtype);
        const size_t eb = qe_rows_table_entry_bytes(p);
        if (!q->mv_rtab) OA_HIP(hipMalloc(&q->mv_rtab, 64 * 64 + 1024));
        const int upload = key != q->mv_rkey;
        size_t off = 0;
        // ESTIMATOR CHAINS: one launch for all estimators, each workgroup loops over its estimator's pieces and keeps the
        // summed products in registers (3 n + 2 transforms per row pair instead of 5 n; no read-modify-write of the product
        // planes).  OA_OPT_MV_CHAIN = 0: the piece-by-piece launches below
        bool chained = false;
        if (q->opt_mv_chain) {
            std::vector<const void*> cgx, cgy, chh;
            std::vector<void*> cpx, cpy;
            std::vector<double> csc;
            std::vector<int> first(nest), cnt(nest);
            int base = 0;
            for (int e = 0; e < nest; base += host_npieces[e], ++e) {
                first[e] = (int)cgx.size(); cnt[e] = host_npieces[e];
                for (int i = 0; i < host_npieces[e]; ++i) {
                    const int idx = base + i;
                    cgx.push_back(plane(2 * gslot[idx])); cgy.push_back(plane(2 * gslot[idx] + 1)); chh.push_back(plane(2 * ng + hslot[idx]));
                    cpx.push_back(prod + 2 * (size_t)e * lbk); cpy.push_back(prod + (2 * (size_t)e + 1) * lbk);
                    csc.push_back(host_signs[idx] * s * s * sy);
                }
            }
            std::vector<unsigned long long> ckey = key;
            ckey.push_back(0xC4A1ull);
            const int up = ckey != q->mv_rkey;
            int rc = qe_rows_chain_w(p, nest, total, cgx.data(), cgy.data(), chh.data(), cpx.data(), cpy.data(), csc.data(), first.data(), cnt.data(),
                                     q->mv_rtab, up, leg_cols, kappa_cols, mrow, pl, pk, st, my);
            if (rc > 0) return rc;
            if (rc == 0) { chained = true; q->mv_rkey = ckey; }
        }
        for (int k = 0; k < maxp && rbatch && !chained; ++k) {
            int rc = qe_rows_table_w(p, count[k], tgx.data() + off, tgy.data() + off, th.data() + off, tpx.data() + off, tpy.data() + off,
                                     tsc.data() + off, (char*)q->mv_rtab + off * eb, upload, k > 0, leg_cols, kappa_cols, mrow, pl, pk, st, my);
            if (rc < 0) { rbatch = false; break; }      // (only possible at k = 0: nothing launched yet)
            if (rc) return rc;
            off += count[k];
        }
        if (rbatch && !chained) q->mv_rkey = key;
    }
    int at = 0;
    for (int e = 0; e < nest && !rbatch; ++e) {
        void* g0 = dbatch ? (void*)(prod + 2 * (size_t)e * lbk) : q->g[0];
        void* g1 = dbatch ? (void*)(prod + (2 * (size_t)e + 1) * lbk) : q->g[1];
        for (int i = 0; i < host_npieces[e]; ++i, ++at) {
            int rc = qe_rows_w(p, plane(2 * gslot[at]), plane(2 * gslot[at] + 1), plane(2 * ng + hslot[at]), g0, g1,
                               host_signs[at] * s * s * sy, i > 0, leg_cols, kappa_cols, mrow, pl, pk, st, my);
            if (rc) return rc;
        }
        if (!dbatch)
            if (int rc = qe_cols_div_w(p, g0, g1, host_Fnorm[e], out, (accumulate || e > 0) ? 1 : 0, kappa_cols, kappa_rows, pk, st, my)) return rc;
    }
    if (dbatch) {
        if (int rc = qe_cols_div_batch_w(p, prod, prod + lbk, host_Fnorm[0], q->c[0], tmp, nest, (long)(2 * lbk / es), fn_moff, (long)(pb / es),
                                         kappa_cols, kappa_rows, pk, st, my)) return rc;
        // (the divergence writes columns <= nx/2 only: the planes' row padding beyond holds whatever the work planes held)
        const int wsum = (kappa_cols > 0 && kappa_cols <= p->nx / 2 + 1) ? kappa_cols : p->nx / 2 + 1;
        return sum_region(p->dtype, q->c[0], (long)(pb / es), nest, out, accumulate ? 1 : 0, p->ny, p->kp, wsum, kappa_rows, st);
    }
    return 0;
}

int oa_qe_tt_splits(oa_plan* p, int nsplits, const void* const* host_kmaps, void* const* host_out, int zero_outside, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG, "oa_qe_tt_splits: call oa_plan_set_filters first");
    OA_REQUIRE(nsplits >= 1 && nsplits <= 64 && host_kmaps && host_out, "oa_qe_tt_splits: bad argument");
    Pipeline* q = (Pipeline*)p->pipe;
    hipStream_t st = (hipStream_t)stream;
    const long pl = work_pitch(p, q->wl), pk = work_pitch(p, q->wk);
    const size_t lb = (size_t)pl * p->ny * 2 * (p->dtype == OA_F32 ? 4 : 8);      // one compact leg plane
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8), lbk = (size_t)pk * p->ny * es;
    const int npairs = nsplits * nsplits;
    // evenly spaced output planes (one (n, n, Ny, kp) block): the divergence of all pairs runs as ONE launch
    bool dbatch = npairs >= 2 && q->opt_mv_batch;
    long out_moff = 0;
    if (dbatch) {
        const long d = (long)((char*)host_out[1] - (char*)host_out[0]);
        dbatch = d > 0 && d % (long)es == 0;
        for (int k = 2; k < npairs && dbatch; ++k) dbatch = host_out[k] && ((char*)host_out[k] - (char*)host_out[0]) == k * d;
        out_moff = d / (long)es;
    }
    // pool: 3 leg planes per split | 2 product planes per pair | 2 pass-1 planes per pair (two-pass divergence)
    if (int rc = ensure_pool(q, 3 * lb * nsplits + (dbatch ? 4 * (size_t)npairs * lbk : 0))) return rc;
    char* const prod = (char*)q->split_legs + 3 * lb * nsplits;
    const int my = q->my;
    auto leg = [&](int i, int c) { return (void*)((char*)q->split_legs + (3 * (size_t)i + c) * lb); };
    for (int i = 0; i < nsplits; ++i) {
        OA_REQUIRE(host_kmaps[i], "oa_qe_tt_splits: NULL split plane");
        if (int rc = qe_legs_cols_w(p, host_kmaps[i], host_kmaps[i], q->FG, q->FH, leg(i, 0), leg(i, 1), leg(i, 2), q->wl, q->rl, pl, st, my)) return rc;
    }
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;
    for (int i = 0; i < nsplits; ++i)
        for (int j = 0; j < nsplits; ++j) {
            const int k = i * nsplits + j;
            void* out = host_out[k];
            OA_REQUIRE(out, "oa_qe_tt_splits: NULL output plane");
            int rc;
            if (zero_outside && (rc = zero_complement(p, out, q->wk, q->rk, st))) return rc;
            void* g0 = dbatch ? (void*)(prod + 2 * (size_t)k * lbk) : q->g[0];
            void* g1 = dbatch ? (void*)(prod + (2 * (size_t)k + 1) * lbk) : q->g[1];
            if ((rc = qe_rows_w(p, leg(i, 0), leg(i, 1), leg(j, 2), g0, g1, s * s * sy, 0, q->wl, q->wk, q->mrow, pl, pk, st, my))) return rc;
            if (!dbatch && (rc = qe_cols_div_w(p, g0, g1, q->Fn, out, 0, q->wk, q->rk, pk, st, my))) return rc;
        }
    if (dbatch)
        return qe_cols_div_batch_w(p, prod, prod + lbk, q->Fn, host_out[0], prod + 2 * (size_t)npairs * lbk, npairs, (long)(2 * lbk / es), 0,
                                   out_moff, q->wk, q->rk, pk, st, my);
    return 0;
}

/* Two Monte-Carlo steps in one call: both maps share every launch behind their row transforms (fft.hip qe_tt_pair_impl);
 * geometries without that path run the two steps one after the other.  n += 2, S += b0 + b1, C += b0 b0^T + b1 b1^T. */
int oa_qe_tt_moments2(oa_plan* p, const void* real_map0, const void* real_map1, int64_t* n, double* S, double* C, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG && ((Pipeline*)p->pipe)->ids, "oa_qe_tt_moments2: call oa_plan_set_filters and oa_plan_set_bins first");
    OA_REQUIRE(real_map0 && real_map1 && n && S && C, "oa_qe_tt_moments2: NULL argument");
    Pipeline* q = (Pipeline*)p->pipe;
    const long pl = work_pitch(p, q->wl), pk = work_pitch(p, q->wk);
    // second kappa plane: the plan-owned input-transform plane (unused on the from-map path); only kappa's active region of
    // it is ever read back (binning)
    if (int rc = ensure_div_tables(p, q, (hipStream_t)stream)) return rc;
    DivBinFuse f = make_fuse(p, q, n, S, C, 0);
    int rc = qe_tt_pair_w(p, real_map0, real_map1, q->FG, q->FH, q->Fn, q->c[0], q->c[1], q->c[2], q->g[0], q->g[1], q->kk, q->kT, q->wl,
                          q->wk, q->rl, q->rk, q->mrow, q->my, pl, pk, (hipStream_t)stream, divbin_enabled(q) ? &f : nullptr,
                          fband_table(p, q, (hipStream_t)stream));
    if (rc > 0) return rc;
    if (rc == 0 && f.done) return 0;           // both maps binned and accumulated (map order) in the divergence launch
    if (rc < 0) {
        if ((rc = oa_qe_tt_moments(p, real_map0, n, S, C, stream))) return rc;
        return oa_qe_tt_moments(p, real_map1, n, S, C, stream);
    }
    // both kappa planes binned in one launch pair (grid y = map; kT sits one plane IN FRONT of kk: stride -1 plane), the moment
    // tail adds the two bandpower vectors in map order
    const long es = 2 * (p->dtype == OA_F32 ? 4 : 8);
    const long back = ((const char*)q->kT - (const char*)q->kk) / es;
    return bin_power_moments(p->dtype, q->kk, q->norm, q->ids, (long)p->ny * p->kp, q->nids, p->kp, p->nx / 2, q->sums, q->counts_tmp,
                             q->bin_scratch, q->wk, q->rk, q->ticket, q->counts_full, n, S, C, (hipStream_t)stream, 2, back);
}

/* One stage of oa_qe_tt_moments on the plan's own work planes, for per-kernel timing (bench.py):
 * 0 = row R2C of the map, 1 = forward column pass 1, 2 = fused forward pass 2 + leg filters + inverse pass 1 and the
 * 3-plane inverse pass 2, 3 = fused row stage, 4 = 2-plane forward pass 1 + divergence kernel, 5 = binned power +
 * moment accumulation (into plan-owned dummies).  Stages read what the previous ones left in the work planes. */
int oa_qe_tt_stage(oa_plan* p, int stage, const void* real_map, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG, "oa_qe_tt_stage: call oa_plan_set_filters first");
    Pipeline* q = (Pipeline*)p->pipe;
    const long pl = work_pitch(p, q->wl), pk = work_pitch(p, q->wk);
    hipStream_t st = (hipStream_t)stream;
    const int my = q->my;
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;
    const int lr = qe_rsplit_lr(p, my, q->wl, q->wk, q->mrow);
    switch (stage) {
        case 0: case 1: case 2:
            OA_REQUIRE(real_map, "oa_qe_tt_stage: stages 0-2 need the map");
            return qe_map_legs_cols_w(p, real_map, q->FG, q->FH, q->c[0], q->c[1], q->c[2], q->wl, q->rl, pl, st, 1 << stage, my, lr,
                                      lr ? fband_table(p, q, st) : nullptr);
        case 3: return qe_rows_w(p, q->c[0], q->c[1], q->c[2], q->g[0], q->g[1], s * s * sy, 0, q->wl, q->wk, q->mrow, pl, pk, st, my, lr);
        case 4: {
            if (q->ids && divbin_enabled(q)) {      // as the one-call entries: binning + moments (into dummies) in the divergence launch
                DivBinFuse f = make_fuse(p, q, (int64_t*)q->kT, (double*)q->kT + 8, (double*)q->kT + 8 + q->nids, 0);
                return qe_cols_div_w(p, q->g[0], q->g[1], q->Fn, q->kk, 0, q->wk, q->rk, pk, st, my, &f);
            }
            return qe_cols_div_w(p, q->g[0], q->g[1], q->Fn, q->kk, 0, q->wk, q->rk, pk, st, my);
        }
        case 5: {
            OA_REQUIRE(q->ids, "oa_qe_tt_stage: stage 5 needs oa_plan_set_bins");
            if (oa_plan_div_fused(p)) return 0;    // nothing left to do: stage 4 binned
            // dummies: the tail of the (nids-long) sums / counts_tmp buffers is not large enough for C: use the kT plane
            int64_t* n = (int64_t*)q->kT;
            double* S = (double*)q->kT + 8;
            double* C = S + q->nids;
            return bandpower_moments(p, q, n, S, C, stream);
        }
        default: return fail("oa_qe_tt_stage: stage must be 0..5");
    }
}

}  // extern "C"

namespace oa {
static int ensure_mc_src(oa_plan* p, Pipeline* q) {
    if (q->mc_cap >= MC_BATCH_MAX) return 0;
    if (q->mc_src) { OA_HIP(hipDeviceSynchronize()); (void)hipFree(q->mc_src); q->mc_src = nullptr; q->mc_cap = 0; }
    OA_HIP(hipMalloc(&q->mc_src, (size_t)MC_BATCH_MAX * plane_bytes(p)));
    q->mc_cap = MC_BATCH_MAX;
    return 0;
}
// pool bytes of a batch of B realisations: 3 B leg planes (gx_b, gy_b at 2b, 2b + 1; h_b at 2B + b) | 2 B product planes | 2 B pass-1 planes
static size_t mc_pool_bytes(const oa_plan* p, const Pipeline* q, int B) {
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8);
    return 3 * (size_t)B * work_pitch(p, q->wl) * p->ny * es + 4 * (size_t)B * work_pitch(p, q->wk) * p->ny * es;
}
/* Everything behind the transforms of a BATCH of B realisations (q->mc_src: their hc planes, leg band filled): leg planes, row
 * stage, divergence, binned power + moments in realisation order, mean-field stack -- each ONE launch for the batch (oa_mc_run and,
 * behind its windowed front end, oa_mc_run_windowed).  *fallback = 1: this geometry's row stage takes one map per launch (nothing
 * was launched). */
static int mc_batch_tail(oa_plan* p, Pipeline* q, int B, int64_t* n, double* S, double* C, double* meanfield_acc, hipStream_t st, int* fallback) {
    *fallback = 0;
    const long pl = work_pitch(p, q->wl), pk = work_pitch(p, q->wk);
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8), pb = plane_bytes(p);
    const size_t lb = (size_t)pl * p->ny * es, lbk = (size_t)pk * p->ny * es;
    const int my = q->my;
    if (int rc = ensure_pool(q, mc_pool_bytes(p, q, B))) return rc;
    char* const legs = (char*)q->split_legs;
    char* const prod = legs + 3 * (size_t)B * lb;
    char* const tmp = prod + 2 * (size_t)B * lbk;
    std::vector<const void*> key;
    for (int b = 0; b < B; ++b) key.push_back(q->FG);
    for (int b = 0; b < B; ++b) key.push_back(q->FH);
    if (!q->mv_ftab) OA_HIP(hipMalloc((void**)&q->mv_ftab, 32 * sizeof(void*)));
    if (key != q->mv_fkey) {
        OA_HIP(hipMemcpyAsync(q->mv_ftab, key.data(), key.size() * sizeof(void*), hipMemcpyHostToDevice, st));
        q->mv_fkey = key;
    }
    unsigned long long sel = 0;                       // field f (gradient fields 0..B-1, H fields B..2B-1) reads realisation f mod B
    for (int f = 0; f < 2 * B; ++f) sel |= (unsigned long long)(f % B) << (4 * f);
    // (the row stage's geometry check first: nothing may have been launched when this batch falls back)
    const double s = 1.0 / ((double)p->ny * p->nx), sy = my ? (double)p->ny / my : 1.0;
    int legs_done = 0, rc;
    if ((rc = qe_legs_batch_w(p, q->mc_src, (long)(pb / es), 0, sel, (const void* const*)q->mv_ftab, B, B, legs, (long)(lb / es), q->wl, q->rl,
                              pl, st, my, 4, &legs_done))) return rc;
    if (!legs_done && (rc = qe_legs_pass2_w(p, legs, 3 * B, (long)(lb / es), q->wl, pl, st, my))) return rc;
    rc = qe_rows_batch_w(p, legs, legs + lb, legs + 2 * (size_t)B * lb, prod, prod + lbk, s * s * sy, q->wl, q->wk, q->mrow, pl, pk, st, my, B,
                         (long)(2 * lb / es), (long)(lb / es), (long)(2 * lbk / es));
    if (rc < 0) { *fallback = 1; return 0; }          // this geometry's row stage takes one map per launch: one-by-one loop
    if (rc) return rc;
    if ((rc = ensure_div_tables(p, q, st))) return rc;
    DivBinFuse f = make_fuse(p, q, n, S, C, meanfield_acc ? 1 : 0);
    if ((rc = qe_cols_div_batch_w(p, prod, prod + lbk, q->Fn, q->c[0], tmp, B, (long)(2 * lbk / es), 0, (long)(pb / es), q->wk, q->rk, pk, st, my,
                                  divbin_enabled(q) ? &f : nullptr)))
        return rc;
    // binned power of the B kappa planes + their moment updates in realisation order: in the divergence launch, else two
    // launches; the mean-field stack: one
    if (!f.done && (rc = bin_power_moments(p->dtype, q->c[0], q->norm, q->ids, (long)p->ny * p->kp, q->nids, p->kp, p->nx / 2, q->sums, q->counts_tmp,
                                q->bin_scratch, q->wk, q->rk, q->ticket, q->counts_full, n, S, C, st, B, (long)(pb / es)))) return rc;
    if (meanfield_acc && (rc = stack_add_region(p->dtype, q->c[0], meanfield_acc, p->ny, p->kp, q->wk, q->rk, st, B, (long)(2 * pb / es)))) return rc;
    return 0;
}
}  // namespace oa

extern "C" {

int oa_mc_run(oa_plan* p, uint64_t base_seed, long sim_lo, long sim_hi, const void* covsqrt_hc, int64_t* n, double* S, double* C,
              double* meanfield_acc, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG && ((Pipeline*)p->pipe)->ids, "oa_mc_run: call oa_plan_set_filters and oa_plan_set_bins first");
    OA_REQUIRE(covsqrt_hc && n && S && C && sim_hi >= sim_lo, "oa_mc_run: bad argument");
    Pipeline* q = (Pipeline*)p->pipe;
    hipStream_t st = (hipStream_t)stream;
    long i = sim_lo;
    // BATCHES of realisations: at 4096^2 a realisation is ~60 MB of traffic behind ~10 launches, i.e. launch latency; with B
    // realisations per launch (grid z) the column and row stages fill the chip.  Same kernels on the same operands in the same
    // order per realisation as the one-by-one loop below: identical moments.
    const int BMAX = std::max(1, std::min(MC_BATCH_MAX, q->opt_mc_batch));     // (OA_OPT_MC_BATCH) 1 / 2 / 4 / 6 per launch at 4096^2: 16.5 / 26.0 / 37.6 / 40.3 k realisations/s
    const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8), pb = plane_bytes(p);
    // (a batch of ONE goes through the same launches -- OA_OPT_MC_BATCH = 1 and the last realisation of an odd shard: the same kernels
    // whatever the batch size, so the moments do not depend on it; the loop further down serves the geometries without them)
    bool batched = p->pow2;
    while (batched && i < sim_hi) {
        const int B = (int)std::min<long>(BMAX, sim_hi - i);
        if (int rc = ensure_mc_src(p, q)) return rc;
        if (int rc = ensure_pool(q, mc_pool_bytes(p, q, B))) return rc;       // (before the draw: growing the pool synchronises the device)
        // the first batch of a geometry without the batched row stage must not have drawn anything it then abandons: the tail checks
        // before its first launch only AFTER the leg launches -- those write plan-owned planes only, harmless
        int rc = grf_hc_band_batch(p, base_seed, (uint64_t)i, B, covsqrt_hc, q->mc_src, (long)(pb / es), q->wl, q->rl, st);
        if (rc) return rc;
        int fallback = 0;
        if ((rc = mc_batch_tail(p, q, B, n, S, C, meanfield_acc, st, &fallback))) return rc;
        if (fallback) { batched = false; break; }
        i += B;
    }
    for (; i < sim_hi; ++i) {
        // only the leg band of the realisation is ever read (col_legs: columns < wl, rows |ky index| < rl)
        int rc = oa_grf_hc_band(p, base_seed, (uint64_t)i, covsqrt_hc, q->kT, q->wl, q->rl, stream);
        if (rc) return rc;
        if ((rc = oa_qe_tt(p, nullptr, q->kT, nullptr, nullptr, 0, stream))) return rc;
        if ((rc = bandpower_moments(p, q, n, S, C, stream))) return rc;
        // kappa_hat vanishes outside its active region (the plan-owned plane was zero-filled once): stack only that
        if (meanfield_acc && (rc = stack_add_region(p->dtype, q->kk, meanfield_acc, p->ny, p->kp, q->wk, q->rk, (hipStream_t)stream))) return rc;
    }
    return 0;
}

/* Monte-Carlo shard with a REAL-SPACE WINDOW (the reference's analysis flow multiplies every map by its apodisation taper
 * before any transform: maps.py:1873-1878 get_taper, maps.py:1350-1361 binned_power(imap * mask) / mean(mask^2)): per
 * realisation a FULL-plane Philox draw (key = (base_seed, sim), the same counters as oa_mc_run's band draw) -> C2R / Npix ->
 * x window -> TT estimator from the real map (row R2C on the active columns ...) -> bandpower moments (+ mean-field stack:
 * with a window the ensemble mean of kappa_hat no longer vanishes -- that is the mean field Statistics.add_stack exists
 * for, stats.py:1123-1150).  The real map lives in the plan's first leg plane, which the estimator overwrites only after its
 * row pass has consumed the map. */
int oa_mc_run_windowed(oa_plan* p, uint64_t base_seed, long sim_lo, long sim_hi, const void* covsqrt_hc, const void* window_real,
                       int64_t* n, double* S, double* C, double* meanfield_acc, void* stream) {
    OA_REQUIRE(p && p->pipe && ((Pipeline*)p->pipe)->FG && ((Pipeline*)p->pipe)->ids, "oa_mc_run_windowed: call oa_plan_set_filters and oa_plan_set_bins first");
    OA_REQUIRE(covsqrt_hc && window_real && n && S && C && sim_hi >= sim_lo, "oa_mc_run_windowed: bad argument");
    OA_NEED_POW2(p, "oa_mc_run_windowed");
    Pipeline* q = (Pipeline*)p->pipe;
    void* tmap = q->c[0];
    const double inv = 1.0 / ((double)p->ny * p->nx);
    const long pl = work_pitch(p, q->wl);
    long i = sim_lo;
    // BATCHES (fused row pass only): per realisation the full-plane part -- draw, inverse columns, C2R x window -> R2C rows onto a
    // compact plane --, then for the batch ONE forward column transform onto the leg band of its hc planes and the launches of
    // oa_mc_run's batches behind it (legs, row stage, divergence + binning + moments, stack): the latency-bound coarse-grid work of a
    // realisation (70 of its 205 us at 4096^2 float) is shared by up to six
    {
        hipStream_t st = (hipStream_t)stream;
        const size_t es = 2 * (p->dtype == OA_F32 ? 4 : 8), pb = plane_bytes(p), lb = (size_t)pl * p->ny * es;
        const int BMAX = std::max(1, std::min(MC_BATCH_MAX, q->opt_mc_batch));
        bool batched = q->opt_win_fused && p->pow2;
        while (batched && i < sim_hi) {
            const int B = (int)std::min<long>(BMAX, sim_hi - i);
            if (int rc = ensure_mc_src(p, q)) return rc;
            const size_t tail = mc_pool_bytes(p, q, B);
            if (int rc = ensure_pool(q, tail + (size_t)B * lb)) return rc;      // the row-transformed planes sit behind the tail's region
            char* const rowT = (char*)q->split_legs + tail;
            for (int b = 0; b < B; ++b) {
                int rc = oa_grf_hc(p, base_seed, (uint64_t)(i + b), covsqrt_hc, q->kT, stream);
                if (rc) return rc;
                if ((rc = qe_windowed_rows_w(p, q->kT, tmap, window_real, q->wl, pl, inv, st, rowT + (size_t)b * lb))) return rc;
            }
            int rc = qe_fwd_cols_batch_w(p, rowT, pl, q->mc_src, B, (long)(lb / es), (long)(pb / es), q->wl, q->rl, st);
            if (rc) return rc;
            int fallback = 0;
            if ((rc = mc_batch_tail(p, q, B, n, S, C, meanfield_acc, st, &fallback))) return rc;
            if (fallback) { batched = false; break; }         // (geometry without the batched row stage: one by one below)
            i += B;
        }
    }
    for (; i < sim_hi; ++i) {
        int rc = oa_grf_hc(p, base_seed, (uint64_t)i, covsqrt_hc, q->kT, stream);
        if (rc) return rc;
        // FUSED ROW PASS (default): inverse columns of the drawn spectrum (into the first leg plane, free until the leg stage), then
        // C2R x window -> R2C per row in ONE kernel -- the real map exists in LDS only -- onto the scratch plane the column stages
        // read.  OA_OPT_WIN_FUSED = 0: C2R with the window at its store -> real map in HBM -> the from-map estimator path.
        const int fused_rows = q->opt_win_fused ? 1 : 0;
        if (fused_rows) rc = qe_windowed_rows_w(p, q->kT, tmap, window_real, q->wl, pl, inv, (hipStream_t)stream);
        else rc = oa_fft_c2r_windowed(p, q->kT, tmap, inv, window_real, stream);     // the window rides on the row pass's store
        if (rc) return rc;
        // binning + moments in the divergence launch where the geometry has that kernel (kappa_hat still stored when the mean-field
        // stack needs it)
        DivBinFuse f = make_fuse(p, q, n, S, C, meanfield_acc ? 1 : 0);
        if ((rc = qe_tt_impl(p, tmap, nullptr, nullptr, nullptr, 0, stream, divbin_enabled(q) ? &f : nullptr, fused_rows))) return rc;
        if (!f.done && (rc = bandpower_moments(p, q, n, S, C, stream))) return rc;
        if (meanfield_acc && (rc = stack_add_region(p->dtype, q->kk, meanfield_acc, p->ny, p->kp, q->wk, q->rk, (hipStream_t)stream))) return rc;
    }
    return 0;
}

// ---- device memory for hosts that bring no GPU array library (the reference is NumPy) -----------------------------
int oa_malloc(void** out, size_t bytes) {
    OA_REQUIRE(out, "oa_malloc: NULL");
    *out = nullptr;
    OA_HIP(hipMalloc(out, bytes ? bytes : 1));
    return 0;
}
int oa_free(void* dptr) {
    if (dptr) OA_HIP(hipFree(dptr));
    return 0;
}
/* kind: 1 = host -> device, 2 = device -> host, 3 = device -> device; stream-ordered (pageable host memory makes the
 * copy synchronous with respect to the host, as hipMemcpyAsync documents) */
int oa_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream) {
    OA_REQUIRE(dst && src, "oa_memcpy: NULL");
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    OA_REQUIRE(kind >= 1 && kind <= 3, "oa_memcpy: kind must be 1 (h2d), 2 (d2h) or 3 (d2d)");
    OA_HIP(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
    return 0;
}
int oa_memset(void* dptr, int value, size_t bytes, void* stream) {
    OA_REQUIRE(dptr, "oa_memset: NULL");
    OA_HIP(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return 0;
}
int oa_stream_synchronize(void* stream) {
    OA_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ---- RCCL all-reduce (Statistics.allreduce, stats.py:1209-1230) for hosts without torch.distributed -------------
typedef struct { char internal[128]; } oa_rccl_id;
struct OaComm { void* comm; };
static void* rccl_sym(const char* name) {
    static void* lib = nullptr;
    if (!lib) {
        lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return nullptr;
    }
    return dlsym(lib, name);
}
#define OA_RCCL(fn, ...)                                                                          \
    do {                                                                                          \
        auto f_ = (int (*)(...))rccl_sym(#fn);                                                    \
        if (!f_) return fail("oa_comm: librccl.so / " #fn " not found");                          \
        int e_ = f_(__VA_ARGS__);                                                                 \
        if (e_ != 0) return fail(std::string(#fn ": RCCL error ") + std::to_string(e_));         \
    } while (0)

int oa_comm_unique_id(void* id128) {
    OA_REQUIRE(id128, "oa_comm_unique_id: NULL");
    OA_RCCL(ncclGetUniqueId, id128);
    return 0;
}
int oa_comm_init(int nranks, int rank, const void* id128, void** comm_out) {
    OA_REQUIRE(id128 && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "oa_comm_init: bad argument");
    oa_rccl_id id;
    memcpy(&id, id128, sizeof(id));
    void* c = nullptr;
    auto f = (int (*)(void**, int, oa_rccl_id, int))rccl_sym("ncclCommInitRank");
    if (!f) return fail("oa_comm_init: librccl.so / ncclCommInitRank not found");
    int e = f(&c, nranks, id, rank);
    if (e != 0) return fail("ncclCommInitRank: RCCL error " + std::to_string(e));
    *comm_out = c;
    return 0;
}
int oa_comm_destroy(void* comm) {
    if (!comm) return 0;
    OA_RCCL(ncclCommDestroy, comm);
    return 0;
}
/* in-place SUM over the ranks of `comm`; dtype_code: 0 = float64, 1 = int64, 2 = float32 */
int oa_allreduce(void* comm, void* buf, long count, int dtype_code, void* stream) {
    OA_REQUIRE(comm && buf && count >= 0, "oa_allreduce: bad argument");
    const int nccl_type = dtype_code == 0 ? 8 /* ncclFloat64 */ : (dtype_code == 1 ? 4 /* ncclInt64 */ : 7 /* ncclFloat32 */);
    auto f = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))rccl_sym("ncclAllReduce");
    if (!f) return fail("oa_allreduce: librccl.so / ncclAllReduce not found");
    int e = f(buf, buf, (size_t)count, nccl_type, 0 /* ncclSum */, comm, (hipStream_t)stream);
    if (e != 0) return fail("ncclAllReduce: RCCL error " + std::to_string(e));
    return 0;
}

}  // extern "C"
