/*
 * orphics_amd C-ABI -- MI355X (gfx950) kernels for the flat-sky CMB lensing
 * quadratic-estimator hot path of msyriac/orphics.
 *
 * The reference has NO native/FFI layer: the hot path is Python
 * calling NumPy / pixell.  Each entry point below therefore replaces a NumPy /
 * pixell expression inside a reference function; the reference file:line is
 * cited per function.  The Python classes in orphics_amd/{maps,stats,lensing}.py
 * mirror the reference signatures and bind these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; oa_last_error()
 *    returns the message of the calling thread's last failure; nothing throws;
 *  - all data pointers are DEVICE pointers unless the name says `host_`;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *    work is stream-ordered, no call synchronises unless documented;
 *  - dtype: OA_F32 (float / complex64 planes) or OA_F64 (double / complex128);
 *  - real planes are (ny, nx) row-major; "hc" (half-complex) planes are
 *    (ny, kpitch) complex row-major with kpitch = oa_plan_kpitch(plan)
 *    = nx/2 + 16; columns [0, nx/2] are valid (numpy rfft2 layout), the pad
 *    columns are never read as data and are left untouched;
 *    "full" complex planes are (ny, nx) row-major (numpy fft2 layout);
 *  - a plan is bound to the device current at creation, owns its twiddle tables
 *    and scratch planes, and is not thread-safe (one plan per stream user).
 */
#ifndef ORPHICS_AMD_H
#define ORPHICS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OA_F32 0
#define OA_F64 1

typedef struct oa_plan oa_plan;

/* ---- library ----------------------------------------------------------- */
const char* oa_last_error(void);
/* ABI version = 100 x the build round that last changed a signature in this header; bindings must refuse a library
 * that reports less than the version they were written against (OA_ABI_VERSION) */
#define OA_ABI_VERSION 402
int oa_version(void);
/* number of HIP devices visible; <0 on error (no compute) */
int oa_device_count(void);

/* ---- plan --------------------------------------------------------------
 * Replaces the per-geometry precomputation of FourierCalc.__init__
 * (maps.py:1600-1607).  ny, nx >= 32.  Powers of two run the LDS-staged FFT kernels and every fused
 * estimator kernel.  Other EVEN sides (<= 8192): sides whose only prime factors are 2, 3 and 5 (the reference notebooks
 * use 600, 750, 1200, 2400) run mixed-radix Stockham transforms (mixed.hip), any other side a chirp-z (Bluestein)
 * evaluation on an inner power-of-two plan -- both exact: oa_fft_r2c / oa_fft_c2r / oa_fft_c2c, oa_lens_maps(_hc) and all
 * per-mode / binning / RNG kernels work, `width` / `rband` hints are ignored, and the fused oa_qe_rows /
 * oa_qe_*_cols / oa_fft_cols / oa_fft_pass calls return an error (use the modular oa_qe_legs .. oa_qe_div chain,
 * as orphics_amd/lensing.py:_reconstruct_hc_modular does). */
int oa_plan_create(int ny, int nx, int dtype, oa_plan** out);
int oa_plan_destroy(oa_plan* p);
long oa_plan_kpitch(const oa_plan* p);
/* bytes of device scratch currently held by the plan */
long oa_plan_scratch_bytes(const oa_plan* p);
/* Upload the signed multipole axes ly[ny], lx[nx] (host float64; pixell
 * enmap.laxes as used by maps.py:1607,1938).  Needed by the oa_qe_* calls. */
int oa_plan_set_laxes(oa_plan* p, const double* host_ly, const double* host_lx);

/* ---- 2-D FFTs ------------------------------------------------------------
 * oa_fft_r2c : real (ny,nx) -> hc, out = scale * sum x e^{-i l.x}
 *              (enmap.fft(normalize=False), maps.py:1613,1636, restricted to
 *              the non-redundant half plane of a real map)
 * oa_fft_c2r : hc -> real, out = scale * sum_k X e^{+i l.x} (input preserved)
 *              (pixell fft.ifft(...,normalize=True) + np.real, maps.py:1633,1923,
 *              with scale = 1/(ny*nx))
 * oa_fft_c2c : full -> full, forward (inverse=0) or inverse (inverse=1),
 *              out != in (MapGen non-Hermitian draws maps.py:1578-1587, pol legs). */
/* ACTIVE COLUMNS (`width`, `win`, `wout` arguments below): band-limited filters (the k-space masks of
 * lensing.Estimator built with maps.mask_kspace, maps.py:1936-1948: the tellmax / kellmax of
 * tutorials/tt_verification.ipynb) leave every hc plane on the estimator path exactly zero beyond some
 * column kx >= width.  A transform told so neither reads nor produces those columns -- same arithmetic on
 * the remaining ones, so results are unchanged; HBM traffic and column-pass work scale with width/(nx/2+1).
 * width <= 0 (or > nx/2+1) means all columns.  r2c: only columns < width of hc_out are written;
 * c2r / inverse cols: columns >= width of the input are taken as zero and never read.
 * ACTIVE ROWS (`rband`): the same masks also confine the planes to the row band |ky index| < rband, i.e. rows
 * y < rband or y > ny - rband.  rband > 0 on oa_fft_r2c: only the band rows hold the transform (the others are left
 * undefined); on oa_qe_cols_div: rows outside the band are not written;
 * on oa_qe_legs_cols: input rows outside the band are not read (the filters vanish there).  0 = all rows. */
int oa_fft_r2c(oa_plan* p, const void* real_in, void* hc_out, double scale, int width, int rband, void* stream);
int oa_fft_c2r(oa_plan* p, const void* hc_in, void* real_out, double scale, int width, void* stream);
/* oa_fft_c2r with a real-space window multiplied into the result at the last pass's store: real_out = window_real *
 * scale * IDFT(hc_in) (window_real: ny x nx reals of the plan's dtype; power-of-two sides).  The apodisation step of the
 * reference's analysis flow (maps.py:1350-1361 binned_power(imap * mask)) without another pass over the map. */
int oa_fft_c2r_windowed(oa_plan* p, const void* hc_in, void* real_out, double scale, const void* window_real, void* stream);
int oa_fft_c2c(oa_plan* p, const void* full_in, void* full_out, int inverse, double scale, void* stream);
/* One constituent pass of the transforms above, for per-kernel timing (bench.py roofline):
 * pass_id 0 = R2C row pass (real in -> hc out), 1 = column pass 1 (hc -> hc, out != in),
 * 2 = column pass 2 (in place on `out`; `in` ignored), 3 = C2R row pass (hc -> real). */
int oa_fft_pass(oa_plan* p, int pass_id, const void* in, void* out, int width, void* stream);

/* Column half of the transforms above on an hc plane (all ny-point column DFTs of the nx/2+1
 * valid columns), out != in.  With oa_qe_rows it forms the fused estimator pipeline. */
int oa_fft_cols(oa_plan* p, const void* hc_in, void* hc_out, int inverse, double scale, int width, void* stream);
/* Fused QE row stage: inputs are the three leg planes AFTER their inverse column transforms
 * (oa_fft_cols(..., inverse=1)); per row h = C2R(H), P_x = R2C(C2R(Gx) * h), P_y = R2C(C2R(Gy) * h),
 * product scaled by `scale` (pass (1/(ny*nx))^2 for normalised inverses).  Outputs are row-transformed
 * planes awaiting oa_fft_cols(..., inverse=0).  Replaces 3 x oa_fft_c2r rows + 2 x oa_mul_real +
 * 2 x oa_fft_r2c rows: the real-space planes never touch HBM.  accumulate != 0 adds the result to
 * the existing px, py (estimators whose weight is a sum of separable terms: cos/sin spin-2 pieces;
 * `scale` carries the sign).
 * ROW GRID (`mrow`): legs that vanish beyond column `win` have real-space products band-limited to 2 (win - 1), so
 * the row transforms may run on any grid of mrow >= 2 win + wout points (a power of two <= nx, or 1536 = 3 x 512 when
 * win <= 512 and nx >= 2048): no aliased product frequency reaches the kept columns k < wout, which therefore equal the
 * full-length result (the grid-size factor is folded into the scale; rounding differs at the 1e-7 level in f32).  The
 * real-space planes live in LDS only.
 * mrow = 0: full length nx;  mrow < 0: the smallest alias-free grid of 1024, 1536, 2048, 4096, 8192, ... points;  otherwise
 * checked against the bound. */
int oa_qe_rows(oa_plan* p, const void* gx, const void* gy, const void* h, void* px, void* py, double scale,
               int accumulate, int win, int wout, int mrow, void* stream);

/* Fused estimator column stages (the fast path of lensing.Estimator.reconstruct_*):
 *  oa_qe_legs_cols : oa_qe_legs + oa_fft_cols(inverse) of the three leg planes in one go -- kX, kY and the
 *                    filters are read once per column tile, the filtered legs never exist in HBM
 *                    (outputs feed oa_qe_rows);
 *  oa_qe_cols_div  : oa_fft_cols(forward) of the two oa_qe_rows outputs + oa_qe_div in one go:
 *                    out (+)= Fnorm * (i lx FFT[Px] + i ly FFT[Py]). */
int oa_qe_legs_cols(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH, void* gx, void* gy, void* h,
                    int width, int rband, void* stream);
/* oa_qe_map_legs_cols: oa_fft_r2c + oa_qe_legs_cols for the common case that BOTH legs come from one real map
 * (kappa_from_map("TT", T), lensing.py:973): the forward column pass 2, the leg filters and the inverse column
 * pass 1 run in one kernel -- the map's transform kT never exists in HBM.  Same outputs as
 * oa_qe_legs_cols(plan, kT, kT, ...). */
int oa_qe_map_legs_cols(oa_plan* p, const void* real_map, const void* FG, const void* FH, void* gx, void* gy, void* h,
                        int width, int rband, void* stream);
int oa_qe_cols_div(oa_plan* p, const void* px_rows, const void* py_rows, const void* Fnorm, void* out, int accumulate,
                   int width, int rband, void* stream);

/* ---- one-call entries (SURVEY.md section 8b export list) ---------------------------------------------------
 * A plan that has been told its TT filters and its radial bins runs whole reconstructions per call; work planes are
 * owned by the plan (allocated once by oa_plan_set_filters, never inside a stream-ordered call afterwards).
 *  oa_plan_set_filters : FG, FH, Fnorm = the estimator's real hc-layout planes (gradient-leg weight C^g/(B C^tot),
 *                        inverse-variance weight 1/(B C^tot), -L(L+1)/2 A_L mask_K; device memory owned by the
 *                        caller, which keeps it alive), their active columns / rows (0 = all) and the row grid
 *                        (mrow as in oa_qe_rows).  Replaces the filter set-up of lensing.qest(...).
 *  oa_plan_set_bins    : radial-bin ids of the hc grid (oa_modl_digitize(..., pitch = kpitch, width = nx/2+1)) and
 *                        the power normalisation area / Npix^2; takes the data-independent mode counts once.
 *                        Replaces stats.bin2D.__init__ (stats.py:783-788).
 *  oa_qe_tt            : qest.kappa_from_map("TT", ...) (lensing.py:973-976) in ONE call: pass a real map (both
 *                        legs from it) XOR Fourier-space leg(s) kX [, kY]; kappa_hat's DFT goes to out_kappa_hc or,
 *                        if NULL, to the plan-owned plane oa_plan_kappa(plan).  The pruned kernels write only
 *                        kappa's active region (kappa_cols x row band); zero_outside != 0 zero-fills the rest of a
 *                        caller-supplied plane first (pass 0 only if it is known to be zero there already, e.g. the
 *                        previous call with the same filters wrote it).
 *  oa_qe_pol           : the general separable estimator (TE, EE, EB, TB; lensing.Estimator.reconstruct_hc): the
 *                        npieces (sign, FG, FH, swap-legs) terms are summed in the row stage, one divergence.
 *                        host_* arrays live on the host; the planes they point to on the device.
 *  oa_filter_map       : maps.filter_map (maps.py:1922-1923): real_out = Re IFFT(FFT(real_in) * filter).
 *  oa_qe_tt_moments    : map -> kappa_hat -> binned auto-power -> n += 1, S += b, C += b b^T (Statistics.add of the
 *                        bandpower vector, stats.py:1068-1090): one Monte-Carlo step per call.
 *  oa_mc_run           : the Gaussian N0 / mean-field Monte-Carlo shard [sim_lo, sim_hi) of
 *                        tutorials/tt_verification.ipynb cell 4: Philox GRF (key = (base_seed, sim)) with per-mode
 *                        amplitude covsqrt_hc -> TT estimator -> bandpower moments (+ mean-field stack of kappa_hat,
 *                        interleaved re/im doubles, if meanfield_acc != NULL).  No host synchronisation.
 *  BINDING: the plan keeps the POINTERS handed to oa_plan_set_filters / oa_plan_set_bins and may keep derived copies of what they
 *  point to (tile-major Fnorm / ids of the fused divergence launch; the (FG, FH) values of the R-split column stage in thread order).  Every call of either entry invalidates those copies -- also
 *  when the addresses are the ones bound before -- so a caller that changes the contents of a bound plane calls the entry again. */
int oa_plan_set_filters(oa_plan* p, const void* FG, const void* FH, const void* Fnorm, int leg_cols, int kappa_cols,
                        int leg_rows, int kappa_rows, int mrow);
/* COLUMN GRID of the one-call TT path (the y-axis counterpart of the ROW GRID above).  Legs confined to the rows
 * |ky index| < leg_rows have real-space products confined to |ky index| <= 2 (leg_rows - 1); evaluated on
 * mcol >= max(2 leg_rows + kappa_rows, 2 kappa_rows) rows (a power of two < ny) no aliased product frequency reaches
 * the kept rows |ky index| < kappa_rows, which therefore equal the full-resolution result (the grid-size factor is
 * folded into the scale).  The input transform is still taken at full resolution (every map pixel is read); the
 * inverse column transforms of the legs, the row stage and the forward column transforms of the products run on
 * mcol instead of ny rows.  Filters, ly axis, kX / kY and the kappa output stay on the full-resolution grid (the
 * kernels address row y + (y >= mcol/2 ? ny - mcol : 0)).  oa_plan_set_filters selects -1 (auto: the smallest such
 * power of two, or none if that is >= ny or the filters have no row band) unless mrow == 0, which selects 0 (the map's
 * own rows, as for the row grid).  oa_plan_set_col_grid overrides: -1 auto, 0 off, > 0 explicit (checked against the
 * bound); oa_plan_col_grid returns the grid resolved for the TT filters (0 = ny).  oa_qe_pol follows the same policy
 * with the row bands of each call (mrow == 0 switches it off there too). */
int oa_plan_set_col_grid(oa_plan* p, int mcol);
int oa_plan_col_grid(const oa_plan* p);
/* R of the R-SPLIT from-map path this plan's one-call TT entries run (0 = not this geometry; 4: 8192^2 / 4096^2 maps at the reference's
 * band limits, 8: 16384^2 float64, 2: 8192^2 with up to 1280 leg columns -- the T filter to ell = 6000): the row R2C carries the first
 * radix-R butterfly of the column transform (R = ny / column grid) and ONE single-pass column kernel goes from its output to
 * the three leg planes -- instead of forward pass 1, [forward pass 2 + filters + inverse pass 1] and inverse pass 2.
 * Same arithmetic up to the order of the column butterflies; results agree with the multi-pass path to rounding. */
int oa_plan_rsplit(const oa_plan* p);
/* 1 when this plan's one-call moment entries (oa_qe_tt_moments, oa_qe_tt_moments2, oa_mc_run) bin |kappa_hat|^2 and update
 * n, S, C in the tail of the single-pass divergence launch (coarse grids of 1024 / 2048 / 4096 rows, bins bound) instead of two more
 * launches over the kappa plane; 0: the separate histogram launches.  Same per-mode arithmetic either way; the order of the
 * float64 sums differs (bandpowers agree to ~1e-15).  When fused, these entries do not write the plan-owned kappa plane
 * (oa_plan_kappa) unless the mean-field stack of oa_mc_run needs it; oa_qe_tt always does.  oa_plan_set_option(p, OA_OPT_DIV_BIN, 0)
 * switches it off. */
int oa_plan_div_fused(const oa_plan* p);
/* Which of several EQUIVALENT launch sequences the one-call entries of this plan run.  Every value is a supported configuration
 * (the non-default ones are what geometries without the batched kernels run anyway) and the GPU tests compare them against each
 * other; the library never reads the environment for this.
 *   OA_OPT_MC_BATCH    realisations per launch in oa_mc_run: 1 .. 6 (0 = default, 6); moments and stack do not depend on it
 *   OA_OPT_MV_BATCH    oa_qe_mv / oa_qe_tt_splits: all leg planes in one launch and all divergences in one launch (default 1);
 *                      0: one launch per distinct filtered field / per estimator
 *   OA_OPT_MV_ROWBATCH oa_qe_mv: the row stage of several pieces per launch (default 1); 0: one launch per piece
 *   OA_OPT_MV_CHAIN    oa_qe_mv: estimator chains -- an estimator's pieces summed in real space inside one row-stage launch
 *                      (default 1); 0: the k-th piece of every estimator per launch, accumulated in Fourier space
 *   OA_OPT_DIV_BIN     moment entries: radial binning + moment update in the tail of the single-pass divergence launch
 *                      (default 1, where the geometry has that kernel: oa_plan_div_fused); 0: the separate histogram launches
 *   OA_OPT_WIN_FUSED   oa_mc_run_windowed: inverse columns, then C2R x window -> R2C of every row in ONE kernel (default 1: the
 *                      real map exists in LDS only); 0: C2R with the window at its store, real map in HBM, from-map estimator path */
enum { OA_OPT_MC_BATCH = 1, OA_OPT_MV_BATCH = 2, OA_OPT_MV_ROWBATCH = 3, OA_OPT_MV_CHAIN = 4, OA_OPT_DIV_BIN = 5, OA_OPT_WIN_FUSED = 6 };
int oa_plan_set_option(oa_plan* p, int option, int value);
int oa_plan_set_bins(oa_plan* p, const int32_t* ids_hc, int nids, double norm, void* stream);
void* oa_plan_kappa(oa_plan* p);
const int64_t* oa_plan_bin_counts(oa_plan* p);
int oa_qe_tt(oa_plan* p, const void* real_map, const void* kX, const void* kY, void* out_kappa_hc, int zero_outside,
             void* stream);
int oa_qe_pol(oa_plan* p, int npieces, const double* host_signs, const void* const* host_FG, const void* const* host_FH,
              const int* host_swap, const void* kX, const void* kY, const void* Fnorm, void* out, int accumulate,
              int leg_cols, int kappa_cols, int leg_rows, int kappa_rows, int mrow, int zero_outside, void* stream);
/* oa_qe_pol for several estimators accumulated into ONE kappa plane (the minimum-variance combination: Fnorm[e] carries
 * weight x normalisation of estimator e), pieces flattened in estimator order (host_npieces[e] each; host_kX/kY/Fnorm per
 * estimator).  Every distinct filtered field -- identified by its (source plane, filter plane) POINTERS -- is transformed
 * once, all in one inverse pass-2 launch: 17 leg planes instead of 30 for TT+TE+EE+EB+TB when the caller shares its filter
 * plane objects.  Same arithmetic per piece as oa_qe_pol, same results.  When the Fnorm planes are evenly spaced in memory
 * (one stacked allocation) the divergence of all estimators is one launch, each into a plan-owned plane, and one pass sums
 * them in estimator order.  oa_qe_mv and oa_qe_tt_splits keep their leg / product planes in a plan-owned pool that is
 * allocated on first use and grown on demand: THAT call synchronises the device once; later calls are stream-ordered. */
int oa_qe_mv(oa_plan* p, int nest, const int* host_npieces, const double* host_signs, const void* const* host_FG,
             const void* const* host_FH, const int* host_swap, const void* const* host_kX, const void* const* host_kY,
             const void* const* host_Fnorm, void* out, int accumulate, int leg_cols, int kappa_cols, int leg_rows, int kappa_rows,
             int mrow, int zero_outside, void* stream);
int oa_filter_map(oa_plan* p, const void* real_in, const void* filt_hcreal, void* real_out, void* stream);
int oa_qe_tt_moments(oa_plan* p, const void* real_map, int64_t* n, double* S, double* C, void* stream);
/* Two Monte-Carlo steps per call (two independent maps): identical results to two oa_qe_tt_moments calls; on the
 * column-grid path every launch behind the two row transforms is shared by both maps. */
int oa_qe_tt_moments2(oa_plan* p, const void* real_map0, const void* real_map1, int64_t* n, double* S, double* C, void* stream);
/* SplitLensing.cross_estimator (lensing.py:980-1003) needs the TT reconstruction of every ordered pair of
 * splits (X / gradient leg from split i, Y leg from split j).  One call: the three filtered leg planes of each split are
 * transformed ONCE (nsplits leg stages instead of nsplits^2), then the row stage + divergence run per pair.
 * host_kmaps: nsplits device hc planes (the splits' transforms); host_out: nsplits^2 device hc planes, [i*nsplits + j];
 * zero_outside as in oa_qe_tt.  Needs oa_plan_set_filters.
 * oa_split_cross_power: the estimator's combination of those planes per mode, in f64 (the QE is bilinear, so the
 * reference's reconstructions involving the split mean are means of the pairwise ones):
 *   out = (n^4 P(kc) - 4 n^2 sum_i P(kic) + 4 sum_{i<j} P(kij)) / (n (n-1)(n-2)(n-3)),  P(x) = |x|^2 norm,
 * written over columns < active_cols and the band rows of the real half-plane `out_hcreal` only (4 <= nsplits <= 8). */
int oa_qe_tt_splits(oa_plan* p, int nsplits, const void* const* host_kmaps, void* const* host_out, int zero_outside, void* stream);
int oa_split_cross_power(int dtype, int nsplits, const void* const* host_kappa, void* out_hcreal, double norm, int ny, long kpitch,
                         int active_cols, int active_rows, void* stream);
/* oa_mc_run: realisations sim_lo .. sim_hi-1 (Philox stream = realisation index): GRF draw -> TT estimator -> bandpowers ->
 * moments [-> mean-field stack], no host work per realisation.  Up to 6 realisations (oa_plan_set_option OA_OPT_MC_BATCH,
 * 1 = one by one) share every launch (grid z / y: draw, leg planes, inverse pass 2, row stage, divergence, binned power, moment tail, stack): at
 * 4096^2 a realisation is launch latency, not bytes.  Same kernels on the same operands in the same order per realisation:
 * the moments and the stack do not depend on the batch size.  The first call allocates the batch's planes (one device
 * synchronisation). */
int oa_mc_run(oa_plan* p, uint64_t base_seed, long sim_lo, long sim_hi, const void* covsqrt_hc, int64_t* n, double* S,
              double* C, double* meanfield_acc, void* stream);
/* The same shard with a real-space window applied to every realisation before its transform (the reference's analysis
 * flow: maps.py:1873-1878 get_taper, maps.py:1350-1361): full-plane draw (same Philox counters as oa_mc_run) -> C2R / Npix
 * -> x window_real (ny x nx reals of the plan's dtype) -> TT estimator -> bandpower moments (+ mean-field stack).  With
 * window == 1 the moments equal oa_mc_run's up to rounding; with a taper the stack holds the window's mean field. */
int oa_mc_run_windowed(oa_plan* p, uint64_t base_seed, long sim_lo, long sim_hi, const void* covsqrt_hc, const void* window_real,
                       int64_t* n, double* S, double* C, double* meanfield_acc, void* stream);
/* One stage of oa_qe_tt_moments on the plan's own work planes, for per-kernel timing (bench.py roofline; the
 * one-call path keeps its intermediates on COMPACT planes -- pitch = active columns rounded up to a 32-column tile
 * -- so its kernels are not the same launches as the fine-grained calls on caller planes of pitch kpitch):
 * 0 = row R2C, 1 = forward column pass 1, 2 = fused leg kernel + 3-plane inverse pass 2, 3 = fused row stage,
 * 4 = 2-plane forward pass 1 + divergence kernel, 5 = binned power + moment accumulation (plan-owned dummies). */
int oa_qe_tt_stage(oa_plan* p, int stage, const void* real_map, void* stream);

/* ---- device memory helpers for hosts without a GPU array library (the reference passes NumPy arrays) --------- */
int oa_malloc(void** out, size_t bytes);
int oa_free(void* dptr);
int oa_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream); /* 1 h2d, 2 d2h, 3 d2d */
int oa_memset(void* dptr, int value, size_t bytes, void* stream);
int oa_stream_synchronize(void* stream);

/* ---- ensemble reduce over GPUs (Statistics.allreduce, stats.py:1209-1230) without torch.distributed ----------
 * RCCL communicator over the ranks of one job (librccl.so is dlopen'ed on first use).  One rank calls
 * oa_comm_unique_id and distributes the 128 bytes (MPI_Bcast in an mpi4py host), every rank then calls
 * oa_comm_init with its rank; oa_allreduce sums a device buffer in place over the ranks, stream-ordered.
 * dtype_code: 0 = float64, 1 = int64, 2 = float32. */
int oa_comm_unique_id(void* id128);
int oa_comm_init(int nranks, int rank, const void* id128, void** comm_out);
int oa_comm_destroy(void* comm);
int oa_allreduce(void* comm, void* device_buf, long count, int dtype_code, void* stream);

/* ---- layout helpers ------------------------------------------------------ */
/* hc -> full by Hermitian symmetry X(-l) = conj X(l) (what the reference's C2C of
 * a real map holds, maps.py:1613) */
int oa_hc_to_full(oa_plan* p, const void* hc_in, void* full_out, void* stream);
/* full -> hc (drops the redundant half) */
int oa_full_to_hc(oa_plan* p, const void* full_in, void* hc_out, void* stream);
/* real-valued hc-layout plane (ny,kpitch) <-> full real (ny,nx), even symmetry */
int oa_hcreal_to_full(oa_plan* p, const void* hcreal_in, void* fullreal_out, void* stream);
int oa_fullreal_to_hc(oa_plan* p, const void* fullreal_in, void* hcreal_out, void* stream);

/* ---- flat elementwise kernels (n = number of elements) --------------------
 * oa_f2power    : out = Re(conj(k1)*k2)*norm        (FourierCalc.f2power, maps.py:1620-1624)
 * oa_cmul_real  : out = k * f (complex * real)      (filter_map's `* kfilter`, maps.py:1923;
 *                                                   MapGen covsqrt*rand scalar case, maps.py:1579)
 * oa_cmul       : out = k * f (complex * complex)   (filter_map with a complex kfilter, maps.py:1923)
 * oa_mul_real   : out = a * b (real)                (QE real-space product)
 * oa_axpby_real : out = alpha*a + beta*b (real)     (observed = beamed + noise, lensing.py:516) */
int oa_f2power(int dtype, const void* k1, const void* k2, void* out_real, double norm, long n, void* stream);
int oa_cmul_real(int dtype, const void* k_in, const void* filt_real, void* k_out, long n, void* stream);
int oa_cmul(int dtype, const void* k_in, const void* filt_complex, void* k_out, long n, void* stream);
int oa_mul_real(int dtype, const void* a, const void* b, void* out, long n, void* stream);
int oa_axpby_real(int dtype, const void* a, const void* b, void* out, double alpha, double beta, long n, void* stream);
/* per-mode 2x2 rotation of two complex planes: [o1;o2] = [[c,-s],[s,c]] [i1;i2]
 * with c,s real planes (QU<->EB, enmap.map_mul(self.rot, ...), maps.py:1614-1615) */
int oa_rot2(int dtype, const void* c, const void* s, const void* i1, const void* i2, void* o1, void* o2, long n, void* stream);

/* ---- quadratic-estimator legs (hc layout) ---------------------------------
 * The reference class (lensing.Estimator/qest) is absent from the snapshot
 * (SURVEY.md F2); the contract is qest.kappa_from_map (lensing.py:973-976).
 * Real-space Hu-DeDeo-Vale form, spin = 0 (TT) or spin-2 phase variants:
 *  oa_qe_legs : from kX (gradient leg) and kY: Gx = i lx FG kX P, Gy = i ly FG kX P,
 *               H = FH kY Q, where FG, FH are real hc-layout filter planes and the
 *               optional spin-2 phase factors P,Q = exp(+-2 i phi_l) are selected by
 *               `phase_g`/`phase_h` (0: none, +1: e^{2i phi}, -1: e^{-2i phi}), and
 *               `h_times_i` multiplies H by i (B-mode leg).
 *  oa_qe_div  : out = Fnorm * (i lx Px + i ly Py)  (divergence * normalisation) */
int oa_qe_legs(oa_plan* p, const void* kX, const void* kY, const void* FG, const void* FH,
               void* Gx, void* Gy, void* H, int phase_g, int phase_h, int h_times_i, void* stream);
int oa_qe_div(oa_plan* p, const void* Px, const void* Py, const void* Fnorm, void* out, int accumulate, void* stream);

/* ---- flat-sky lensing of simulated maps (lensing.flat_taylens, lensing.py:395-440) ------------------
 * oa_lens_split  : shift[i] = rint(alpha[i]/step), delta[i] = alpha[i] - shift[i]*step (nearest pixel +
 *                  sub-pixel remainder of a displacement component; lensing.py:420-424)
 * oa_lens_gather : out[y,x] (+)= coef * src[(y+sy) mod ny, (x+sx) mod nx] * dx^pow_x * dy^pow_y
 *                  (integer-pixel remap and one Taylor term, lensing.py:428-438) */
int oa_lens_split(int dtype, const void* alpha, double step, int32_t* shift, void* delta, long n, void* stream);
int oa_lens_gather(oa_plan* p, const void* src, const int32_t* shift_x, const int32_t* shift_y, const void* dx, const void* dy,
                   int pow_x, int pow_y, double coef, void* out, int accumulate, void* stream);
/* The same Taylor series with every term of every order in two launches instead of one derivative kernel, one C2R and
 * one gather per term (lensing.py:395-440; order 5 = 14 terms per map):
 *  oa_hc_derivs   : hc_out_planes[idx(a, b)] = (i lx)^a (i ly)^b hc_in for 1 <= a + b < order, idx(a, b) = n (n + 1) / 2 - 1 + b,
 *                   n = a + b, planes plane_stride complex elements apart (order (order + 1) / 2 - 1 of them).
 *  oa_lens_taylor : out = sum_{a + b < order} dx^a dy^b / (a! b!) D_ab[(y + shift_y) % ny, (x + shift_x) % nx] with D_00 = src
 *                   and D_ab the REAL planes (the C2R of the planes above, normalised) in the same order, plane_stride
 *                   real elements apart.  shift / dx / dy as produced by oa_lens_split. */
int oa_hc_derivs(oa_plan* p, const void* hc_in, int order, void* hc_out_planes, long plane_stride, void* stream);
int oa_lens_taylor(oa_plan* p, const void* src, const void* deriv_planes, long plane_stride, int order, const int32_t* shift_x,
                   const int32_t* shift_y, const void* dx, const void* dy, void* out, void* stream);
/* flat_taylens (lensing.py:395-440) of nmaps real maps (in_stride / out_stride elements apart) by ONE deflection field, given as
 * its nearest-pixel shifts and sub-pixel remainders (oa_lens_split; shared by all maps: T, Q, U of a realisation): nmaps R2Cs, then
 * the inverse transforms of all nmaps * nd derivative fields (nd = order (order + 1) / 2 - 1), SEPARABLY: per map and y-derivative
 * order b one column transform of (i ly)^b k (passes 1 and 2 on ONE plan-owned hc plane, which stays in the infinity cache) and one
 * row launch that takes every x-derivative (i lx)^a at its load (the derivative spectra never exist in HBM) -- and one
 * gather pass per map (oa_lens_taylor).  Same results as oa_hc_derivs + C2R per term + oa_lens_taylor.  The planes live in a
 * plan-owned pool allocated / grown on first use (that call synchronises the device once): nmaps * (1 + nd) planes + one hc plane,
 * e.g. 6.2 GB for T, Q, U at 4096^2 float64 and order 5; oa_plan_release_pools frees it (and the pools of oa_qe_mv /
 * oa_qe_tt_splits / oa_mc_run), the next call reallocates. */
int oa_lens_maps(oa_plan* p, int nmaps, const void* real_in, long in_stride, int order, const int32_t* shift_x, const int32_t* shift_y,
                 const void* dx, const void* dy, void* real_out, long out_stride, void* stream);
/* The same from the maps' hc transforms (a simulation that DREW its fields in harmonic space has them already: MapGen.get_map's
 * transform, lensing.py:499-512): `hc_in` holds nmaps planes hc_stride complex elements apart, map = scale * C2R(hc_in)
 * (scale = 1 / sqrt(Ny Nx) for MapGen's unitary draws).  No forward transform is taken and the undisplaced map itself -- the
 * (a, b) = (0, 0) term -- leaves the same batched row launch as the x-derivatives of the b = 0 column transform: nmaps fewer
 * R2Cs and nmaps fewer separate C2Rs than oa_lens_maps on the inverse-transformed maps, same results to rounding. */
int oa_lens_maps_hc(oa_plan* p, int nmaps, const void* hc_in, long hc_stride, double scale, int order, const int32_t* shift_x,
                    const int32_t* shift_y, const void* dx, const void* dy, void* real_out, long out_stride, void* stream);
int oa_plan_release_pools(oa_plan* p);

/* ---- radial binning (stats.bin2D, stats.py:782-811) ------------------------
 * oa_digitize : ids[i] = np.digitize(x[i], edges, right=True) (stats.py:786):
 *               e[id-1] < x <= e[id]; 0 = underflow, nedges = overflow. x, edges
 *               are float64 (bit-exact comparisons); ids are int32.
 * oa_modl_digitize : same, with x = sqrt(ly[y]^2 + lx[x]^2) computed on device in
 *               float64 with IEEE round-to-nearest mul/add/sqrt and no FMA
 *               contraction (NumPy op order of enmap.modlmap).  `pitch`/`width`
 *               select the plane layout: full plane pitch=width=nx; hc plane
 *               pitch=kpitch,width=nx/2+1 (pad columns get id -1 = ignored).
 * oa_bin      : sums[id] += v, counts[id] += m over all i with ids[i] >= 0 where
 *                 v = data[i] (mode 0) | (data[i]-aux[id])^2 (mode 1), times
 *                 weights[i] if weights != NULL, times the Hermitian multiplicity m,
 *                 m = 1 (herm_nxh < 0) or {1 for col 0 and col nxh, 2 for 0<col<nxh}
 *                 with col = i % herm_pitch;
 *               counts[] are exact int64 (unweighted) ; wsums[] = sum of weights*m (f64).
 *               skip_nan != 0 drops NaN data (mask_nan=True, stats.py:793).
 *               Output arrays have nids = nedges+1 entries and are OVERWRITTEN.
 *               Deterministic: per-workgroup partials reduced in fixed order.
 *               `scratch` must hold oa_bin_scratch_bytes(nids) bytes. */
int oa_digitize(const double* x, long n, const double* edges, int nedges, int32_t* ids, void* stream);
int oa_modl_digitize(const double* ly, const double* lx, int ny, int nx, long pitch, int width,
                     const double* edges, int nedges, int32_t* ids, double* modl_out, void* stream);
long oa_bin_scratch_bytes(int nids);
int oa_bin(int dtype, const void* data, const int32_t* ids, const void* weights, const double* aux, long n,
           int nids, int mode, int skip_nan, long herm_pitch, int herm_nxh, double* sums, int64_t* counts,
           double* wsums, void* scratch, void* stream);

/* oa_bin_power: FourierCalc.f2power (maps.py:1620-1624) fused into oa_bin: the binned value is
 * Re(conj(k1[i]) k2[i]) * norm for complex planes k1, k2 (k1 == k2 for auto spectra); the 2-D power
 * plane is never written.  Same ids / multiplicity / determinism contract as oa_bin (mode 0).
 * active_cols > 0 (Hermitian mode only): only columns < active_cols of each row are visited (planes that
 * vanish beyond them): sums are unchanged, COUNTS then cover the visited columns only -- take the
 * data-independent counts from one full oa_bin_power / oa_bin call at plan time.  active_rows > 0 (with
 * active_cols): additionally only the rows of the band y < active_rows or y > ny - active_rows. */
int oa_bin_power(int dtype, const void* k1, const void* k2, double norm, const int32_t* ids, const void* weights, long n,
                 int nids, long herm_pitch, int herm_nxh, double* sums, int64_t* counts, double* wsums, void* scratch, int active_cols, int active_rows,
                 void* stream);

/* ---- Gaussian random fields (MapGen.get_map, maps.py:1576-1587) --------------
 * Fills an hc plane with Hermitian-consistent complex white noise of unit
 * variance per full-plane mode, scaled per mode by the real hc-layout plane
 * `covsqrt` (may be NULL): the C2R of the result with the unitary scale
 * 1/sqrt(ny*nx) is statistically identical to
 * enmap.ifft(covsqrt*rand_gauss_harm).real.  Counter-based Philox4x32-10,
 * key = (seed, stream_id), so realisations are independent of launch geometry. */
int oa_grf_hc(oa_plan* p, uint64_t seed, uint64_t stream_id, const void* covsqrt_hc, void* hc_out, void* stream);
/* The same draw restricted to the ACTIVE region (columns < width, rows y < rband or y > ny - rband; 0 = all): a
 * bit-identical subset of oa_grf_hc's plane (every mode keeps its Philox counter), the rest of hc_out is not written.
 * A Monte-Carlo loop whose estimator reads only its leg band (oa_mc_run does this itself) draws ~1 % of the modes. */
int oa_grf_hc_band(oa_plan* p, uint64_t seed, uint64_t stream_id, const void* covsqrt_hc, void* hc_out, int width, int rband,
                   void* stream);
/* MapGen.get_map's draw in ONE pass (maps.py:1579-1587: covsqrt * rand_gauss_harm, then harm2map's rotation): ncomp (1..3) white
 * fields of streams (seed, stream_id0 + j) -- bit-identical to oa_grf_hc's -- mixed by the covariance square root,
 *     v_i = sum_j covsqrt_hc[i * ncomp + j] * w_j        (hc-real planes; a NULL entry is a zero block),
 * then, with rotation planes (ncomp == 3: components 1, 2 <- (x1 c - x2 s, x1 s + x2 c), oa_rot2's convention),
 *     hc_in == NULL : hc_out[i] = scale * rot(v)_i                       (unlensed T, Q, U transforms; kappa)
 *     hc_in != NULL : hc_out[i] = rot(hc_in * filt_hcreal)_i + scale * v_i   (beam x lensed Q, U -> E, B, + noise: the observed
 *                                                                          transforms of lensing.py:513-516 in one pass)
 * filt_hcreal may be NULL (= 1); hc_out[i] may alias hc_in[i]. */
int oa_grf_mix(oa_plan* p, uint64_t seed, uint64_t stream_id0, int ncomp, const void* const* covsqrt_hc, const void* rot_c, const void* rot_s,
               const void* const* hc_in, const void* filt_hcreal, double scale, void* const* hc_out, void* stream);
/* real white noise N(0,1) plane of n elements (enmap.rand_gauss) */
int oa_randn(int dtype, uint64_t seed, uint64_t stream_id, void* out, long n, void* stream);

/* ---- one-pass moment accumulation (Statistics.add, stats.py:1068-1090) ---------
 * n += 1 ; S += x ; C += x x^T  for a device vector x of length d (float64). */
int oa_moments_add(const double* x, int d, int64_t* n, double* S, double* C, void* stream);
/* Same with x_a = sums[a] / counts[a] (the bin means of stats.bin2D.bin, stats.py:790-811) formed in the
 * kernel: feeds the interior slots of oa_bin / oa_bin_power output directly (pass sums+1, counts+1, d=nbins). */
int oa_moments_add_binned(const double* sums, const int64_t* counts, int d, int64_t* n, double* S, double* C, void* stream);
/* stack accumulation (Statistics.add_stack, stats.py:1124-1150): acc(f64) += x (dtype) */
int oa_stack_add(int dtype, const void* x, double* acc, long n, void* stream);

/* ---- measurement helpers -------------------------------------------------------
 * Streaming device copy / read of `bytes` (a multiple of 16) with 16-byte accesses: the bandwidth ceiling bench.py
 * measures in the same run as the kernels it prices against it.  `sink`: >= 8 MiB of device scratch (one word per
 * thread of the read probe). */
int oa_probe_copy(void* dst, const void* src, size_t bytes, void* stream);
int oa_probe_read(const void* src, size_t bytes, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORPHICS_AMD_H */
