// K7: radial binning (stats.bin2D).  digitize = bit-exact float64 comparisons;
// bin = streaming histogram: per-lane run merge -> wavefront match/shuffle
// reduce -> wave-private LDS rows (no atomics) -> per-workgroup partials ->
// fixed-order final reduce (deterministic float64 sums, exact int64 counts).
#include "common.hpp"

#pragma clang fp contract(off)

namespace oa {

constexpr int BIN_BLOCK = 256;
constexpr int BIN_WAVES = BIN_BLOCK / 64;
constexpr int BIN_GMAX = 1024;
constexpr int BIN_MAX_IDS = 1024;
constexpr int DIG_MAX_EDGES = 4096;

// np.digitize(x, edges, right=True) for increasing edges: #edges strictly < x; NaN -> nedges
OA_D int digitize_one(double x, const double* e, int ne) {
    if (x != x) return ne;
    int lo = 0, hi = ne;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (e[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void digitize_kernel(const double* __restrict__ x, long n, const double* __restrict__ edges,
                                                       int ne, int32_t* __restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    double* e = reinterpret_cast<double*>(sm_raw);
    for (int i = threadIdx.x; i < ne; i += blockDim.x) e[i] = edges[i];
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ids[i] = digitize_one(x[i], e, ne);
}

__global__ __launch_bounds__(256) void modl_digitize_kernel(const double* __restrict__ ly, const double* __restrict__ lx,
                                                            int ny, int nx, long pitch, int width,
                                                            const double* __restrict__ edges, int ne,
                                                            int32_t* __restrict__ ids, double* __restrict__ modl_out) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    double* e = reinterpret_cast<double*>(sm_raw);
    for (int i = threadIdx.x; i < ne; i += blockDim.x) e[i] = edges[i];
    __syncthreads();
    const int y = blockIdx.y;
    for (long x = (long)blockIdx.x * blockDim.x + threadIdx.x; x < pitch; x += (long)gridDim.x * blockDim.x) {
        const long i = (long)y * pitch + x;
        if (x >= width) {
            ids[i] = -1;
            if (modl_out) modl_out[i] = 0.0;
            continue;
        }
        // NumPy order of enmap.modlmap: sqrt(ly**2 + lx**2), each op rounded to nearest
        const double a = __dmul_rn(ly[y], ly[y]);
        const double b = __dmul_rn(lx[x], lx[x]);
        const double m = __dsqrt_rn(__dadd_rn(a, b));
        ids[i] = digitize_one(m, e, ne);
        if (modl_out) modl_out[i] = m;
    }
}

OA_D double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// All lanes call; lanes with valid==false contribute nothing.
template <bool WEIGHTED>
OA_D void wave_accum(bool valid, int id, double v, double cw, int ci, double* row_sum, double* row_w,
                     unsigned long long* row_cnt, int lane) {
    unsigned long long act = __ballot(valid);
    while (act) {
        const int leader = __ffsll((long long)act) - 1;
        const int lid = __shfl(id, leader, 64);
        const bool mine = valid && (id == lid);
        const unsigned long long mm = __ballot(mine);
        const double sv = wave_sum(mine ? v : 0.0);
        if (WEIGHTED) {
            const double sw = wave_sum(mine ? cw : 0.0);
            if (lane == leader) { row_sum[lid] += sv; row_w[lid] += sw; }
        } else {
            // counts are 1, 2 (merged runs: up to 8): reduce as integers, exactly
            int c = mine ? ci : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == leader) { row_sum[lid] += sv; row_cnt[lid] += (unsigned long long)c; }
        }
        act &= ~mm;
    }
}

// Sorted fast path.  Radial ids are non-decreasing along a row of the half plane, so the keys of a wave-instruction are
// (almost always) SORTED over the lanes: one segmented inclusive scan (6 shuffle steps for every id at once) replaces
// the match loop above (6+ dependent shuffle steps PER DISTINCT ID -- the latency that dominated the active-region
// launch).  Every lane passes a key, also lanes without a value (`valid` false: they contribute zero and must carry a
// key that keeps the sequence sorted); the last lane of each run of equal keys adds the run total to the wave's
// private LDS row -- distinct keys, distinct addresses, fixed order: deterministic.  Returns false (nothing done)
// when the keys are not sorted; the caller then falls back to wave_accum.
template <bool WEIGHTED>
OA_D bool wave_accum_sorted(bool valid, int key, double v, double cw, int ci, int nids, double* row_sum, double* row_w,
                            unsigned long long* row_cnt, int lane) {
    const int prev = __shfl_up(key, 1, 64);
    if (__ballot(lane > 0 && prev > key)) return false;
    if (!valid) { v = 0.0; cw = 0.0; ci = 0; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int pk = __shfl_up(key, d, 64);
        const double pv = __shfl_up(v, d, 64);
        const bool same = lane >= d && pk == key;
        if (WEIGHTED) {
            const double pw = __shfl_up(cw, d, 64);
            if (same) { v += pv; cw += pw; }
        } else {
            const int pc = __shfl_up(ci, d, 64);
            if (same) { v += pv; ci += pc; }
        }
    }
    const int next = __shfl_down(key, 1, 64);
    const bool tail = lane == 63 || next != key;
    if (tail && key >= 0 && key < nids) {
        if (WEIGHTED) { row_sum[key] += v; row_w[key] += cw; }
        else { row_sum[key] += v; row_cnt[key] += (unsigned long long)ci; }
    }
    return true;
}

// Tail of the one-call Monte-Carlo step (pipeline.hip): the workgroup of bin_final_kernel that finishes LAST (a ticket
// counter over its nids workgroups; nobody waits for anybody) also does what moments_add_binned_kernel would do in one
// more launch: n += 1, S += b, C += b b^T with b = sums / mode counts.  The final sums travel write-through (sc1
// stores, agent-scope loads) instead of behind a device-scope fence: on this multi-XCD part a release fence writes back
// the XCD's whole L2.  Measured and dropped: the same ticket in bin_kernel itself (~1000 workgroups: 95 us with
// __threadfence(), 30 us of same-address atomics without) and a one-workgroup tail kernel (30 us: one CU cannot keep
// enough of the strided partial loads in flight).
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx942__) || defined(__gfx950__))
#error "BinTail: the ticket hand-over below relies on gfx942 / gfx950 behaviour (agent-scope relaxed atomic stores are sc1 write-through, and stores count in vmcnt); on another target use release / acquire on the ticket or the two-launch moments_add_binned path"
#endif
struct BinTail {
    unsigned* ticket;            // zero before the launch; reset to zero by the last workgroup.  nullptr: no tail
    const int64_t* mcounts;      // moments: data-independent mode counts per id (nullptr: no moments)
    int64_t* n; double* S; double* C;
};

// POWER: `data`/`data2` are complex planes and the binned value is Re(conj(k1) k2) * pnorm
// (FourierCalc.f2power fused into the histogram: the 2-D power plane never exists in HBM)
template <typename T, bool WEIGHTED, bool POWER>
__global__ __launch_bounds__(BIN_BLOCK) void bin_kernel(const T* __restrict__ data, const T* __restrict__ data2, double pnorm,
                                                        const int32_t* __restrict__ ids,
                                                        const T* __restrict__ w, const double* __restrict__ aux, long n,
                                                        int nids, int mode, int skip_nan, unsigned hp, int nxh, unsigned wq,
                                                        unsigned rb, double* __restrict__ part_sum, double* __restrict__ part_w,
                                                        unsigned long long* __restrict__ part_cnt, long dstride, long pstride) {
    // grid y = plane of a batch (oa_mc_run): data planes dstride elements apart, partial arrays pstride apart
    data += (long)blockIdx.y * dstride; data2 += (long)blockIdx.y * dstride;
    part_sum += (long)blockIdx.y * pstride; part_w += (long)blockIdx.y * pstride; part_cnt += (long)blockIdx.y * pstride;
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    double* s_sum = reinterpret_cast<double*>(sm_raw);                        // [WAVES][nids]
    double* s_w = s_sum + BIN_WAVES * nids;                                   // [WAVES][nids] (weighted)
    unsigned long long* s_cnt = reinterpret_cast<unsigned long long*>(s_w);   // aliases s_w (unweighted)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 2 * BIN_WAVES * nids; i += BIN_BLOCK) s_sum[i] = 0.0;  // zero bits == 0ull
    __syncthreads();
    double* row_sum = s_sum + wv * nids;
    double* row_w = s_w + wv * nids;
    unsigned long long* row_cnt = s_cnt + wv * nids;

    // wq > 0: visit only the first wq 4-element chunks of every row of pitch hp (active columns); rb > 0: and
    // only the rows of the band |ky index| < rb, i.e. y < rb or y > ny - rb  (2 rb - 1 rows)
    const long nyf = n / hp;
    const long nrows = (wq && rb && 2L * rb - 1 < nyf) ? 2L * rb - 1 : nyf;
    // Hermitian (row-structured) planes: the chunks of a visited row are padded to a whole number of waves in the
    // VIRTUAL index, so a wave never straddles two rows and its ids stay sorted (wave_accum_sorted)
    const unsigned rowc = wq ? wq : (nxh >= 0 ? (hp >> 2) : 0u);     // chunks per visited row (0: flat data)
    const unsigned rowp = (rowc + 63u) & ~63u;
    const long nchunks = rowc ? nrows * (long)rowp : (n + 3) / 4;
    for (long base = (long)blockIdx.x * BIN_BLOCK; base < nchunks; base += (long)gridDim.x * BIN_BLOCK) {
        const long cv = base + tid;      // virtual chunk index over the visited region
        bool cin = cv < nchunks;
        long c = cv;
        unsigned col0 = 0;               // column of the chunk's first element
        if (rowc) {
            const unsigned r = (unsigned)(cv / rowp);      // < 2^31 rows x padded chunks: the quotient fits 32 bits
            const unsigned q = (unsigned)(cv - (long)r * rowp);
            const long y = (nrows == nyf || (long)r < (long)rb) ? (long)r : nyf - nrows + (long)r;
            c = y * (long)(hp >> 2) + q;
            cin = cin && q < rowc;
            col0 = q * 4u;
        }
        const long i0 = c * 4;
        int id[4];
        double v[4], cw[4];
        int ci[4];
        bool ok[4];
        if (cin && i0 + 3 < n) {
            const Arr<int32_t, 4> I = reinterpret_cast<const Arr<int32_t, 4>*>(ids)[c];
            Arr<T, 4> W;
            if (WEIGHTED) W = reinterpret_cast<const Arr<T, 4>*>(w)[c];
            if (POWER) {
                const Arr<T, 8> K1 = reinterpret_cast<const Arr<T, 8>*>(data)[c];
                const Arr<T, 8> K2 = (data2 == data) ? K1 : reinterpret_cast<const Arr<T, 8>*>(data2)[c];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = (double)((K1.v[2 * j] * K2.v[2 * j] + K1.v[2 * j + 1] * K2.v[2 * j + 1]) * (T)pnorm);
            } else {
                const Arr<T, 4> D = reinterpret_cast<const Arr<T, 4>*>(data)[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (double)D.v[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { id[j] = I.v[j]; cw[j] = WEIGHTED ? (double)W.v[j] : 1.0; ok[j] = true; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[j] = cin && (i0 + j < n);
                id[j] = ok[j] ? ids[i0 + j] : -1;
                if (POWER) {
                    const long e = 2 * (i0 + j);
                    v[j] = ok[j] ? (double)((data[e] * data2[e] + data[e + 1] * data2[e + 1]) * (T)pnorm) : 0.0;
                } else {
                    v[j] = ok[j] ? (double)data[i0 + j] : 0.0;
                }
                cw[j] = (WEIGHTED && ok[j]) ? (double)w[i0 + j] : 1.0;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = 1;
            if (nxh >= 0) {
                const int col = (int)col0 + j;
                m = (col == 0 || col == nxh) ? 1 : (col < nxh ? 2 : 0);
            }
            ok[j] = ok[j] && id[j] >= 0 && id[j] < nids && m > 0;
            if (skip_nan && v[j] != v[j]) ok[j] = false;
            if (ok[j] && mode == 1) { const double d = v[j] - aux[id[j]]; v[j] = d * d; }
            if (WEIGHTED) { v[j] = v[j] * cw[j] * (double)m; cw[j] = cw[j] * (double)m; }
            else v[j] = v[j] * (double)m;
            ci[j] = m;
        }
        // per-lane run merge: up to 4 runs of equal ids (a lane holds 4 consecutive columns: usually 1 run, 2 where
        // a bin edge falls inside the chunk)
        // (all indices below are compile-time after unrolling: a runtime-indexed local array would live in scratch)
        int rid[4], rc[4], nr = 0, last = 0;
        double rv[4], rw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { rid[j] = 0; rc[j] = 0; rv[j] = 0.0; rw[j] = 0.0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ok[j]) {
                if (!(nr > 0 && last == id[j])) { ++nr; last = id[j]; }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (nr - 1 == t) { rid[t] = id[j]; rv[t] += v[j]; rw[t] += cw[j]; rc[t] += ci[j]; }
            }
        }
        // run k of every lane in one wave-level step: sorted keys -> segmented scan, anything else -> match loop.
        // A lane without a k-th run passes its last key (keeps the sequence sorted) and no value; a lane without any
        // run sits beyond the end of its row: key = INT_MAX.
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!__ballot(nr > k)) break;
            const bool has = k < nr;
            int key = 0x7fffffff, kc = 0;
            double kv = 0.0, kw = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool pick = has ? (t == k) : (t == nr - 1);     // own k-th run, else the lane's last run (key only)
                if (pick) { key = rid[t]; kv = rv[t]; kw = rw[t]; kc = rc[t]; }
            }
            if (!wave_accum_sorted<WEIGHTED>(has, key, kv, kw, kc, nids, row_sum, row_w, row_cnt, lane))
                wave_accum<WEIGHTED>(has, key, kv, kw, kc, row_sum, row_w, row_cnt, lane);
        }
    }
    __syncthreads();
    for (int i = tid; i < nids; i += BIN_BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < BIN_WAVES; ++k) s += s_sum[k * nids + i];
        part_sum[(long)blockIdx.x * nids + i] = s;
        if (WEIGHTED) {
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < BIN_WAVES; ++k) q += s_w[k * nids + i];
            part_w[(long)blockIdx.x * nids + i] = q;
        } else {
            unsigned long long q = 0;
#pragma unroll
            for (int k = 0; k < BIN_WAVES; ++k) q += s_cnt[k * nids + i];
            part_cnt[(long)blockIdx.x * nids + i] = q;
        }
    }
}

// one workgroup per id; fixed-order strided partial sums + fixed-order LDS tree -> deterministic
__global__ __launch_bounds__(256) void bin_final_kernel(const double* __restrict__ part_sum, const double* __restrict__ part_w,
                                                        const unsigned long long* __restrict__ part_cnt, int nblocks,
                                                        int nids, int weighted, double* __restrict__ sums,
                                                        int64_t* __restrict__ counts, double* __restrict__ wsums, BinTail tail,
                                                        long pstride) {
    // grid y = plane of a batch: partial arrays pstride apart, sums / counts nids apart; the LAST workgroup of the whole grid
    // adds the planes' bandpower vectors to the moments one after the other, in plane order
    double* const sums0 = sums;
    part_sum += (long)blockIdx.y * pstride; part_w += (long)blockIdx.y * pstride; part_cnt += (long)blockIdx.y * pstride;
    sums += (long)blockIdx.y * nids;
    if (counts) counts += (long)blockIdx.y * nids;
    __shared__ double sh_s[256];
    __shared__ double sh_q[256];
    __shared__ unsigned long long sh_c[256];
    __shared__ int s_last;
    const int i = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    unsigned long long c = 0;
    for (int b = t; b < nblocks; b += 256) {
        s += part_sum[(long)b * nids + i];
        if (weighted) q += part_w[(long)b * nids + i];
        else c += part_cnt[(long)b * nids + i];
    }
    sh_s[t] = s; sh_q[t] = q; sh_c[t] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { sh_s[t] += sh_s[t + o]; sh_q[t] += sh_q[t + o]; sh_c[t] += sh_c[t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        if (tail.ticket) __hip_atomic_store(sums + i, sh_s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else sums[i] = sh_s[0];
        if (weighted) { if (wsums) wsums[i] = sh_q[0]; }
        else if (counts) counts[i] = (int64_t)sh_c[0];
    }
    if (!tail.ticket) return;
    if (t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through store of sums[i] has been acknowledged
        s_last = (__hip_atomic_fetch_add(tail.ticket, 1u, OA_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // moments of the bandpower vector b[a] = sums[1 + a] / mcounts[1 + a], a < d = nids - 2 (stats.bin2D [1:-1])
    const int d = nids - 2;
    for (int zb = 0; zb < (int)gridDim.y; ++zb) {
        const double* sm = sums0 + (long)zb * nids;
        for (int a0 = 0; a0 < d; a0 += 256) {              // S, and b staged through LDS in chunks of 256
            const int a = a0 + t;
            double bv = 0.0;
            if (a < d) {
                bv = __hip_atomic_load(sm + 1 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (double)tail.mcounts[1 + a];
                tail.S[a] += bv;
            }
            __syncthreads();
            sh_s[t] = bv;
            __syncthreads();
            // C rows a0 .. a0+255 need every b: column values are re-read from global (write-through, agent scope)
            for (long e = t; e < (long)((d - a0 < 256) ? d - a0 : 256) * d; e += 256) {
                const int ra = (int)(e / d), cb = (int)(e - (long)ra * d);
                const double bb = __hip_atomic_load(sm + 1 + cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (double)tail.mcounts[1 + cb];
                tail.C[(long)(a0 + ra) * d + cb] += sh_s[ra] * bb;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        tail.n[0] += (int64_t)gridDim.y;
        __hip_atomic_store(tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
}

template <typename T>
static int bin_impl(const void* data, const void* data2, double pnorm, bool power, const int32_t* ids, const void* weights,
                    const double* aux, long n, int nids, int mode,
                    int skip_nan, long hp, int nxh, double* sums, int64_t* counts, double* wsums, void* scratch,
                    hipStream_t st, int active_cols = 0, int active_rows = 0, const BinTail* fused = nullptr, int nbatch = 1,
                    long dstride = 0) {
    // nbatch > 1 (bin_power_moments): planes dstride elements of T apart; scratch, sums and counts hold nbatch sets
    const long pstride = (long)2 * BIN_GMAX * nids;
    unsigned wq = 0;                                  // 4-element chunks visited per row (0 = whole rows)
    if (active_cols > 0 && nxh >= 0 && hp > 0 && (long)active_cols < hp && n % hp == 0) wq = (unsigned)((active_cols + 3) / 4);
    const unsigned rb = (wq && active_rows > 0 && 2L * active_rows - 1 < n / hp) ? (unsigned)active_rows : 0u;
    const unsigned rowc = wq ? wq : ((nxh >= 0 && hp > 0) ? (unsigned)(hp >> 2) : 0u);   // as in the kernel
    const long rowp = (long)((rowc + 63u) & ~63u);
    const long nchunks = rowc ? (rb ? 2L * rb - 1 : n / hp) * rowp : (n + 3) / 4;
    int G = (int)((nchunks + BIN_BLOCK - 1) / BIN_BLOCK);
    if (G < 1) G = 1;
    if (G > BIN_GMAX) G = BIN_GMAX;
    double* part_sum = reinterpret_cast<double*>(scratch);
    double* part_w = part_sum + (long)BIN_GMAX * nids;
    unsigned long long* part_cnt = reinterpret_cast<unsigned long long*>(part_w);
    const size_t smem = (size_t)2 * BIN_WAVES * nids * sizeof(double);
    const bool weighted = weights != nullptr;
    BinTail tail{};
    if (fused && !weighted) tail = *fused;
#define OA_BIN_LAUNCH(W, P)                                                                                          \
    {                                                                                                                \
        auto k = bin_kernel<T, W, P>;                                                                                \
        if (smem > 48 * 1024)                                                                                        \
            OA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)smem));                                                                  \
        hipLaunchKernelGGL(k, dim3(G, nbatch), dim3(BIN_BLOCK), smem, st, (const T*)data, (const T*)data2, pnorm, ids, \
                           (const T*)weights, aux, n, nids, mode, skip_nan, (unsigned)(hp > 0 ? hp : 4), nxh, wq, rb, \
                           part_sum, part_w, part_cnt, dstride, pstride);                                            \
    }
    if (weighted && power) OA_BIN_LAUNCH(true, true)
    else if (weighted) OA_BIN_LAUNCH(true, false)
    else if (power) OA_BIN_LAUNCH(false, true)
    else OA_BIN_LAUNCH(false, false)
#undef OA_BIN_LAUNCH
    OA_LAUNCH_CHECK();
    hipLaunchKernelGGL(bin_final_kernel, dim3(nids, nbatch), dim3(256), 0, st, part_sum, part_w, part_cnt, G, nids,
                       weighted ? 1 : 0, sums, counts, wsums, tail, pstride);
    OA_LAUNCH_CHECK();
    return 0;
}

// kappa_hat -> binned auto-power -> moment accumulation in ONE launch (pipeline.hip)
int bin_power_moments(int dtype, const void* k, double norm, const int32_t* ids, long n, int nids, long hp, int nxh, double* sums,
                      int64_t* counts, void* scratch, int active_cols, int active_rows, unsigned* ticket,
                      const int64_t* mcounts, int64_t* mn, double* S, double* C, hipStream_t st, int nbatch, long kstride) {
    // nbatch planes kstride complex elements apart (scratch / sums / counts sized for nbatch sets): their bandpower vectors
    // are added to (n, S, C) in plane order by the last workgroup
    BinTail t{ticket, mcounts, mn, S, C};
    if (dtype == OA_F32)
        return bin_impl<float>(k, k, norm, true, ids, nullptr, nullptr, n, nids, 0, 0, hp, nxh, sums, counts, nullptr, scratch, st,
                               active_cols, active_rows, &t, nbatch, 2 * kstride);
    return bin_impl<double>(k, k, norm, true, ids, nullptr, nullptr, n, nids, 0, 0, hp, nxh, sums, counts, nullptr, scratch, st,
                            active_cols, active_rows, &t, nbatch, 2 * kstride);
}

}  // namespace oa

using namespace oa;

extern "C" {

int oa_digitize(const double* x, long n, const double* edges, int nedges, int32_t* ids, void* stream) {
    OA_REQUIRE(x && edges && ids && n >= 0, "oa_digitize: bad argument");
    OA_REQUIRE(nedges >= 1 && nedges <= DIG_MAX_EDGES, "oa_digitize: nedges must be in [1,4096]");
    hipLaunchKernelGGL(digitize_kernel, dim3(flat_grid(n > 0 ? n : 1)), dim3(256), nedges * sizeof(double),
                       (hipStream_t)stream, x, n, edges, nedges, ids);
    OA_LAUNCH_CHECK();
    return 0;
}

int oa_modl_digitize(const double* ly, const double* lx, int ny, int nx, long pitch, int width, const double* edges,
                     int nedges, int32_t* ids, double* modl_out, void* stream) {
    OA_REQUIRE(ly && lx && edges && ids, "oa_modl_digitize: NULL argument");
    OA_REQUIRE(ny > 0 && nx > 0 && width > 0 && width <= nx && pitch >= width, "oa_modl_digitize: bad geometry");
    OA_REQUIRE(nedges >= 1 && nedges <= DIG_MAX_EDGES, "oa_modl_digitize: nedges must be in [1,4096]");
    int gx = (int)((pitch + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(modl_digitize_kernel, dim3(gx, ny), dim3(256), nedges * sizeof(double), (hipStream_t)stream, ly, lx,
                       ny, nx, pitch, width, edges, nedges, ids, modl_out);
    OA_LAUNCH_CHECK();
    return 0;
}

long oa_bin_scratch_bytes(int nids) {
    if (nids < 1) return -1;
    return (long)2 * BIN_GMAX * nids * (long)sizeof(double);
}

int oa_bin(int dtype, const void* data, const int32_t* ids, const void* weights, const double* aux, long n, int nids,
           int mode, int skip_nan, long herm_pitch, int herm_nxh, double* sums, int64_t* counts, double* wsums,
           void* scratch, void* stream) {
    OA_REQUIRE(data && ids && sums && scratch && n >= 0, "oa_bin: bad argument");
    OA_REQUIRE(nids >= 1 && nids <= BIN_MAX_IDS, "oa_bin: nids (= nedges+1) must be in [1,1024]");
    OA_REQUIRE(mode == 0 || (mode == 1 && aux != nullptr && weights == nullptr), "oa_bin: mode 1 needs aux and no weights");
    OA_REQUIRE(weights ? (wsums != nullptr) : (counts != nullptr), "oa_bin: counts (unweighted) / wsums (weighted) required");
    if (herm_nxh >= 0) OA_REQUIRE(herm_pitch > 0 && herm_pitch % 4 == 0, "oa_bin: herm_pitch must be a positive multiple of 4");
    if (dtype == OA_F32)
        return bin_impl<float>(data, data, 1.0, false, ids, weights, aux, n, nids, mode, skip_nan, herm_pitch, herm_nxh, sums,
                               counts, wsums, scratch, (hipStream_t)stream);
    if (dtype == OA_F64)
        return bin_impl<double>(data, data, 1.0, false, ids, weights, aux, n, nids, mode, skip_nan, herm_pitch, herm_nxh, sums,
                                counts, wsums, scratch, (hipStream_t)stream);
    return fail("oa_bin: bad dtype");
}

int oa_bin_power(int dtype, const void* k1, const void* k2, double norm, const int32_t* ids, const void* weights, long n,
                 int nids, long herm_pitch, int herm_nxh, double* sums, int64_t* counts, double* wsums, void* scratch,
                 int active_cols, int active_rows, void* stream) {
    OA_REQUIRE(k1 && k2 && ids && sums && scratch && n >= 0, "oa_bin_power: bad argument");
    OA_REQUIRE(nids >= 1 && nids <= BIN_MAX_IDS, "oa_bin_power: nids (= nedges+1) must be in [1,1024]");
    OA_REQUIRE(weights ? (wsums != nullptr) : (counts != nullptr), "oa_bin_power: counts (unweighted) / wsums (weighted) required");
    if (herm_nxh >= 0) OA_REQUIRE(herm_pitch > 0 && herm_pitch % 4 == 0, "oa_bin_power: herm_pitch must be a positive multiple of 4");
    if (dtype == OA_F32)
        return bin_impl<float>(k1, k2, norm, true, ids, weights, nullptr, n, nids, 0, 0, herm_pitch, herm_nxh, sums, counts,
                               wsums, scratch, (hipStream_t)stream, active_cols, active_rows);
    if (dtype == OA_F64)
        return bin_impl<double>(k1, k2, norm, true, ids, weights, nullptr, n, nids, 0, 0, herm_pitch, herm_nxh, sums, counts,
                                wsums, scratch, (hipStream_t)stream, active_cols, active_rows);
    return fail("oa_bin_power: bad dtype");
}

}  // extern "C"
