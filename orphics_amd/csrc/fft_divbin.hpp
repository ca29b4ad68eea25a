// Tail of the single-pass divergence kernel (col_div_body<.., DivBinTail>): radial binning of |kappa|^2 and the moment update,
// GPU only (wave-level reductions).  Replaces bin_kernel<POWER> + bin_final_kernel on the one-call paths: the kappa plane is
// not re-read (14 MB per 8192^2 reconstruction) and two dependent launches (15 us) go away.
//
// Per value the arithmetic is bin_kernel's: v = (double)((re^2 + im^2) * (T)pnorm) * multiplicity, multiplicity 1 on the columns
// 0 and nx/2 of the half plane, 2 between.  Per wave-step the lanes that share an id are summed in a fixed order (DPP inside rows of 16 lanes, then the row totals) and the
// lowest such lane adds the total to the wave's private LDS row (no atomics, fixed order); the rows are summed per workgroup
// in wave order, the workgroups' partials by the LAST workgroup (ticket) in workgroup order: deterministic.
#pragma once
#include "fft_launch.hpp"

namespace oa {

#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx942__) || defined(__gfx950__))
#error "DivBinTail: the ticket hand-over relies on gfx942 / gfx950 behaviour (agent-scope relaxed atomic stores are sc1 write-through, stores count in vmcnt)"
#endif

// Sum of v over the 64 lanes, returned in every lane (wave-uniform): four DPP steps inside each row of 16 lanes (pairs, quads,
// half-row mirror, row mirror: 2 v_mov_dpp + 1 v_add_f64 each -- float64 has no DPP add), then the four row totals in row
// order through v_readlane.  Fixed order: deterministic.  (64-lane butterflies through ds_bpermute were a chain of six LDS
// round trips per step: the tail took longer than the divergence itself.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ double wave_total_f64(double v) {
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

template <typename T>
struct DivBinTail {
    static constexpr bool active = true;
    DivBinFuse f;
    double* rows = nullptr;     // LDS: [waves][nids]

    __device__ __forceinline__ void begin(GpuCtx&, void* lds) {
        rows = reinterpret_cast<double*>(lds);
        const int nw = blockDim.x >> 6;
        for (int i = threadIdx.x; i < nw * f.nids; i += blockDim.x) rows[i] = 0.0;
    }
    __device__ __forceinline__ int id_at(unsigned yfull, int col) const { return f.ids[(long)yfull * f.ipitch + col]; }
    // tile-major copy of the ids on the coarse grid ([tile][coarse row][C], DivBinFuse::ids_t): contiguous per tile instead of 16- /
    // 32-byte row segments of the full-pitch plane (7.3 / 15.3 MB fetched for 3.5 MB: profiles/r04_overfetch.txt)
    __device__ __forceinline__ bool packed_ids() const { return f.ids_t != nullptr; }
    __device__ __forceinline__ int id_at_t(long i) const { return f.ids_t[i]; }
    // id < 0: this lane has no value in this step; pw = re^2 + im^2 of kappa
    __device__ __forceinline__ void add(GpuCtx&, int id, T pw, int col) {
        const int m = (col == 0 || col == f.nxh) ? 1 : (col < f.nxh ? 2 : 0);
        const bool ok = id >= 0 && id < f.nids && m > 0;
        const double v = ok ? (double)(pw * (T)f.pnorm) * (double)m : 0.0;
        const int lane = threadIdx.x & 63;
        double* row = rows + (threadIdx.x >> 6) * f.nids;
        unsigned long long act = __ballot(ok);
        while (act) {
            const int leader = __ffsll((long long)act) - 1;
            const int lid = __shfl(id, leader, 64);
            const bool mine = ok && id == lid;
            const unsigned long long mm = __ballot(mine);
            const double sv = wave_total_f64(mine ? v : 0.0);
            if (lane == leader) row[lid] += sv;
            act &= ~mm;
        }
    }
    __device__ __forceinline__ void finish(GpuCtx&) {
        __shared__ int s_last;
        const int t = threadIdx.x, nt = blockDim.x, nw = nt >> 6, nids = f.nids;
        __syncthreads();
        const long wg = (long)blockIdx.z * gridDim.x + blockIdx.x;
        for (int i = t; i < nids; i += nt) {
            double s = 0.0;
            for (int k = 0; k < nw; ++k) s += rows[k * nids + i];
            __hip_atomic_store(f.part + wg * nids + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's partials have been acknowledged
        __syncthreads();
        if (t == 0)
            s_last = (__hip_atomic_fetch_add(f.ticket, 1u, OA_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.z - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        // last workgroup: per map, in map order: sums[i] = the workgroups' partials in a fixed two-level order -- thread (g, i),
        // i = t mod NI, g = t / NI, adds the partials of workgroups b = g, g + G, ... (four loads in flight), then the G group sums
        // are added in group order; b = sums[1 .. nids-2] / mode counts; moments.  (One thread per id walking all workgroups was
        // a chain of ~100 dependent L2-miss loads: 30-50 us.)
        const int d = nids - 2;
        int NI = 1;
        while (NI < nids && NI < nt) NI <<= 1;
        const int G = nt / NI > 0 ? nt / NI : 1, gi = t / NI, ii = t - gi * NI;
        double* red = rows;                                  // [G][NI] group sums, then b in red[0 .. d) (the rows are spent)
        const int gx = (int)gridDim.x;
        for (int z = 0; z < (int)gridDim.z; ++z) {
            for (int i0 = 0; i0 < nids; i0 += NI) {          // (one round unless nids > workgroup size)
                const int i = i0 + ii;
                __syncthreads();
                if (gi < G) {
                    double s = 0.0;
                    if (i < nids) {
                        const double* pz = f.part + ((long)z * gx) * nids + i;
                        int b = gi;
                        for (; b + 3 * G < gx; b += 4 * G) {
                            const double p0 = __hip_atomic_load(pz + (long)b * nids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const double p1 = __hip_atomic_load(pz + (long)(b + G) * nids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const double p2 = __hip_atomic_load(pz + (long)(b + 2 * G) * nids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const double p3 = __hip_atomic_load(pz + (long)(b + 3 * G) * nids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s += (p0 + p1) + (p2 + p3);
                        }
                        for (; b < gx; b += G) s += __hip_atomic_load(pz + (long)b * nids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    red[gi * NI + ii] = s;
                }
                __syncthreads();
                double tot = 0.0;
                if (gi == 0 && i < nids) {
                    for (int k = 0; k < G; ++k) tot += red[k * NI + ii];
                    f.sums[(long)z * nids + i] = tot;
                }
                __syncthreads();
                if (gi == 0 && i >= 1 && i <= d) red[G * NI + i - 1] = tot / (double)f.mcounts[i];     // b, behind the group sums
            }
            __syncthreads();
            const double* bv = red + G * NI;
            if (f.S) {
                for (int a = t; a < d; a += nt) f.S[a] += bv[a];
                for (long e = t; e < (long)d * d; e += nt) {
                    const int ra = (int)(e / d), cb = (int)(e - (long)ra * d);
                    f.C[e] += bv[ra] * bv[cb];
                }
            }
        }
        if (t == 0) {
            if (f.n) f.n[0] += (int64_t)gridDim.z;
            __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        }
    }
};

}  // namespace oa
