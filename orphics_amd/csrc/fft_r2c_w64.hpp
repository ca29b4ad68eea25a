// One-wave-per-row R2C pass for band-limited consumers (f32, rows of 8192 reals = 4096 packed complex points).
//
// The general row pass (row_fft_body) spends its time in LDS and barriers, not in HBM: 16 points per thread means three
// radix-16 stages = three full-row LDS round trips (192 KB per row against 128 B/clk/CU) and seven workgroup barriers
// per row; with the global loads removed it still takes 54 of its 66 us at 8192^2 (tools/r2c_bench.py, OA_R2C_NOLOAD).
// Here ONE wave owns a row and every lane holds 64 points: 4096 = 64 x 64 is two in-register radix-64 stages (each
// 8 x 8, constants W64^m) around ONE LDS transpose, with no workgroup barrier at all (a single wave's LDS operations
// execute in order).  The second stage computes only the output bins the consumers keep (columns < 512 and their
// mirror images, which the real-transform untangle needs): 2 of the 8 outputs of each outer radix-8.
// Lane j:  stage 1: B_j[k1] = sum_t z[j + 64 t] W64^(t k1), times W4096^(j k1)  -> LDS[k1][j]
//          stage 2 (lane k1): Z[k1 + 64 k2] = sum_j LDS[k1][j] W64^(j k2),  k2 in [0,8) and [56,64)
//          untangle: X[k] = E + W8192^k O,  E = (Z[k] + conj Z[4096-k]) / 2,  O = -i (Z[k] - conj Z[4096-k]) / 2
#pragma once
#include "fft_kernels.hpp"

namespace oa {

struct W64Tab {
    static constexpr float c[64] = {1.0f, 0.9951847266721969f, 0.9807852804032304f, 0.9569403357322088f, 0.9238795325112867f, 0.881921264348355f, 0.8314696123025452f, 0.773010453362737f, 0.7071067811865476f, 0.6343932841636455f, 0.5555702330196023f, 0.4713967368259978f, 0.38268343236508984f, 0.29028467725446233f, 0.19509032201612833f, 0.09801714032956077f, 6.123233995736766e-17f, -0.09801714032956065f, -0.1950903220161282f, -0.29028467725446216f, -0.3826834323650897f, -0.4713967368259977f, -0.555570233019602f, -0.6343932841636454f, -0.7071067811865475f, -0.773010453362737f, -0.8314696123025453f, -0.8819212643483549f, -0.9238795325112867f, -0.9569403357322088f, -0.9807852804032304f, -0.9951847266721968f, -1.0f, -0.9951847266721969f, -0.9807852804032304f, -0.9569403357322089f, -0.9238795325112868f, -0.881921264348355f, -0.8314696123025455f, -0.7730104533627371f, -0.7071067811865477f, -0.6343932841636459f, -0.5555702330196022f, -0.47139673682599786f, -0.38268343236509034f, -0.29028467725446244f, -0.19509032201612866f, -0.09801714032956045f, -1.8369701987210297e-16f, 0.09801714032956009f, 0.1950903220161283f, 0.29028467725446205f, 0.38268343236509f, 0.4713967368259976f, 0.5555702330196018f, 0.6343932841636456f, 0.7071067811865474f, 0.7730104533627367f, 0.8314696123025452f, 0.8819212643483548f, 0.9238795325112865f, 0.9569403357322088f, 0.9807852804032303f, 0.9951847266721969f};
    static constexpr float s[64] = {0.0f, 0.0980171403295606f, 0.19509032201612825f, 0.29028467725446233f, 0.3826834323650898f, 0.47139673682599764f, 0.5555702330196022f, 0.6343932841636455f, 0.7071067811865475f, 0.773010453362737f, 0.8314696123025452f, 0.8819212643483549f, 0.9238795325112867f, 0.9569403357322089f, 0.9807852804032304f, 0.9951847266721968f, 1.0f, 0.9951847266721969f, 0.9807852804032304f, 0.9569403357322089f, 0.9238795325112867f, 0.881921264348355f, 0.8314696123025455f, 0.7730104533627371f, 0.7071067811865476f, 0.6343932841636455f, 0.5555702330196022f, 0.47139673682599786f, 0.3826834323650899f, 0.2902846772544624f, 0.1950903220161286f, 0.09801714032956083f, 1.2246467991473532e-16f, -0.09801714032956059f, -0.19509032201612836f, -0.2902846772544621f, -0.38268343236508967f, -0.47139673682599764f, -0.555570233019602f, -0.6343932841636453f, -0.7071067811865475f, -0.7730104533627367f, -0.8314696123025452f, -0.8819212643483549f, -0.9238795325112865f, -0.9569403357322088f, -0.9807852804032303f, -0.9951847266721969f, -1.0f, -0.9951847266721969f, -0.9807852804032304f, -0.9569403357322089f, -0.9238795325112866f, -0.881921264348355f, -0.8314696123025455f, -0.7730104533627369f, -0.7071067811865477f, -0.6343932841636459f, -0.5555702330196022f, -0.4713967368259979f, -0.3826834323650904f, -0.2902846772544625f, -0.19509032201612872f, -0.0980171403295605f};
};

// the input map is read exactly once per reconstruction: non-temporal loads keep it from displacing the small work
// planes in L2 / infinity cache (+2 % reconstructions/s, the column stages 5-10 % shorter; -DOA_W64_PLAIN_LOADS: A/B)
OA_HD cx<float> ld_once(const cx<float>* p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OA_W64_PLAIN_LOADS)
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
    return mk<float>(v.x, v.y);
#else
    return *p;
#endif
}

// W64^m = exp(-2 pi i m / 64), m a compile-time constant after unrolling
OA_HD cx<float> w64(int m) { return mk<float>(W64Tab::c[m & 63], -W64Tab::s[m & 63]); }

// In-register DFT of 64 points, in place.  Input natural order; output bin k = a + 8 b is left in v[8 a + b].
// PRUNE: only the bins k in [0,8) (v[8 a]) and [56,64) (v[8 a + 7]) are produced.  (Callers that need a few more
// bins -- b = 1, 6 -- use the full transform and simply do not read the rest: after unrolling the unused butterfly
// outputs are dead code.)
template <bool PRUNE>
OA_HD void dft64(cx<float>* v) {
    // inner layer: for each s0, DFT-8 over s1 of v[8 s1 + s0] -> bin a, times W64^(s0 a), stored at v[8 a + s0]
#pragma unroll
    for (int s0 = 0; s0 < 8; ++s0) {
        cx<float> t[8];
#pragma unroll
        for (int s1 = 0; s1 < 8; ++s1) t[s1] = v[8 * s1 + s0];
        Dft<float, 8>::run(t);
#pragma unroll
        for (int a = 0; a < 8; ++a) v[8 * a + s0] = (s0 * a) ? t[a] * w64(s0 * a) : t[a];
    }
    // outer layer: for each a, DFT-8 over s0 of v[8 a + s0] -> bin b, i.e. output k = a + 8 b, stored at v[8 a + b]
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        cx<float>* t = v + 8 * a;
        if (!PRUNE) {
            Dft<float, 8>::run(t);
        } else {
            const float h = 0.70710678118654752440f;
            // b = 0: plain sum;  b = 7: sum_s t[s] W8^(-s)
            const cx<float> p04 = t[0] + t[4], m04 = t[0] - t[4], p26 = t[2] + t[6], m26 = t[2] - t[6];
            const cx<float> p15 = t[1] + t[5], m15 = t[1] - t[5], p37 = t[3] + t[7], m37 = t[3] - t[7];
            const cx<float> x0 = (p04 + p26) + (p15 + p37);
            // W8^-1 = (1+i)h, W8^-2 = i, W8^-3 = (-1+i)h:  m15 (1+i) h + m37 (-1+i) h = h [(m15 - m37) + i (m15 + m37)]
            const cx<float> d = m15 - m37, e = m15 + m37;
            const cx<float> x7 = add_pi(m04, m26) + add_pi(d, e) * h;
            t[0] = x0;
            t[7] = x7;
        }
    }
}

struct RowW64Args {
    const cx<float>* in;      // real rows viewed as packed complex: z[n] = x[2n] + i x[2n+1]
    cx<float>* out;
    long in_pitch, out_pitch; // complex elements
    const cx<float>* tw;      // W_M^k, M = 2^logTw >= 8192
    int logTw;
    float scale;
    int wcols;                // columns produced (<= 512)
    int ny, nwg;
};

// LDS: the 64 x 64 transpose goes through ONE 64 x 65 plane of 4-byte words, twice (real parts, then imaginary parts:
// the b32 writes of consecutive lanes and the row-strided reads are both conflict-free, and two b32 passes cost the LDS
// the same cycles as one b64 pass).  16.6 KB per wave instead of 33: the four persistent waves of a CU (one per SIMD,
// each owning the whole register file: 512) hold 66 KB, so the coarse-grid kernels of ANOTHER reconstruction (32-column
// tiles: 34 KB; row stage 18 KB) find LDS on every CU while this kernel streams the map -- with the 33 KB layout the
// four waves held 132 of the 160 KB for the whole launch and every launch of the other streams waited for its last
// row (tools/overlap_trace.sh: col_fft 10.6 -> 42 us, col_div 19.7 -> 62 us under an R2C; profiles/r03b_overlap.txt).
// The real-transform untangle partners use the first 4 KB of the same buffer.
// Tried before: halving the LDS by transposing in two halves of 32 rows needs v and u live together within 256
// registers and spills (176 us at 8192^2).  Prefetching the next row -- all of it into 128 more registers, or into the
// first stage's registers once they are dead -- measured slower than not prefetching (68.6 / 68.9 us vs 61.1; the four
// waves of a CU cover each other's load phases), so each row simply loads, transforms, stores.
constexpr int W64_LDS_STRIDE = 65;
constexpr size_t W64_LDS_BYTES = (size_t)64 * W64_LDS_STRIDE * sizeof(float);
constexpr size_t W64_LDS_BYTES_CX = (size_t)64 * W64_LDS_STRIDE * sizeof(cx<float>);   // two-waves-per-row kernel below

// one row: 4096 packed samples at src (lane j reads src[64 t]) -> X[m] = untangled output column j + 64 m, m < 8 (times scale)
// W128^m = exp(-2 pi i m / 128): U[m] = W8192^(j + 64 m) = U[0] W128^m
OA_HD cx<float> w128(int m) {
    constexpr float c[8] = {1.0f, 0.99879545620517241f, 0.99518472667219693f, 0.98917650996478101f, 0.98078528040323043f, 0.97003125319454397f,
                            0.95694033573220882f, 0.94154406518302081f};
    constexpr float sn[8] = {0.0f, 0.049067674327418015f, 0.098017140329560604f, 0.14673047445536175f, 0.19509032201612825f, 0.24298017990326387f,
                             0.29028467725446233f, 0.33688985339222005f};
    return mk<float>(c[m & 7], -sn[m & 7]);
}

template <class Ctx>
OA_HD void w64_row(Ctx& ctx, const RowW64Args& a, const cx<float>* src, const cx<float>* Pin, const cx<float>* Qin, const cx<float>* Uin, cx<float>* X) {
#ifdef OA_W64_DERIVE_TW
    // only the bases P[1], Q[1], U[0] are kept across rows; the rest is rebuilt per row (19 complex products, 3 deep): 42
    // registers less while the 64 points are live -- what two waves per SIMD need
    cx<float> P[8], Q[8], U[8];
    P[0] = Q[0] = mk<float>(1.f, 0.f);
    P[1] = Pin[1]; Q[1] = Qin[1];
    P[2] = P[1] * P[1]; P[3] = P[2] * P[1]; P[4] = P[2] * P[2]; P[5] = P[4] * P[1]; P[6] = P[4] * P[2]; P[7] = P[4] * P[3];
    Q[2] = Q[1] * Q[1]; Q[3] = Q[2] * Q[1]; Q[4] = Q[2] * Q[2]; Q[5] = Q[4] * Q[1]; Q[6] = Q[4] * Q[2]; Q[7] = Q[4] * Q[3];
#else
    const cx<float>* P = Pin; const cx<float>* Q = Qin; const cx<float>* U = Uin;
#endif
    float* sf = reinterpret_cast<float*>(ctx.smem());
    cx<float>* s = reinterpret_cast<cx<float>*>(ctx.smem());   // untangle exchange (first 4 KB)
    const int j = ctx.tid();
    const int jm = (64 - j) & 63;
    cx<float> v[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) v[t] = ld_once(src + 64 * t);
    dft64<false>(v);                           // bin k1 = aa + 8 b sits in v[8 aa + b]
#pragma unroll
    for (int aa = 0; aa < 8; ++aa) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int k1 = aa + 8 * b;
            if (k1) v[8 * aa + b] = v[8 * aa + b] * ((aa && b) ? P[aa] * Q[b] : (aa ? P[aa] : Q[b]));
        }
    }
    // transpose, real parts: lane j writes word [k1][j], lane k1 = j reads [j][t]
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
        for (int b = 0; b < 8; ++b) sf[(aa + 8 * b) * W64_LDS_STRIDE + j] = v[8 * aa + b].x;
    ctx.sync();
    // (the transposed real parts go straight into the .x slots -- dead once written -- while the .y slots still hold the
    // untransposed imaginary parts of the first stage: no second 64-register array, which the R-split build cannot afford)
#pragma unroll
    for (int t = 0; t < 64; ++t) v[t].x = sf[j * W64_LDS_STRIDE + t];
    ctx.sync();                                // every read of the plane precedes the writes below
#pragma unroll
    for (int aa = 0; aa < 8; ++aa)
#pragma unroll
        for (int b = 0; b < 8; ++b) sf[(aa + 8 * b) * W64_LDS_STRIDE + j] = v[8 * aa + b].y;
    ctx.sync();
#pragma unroll
    for (int t = 0; t < 64; ++t) v[t].y = sf[j * W64_LDS_STRIDE + t];
    ctx.sync();
    dft64<true>(v);                            // lane k1 = j: Z[k1 + 64 m] in v[8 m], Z[k1 + 64 (56 + m)] in v[8 m + 7]
#pragma unroll
    for (int m = 0; m < 8; ++m) s[m * 64 + j] = v[8 * m + 7];
    ctx.sync();
#ifdef OA_W64_DERIVE_TW
    U[0] = Uin[0];
#pragma unroll
    for (int m = 1; m < 8; ++m) U[m] = U[0] * w128(m);
#endif
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const cx<float> Zk = v[8 * m];
        // partner Z[4096 - k]: lane (64 - j) & 63, high slot 63 - m (j > 0) or 64 - m (j = 0; m = 0: Z[0] itself)
        const int idx = j ? (7 - m) : (8 - m);
        cx<float> Zm = Zk;
        if (m > 0 || j) Zm = s[(idx & 7) * 64 + jm];
        const cx<float> E = (Zk + conj(Zm)) * 0.5f;
        const cx<float> O = mul_mi(Zk - conj(Zm)) * 0.5f;
        X[m] = (E + U[m] * O) * a.scale;
    }
    ctx.sync();                                // the partner reads precede the next row's transpose writes
}

template <class Ctx>
OA_HD void row_r2c_w64_body(Ctx& ctx, const RowW64Args& a) {
    const int j = ctx.tid();                       // lane = point residue (stage 1) = bin residue k1 (stage 2)
    const int sh = a.logTw - 12;                   // W4096^e = tw[e << sh]
    // per-lane twiddle bases, kept across the rows of this wave: P[a] = W4096^(j a), Q[b] = W4096^(8 j b);
    // untangle factors U[m] = W8192^(j + 64 m)
    cx<float> P[8], Q[8], U[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        P[i] = a.tw[(unsigned)((j * i) & 4095) << sh];
        Q[i] = a.tw[(unsigned)((8 * j * i) & 4095) << sh];
        U[i] = a.tw[(unsigned)(j + 64 * i) << (sh - 1)];
    }
    for (long row = ctx.bid_x(); row < a.ny; row += a.nwg) {
        cx<float> X[8];
        w64_row(ctx, a, a.in + row * a.in_pitch + j, P, Q, U, X);
        cx<float>* dst = a.out + row * a.out_pitch + j;
#pragma unroll
        for (int m = 0; m < 8; ++m)
            if (j + 64 * m < a.wcols) dst[64 * m] = X[m];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 16384-point rows (8192 packed complex points): TWO waves per row.  Wave 0 transforms the even packed samples, wave 1
// the odd ones -- each exactly the 4096-point two-stage transform above -- and a radix-2 combine
//   Z[k] = Ze[k] + W8192^k Zo[k],   Z[8192 - k] = Ze[4096 - k] + conj(W8192^k) Zo[4096 - k]
// in front of the untangle  X[k] = E + W16384^k O  finishes the row.  Columns k < 768 only (bins k2 <= 11 and their
// mirror images k2 >= 52 of each half transform).  One workgroup barrier per exchange (two waves).
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t W64X2_LDS_BYTES = 2 * W64_LDS_BYTES_CX;
constexpr int W64X2_KEEP = 12;     // kept bins k2 per side of each half transform

template <class Ctx>
OA_HD void row_r2c_w64x2_body(Ctx& ctx, const RowW64Args& a) {
    cx<float>* s = reinterpret_cast<cx<float>*>(ctx.smem());
    const int tid = ctx.tid(), w = tid >> 6, j = tid & 63;
    cx<float>* T = s + w * (64 * W64_LDS_STRIDE);          // this wave's transpose buffer
    const int sh = a.logTw - 12;                            // W4096^e = tw[e << sh]
    const int jm = (64 - j) & 63;
    constexpr int NM = W64X2_KEEP / 2;                      // output column groups m per wave
    cx<float> P[8], Q[8], Wk[NM], U[NM];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        P[i] = a.tw[(unsigned)((j * i) & 4095) << sh];
        Q[i] = a.tw[(unsigned)((8 * j * i) & 4095) << sh];
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int k = j + 64 * (w * NM + i);
        Wk[i] = a.tw[(unsigned)k << (sh - 1)];             // W8192^k
        U[i] = a.tw[(unsigned)k << (sh - 2)];              // W16384^k
    }
    for (long row = ctx.bid_x(); row < a.ny; row += a.nwg) {
        cx<float> v[64];
        const cx<float>* src = a.in + row * a.in_pitch + w + 2 * j;
#pragma unroll
        for (int t = 0; t < 64; ++t) v[t] = ld_once(src + 128 * t);  // packed sample 2 (j + 64 t) + w
        dft64<false>(v);
#pragma unroll
        for (int aa = 0; aa < 8; ++aa) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int k1 = aa + 8 * b;
                cx<float> x = v[8 * aa + b];
                if (k1) x = x * ((aa && b) ? P[aa] * Q[b] : (aa ? P[aa] : Q[b]));
                T[k1 * W64_LDS_STRIDE + j] = x;
            }
        }
        ctx.sync();
#pragma unroll
        for (int t = 0; t < 64; ++t) v[t] = T[j * W64_LDS_STRIDE + t];
        ctx.sync();                                        // every read of the transposes precedes the writes below
        dft64<false>(v);                                   // lane k1 = j: Zw[k1 + 64 k2], k2 = a + 8 b, in v[8 a + b]
        // exchange area (aliases the transposes): EX[half][slot][lane]; slots 0..11 = bins k2 0..11, 12..23 = bins 52..63
        cx<float>* EX = s + w * (2 * W64X2_KEEP * 64);
#pragma unroll
        for (int k2 = 0; k2 < W64X2_KEEP; ++k2) EX[k2 * 64 + j] = v[8 * (k2 & 7) + (k2 >> 3)];
#pragma unroll
        for (int q = 0; q < W64X2_KEEP; ++q) {
            const int k2 = 64 - W64X2_KEEP + q;
            EX[(W64X2_KEEP + q) * 64 + j] = v[8 * (k2 & 7) + (k2 >> 3)];
        }
        ctx.sync();
        const cx<float>* Ee = s;                           // even-sample half
        const cx<float>* Eo = s + 2 * W64X2_KEEP * 64;     // odd-sample half
        cx<float>* dst = a.out + row * a.out_pitch + j;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int m = w * NM + i, k = j + 64 * m;
            const cx<float> Zk = Ee[m * 64 + j] + Wk[i] * Eo[m * 64 + j];
            // partner bin 4096 - k of the half transforms: lane (64 - j) & 63, bin 63 - m (j > 0) or 64 - m (j = 0)
            const int k2p = j ? (63 - m) : (64 - m);
            const int sp = W64X2_KEEP + (k2p - (64 - W64X2_KEEP));
            cx<float> Zm = Zk;
            if (m > 0 || j) Zm = Ee[(sp % (2 * W64X2_KEEP)) * 64 + jm] + conj(Wk[i]) * Eo[(sp % (2 * W64X2_KEEP)) * 64 + jm];
            const cx<float> E = (Zk + conj(Zm)) * 0.5f;
            const cx<float> O = mul_mi(Zk - conj(Zm)) * 0.5f;
            const cx<float> X = (E + U[i] * O) * a.scale;
            if (k < a.wcols) dst[64 * m] = X;
        }
        ctx.sync();                                        // the exchange reads precede the next row's transpose writes
    }
}

}  // namespace oa
