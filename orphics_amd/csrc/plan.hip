// Plan lifetime + library-level entry points of the C-ABI.
#include <vector>
#include "common.hpp"
#include "fft_plan.hpp"

namespace oa {
std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}

int plan_ensure_scratch(oa_plan* p, size_t bytes) {
    if (p->scratch_bytes >= bytes) return 0;
    if (p->scratch) {
        OA_HIP(hipDeviceSynchronize());
        OA_HIP(hipFree(p->scratch));
        p->scratch = nullptr;
        p->scratch_bytes = 0;
    }
    OA_HIP(hipMalloc(&p->scratch, bytes));
    p->scratch_bytes = bytes;
    return 0;
}

template <typename T>
static int upload_tables(oa_plan* p) {
    auto tx = make_twiddles<T>(p->nx);
    auto ty = make_twiddles<T>(p->ny);
    OA_HIP(hipMalloc(&p->tw_x, tx.size() * sizeof(cx<T>)));
    OA_HIP(hipMalloc(&p->tw_y, ty.size() * sizeof(cx<T>)));
    OA_HIP(hipMemcpy(p->tw_x, tx.data(), tx.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    OA_HIP(hipMemcpy(p->tw_y, ty.data(), ty.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    for (int i = 0; i < RQ8_NGRIDS; ++i) {       // row grids of the fused row stage (512 points per wave) shorter than or equal to the rows
        if (512 * RQ8_WAVES[i] > p->nx) continue;
        auto t = rq8_make_consts<T>(RQ8_WAVES[i]);
        OA_HIP(hipMalloc(&p->rq8c[i], t.size() * sizeof(cx<T>)));
        OA_HIP(hipMemcpy(p->rq8c[i], t.data(), t.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    }
    return 0;
}
}  // namespace oa

using namespace oa;

extern "C" {

const char* oa_last_error(void) { return last_error_ref().c_str(); }

int oa_version(void) { return OA_ABI_VERSION; }   // round number x 100: bumped whenever a signature in include/orphics_amd.h changes

int oa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int oa_plan_create(int ny, int nx, int dtype, oa_plan** out) {
    OA_REQUIRE(out != nullptr, "oa_plan_create: out is NULL");
    *out = nullptr;
    const bool pow2 = is_pow2(ny) && is_pow2(nx);
    OA_REQUIRE(ny >= 32 && nx >= 32, "oa_plan_create: ny and nx must be >= 32");
    OA_REQUIRE(ny <= 32768 && nx <= 32768, "oa_plan_create: ny and nx must be <= 32768");
    OA_REQUIRE(dtype == OA_F32 || dtype == OA_F64, "oa_plan_create: dtype must be OA_F32 or OA_F64");
    if (!pow2) {
        OA_REQUIRE(ny % 2 == 0 && nx % 2 == 0, "oa_plan_create: sides that are not powers of two must be even");
        OA_REQUIRE(ny <= 8192 && nx <= 8192, "oa_plan_create: sides that are not powers of two must be <= 8192 (chirp-z work planes)");
    }
    if (dtype == OA_F64) OA_REQUIRE(nx <= 16384, "oa_plan_create: float64 plans support nx <= 16384 (LDS row budget)");
    int ndev = 0;
    OA_HIP(hipGetDeviceCount(&ndev));
    OA_REQUIRE(ndev > 0, "oa_plan_create: no HIP device available (the product path has no CPU fallback)");
    oa_plan* p = new oa_plan();
    memset(p, 0, sizeof(*p));
    p->ny = ny; p->nx = nx; p->logNy = ilog2(ny); p->logNx = ilog2(nx);
    p->dtype = dtype; p->kp = kpitch_for(nx); p->pow2 = pow2;
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; return fail("hipGetDevice failed"); }
    p->mixed = !pow2 && mixed_sides_ok(ny, nx);
    int rc = pow2 ? ((dtype == OA_F32) ? upload_tables<float>(p) : upload_tables<double>(p)) : (p->mixed ? mixed_setup(p) : czt_setup(p));
    // scratch for every transform of a power-of-two plan (two hc planes >= one full complex plane) is taken HERE, so
    // that no stream-ordered entry point ever synchronises the device or frees memory (plan_ensure_scratch grows it
    // only for the chirp-z work planes, at set-up time)
    if (!rc && pow2) rc = plan_ensure_scratch(p, 2 * (size_t)ny * p->kp * 2 * (dtype == OA_F32 ? sizeof(float) : sizeof(double)));
    if (rc) { oa_plan_destroy(p); return rc; }
    *out = p;
    return 0;
}

int oa_plan_destroy(oa_plan* p) {
    if (!p) return 0;
    (void)hipDeviceSynchronize();
    pipeline_release(p);
    czt_release(p);
    mixed_release(p);
    if (p->tw_x) (void)hipFree(p->tw_x);
    if (p->tw_y) (void)hipFree(p->tw_y);
    for (void* t : p->rq8c) if (t) (void)hipFree(t);
    for (void* t : p->tw_y_small) if (t) (void)hipFree(t);
    if (p->scratch) (void)hipFree(p->scratch);
    if (p->ly) (void)hipFree(p->ly);
    if (p->lx) (void)hipFree(p->lx);
    if (p->lyd) (void)hipFree(p->lyd);
    if (p->lxd) (void)hipFree(p->lxd);
    if (p->ly64) (void)hipFree(p->ly64);
    if (p->lx64) (void)hipFree(p->lx64);
    delete p;
    return 0;
}

long oa_plan_kpitch(const oa_plan* p) { return p ? p->kp : -1; }

long oa_plan_scratch_bytes(const oa_plan* p) { return p ? (long)p->scratch_bytes : -1; }

int oa_plan_set_laxes(oa_plan* p, const double* host_ly, const double* host_lx) {
    OA_REQUIRE(p && host_ly && host_lx, "oa_plan_set_laxes: NULL argument");
    const size_t es = p->dtype == OA_F32 ? sizeof(float) : sizeof(double);
    if (!p->ly) {
        OA_HIP(hipMalloc(&p->ly, p->ny * es));
        OA_HIP(hipMalloc(&p->lx, p->nx * es));
        OA_HIP(hipMalloc(&p->lyd, p->ny * es));
        OA_HIP(hipMalloc(&p->lxd, p->nx * es));
        OA_HIP(hipMalloc((void**)&p->ly64, p->ny * sizeof(double)));
        OA_HIP(hipMalloc((void**)&p->lx64, p->nx * sizeof(double)));
    }
    OA_HIP(hipMemcpy(p->ly64, host_ly, p->ny * sizeof(double), hipMemcpyHostToDevice));
    OA_HIP(hipMemcpy(p->lx64, host_lx, p->nx * sizeof(double), hipMemcpyHostToDevice));
    // derivative axes: the self-conjugate Nyquist frequency carries no odd (i*l) component of a
    // real field (what np.real() of the reference's full-plane C2C inverse discards)
    std::vector<double> dy(host_ly, host_ly + p->ny), dx(host_lx, host_lx + p->nx);
    dy[p->ny / 2] = 0.0;
    dx[p->nx / 2] = 0.0;
    if (p->dtype == OA_F32) {
        std::vector<float> a(p->ny), b(p->nx), c(p->ny), d(p->nx);
        for (int i = 0; i < p->ny; ++i) { a[i] = (float)host_ly[i]; c[i] = (float)dy[i]; }
        for (int i = 0; i < p->nx; ++i) { b[i] = (float)host_lx[i]; d[i] = (float)dx[i]; }
        OA_HIP(hipMemcpy(p->ly, a.data(), p->ny * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lx, b.data(), p->nx * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lyd, c.data(), p->ny * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lxd, d.data(), p->nx * es, hipMemcpyHostToDevice));
    } else {
        OA_HIP(hipMemcpy(p->ly, host_ly, p->ny * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lx, host_lx, p->nx * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lyd, dy.data(), p->ny * es, hipMemcpyHostToDevice));
        OA_HIP(hipMemcpy(p->lxd, dx.data(), p->nx * es, hipMemcpyHostToDevice));
    }
    p->have_laxes = true;
    return 0;
}

}  // extern "C"
