"""ctypes binding of liborphics_amd.so (C-ABI declared in include/orphics_amd.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
product entry point raises.  Build it with ``python -c "import __graft_entry__
as g; g.build()"`` or ``make -C orphics_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORPHICS_AMD_LIB", os.path.join(_HERE, "liborphics_amd.so"))  # override: tuning builds only

OA_F32 = 0
OA_F64 = 1
ABI_VERSION = 402     # include/orphics_amd.h OA_ABI_VERSION: the signatures below are those of this version

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_long = ctypes.c_long
c_double = ctypes.c_double
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/orphics_amd.h one to one
SIGNATURES = {
    "oa_last_error": (ctypes.c_char_p, []),
    "oa_version": (c_int, []),
    "oa_device_count": (c_int, []),
    "oa_plan_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "oa_plan_destroy": (c_int, [c_void_p]),
    "oa_plan_kpitch": (c_long, [c_void_p]),
    "oa_plan_scratch_bytes": (c_long, [c_void_p]),
    "oa_plan_set_laxes": (c_int, [c_void_p, c_void_p, c_void_p]),
    "oa_fft_r2c": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_void_p]),
    "oa_fft_c2r": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p]),
    "oa_fft_c2r_windowed": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "oa_fft_c2c": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p]),
    "oa_fft_pass": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "oa_fft_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_void_p]),
    "oa_qe_rows": (c_int, [c_void_p] * 6 + [c_double, c_int, c_int, c_int, c_int, c_void_p]),
    "oa_qe_legs_cols": (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p]),
    "oa_qe_map_legs_cols": (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p]),
    "oa_qe_cols_div": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p]),
    "oa_plan_set_filters": (c_int, [c_void_p] * 4 + [c_int] * 5),
    "oa_plan_set_col_grid": (c_int, [c_void_p, c_int]),
    "oa_plan_col_grid": (c_int, [c_void_p]),
    "oa_plan_rsplit": (c_int, [c_void_p]),
    "oa_plan_div_fused": (c_int, [c_void_p]),
    "oa_plan_set_option": (c_int, [c_void_p, c_int, c_int]),
    "oa_plan_set_bins": (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p]),
    "oa_plan_kappa": (c_void_p, [c_void_p]),
    "oa_plan_bin_counts": (c_void_p, [c_void_p]),
    "oa_qe_tt": (c_int, [c_void_p] * 5 + [c_int, c_void_p]),
    "oa_qe_pol": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oa_qe_mv": (c_int, [c_void_p, c_int] + [c_void_p] * 9 + [c_int] * 7 + [c_void_p]),
    "oa_filter_map": (c_int, [c_void_p] * 5),
    "oa_qe_tt_moments": (c_int, [c_void_p] * 6),
    "oa_qe_tt_moments2": (c_int, [c_void_p] * 7),
    "oa_qe_tt_splits": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "oa_split_cross_power": (c_int, [c_int, c_int, c_void_p, c_void_p, c_double, c_int, c_long, c_int, c_int, c_void_p]),
    "oa_qe_tt_stage": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "oa_mc_run": (c_int, [c_void_p, c_u64, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_mc_run_windowed": (c_int, [c_void_p, c_u64, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_malloc": (c_int, [ctypes.POINTER(c_void_p), ctypes.c_size_t]),
    "oa_free": (c_int, [c_void_p]),
    "oa_memcpy": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p]),
    "oa_memset": (c_int, [c_void_p, c_int, ctypes.c_size_t, c_void_p]),
    "oa_stream_synchronize": (c_int, [c_void_p]),
    "oa_comm_unique_id": (c_int, [c_void_p]),
    "oa_comm_init": (c_int, [c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "oa_comm_destroy": (c_int, [c_void_p]),
    "oa_allreduce": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "oa_hc_to_full": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_full_to_hc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_hcreal_to_full": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_fullreal_to_hc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_f2power": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_long, c_void_p]),
    "oa_cmul_real": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_cmul": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_mul_real": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_axpby_real": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_double, c_long, c_void_p]),
    "oa_rot2": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_qe_legs": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "oa_qe_div": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "oa_lens_split": (c_int, [c_int, c_void_p, c_double, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_lens_maps": (c_int, [c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_lens_maps_hc": (c_int, [c_void_p, c_int, c_void_p, c_long, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_plan_release_pools": (c_int, [c_void_p]),
    "oa_lens_gather": (c_int, [c_void_p] * 6 + [c_int, c_int, c_double, c_void_p, c_int, c_void_p]),
    "oa_hc_derivs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "oa_lens_taylor": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_digitize": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p]),
    "oa_modl_digitize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "oa_bin_scratch_bytes": (c_long, [c_int]),
    "oa_bin": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_long, c_int,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_bin_power": (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_long, c_int, c_long, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "oa_grf_hc": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_void_p, c_void_p]),
    "oa_grf_hc_band": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "oa_grf_mix": (c_int, [c_void_p, c_u64, c_u64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "oa_randn": (c_int, [c_int, c_u64, c_u64, c_void_p, c_long, c_void_p]),
    "oa_moments_add": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_moments_add_binned": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oa_stack_add": (c_int, [c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "oa_probe_copy": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "oa_probe_read": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
}

_lib = None


class OrphicsAmdError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it is absent -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrphicsAmdError(
            "orphics_amd: HIP extension %s not built; run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; import it FIRST so this library binds to the same HIP
    # runtime instance (device pointers and streams are shared with torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.oa_version()
    # version = 100 x MAJOR + minor: MAJOR changes whenever an entry is removed or a signature changes, the minor number when entries
    # are added.  A library of another MAJOR could be called with the wrong argument types; an older minor lacks entries.
    if got // 100 != ABI_VERSION // 100 or got < ABI_VERSION:
        raise OrphicsAmdError("orphics_amd: %s reports C-ABI version %d; this binding was written against %d and accepts %d..%d "
                              "(stale build? run __graft_entry__.build())" % (LIB_PATH, got, ABI_VERSION, ABI_VERSION, ABI_VERSION // 100 * 100 + 99))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().oa_last_error()
        raise OrphicsAmdError(msg.decode() if msg else "orphics_amd: unknown error %d" % rc)


def require_gpu():
    lib = load()
    n = lib.oa_device_count()
    if n <= 0:
        raise OrphicsAmdError("orphics_amd: no HIP device visible; the product path has no CPU fallback")
    return n
