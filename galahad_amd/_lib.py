"""ctypes binding of libgsls.so (the C ABI declared in include/gsls.h).

This is plumbing only: every numeric call goes into the HIP library.  If the library is missing the
import fails loudly -- there is no Python/NumPy fallback for factorize or solve.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsls.so")

i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
p_i32, p_i64, p_f64 = C.POINTER(i32), C.POINTER(i64), C.POINTER(f64)


class Options(C.Structure):
    """struct gsls_options (include/gsls.h) <-> type(ssids_options), src/ssids/datatypes.f90:187-283"""
    _fields_ = [("print_level", i32), ("ordering", i32), ("nemin", i32), ("scaling", i32),
                ("action", i32), ("device", i32), ("reserved2", i32), ("reserved0", i32),
                ("u", f64), ("small", f64), ("multiplier", f64), ("reserved1", f64)]


class Inform(C.Structure):
    """struct gsls_inform (include/gsls.h) <-> type(ssids_inform), src/ssids/inform.f90:17-44"""
    _fields_ = [("flag", i32), ("matrix_dup", i32), ("matrix_missing_diag", i32),
                ("matrix_outrange", i32), ("matrix_rank", i32), ("maxdepth", i32),
                ("maxfront", i32), ("num_delay", i32), ("num_factor", i64), ("num_flops", i64),
                ("num_neg", i32), ("num_sup", i32), ("num_two", i32), ("stat", i32),
                ("hip_error", i32), ("not_first_pass", i32), ("nlevels", i32), ("reserved0", i32),
                ("factor_bytes", i64), ("solve_bytes", i64), ("time_analyse", f64),
                ("time_factor", f64), ("time_solve", f64), ("reserved1", f64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


# every symbol include/gsls.h declares: (restype, argtypes)
SIGNATURES = {
    "gsls_default_options": (None, [C.POINTER(Options)]),
    "gsls_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gsls_destroy": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gsls_analyse": (C.c_int, [C.c_void_p, i32, p_i64, p_i32, p_i32, C.POINTER(Options),
                               C.POINTER(Inform)]),
    "gsls_analyse_matching": (C.c_int, [C.c_void_p, i32, p_i64, p_i32, C.c_void_p, p_i32, C.POINTER(Options),
                                        C.POINTER(Inform)]),
    "gsls_factor": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.POINTER(Options),
                              C.POINTER(Inform)]),
    "gsls_factor_dev": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.POINTER(Options),
                                  C.POINTER(Inform)]),
    "gsls_set_coo": (C.c_int, [C.c_void_p, i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsls_factor_coo": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.POINTER(Options),
                                  C.POINTER(Inform)]),
    "gsls_set_value_part": (C.c_int, [C.c_void_p, i32, C.c_void_p, i64, f64]),
    "gsls_factor_coo_dev": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.POINTER(Options),
                                      C.POINTER(Inform)]),
    "gsls_residual": (C.c_int, [C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, i32, C.c_void_p, i32,
                                C.POINTER(Inform)]),
    "gsls_solve_ir": (C.c_int, [C.c_void_p, C.c_void_p, i32, f64, f64, C.POINTER(i32), C.POINTER(Options),
                                C.POINTER(Inform)]),
    "gsls_solve": (C.c_int, [C.c_void_p, i32, i32, C.c_void_p, i32, C.POINTER(Options),
                             C.POINTER(Inform)]),
    "gsls_solve_dev_rhs": (C.c_int, [C.c_void_p, i32, i32, C.c_void_p, C.c_void_p, i32, C.POINTER(Options),
                                     C.POINTER(Inform)]),
    "gsls_solve_dev": (C.c_int, [C.c_void_p, i32, i32, C.c_void_p, i32, C.POINTER(Options),
                                 C.POINTER(Inform)]),
    "gsls_enquire_posdef": (C.c_int, [C.c_void_p, p_f64, C.POINTER(Inform)]),
    "gsls_enquire_indef": (C.c_int, [C.c_void_p, p_i32, p_f64, C.POINTER(Inform)]),
    "gsls_alter": (C.c_int, [C.c_void_p, p_f64, C.POINTER(Inform)]),
    "gsls_get_symbolic_sizes": (C.c_int, [C.c_void_p, p_i32, p_i64, p_i64]),
    "gsls_get_symbolic": (C.c_int, [C.c_void_p, p_i32, p_i32, p_i64, p_i32, p_i64, p_i64]),
    "gsls_shard": (C.c_int, [C.c_void_p, i32, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gsls_shard_factor_dev": (C.c_int, [C.c_void_p, i32, i32, C.c_void_p, C.c_void_p, C.POINTER(Options),
                                        C.POINTER(Inform)]),
    "gsls_shard_solve_dev": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_shard_failed": (C.c_int, [C.c_void_p, C.POINTER(i32), p_i32]),
    "gsls_shard_repair": (C.c_int, [C.c_void_p, i32, p_i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gsls_shard_get": (C.c_int, [C.c_void_p, p_i32, C.POINTER(i32), p_i32]),
    "gsls_get_layout_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gsls_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gsls_comm_init": (C.c_int, [C.c_void_p, i32, i32, C.c_char_p, C.POINTER(Options)]),
    "gsls_comm_factor_dev": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.POINTER(Options), C.POINTER(Inform)]),
    "gsls_comm_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_comm_collect_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_comm_destroy": (C.c_int, [C.c_void_p]),
    "gsls_comm_factor": (C.c_int, [C.c_void_p, i32, C.c_void_p, C.POINTER(Options), C.POINTER(Inform)]),
    "gsls_comm_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_comm_init_env": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "gsls_get_order": (C.c_int, [C.c_void_p, p_i32]),
    "gsls_get_scaling": (C.c_int, [C.c_void_p, p_f64]),
    "gsls_shard_fast": (C.c_int, [C.c_void_p, i32]),
    "gsls_scale_sym": (C.c_int, [i32, i32, p_i64, p_i32, p_f64, i32, p_f64]),
    "gsls_refine_order_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_refine_order": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Inform)]),
    "gsls_get_factor_stats": (C.c_int, [C.c_void_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "gsls_get_stream": (C.c_void_p, [C.c_void_p]),
    "gsls_last_solve_kernel_seconds": (C.c_int, [C.c_void_p, p_f64, p_f64, p_f64]),
    "gsls_device_count": (C.c_int, []),
    "gsls_version": (C.c_char_p, []),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "galahad_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI and the header drifted apart
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
