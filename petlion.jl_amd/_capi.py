"""ctypes binding of the C ABI in include/petlion_hip.h (libpetlion_hip.so).

The product path has NO CPU fallback: `load()` raises if the HIP library is missing, and `plh_model_create` fails loudly
when no GPU is visible.  (tests may pass an explicit `path=` to exercise the same C ABI built against the test-only wave
emulator; the package itself never does.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PETLION_HIP_LIB") or os.path.join(HERE, "libpetlion_hip.so")     # the override is for build experiments (tools/opt_level_check.sh)

PLH_HOST, PLH_DEVICE, PLH_HOST_ASYNC = 0, 1, 2
PREC_F64, PREC_MIXED, PREC_F64_REFORDER = 0, 1, 2
PART_BLOCK, PART_CYCLIC = 0, 1
MODE_I, MODE_V, MODE_DT, MODE_P, MODE_ETA_P, MODE_RES, MODE_DSTATE = 0, 1, 2, 3, 4, 5, 6
DSTATE = {"dc_s_p_max": 1, "dc_s_p_min": 2, "dc_s_n_max": 3, "dc_s_n_min": 4, "dc_e_max": 5, "dc_e_min": 6}
VAL_CONST, VAL_HOLD, VAL_REST, VAL_TABLE, VAL_EXPR = 0, 1, 2, 3, 4
CHEM_LCO, CHEM_NMC, CHEM_LGM50 = 0, 1, 2
FLAG_RUNNING, ERR_INIT, ERR_STALL, ERR_MAXITERS, ERR_OUTPUT_FULL = -1, -11, -12, -13, -14

BOUND_FIELDS = ["V_max", "V_min", "SOC_max", "SOC_min", "T_max", "c_s_n_max", "I_max", "I_min", "eta_plating_min",
                "c_e_min", "dfilm_max"]


class ModelDesc(C.Structure):
    _fields_ = [(f, C.c_int) for f in ["chemistry", "N_p", "N_s", "N_n", "N_a", "N_z", "N_r_p", "N_r_n", "temperature",
                                      "aging_SEI", "real_bytes", "precision", "device", "solid_diffusion", "thermodynamic_factor", "rxn", "waves_per_cell"]]


class Bounds(C.Structure):
    _fields_ = [(f, C.c_double) for f in BOUND_FIELDS]


class Run(C.Structure):
    _fields_ = [("mode", C.c_int), ("value_kind", C.c_int), ("value", C.c_double), ("tf", C.c_double), ("bounds", Bounds),
                ("n_tab", C.c_int), ("closure_id", C.c_int), ("tab_t", C.POINTER(C.c_double)), ("tab_v", C.POINTER(C.c_double)),
                ("value_cell", C.POINTER(C.c_double)), ("tf_cell", C.POINTER(C.c_double)),
                ("n_dcol", C.c_int), ("dstate", C.c_int), ("dcol", C.POINTER(C.c_int)), ("dofs", C.POINTER(C.c_int))]


class Opts(C.Structure):
    _fields_ = [("abstol", C.c_double), ("reltol", C.c_double), ("abstol_init", C.c_double), ("reltol_init", C.c_double),
                ("maxiters", C.c_int), ("check_bounds", C.c_int), ("interp_final", C.c_int), ("max_order", C.c_int),
                ("jac_every_step", C.c_int), ("init_step", C.c_double), ("n_tdiscon", C.c_int), ("tdiscon", C.POINTER(C.c_double)),
                ("refine", C.c_int), ("n_tstops", C.c_int), ("tstops", C.POINTER(C.c_double)), ("yp_alg_zero", C.c_int),
                ("n_stop", C.c_int), ("stop_ops", C.POINTER(C.c_double)), ("stop_args", C.POINTER(C.c_double))]


class RunInfo(C.Structure):
    _fields_ = [("flag", C.c_int), ("iterations", C.c_int), ("t_end", C.c_double), ("V", C.c_double), ("I", C.c_double),
                ("SOC", C.c_double), ("T_avg", C.c_double)]


COUNTER_FIELDS = ["n_steps", "n_res", "n_jac", "n_fact", "n_solve", "n_newton", "n_errfail", "n_convfail", "sum_kp2",
                  "n_init_iters"]


class CountersS(C.Structure):
    _fields_ = [(f, C.c_longlong) for f in COUNTER_FIELDS] + [("cyc", C.c_longlong * 8)]


class Outputs(C.Structure):
    _fields_ = [("max_pts", C.c_int), ("t", C.c_void_p), ("V", C.c_void_p), ("I", C.c_void_p), ("SOC", C.c_void_p),
                ("T_avg", C.c_void_p), ("n_pts", C.c_void_p), ("Y_final", C.c_void_p), ("YP_final", C.c_void_p),
                ("run_info", C.c_void_p), ("counters", C.c_void_p), ("Y_all", C.c_void_p)]


RUN_INFO_DTYPE = np.dtype([("flag", np.int32), ("iterations", np.int32), ("t_end", np.float64), ("V", np.float64),
                           ("I", np.float64), ("SOC", np.float64), ("T_avg", np.float64)], align=True)
COUNTERS_DTYPE = np.dtype([(f, np.int64) for f in COUNTER_FIELDS] + [("cyc", np.int64, (8,))], align=True)
assert RUN_INFO_DTYPE.itemsize == C.sizeof(RunInfo) and COUNTERS_DTYPE.itemsize == C.sizeof(CountersS)

EXPORTS = ["plh_model_create", "plh_model_destroy", "plh_register_grid_library", "plh_n_states", "plh_n_diff", "plh_n_theta", "plh_theta_key",
           "plh_theta_default", "plh_lds_bytes", "plh_n_sections", "plh_section", "plh_jac_pattern", "plh_jac_alg_pattern", "plh_last_error", "plh_build_info", "plh_device_count", "plh_abi_layout",
           "plh_initial_guess", "plh_residual", "plh_jacobian", "plh_linear_solve", "plh_linear_solve_refined", "plh_residual_diff", "plh_residual_alg",
           "plh_jacobian_alg", "plh_init_consistent", "plh_integrate", "plh_integrate_sens", "plh_model_attach_closure_library", "plh_last_integrate_compiled", "plh_closure_digest", "plh_last_kernel_ms", "plh_host_alloc", "plh_host_free", "plh_synchronize",
           "plh_comm_unique_id", "plh_comm_create", "plh_comm_destroy", "plh_comm_rank", "plh_comm_size", "plh_ensemble_run"]


class PetlionHipError(RuntimeError):
    pass


_cache = {}


def load(path=None):
    """Load the C-ABI library.  Fails loudly if it has not been built (`python __graft_entry__.py build`)."""
    path = path or LIB_PATH
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise PetlionHipError("%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
                              "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.plh_theta_key.restype = C.c_char_p
    lib.plh_theta_key.argtypes = [C.c_void_p, C.c_int]
    lib.plh_theta_default.restype = C.c_double
    lib.plh_theta_default.argtypes = [C.c_void_p, C.c_int]
    lib.plh_last_error.restype = C.c_char_p
    lib.plh_build_info.restype = C.c_char_p
    lib.plh_last_kernel_ms.restype = C.c_double
    lib.plh_last_kernel_ms.argtypes = [C.c_void_p]
    lib.plh_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
    lib.plh_model_destroy.argtypes = [C.c_void_p]
    lib.plh_register_grid_library.argtypes = [C.c_char_p]
    lib.plh_section.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for f in ("plh_n_states", "plh_n_diff", "plh_n_theta", "plh_n_sections", "plh_lds_bytes"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.plh_jac_pattern.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    lib.plh_jac_alg_pattern.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    lib.plh_abi_layout.argtypes = [C.c_void_p, C.c_int]
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    lib.plh_initial_guess.argtypes = [vp, i, vp, vp, vp, i, vp]
    lib.plh_residual.argtypes = [vp, i, vp, vp, vp, i, d, vp, i, vp]
    lib.plh_jacobian.argtypes = [vp, i, vp, vp, vp, d, i, vp, i, vp]
    lib.plh_linear_solve.argtypes = [vp, i, vp, vp, vp, d, i, vp, i, vp]
    lib.plh_linear_solve_refined.argtypes = [vp, i, vp, vp, vp, d, i, vp, i, i, vp]
    lib.plh_residual_diff.argtypes = [vp, i, vp, vp, vp, vp, i, vp]
    lib.plh_residual_alg.argtypes = [vp, i, vp, vp, vp, vp, i, vp]
    lib.plh_jacobian_alg.argtypes = [vp, i, vp, vp, vp, i, vp, i, vp]
    lib.plh_host_alloc.argtypes = [C.POINTER(vp), C.c_ulonglong]
    lib.plh_host_free.argtypes = [vp]
    lib.plh_host_free.restype = None
    lib.plh_synchronize.argtypes = [vp, vp]
    lib.plh_comm_unique_id.argtypes = [C.c_char_p]
    lib.plh_comm_create.argtypes = [i, i, C.c_char_p, i, C.POINTER(vp)]
    lib.plh_comm_destroy.argtypes = [vp]
    lib.plh_comm_destroy.restype = None
    lib.plh_comm_rank.argtypes = [vp]
    lib.plh_comm_size.argtypes = [vp]
    lib.plh_ensemble_run.argtypes = [vp, vp, i, vp, vp, i, C.POINTER(Run), C.POINTER(Opts), i, vp, vp, vp, vp]
    lib.plh_init_consistent.argtypes = [vp, i, vp, i, d, d, vp, vp, vp, vp, i, vp]
    lib.plh_integrate.argtypes = [vp, i, vp, vp, vp, vp, i, C.POINTER(Run), C.POINTER(Opts), C.POINTER(Outputs), i, vp]
    lib.plh_model_attach_closure_library.argtypes = [vp, C.c_char_p]
    lib.plh_last_integrate_compiled.argtypes = [vp]
    lib.plh_closure_digest.argtypes = [i, C.POINTER(Run)]
    lib.plh_closure_digest.restype = C.c_ulonglong
    lib.plh_integrate_sens.argtypes = [vp, i, vp, vp, i, C.POINTER(Run), C.POINTER(Opts), C.POINTER(Outputs), i, vp, vp, vp, vp, i, vp]
    _cache[path] = lib
    return lib


def check(lib, rc, what):
    if rc != 0:
        raise PetlionHipError("%s failed (%d): %s" % (what, rc, lib.plh_last_error().decode("utf-8", "replace")))


def ptr(a):
    """void* of a numpy array (host) or a torch tensor (device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()
