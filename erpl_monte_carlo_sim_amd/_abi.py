"""ctypes mirror of include/erpl_mc.h and the loader of the HIP library.

The product path has NO CPU fallback: `load_library()` raises if the HIP shared object has not
been built (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C erpl_monte_carlo_sim_amd/csrc`).
"""
import ctypes as C
import os

ABI_VERSION = 3
STATE_DIM = 14
IC_DIM = 13
ROCKET_DIM = 2
MOTOR_DIM = 4
SUMMARY_DIM = 16
TRAJ_DIM = 15
DIAG_DIM = 17
MAX_MACH_KNOTS = 16
MAX_CURVE_KNOTS = 32
MAX_WIND_KNOTS = 1024
PROFILE_RING = 256

MOTOR_LIQUID, MOTOR_SOLID = 0, 1
PREC_F64, PREC_F32, PREC_F64_FAST = 0, 1, 2
PRECISIONS = {"f64": PREC_F64, "f32": PREC_F32, "f64_fast": PREC_F64_FAST}
MAX_OVERLAP = 8
DBG_ATMOSPHERE, DBG_AERO, DBG_RHS = 0, 1, 2
FLAG_STOP_AT_APOGEE = 1
FLAG_CAPTURE_POSITION_ONLY = 2

# rows of the summary
(SUM_APOGEE_ALT, SUM_APOGEE_TIME, SUM_FIRST_APOGEE_ALT, SUM_FIRST_APOGEE_TIME, SUM_RANGE,
 SUM_FLIGHT_TIME, SUM_RAIL_EXIT_TIME, SUM_RAIL_EXIT_SPEED, SUM_IMPACT_X, SUM_IMPACT_Y,
 SUM_IMPACT_Z, SUM_STEPS, SUM_RAIL_EXIT_AOA, SUM_RAIL_EXIT_SIDESLIP, SUM_FINAL_VZ,
 SUM_MAX_SPEED) = range(16)

END_MAX_TIME, END_GROUND, END_ALTITUDE, END_COAST, END_APOGEE = range(5)
ST_APOGEE_LATCHED, ST_CHUTE, ST_NAN, ST_INCOMPLETE = 1 << 8, 1 << 9, 1 << 10, 1 << 11
ERR_INCOMPLETE = -5

_M = C.c_double * MAX_MACH_KNOTS
_T = C.c_double * MAX_CURVE_KNOTS


class ErplConfig(C.Structure):
    _fields_ = [
        ("diameter", C.c_double), ("center_of_mass_dry", C.c_double),
        ("Ixx_dry", C.c_double), ("Iyy_dry", C.c_double),
        ("reference_area", C.c_double), ("reference_diameter", C.c_double),
        ("cp_location", C.c_double),
        ("fin_root_chord", C.c_double), ("fin_tip_chord", C.c_double),
        ("fin_span", C.c_double), ("fin_sweep_angle", C.c_double),
        ("parachute_area", C.c_double), ("parachute_cd", C.c_double),
        ("parachute_deployment_altitude", C.c_double),
        ("power_off_drag_factor", C.c_double),
        ("n_cd", C.c_int32), ("n_cp", C.c_int32),
        ("cd_mach", _M), ("cd0", _M), ("cda", _M), ("cp_mach", _M), ("cp_shift", _M),
        ("motor_kind", C.c_int32), ("n_curve", C.c_int32),
        ("curve_time", _T), ("curve_thrust", _T),
        ("sea_level_pressure", C.c_double), ("sea_level_temperature", C.c_double),
        ("temperature_lapse_rate", C.c_double), ("gas_constant", C.c_double),
        ("gravity", C.c_double), ("troposphere_height", C.c_double),
        ("stratosphere_height", C.c_double), ("stratosphere_temp", C.c_double),
        ("dt_initial", C.c_double), ("max_time", C.c_double), ("rail_length", C.c_double),
        ("pitch_damping", C.c_double), ("yaw_damping", C.c_double),
    ]


class ErplBatch(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("precision", C.c_int32), ("k_wind", C.c_int32),
        ("flags", C.c_int32), ("reserved", C.c_int32),
        ("ic", C.c_void_p), ("rocket", C.c_void_p), ("motor", C.c_void_p),
        ("alt_grid", C.c_void_p), ("wind", C.c_void_p),
    ]


class ErplOut(C.Structure):
    _fields_ = [
        ("summary", C.c_void_p), ("status", C.c_void_p),
        ("n_traj", C.c_int64), ("traj_ids", C.c_void_p),
        ("traj_stride", C.c_int64), ("traj_cap", C.c_int64),
        ("traj", C.c_void_p), ("traj_len", C.c_void_p),
    ]


LIB_NAME = "liberpl_mc.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)

# every symbol include/erpl_mc.h declares
RS_GAUSS, RS_DOUBLE = 0, 1   # erpl_mc_legacy_random_streams ops

EXPORTS = ("erpl_mc_abi_version", "erpl_mc_last_error", "erpl_mc_create", "erpl_mc_destroy",
           "erpl_mc_set_config", "erpl_mc_reserve", "erpl_mc_run_batch", "erpl_mc_set_launch",
           "erpl_mc_last_stats", "erpl_mc_ticket_stats", "erpl_mc_set_profiling", "erpl_mc_last_kernel_ms",
           "erpl_mc_kernel_ms_history", "erpl_mc_debug_counters", "erpl_mc_extract_histories", "erpl_mc_set_chunk",
           "erpl_mc_legacy_random_streams", "erpl_mc_legacy_wind_profiles", "erpl_mc_set_waves_per_simd",
           "erpl_mc_set_overlap", "erpl_mc_submit_batch", "erpl_mc_wait_batch", "erpl_mc_synchronize",
           "erpl_mc_debug_eval", "erpl_mc_synth_wind", "erpl_mc_set_adopt", "erpl_mc_get_overlap",
           "erpl_mc_check_batch", "erpl_mc_set_adopt_spin", "erpl_mc_set_short_flight_overlap")

_lib = None


class ErplError(RuntimeError):
    pass


class IncompleteBatch(ErplError):
    """A lane hand-over of a batch timed out (ERPL_ERR_INCOMPLETE): the samples whose status word carries
    ST_INCOMPLETE were not integrated."""


def load_library(path=None):
    """dlopen the HIP library and declare prototypes.  Fails loudly when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ErplError(
            f"{p} not found: the HIP extension is not built. There is no CPU fallback; "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm ships its own copy of the HIP / HSA runtime.  Whichever copy a process maps first
    # serves every later dlopen of that SONAME, and the device buffers torch allocates must come from
    # the runtime that launches this library's kernels: load torch's first, always.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.erpl_mc_abi_version.restype = C.c_int
    lib.erpl_mc_last_error.restype = C.c_char_p
    lib.erpl_mc_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.erpl_mc_destroy.argtypes = [C.c_void_p]
    lib.erpl_mc_set_config.argtypes = [C.c_void_p, C.POINTER(ErplConfig)]
    lib.erpl_mc_reserve.argtypes = [C.c_void_p, C.c_int64]
    lib.erpl_mc_run_batch.argtypes = [C.c_void_p, C.POINTER(ErplBatch), C.POINTER(ErplOut), C.c_void_p]
    lib.erpl_mc_set_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.erpl_mc_set_overlap.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_submit_batch.argtypes = [C.c_void_p, C.POINTER(ErplBatch), C.POINTER(ErplOut), C.c_void_p,
                                         C.POINTER(C.c_int64)]
    lib.erpl_mc_wait_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.erpl_mc_synchronize.argtypes = [C.c_void_p]
    lib.erpl_mc_synth_wind.argtypes = [C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 9 + [C.c_int32, C.c_void_p]
    lib.erpl_mc_debug_eval.argtypes = [C.c_void_p, C.POINTER(ErplBatch), C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    lib.erpl_mc_set_chunk.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_set_adopt.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_set_adopt_spin.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_check_batch.argtypes = [C.c_void_p, C.c_int64]
    lib.erpl_mc_get_overlap.argtypes = [C.c_void_p]
    lib.erpl_mc_get_overlap.restype = C.c_int
    lib.erpl_mc_set_waves_per_simd.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.erpl_mc_ticket_stats.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.erpl_mc_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.erpl_mc_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.erpl_mc_kernel_ms_history.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                              C.POINTER(C.c_int)]
    lib.erpl_mc_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.erpl_mc_extract_histories.argtypes = [C.c_void_p, C.POINTER(ErplBatch), C.c_int64, C.c_void_p, C.c_int64,
                                              C.c_double, C.c_void_p, C.c_void_p]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if a declared symbol is not exported
        if name not in ("erpl_mc_last_error",):
            getattr(lib, name).restype = C.c_int
    v = lib.erpl_mc_abi_version()
    if v != ABI_VERSION:
        raise ErplError(f"ABI version mismatch: library {v}, python {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib, rc, what):
    if rc != 0:
        msg = lib.erpl_mc_last_error()
        if rc == ERR_INCOMPLETE:
            raise IncompleteBatch(f"{what}: {msg.decode() if msg else ''}")
        raise ErplError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
