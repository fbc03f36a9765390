"""ctypes binding of libepiekf.so (the C ABI of include/epiekf.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the
compute API raises -- the product path never routes through oracle/ or NumPy."""
from __future__ import annotations

import ctypes as C
import os

from . import layout as L

HERE = os.path.dirname(os.path.abspath(__file__))
# EPIEKF_LIB: load another build of the same ABI instead (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("EPIEKF_LIB") or os.path.join(HERE, "libepiekf.so")

ABI_VERSION = 6      # EPIEKF_ABI_VERSION of include/epiekf.h
ABI_SYMBOLS = [
    "epi_abi_version", "epi_status_string", "epi_model_dim", "epi_ekf_validate", "epi_ekf_workspace_bytes",
    "epi_ekf_precheck_device", "epi_ekf_time_stages_device", "epi_ekf_preferred_lane_block", "epi_ekf_run_device", "epi_ekf_run_host", "epi_ekf_run_host_multi", "epi_host_pool_release", "epi_sialpha_sim_device", "epi_sialpha_score_device", "epi_seirp_sim_device",
    "epi_random_npi_mc_device", "epi_pareto_front_device", "epi_npi_cost_device", "epi_si_controlled_device", "epi_si_controlled_host", "epi_sialpha_sim_host", "epi_seirp_sim_host", "epi_npi_cost_host", "epi_calib_copy_f64_device",
    "epi_rt_expfit_validate", "epi_rt_expfit_run_device", "epi_rt_expfit_run_host",
    "epi_preprocess_workspace_bytes", "epi_preprocess_device", "epi_nnls_affine_fit_device",
    "epi_sweep_run_device", "epi_sweep_prescribe_host", "epi_preprocess_host", "epi_nnls_affine_fit_host", "epi_random_npi_mc_host",
    "epi_sir_sim_device", "epi_sir_sim_host",
]


class EpiError(RuntimeError):
    """Raised for a negative epi_status; .status holds the code, the message is the reference's
    own error() text for the four reference errors (include/epiekf.h)."""

    def __init__(self, status: int, msg: str):
        super().__init__(msg)
        self.status = status


class BatchDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "model", "B", "T", "Sx", "Su", "n_npi", "L", "order",
                                          "obs_type", "r_mode", "q_mode")] + [
        ("out_mask", C.c_uint32), ("phase", C.c_int32), ("path_hint", C.c_int32), ("time_pipe", C.c_int32),
        ("lane_block", C.c_int32), ("shape", C.c_int32), ("storage", C.c_int32), ("exact_nonfinite", C.c_int32),
        ("placement_tries", C.c_int32),                                  # host-pointer entry points: candidate arenas (include/epiekf.h)
        ("test_window", C.c_int32), ("test_flags", C.c_int32)]           # test hooks, 0 in production


class Inputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_series", "u_series", "x", "u", "R_series", "R_scalar", "prm",
                                           "s_init", "Ps_init", "s_final", "Ps_final", "Q")]


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("u_opt", "u_opt_smooth", "S_MINUS", "S_PLUS", "S_SMOOTH", "P_MINUS",
                                           "P_PLUS", "P_SMOOTH", "K_GAIN", "innovations", "rho", "pinv_rank",
                                           "status", "placement")]


class PlacementReport(C.Structure):
    _fields_ = [("tries", C.c_int32), ("chosen", C.c_int32), ("ms", C.c_float * 8)]


class SweepDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "R", "P", "t_hist")]


class PrescribeDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "R", "P", "T", "t_hist", "n_npi", "L", "order", "obs_type")] + [
        ("out_mask", C.c_uint32), ("shape", C.c_int32), ("time_pipe", C.c_int32), ("placement_tries", C.c_int32)]


class PrescribeInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "u", "R_series", "prm", "s_init", "Ps_init", "s_final", "Ps_final", "Q", "eps",
                                           "sp", "J0_prefix", "J1_prefix")]


class PrescribeOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("J0", "J1", "on_front", "i_opt", "u_opt", "S_opt")] + [("extras", Outputs), ("placement", C.c_void_p)]


class SimDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "B", "K", "Su", "n_npi", "noise", "with_cost", "prefix_days",
                                          "u_block")]


class McDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "R", "n_scen", "K", "n_npi", "noise", "prefix_days")] + [
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32)]


class RtDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "B", "T", "Sx", "L", "order")]


RT_OUT_NAMES = ("S_MINUS", "S_PLUS", "P_MINUS", "P_PLUS", "K_GAIN", "S_SMOOTH", "P_SMOOTH", "innovations", "rho")


class RtOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in RT_OUT_NAMES]


class PreDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "S", "T", "n_npi", "W", "first_num_days")] + [("min_cases", C.c_double)]


PRE_OUT_NAMES = ("new_refined", "new_smoothed", "zero_lag", "x_new", "x_total", "R_v", "fatality", "I0", "ip_filled")


class PreOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PRE_OUT_NAMES]


class NnlsDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "S", "D", "n", "max_iters")]


_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as
    /opt/rocm's); device pointers and streams handed to libepiekf.so come from torch, so the library
    must bind to torch's runtime, whichever of the two is loaded first.  Loading torch's copy by path
    here (no `import torch` needed) makes the dynamic linker resolve libepiekf.so's NEEDED
    libamdhip64.so.7 to it, and a later `import torch` reuses the same mapping.  Without torch in the
    environment (e.g. a MEX host) the system runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """Load libepiekf.so; raises if absent (build it with __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build the HIP library first "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        _preload_torch_hip_runtime()
        h = C.CDLL(LIB_PATH)
        h.epi_abi_version.restype = C.c_int
        h.epi_status_string.restype = C.c_char_p
        h.epi_status_string.argtypes = [C.c_int]
        h.epi_model_dim.restype = C.c_int
        h.epi_model_dim.argtypes = [C.c_int]
        h.epi_ekf_validate.restype = C.c_int
        h.epi_ekf_validate.argtypes = [C.POINTER(BatchDesc), C.c_char_p]
        h.epi_ekf_workspace_bytes.restype = C.c_size_t
        h.epi_ekf_workspace_bytes.argtypes = [C.POINTER(BatchDesc)]
        h.epi_ekf_precheck_device.restype = C.c_int
        h.epi_ekf_precheck_device.argtypes = [C.POINTER(BatchDesc), C.POINTER(Inputs), C.c_void_p, C.POINTER(C.c_int),
                                              C.c_char_p]
        h.epi_ekf_preferred_lane_block.restype = C.c_int
        h.epi_ekf_preferred_lane_block.argtypes = [C.POINTER(BatchDesc)]
        h.epi_ekf_run_device.restype = C.c_int
        h.epi_ekf_run_device.argtypes = [C.POINTER(BatchDesc), C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_char_p]
        h.epi_ekf_time_stages_device.restype = C.c_int
        h.epi_ekf_time_stages_device.argtypes = [C.POINTER(BatchDesc), C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p,
                                                 C.c_size_t, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_char_p]
        h.epi_ekf_run_host.restype = C.c_int
        h.epi_ekf_run_host.argtypes = [C.POINTER(BatchDesc), C.POINTER(Inputs), C.POINTER(Outputs), C.c_int,
                                       C.c_char_p]
        h.epi_sialpha_sim_device.restype = C.c_int
        h.epi_sialpha_sim_device.argtypes = [C.POINTER(SimDesc)] + [C.c_void_p] * 9 + [C.c_void_p, C.c_char_p]
        h.epi_sialpha_score_device.restype = C.c_int
        h.epi_sialpha_score_device.argtypes = [C.POINTER(SimDesc)] + [C.c_void_p] * 11 + [C.c_void_p, C.c_char_p]
        h.epi_seirp_sim_device.restype = C.c_int
        h.epi_seirp_sim_device.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
        h.epi_random_npi_mc_device.restype = C.c_int
        h.epi_random_npi_mc_device.argtypes = [C.POINTER(McDesc)] + [C.c_void_p] * 8 + [C.c_void_p, C.c_char_p]
        h.epi_pareto_front_device.restype = C.c_int
        h.epi_pareto_front_device.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_void_p, C.c_char_p]
        h.epi_npi_cost_device.restype = C.c_int
        h.epi_npi_cost_device.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 6 + [C.c_void_p, C.c_char_p]
        h.epi_sialpha_sim_host.restype = C.c_int
        h.epi_sialpha_sim_host.argtypes = [C.POINTER(SimDesc)] + [C.c_void_p] * 9 + [C.c_int, C.c_char_p]
        h.epi_seirp_sim_host.restype = C.c_int
        h.epi_seirp_sim_host.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        h.epi_si_controlled_device.restype = C.c_int
        h.epi_si_controlled_device.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double] + [C.c_void_p] * 5 + [C.c_void_p, C.c_char_p]
        h.epi_si_controlled_host.restype = C.c_int
        h.epi_si_controlled_host.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double] + [C.c_void_p] * 5 + [C.c_int, C.c_char_p]
        h.epi_npi_cost_host.restype = C.c_int
        h.epi_npi_cost_host.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 6 + [C.c_int, C.c_char_p]
        h.epi_rt_expfit_validate.restype = C.c_int
        h.epi_rt_expfit_validate.argtypes = [C.POINTER(RtDesc), C.c_char_p]
        h.epi_rt_expfit_run_device.restype = C.c_int
        h.epi_rt_expfit_run_device.argtypes = [C.POINTER(RtDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtOutputs),
                                               C.c_void_p, C.c_char_p]
        h.epi_rt_expfit_run_host.restype = C.c_int
        h.epi_rt_expfit_run_host.argtypes = [C.POINTER(RtDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RtOutputs),
                                             C.c_int, C.c_char_p]
        h.epi_preprocess_workspace_bytes.restype = C.c_size_t
        h.epi_preprocess_workspace_bytes.argtypes = [C.POINTER(PreDesc)]
        h.epi_preprocess_device.restype = C.c_int
        h.epi_preprocess_device.argtypes = [C.POINTER(PreDesc)] + [C.c_void_p] * 4 + [C.POINTER(PreOutputs), C.c_void_p,
                                                                                        C.c_size_t, C.c_void_p, C.c_char_p]
        h.epi_nnls_affine_fit_device.restype = C.c_int
        h.epi_nnls_affine_fit_device.argtypes = [C.POINTER(NnlsDesc)] + [C.c_void_p] * 7 + [C.c_void_p, C.c_char_p]
        h.epi_calib_copy_f64_device.restype = C.c_int
        h.epi_calib_copy_f64_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_char_p]
        h.epi_ekf_run_host_multi.restype = C.c_int
        h.epi_ekf_run_host_multi.argtypes = [C.POINTER(BatchDesc), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_char_p]
        h.epi_host_pool_release.restype = None
        h.epi_sweep_run_device.restype = C.c_int
        h.epi_sweep_run_device.argtypes = [C.POINTER(BatchDesc), C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p, C.c_size_t,
                                           C.POINTER(SweepDesc)] + [C.c_void_p] * 7 + [C.c_void_p, C.c_char_p]
        h.epi_sweep_prescribe_host.restype = C.c_int
        h.epi_sweep_prescribe_host.argtypes = [C.POINTER(PrescribeDesc), C.POINTER(PrescribeInputs), C.POINTER(PrescribeOutputs),
                                               C.c_int, C.POINTER(C.c_int), C.c_char_p]
        h.epi_sir_sim_device.restype = C.c_int
        h.epi_sir_sim_device.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
        h.epi_sir_sim_host.restype = C.c_int
        h.epi_sir_sim_host.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
        h.epi_preprocess_host.restype = C.c_int
        h.epi_preprocess_host.argtypes = [C.POINTER(PreDesc)] + [C.c_void_p] * 4 + [C.POINTER(PreOutputs), C.c_int, C.c_char_p]
        h.epi_nnls_affine_fit_host.restype = C.c_int
        h.epi_nnls_affine_fit_host.argtypes = [C.POINTER(NnlsDesc)] + [C.c_void_p] * 7 + [C.c_int, C.c_char_p]
        h.epi_random_npi_mc_host.restype = C.c_int
        h.epi_random_npi_mc_host.argtypes = [C.POINTER(McDesc)] + [C.c_void_p] * 8 + [C.c_int, C.c_char_p]
        if h.epi_abi_version() != ABI_VERSION:
            raise ImportError("libepiekf.so ABI version mismatch")
        _lib = h
    return _lib


def check(rc: int, err_buf) -> None:
    if rc != 0:
        msg = err_buf.value.decode(errors="replace") if err_buf is not None and err_buf.value else \
            lib().epi_status_string(rc).decode()
        raise EpiError(rc, msg)


def make_desc(model, B, T, Sx, Su, n_npi, L_, order, obs_type, r_mode, out_mask, q_mode=0) -> BatchDesc:
    d = BatchDesc()
    d.abi_version = ABI_VERSION
    d.model = L.MODEL_IDS[model] if isinstance(model, str) else int(model)
    d.B, d.T, d.Sx, d.Su, d.n_npi, d.L, d.order = int(B), int(T), int(Sx), int(Su), int(n_npi), int(L_), int(order)
    if isinstance(obs_type, str):
        d.obs_type = L.OBS_IDS.get(obs_type, 99)   # unknown strings reach the library's own check
    else:
        d.obs_type = int(obs_type)
    d.r_mode, d.q_mode, d.out_mask, d.phase = int(r_mode), int(q_mode), int(out_mask), 0
    d.path_hint, d.time_pipe, d.lane_block, d.shape, d.storage, d.exact_nonfinite = 0, 0, 0, 0, 0, 0
    d.placement_tries, d.test_window, d.test_flags = 0, 0, 0
    return d
