"""Python mirror of include/epiekf_layout.h and the enums of include/epiekf.h."""

MAX_NPI = 12

PRM_DT, PRM_BETA, PRM_GAMMA, PRM_SIGMA, PRM_B, PRM_EPSILON = 0, 1, 2, 3, 4, 5
PRM_S_MIN, PRM_I_MIN, PRM_ALPHA_MIN, PRM_ALPHA_MAX = 6, 7, 8, 9
PRM_A, PRM_U_MIN, PRM_U_MAX, PRM_W_EFF = 10, 22, 34, 46
PRM_V_BAR, PRM_BETA_EKF, PRM_GAMMA_EKF = 58, 59, 60
PRM_COUNT = 61

# model ids (enum epi_model)
MODEL_IDS = {
    "SIAlphaModelEKF": 0,
    "SIAlphaModelEKFOptControlled": 1,
    "SIAlphaModelBackwardEKF": 2,
    "SIAlphaModelBackwardEKFOptControlled": 3,
    "NewCaseEKFEstimatorWithOptimalNPI": 4,
    "NewCaseEKFEstimatorWithOptimalNPI_codegen": 5,
}
MODEL_DIM = {
    "SIAlphaModelEKF": 3,
    "SIAlphaModelEKFOptControlled": 6,
    "SIAlphaModelBackwardEKF": 3,
    "SIAlphaModelBackwardEKFOptControlled": 6,
    "NewCaseEKFEstimatorWithOptimalNPI": 6,
    "NewCaseEKFEstimatorWithOptimalNPI_codegen": 6,
}
OBS_IDS = {"NEWCASES": 0, "TOTALCASES": 1}

# output selection bits (enum epi_out)
OUT_BITS = {
    "u_opt": 1 << 0, "u_opt_smooth": 1 << 1,
    "S_MINUS": 1 << 2, "S_PLUS": 1 << 3, "S_SMOOTH": 1 << 4,
    "P_MINUS": 1 << 5, "P_PLUS": 1 << 6, "P_SMOOTH": 1 << 7,
    "K_GAIN": 1 << 8, "innovations": 1 << 9, "rho": 1 << 10,
}
OUT_ALL = (1 << 11) - 1


def out_rows(name: str, m: int, n_npi: int) -> int:
    """Rows per time step of output `name` ([T, rows, B]); 0 means the array is [T, B]."""
    if name in ("u_opt", "u_opt_smooth"):
        return n_npi
    if name in ("S_MINUS", "S_PLUS", "S_SMOOTH", "K_GAIN"):
        return m
    if name in ("P_MINUS", "P_PLUS", "P_SMOOTH"):
        return m * m
    return 0
