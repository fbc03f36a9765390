/*
 * epiekf_layout.h -- data-format constants of the batched (SoA) interface.
 *
 * Per-chain model parameters travel as one SoA block  prm[EPI_PRM_COUNT][B]
 * (field-major, chain-minor).  The fields are those of the reference's `params`
 * struct (Tools/TrainPredictPrescribeNPI.m:202-224) plus the three per-call
 * scalar filter knobs v_bar / beta / gamma of the L2 signature
 * (Tools/SIAlphaModelEKF.m:1).  Vector fields are always EPI_MAX_NPI wide;
 * entries >= n_npi must be zero.
 */
#ifndef EPIEKF_LAYOUT_H
#define EPIEKF_LAYOUT_H

#define EPI_MAX_NPI 12

enum {
    EPI_PRM_DT = 0,        /* params.dt        */
    EPI_PRM_BETA = 1,      /* params.beta      */
    EPI_PRM_GAMMA = 2,     /* params.gamma     */
    EPI_PRM_SIGMA = 3,     /* params.sigma     */
    EPI_PRM_B = 4,         /* params.b         */
    EPI_PRM_EPSILON = 5,   /* params.epsilon   */
    EPI_PRM_S_MIN = 6,     /* params.s_min     */
    EPI_PRM_I_MIN = 7,     /* params.i_min     */
    EPI_PRM_ALPHA_MIN = 8, /* params.alpha_min */
    EPI_PRM_ALPHA_MAX = 9, /* params.alpha_max */
    EPI_PRM_A = 10,        /* params.a(1:12)     */
    EPI_PRM_U_MIN = 22,    /* params.u_min(1:12) */
    EPI_PRM_U_MAX = 34,    /* params.u_max(1:12) */
    EPI_PRM_W_EFF = 46,    /* params.w resolved per SURVEY.md A.3 (implicit expansion + linear index) */
    EPI_PRM_V_BAR = 58,    /* v_bar  (argument 9 of the L2 signature)  */
    EPI_PRM_BETA_EKF = 59, /* beta   (argument 12): observation-noise update factor */
    EPI_PRM_GAMMA_EKF = 60,/* gamma  (argument 13): fading-memory factor */
    EPI_PRM_COUNT = 61
};

#endif
