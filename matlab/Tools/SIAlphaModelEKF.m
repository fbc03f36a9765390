function [u_opt, u_opt_smooth, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH, K_GAIN, innovations, rho] = SIAlphaModelEKF(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order)
% Drop-in replacement of the reference's Tools/SIAlphaModelEKF.m (same signature, same outputs): put this
% directory before the reference's Tools/ on the MATLAB path.  Runs on an MI355X through epiekf_mex.
m = length(s_init);
Q_w = epiekf_expand_Q(Q_w, m, size(x, 2));   % m x m, or m x m x T for a time-varying Q_w (GenericEKF.m:63-76)
if isequal(params.obs_type, 'NEWCASES'), ot = 0; elseif isequal(params.obs_type, 'TOTALCASES'), ot = 1; else, error('unknown observation type'); end
prm = epiekf_pack_params(params, size(u, 1), v_bar, beta, gamma, 0);
o = epiekf_mex(0, u, x, prm, s_init(:), Ps_init, s_final(:), Ps_final, Q_w, R_v, inv_monitor_len, order, ot);
u_opt = o.u_opt; u_opt_smooth = o.u_opt_smooth; S_MINUS = o.S_MINUS; S_PLUS = o.S_PLUS; S_SMOOTH = o.S_SMOOTH;
P_MINUS = o.P_MINUS; P_PLUS = o.P_PLUS; P_SMOOTH = o.P_SMOOTH; K_GAIN = o.K_GAIN; innovations = o.innovations; rho = o.rho;
end
