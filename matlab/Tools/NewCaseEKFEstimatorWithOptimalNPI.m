function [u_opt, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH, K_GAIN, innovations, rho] = NewCaseEKFEstimatorWithOptimalNPI(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order)
% Drop-in replacement of the reference's Tools/NewCaseEKFEstimatorWithOptimalNPI.m (Tools/ output order, 10 outputs).
% For the MATLAB-Coder twin's output order (MatlabCodeGenerator/NewCaseEKFEstimatorWithOptimalNPI.m:1) call
% epiekf_mex with model id 5 and reorder: [u_opt, S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho].
if isequal(params.obs_type, 'NEWCASES'), ot = 0; elseif isequal(params.obs_type, 'TOTALCASES'), ot = 1; else, error('unknown observation type'); end
prm = epiekf_pack_params(params, size(u, 1), v_bar, beta, gamma, 4);
o = epiekf_mex(4, u, x, prm, s_init(:), Ps_init, s_final(:), Ps_final, Q_w, R_v, inv_monitor_len, order, ot);
u_opt = o.u_opt; S_MINUS = o.S_MINUS; S_PLUS = o.S_PLUS; S_SMOOTH = o.S_SMOOTH;
P_MINUS = o.P_MINUS; P_PLUS = o.P_PLUS; P_SMOOTH = o.P_SMOOTH; K_GAIN = o.K_GAIN; innovations = o.innovations; rho = o.rho;
end
