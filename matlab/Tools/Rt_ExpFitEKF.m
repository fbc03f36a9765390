function [S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho] = Rt_ExpFitEKF(x, s_init, params, w_bar, v_bar, Ps_init, Q_w, R_v, beta, gamma, inv_monitor_len, order)
% Drop-in replacement of the reference's Tools/Rt_ExpFitEKF.m (same signature, same outputs): put this directory
% before the reference's Tools/ on the MATLAB path.  Runs on an MI355X through epiekf_rt_mex.
if size(x, 1) ~= 1, error('epiekf:unsupported', 'scalar-observation filter: size(x,1) must be 1'); end
rp = zeros(19, 1);                       % EPI_RT_* rows of include/epiekf.h (1-based here)
rp(1:3) = params(1:3); rp(4:5) = w_bar(1:2); rp(6) = v_bar; rp(7) = R_v; rp(8) = beta; rp(9) = gamma;
rp(10:11) = s_init(1:2); rp(12:15) = Ps_init(:); rp(16:19) = Q_w(:);
o = epiekf_rt_mex(x, rp, inv_monitor_len, order);
S_MINUS = o.S_MINUS; S_PLUS = o.S_PLUS; P_MINUS = o.P_MINUS; P_PLUS = o.P_PLUS; K_GAIN = o.K_GAIN;
S_SMOOTH = o.S_SMOOTH; P_SMOOTH = o.P_SMOOTH; innovations = o.innovations; rho = o.rho;
end
