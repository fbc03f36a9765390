function [s, i, alpha] = SIalpha_Controlled(u, s0, i0, alpha0, u_max, alpha_min, alpha_max, gamma, a, b, beta, s_noise_std, i_noise_std, alpha_noise_std, K, dt)
% Drop-in replacement of the reference's Tools/SIalpha_Controlled.m (same signature, same outputs): put this directory
% before the reference's Tools/ on the MATLAB path.  Runs on an MI355X through epiekf_sim_mex.
% The reference draws randn three times per step (s, i, alpha); randn(3, K) takes the same numbers from the global
% stream in the same order, so a seeded script sees identical noise.
n = size(u, 1);
if n > 12, error('epiekf:unsupported', 'at most 12 NPIs'); end
sp = zeros(48, 1);                       % EPI_SIM_* rows of include/epiekf.h (1-based here)
sp(1:12) = [s0; i0; alpha0; alpha_min; alpha_max; gamma; b; beta; s_noise_std; i_noise_std; alpha_noise_std; dt];
sp(12 + (1:n)) = a(:); sp(24 + (1:n)) = u_max(:);
z = randn(3, K);
[s, i, alpha] = epiekf_sim_mex('sialpha', u(:, 1:K), sp, z);
end
