function [s, e, i, r, p] = SEIRP(alpha_e, alpha_i, kappa, rho, beta, mu, gamma, s0, e0, i0, r0, p0, T, dt)
% Drop-in replacement of the reference's Tools/SEIRP.m (same signature, same outputs).  Runs on an MI355X through
% epiekf_sim_mex; the seven parameters are per-step arrays with at least K-1 = round(T/dt)-1 elements, as in the reference.
K = round(T / dt);
par = zeros(7, K);
names = {alpha_e, alpha_i, kappa, rho, beta, mu, gamma};
for j = 1 : 7
    v = names{j};
    par(j, 1 : K - 1) = v(1 : K - 1);        % the reference reads elements 1 : K-1 (an index error if there are fewer)
end
o = epiekf_sim_mex('seirp', par, [s0; e0; i0; r0; p0], dt, []);
s = o(1, :); e = o(2, :); i = o(3, :); r = o(4, :); p = o(5, :);
end
