function [s, e, i, r, p] = SEIRPSaturatedResource(alpha_e, alpha_i, kappa, rho, gamma, s0, e0, i0, r0, p0, T, dt, beta_0, beta_s, mu_0, mu_s, sigma, i_0)
% Drop-in replacement of the reference's Tools/SEIRPSaturatedResource.m (same signature, same outputs).  Runs on an
% MI355X through epiekf_sim_mex; recovery and death rates follow tanh((i - i_0)/sigma) as in the reference.
K = round(T / dt);
par = zeros(7, K);
names = {alpha_e, alpha_i, kappa, rho, [], [], gamma};
for j = [1 2 3 4 7]
    v = names{j};
    par(j, 1 : K - 1) = v(1 : K - 1);
end
o = epiekf_sim_mex('seirp', par, [s0; e0; i0; r0; p0], dt, [beta_0; beta_s; mu_0; mu_s; sigma; i_0]);
s = o(1, :); e = o(2, :); i = o(3, :); r = o(4, :); p = o(5, :);
end
