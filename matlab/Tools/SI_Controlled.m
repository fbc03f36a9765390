function [s, i] = SI_Controlled(alpha, beta, s0, i0, K, dt)
% Drop-in replacement of the reference's Tools/SI_Controlled.m (same signature, same outputs): 2-state forward Euler with
% the time-dependent infection rate alpha(1 : K-1), on an MI355X through epiekf_sim_mex.
[s, i] = epiekf_sim_mex('si', alpha(:), [beta; s0; i0], K, dt);
end
