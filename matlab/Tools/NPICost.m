function [J0, J1] = NPICost(newcases, inputs, weights)
% Drop-in replacement of the reference's Tools/NPICost.m (same signature): both means on the MI355X through
% epiekf_sim_mex, summed in column-major element order.  weights may be n x T or n x 1 (implicit expansion).
[J0, J1] = epiekf_sim_mex('npicost', newcases(:).', inputs, weights);
end
