function out = SIAlphaModelEKFOptControlledSweep(u, x, params, s_init, Ps_init, s_final, Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order, epsilons)
% The Pareto sweep of one region in ONE call: what the reference does with
%     for ll = 1 : num_pareto_front_points
%         params.epsilon = human_npi_cost_factor(ll);
%         [~, opt_control_input_smooth, ...] = SIAlphaModelEKFOptControlled(control_input, observations, params, ...);
%     end                                                      (Tools/TrainPredictPrescribeNPI.m:421-460)
% with the same arguments as SIAlphaModelEKFOptControlled plus the vector of cost weights `epsilons`.  Every chain
% shares the region's u and x (x_series = u_series = 0); only row EPI_PRM_EPSILON of the parameter block differs.
% out.<name>(ll, :, :) is what the ll-th call of the loop returns (P_* with vec'd matrices: reshape(.., 6, 6, [])).
B = numel(epsilons);
n_npi = size(u, 1); T = size(x, 2); m = 6;
if ~isequal(size(Q_w), [m m]), Q_w = epiekf_expand_Q(Q_w, m, T); end
if ndims(Q_w) == 3, error('epiekf:unsupported', 'the batched gateway takes a fixed Q_w'); end
if isequal(params.obs_type, 'NEWCASES'), ot = 0; elseif isequal(params.obs_type, 'TOTALCASES'), ot = 1; else, error('unknown observation type'); end
prm = zeros(B, 61);
for ll = 1 : B
    params.epsilon = epsilons(ll);
    prm(ll, :) = epiekf_pack_params(params, n_npi, v_bar, beta, gamma, 1)';
end
rep = @(v) repmat(v(:)', B, 1);                                   % B x numel(v), chain index first
if isscalar(R_v), Rv = repmat(R_v, B, 1); else, Rv = reshape(R_v, 1, T); end   % fixed per chain, or 1 x T for the one series
zero = zeros(B, 1, 'int32');
out = epiekf_batch_mex(1, reshape(u, 1, n_npi, T), reshape(x, 1, T), prm, rep(s_init), rep(Ps_init), rep(s_final), ...
                       rep(Ps_final), rep(Q_w), Rv, inv_monitor_len, order, ot, zero, zero);
end
