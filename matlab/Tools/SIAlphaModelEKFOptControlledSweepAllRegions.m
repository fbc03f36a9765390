function out = SIAlphaModelEKFOptControlledSweepAllRegions(u, x, params, s_init, Ps_init, s_final, Ps_final, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order, epsilons, scoring, devices)
% The Pareto sweep of ALL regions in ONE call: what Tools/TrainPredictPrescribeNPI.m does with the two nested loops
%     for k = 1 : NumGeoLocations                                   (:93)
%         for ll = 1 : num_pareto_front_points                      (:421)
%             params.epsilon = human_npi_cost_factor(ll);
%             [~, opt_control_input_smooth, ...] = SIAlphaModelEKFOptControlled(control_input, observations, params, ...);   (:460)
%             [s, i, alpha] = SIalpha_Controlled(opt_control_input_smooth(:, NumNPIdays + 1 : end), ...);                  (:481)
%             [J0_opt_control(ll), J1_opt_control(ll)] = NPICost(...);                                                       (:493)
%         end
%         is_on_pareto_front(ii) = ...;  [~, I_opt] = min(...);                                                             (:624-633)
%     end
% Arguments: cell arrays / stacked arrays with one entry per region r = 1 .. R --
%   u{r}         n_npi x T   control_input = [IP, nan(NumNPI, num_forecast_days)]
%   x{r}         1 x T       observations  = [NewCasesSmoothedNormalized, nan(1, num_forecast_days)]
%   params(r)    struct array (params.epsilon is ignored), s_init{r} 6 x 1, Ps_init{r}, Ps_final{r}, Q_w{r} 6 x 6, s_final{r} 6 x 1,
%   R_v{r}       1 x T
%   epsilons     P x 1 = human_npi_cost_factor
%   scoring      struct with fields (one row per region): s_end, i_end, alpha_end (s/i/alpha_historic(end)), a (R x n_npi),
%                b, alpha_min, alpha_max, gamma, beta, u_max (R x n_npi = NPI_MAXES), weights (R x n_npi = npi_weights),
%                J0_prefix (= sum(s_historic .* i_historic .* alpha_historic)), J1_prefix (= sum of weights .* IP, column-major),
%                t_hist (= NumNPIdays)
%   devices      zero-based GPU ids, [] = device 0; whole regions are cut into blocks over them
% Returns out.J0, out.J1, out.on_front (P x R), out.I_opt (R x 1, one-based), out.u_opt (R x n_npi x T: the optimum's
% opt_control_input_smooth), out.S_opt (R x 6 x T: its S_SMOOTH).  The per-chain filter outputs never leave the device.
R = numel(u);
n_npi = size(u{1}, 1); T = size(x{1}, 2);
U = zeros(R, n_npi, T); X = zeros(R, T); RV = zeros(R, T); prm = zeros(R, 61);
S0 = zeros(R, 6); P0 = zeros(R, 36); SF = zeros(R, 6); PF = zeros(R, 36); Q = zeros(R, 36);
ot = -1;
for r = 1 : R
    U(r, :, :) = reshape(u{r}, 1, n_npi, T); X(r, :) = x{r}; RV(r, :) = R_v{r};
    prm(r, :) = epiekf_pack_params(params(r), n_npi, v_bar, beta, gamma, 1)';
    S0(r, :) = s_init{r}(:)'; P0(r, :) = Ps_init{r}(:)'; SF(r, :) = s_final{r}(:)'; PF(r, :) = Ps_final{r}(:)';
    Qr = Q_w{r}; if ~isequal(size(Qr), [6 6]), Qr = epiekf_expand_Q(Qr, 6, T); end
    if ndims(Qr) == 3, error('epiekf:unsupported', 'the sweep takes a fixed Q_w'); end
    Q(r, :) = Qr(:)';
    if isequal(params(r).obs_type, 'NEWCASES'), o = 0; elseif isequal(params(r).obs_type, 'TOTALCASES'), o = 1; else, error('unknown observation type'); end
    if ot >= 0 && o ~= ot, error('epiekf:unsupported', 'one observation type per call'); end
    ot = o;
end
sp = zeros(R, 48);                                   % EPI_SIM_* columns of include/epiekf.h, zero-based offsets + 1
sp(:, 1) = scoring.s_end(:); sp(:, 2) = scoring.i_end(:); sp(:, 3) = scoring.alpha_end(:);
sp(:, 4) = scoring.alpha_min(:); sp(:, 5) = scoring.alpha_max(:); sp(:, 6) = scoring.gamma(:);
sp(:, 7) = scoring.b(:); sp(:, 8) = scoring.beta(:); sp(:, 12) = 1;        % noise-free scoring, dt = 1
sp(:, 12 + (1 : n_npi)) = scoring.a; sp(:, 24 + (1 : n_npi)) = scoring.u_max; sp(:, 36 + (1 : n_npi)) = scoring.weights;
out = epiekf_pipeline_mex('prescribe', X, U, RV, prm, S0, P0, SF, PF, Q, epsilons(:), sp, scoring.J0_prefix(:), scoring.J1_prefix(:), ...
                          scoring.t_hist, inv_monitor_len, order, ot, devices(:));
end
