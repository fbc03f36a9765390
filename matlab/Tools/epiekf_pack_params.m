function prm = epiekf_pack_params(params, n_npi, v_bar, beta, gamma, model_id)
% Packs the reference's `params` struct (+ v_bar, beta, gamma of the L2 signature) into the
% EPI_PRM_COUNT x 1 column of include/epiekf_layout.h (1-based here, 0-based there).  Only the fields the
% reference model actually reads are touched, so a missing field raises MATLAB's own
% "Reference to non-existent field" error, as it would in the reference.
prm = zeros(61, 1);
prm(1) = params.dt; prm(2) = params.beta; prm(3) = params.gamma; prm(5) = params.b;
prm(9) = params.alpha_min; prm(10) = params.alpha_max;
prm(10 + (1:n_npi)) = params.a(:);
prm(34 + (1:n_npi)) = params.u_max(:);
if model_id == 0                      % Tools/SIAlphaModelEKF.m:28-29 is the only model that reads s_min / i_min
    prm(7) = params.s_min; prm(8) = params.i_min;
end
if model_id == 1 || model_id >= 3     % 6-state models
    prm(4) = params.sigma; prm(6) = params.epsilon;
    prm(22 + (1:n_npi)) = params.u_min(:);
    % phi(kk) is a LINEAR index into  epsilon*w - gamma*s6*a  with a n x 1
    % (Tools/SIAlphaModelEKFOptControlled.m:49,107): implicit expansion turns a 1 x n `w` into an n x n matrix
    % whose linear indices 1..n are its first column, i.e. w(1) for every NPI; an n x D `w` uses w(kk,1).
    w = params.w;
    if isscalar(w)
        w_eff = repmat(w, n_npi, 1);
    elseif size(w, 1) == 1
        w_eff = repmat(w(1), n_npi, 1);
    else
        w_eff = w(1:n_npi, 1);
    end
    prm(46 + (1:n_npi)) = w_eff(:);
end
prm(59) = v_bar; prm(60) = beta; prm(61) = gamma;
end
