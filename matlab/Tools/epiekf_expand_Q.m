function Q = epiekf_expand_Q(Q_w, m, T)
% The forms of Q_w the reference's generic filter accepts (Tools/GenericExtendedKalmanFilter.m:63-76), brought to
% what epiekf_mex takes: m x m (fixed) or m x m x T (page k is added at filter step k).
%  * scalar               -> q*eye(m)                (Bk*Q*Bk' with Bk = eye(m), :158)
%  * m x m                -> as is
%  * m x m x D, 1 x 1 x D -> the first test (size(Q_w,1) == size(Q_w,2)) catches these too and repmat's them, so
%                            page k of the result is Q_w(:,:,mod(k-1,D)+1): pages are used cyclically
%  * vector of length T   -> q(k)*eye(m)
if size(Q_w, 1) == size(Q_w, 2)
    D = size(Q_w, 3);
    if D == 1
        if isscalar(Q_w), Q = Q_w * eye(m); else, Q = Q_w; end
    else
        idx = mod(0 : T - 1, D) + 1;
        if size(Q_w, 1) == 1
            Q = repmat(eye(m), 1, 1, T) .* reshape(Q_w(1, 1, idx), 1, 1, T);
        else
            Q = Q_w(:, :, idx);
        end
    end
    if size(Q, 1) ~= m, error('Process noise covariance noise mismatch'); end
elseif isvector(Q_w) && length(Q_w) == T
    Q = repmat(eye(m), 1, 1, T) .* reshape(Q_w, 1, 1, T);
else
    error('Process noise covariance noise mismatch');
end
end
