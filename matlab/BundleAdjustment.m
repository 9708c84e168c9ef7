function [R_t,Reconst,iter,repr_err]=BundleAdjustment(CalM,R_t_0,Corresp,Reconst0)
% MI355X drop-in for the reference's BundleAdjustment: three views (CalM 9x3, R_t_0 9x4 with R_t_0(1:3,:)=eye(3,4),
% Corresp 6xN without NaN entries), Reconst0 3xN optional.
if nargin<4
    [R_t,Reconst,iter,repr_err]=tftfund_mex('bundle_adjustment',Corresp,CalM,R_t_0);
else
    [R_t,Reconst,iter,repr_err]=tftfund_mex('bundle_adjustment',Corresp,CalM,R_t_0,Reconst0);
end
end
