function [R_t,Reconst,iter,repr_err]=BundleAdjustment(CalM,R_t_0,Corresp,Reconst0)
% MI355X drop-in for the reference's BundleAdjustment: M = 2 .. 6 views (CalM 3Mx3, R_t_0 3Mx4, Corresp 2MxN, NaN = not seen),
% Reconst0 3xN optional.
if nargin<4
    [R_t,Reconst,iter,repr_err]=tftfund_mex('bundle_adjustment',Corresp,CalM,R_t_0);
else
    [R_t,Reconst,iter,repr_err]=tftfund_mex('bundle_adjustment',Corresp,CalM,R_t_0,Reconst0);
end
end
