function [R_t_2,R_t_3,Reconst,T,iter]=OptimFPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's OptimFPoseEstimation (needs N>=8; iter = it1+it2).
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('optim_f',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('optim_f',Corresp,CalM);
end
end
