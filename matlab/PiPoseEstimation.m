function [R_t_2,R_t_3,Reconst,T,iter]=PiPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's PiPoseEstimation (iter = Gauss-Helmert iterations).
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('pi',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('pi',Corresp,CalM);
end
end
