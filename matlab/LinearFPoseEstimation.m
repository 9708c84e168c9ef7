function [R_t_2,R_t_3,Reconst,T,iter]=LinearFPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's LinearFPoseEstimation (needs N>=8).
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('linear_f',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('linear_f',Corresp,CalM);
end
end
