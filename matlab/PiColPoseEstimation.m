function [R_t_2,R_t_3,Reconst,T,iter]=PiColPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's PiColPoseEstimation (collinear camera centres; iter = Gauss-Helmert iterations).
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('picol',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('picol',Corresp,CalM);
end
end
