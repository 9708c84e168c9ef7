function [R_t_2,R_t_3,Reconst,T,iter]=NordbergTFTPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's NordbergTFTPoseEstimation (iter = Gauss-Helmert iterations).
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('nordberg_tft',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('nordberg_tft',Corresp,CalM);
end
end
