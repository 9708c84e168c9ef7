function [R_t_2,R_t_3,Reconst,T,iter]=LinearTFTPoseEstimation(Corresp,CalM)
% MI355X drop-in for the reference's LinearTFTPoseEstimation: same inputs
% (Corresp 6xN, CalM 9x3) and outputs, computed by libtftfund.so through the
% MEX gateway.  Corresp may also be 6xNxB for a batch of B triplets.
if nargout>=3
    [R_t_2,R_t_3,Reconst,T,iter]=tftfund_mex('linear_tft',Corresp,CalM);
else
    [R_t_2,R_t_3]=tftfund_mex('linear_tft',Corresp,CalM);
end
end
