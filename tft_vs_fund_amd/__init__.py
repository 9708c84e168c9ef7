"""tft_vs_fund_amd: MI355X-native batched trifocal-tensor / fundamental-matrix pose estimation."""
