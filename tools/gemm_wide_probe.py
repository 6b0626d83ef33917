"""Where does the K = 256 NT shape lose time?  Variants of one launch (M 48000, N 256), tile 13 (128x128) and 14 (64x128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_wide_check import case, NT
from sdumc_amd import ops
T = (0, 13, 14)
case("plain K256", NT, 48000, 256, 256, T)
case("plain K256 rowmod", NT, 48000, 256, 256, T, row_mod=24000)
case("+bias", NT, 48000, 256, 256, T, bias=True)
case("+bias+tanh", NT, 48000, 256, 256, T, bias=True, act=ops.ACT_TANH)
case("+mask", NT, 48000, 256, 256, T, drop=True)
case("+mask+bias+tanh+rowmod", NT, 48000, 256, 256, T, bias=True, act=ops.ACT_TANH, drop=True, row_mod=24000)
case("plain K512", NT, 48000, 256, 512, T)
case("plain K1024", NT, 48000, 256, 1024, T)
case("plain K2048", NT, 48000, 256, 2048, T)
case("plain M12000 K1024", NT, 12000, 256, 1024, T)
case("plain M96000 K256", NT, 96000, 256, 256, T)
case("plain N512 K256", NT, 48000, 512, 256, T)
case("plain N1024 K256", NT, 48000, 1024, 256, T)
