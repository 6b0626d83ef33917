"""Experiment: persistent NT kernels, grid size x start stagger (env knobs read once per process -> one subprocess per point)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
from tools.gemm_wide_check import case, NT
from sdumc_amd import ops
import os
tag = "G=%%s st=%%s" %% (os.environ.get("SDUMC_WIDE_G"), os.environ.get("SDUMC_WIDE_STAGGER"))
T = (0, 15, 16, 17, 18)
case(tag + " keys audio", NT, 48000, 256, 256, T, bias=True, act=ops.ACT_TANH, drop=True, row_mod=24000)
case(tag + " frame audio", NT, 24000, 256, 1024, T, bias=True)
case(tag + " keys video", NT, 28800, 256, 256, T, bias=True, act=ops.ACT_TANH, drop=True, row_mod=14400)
''' % ROOT
for G in (256, 512, 768):
    for st in (0, 2, 4, 8):
        env = dict(os.environ, SDUMC_WIDE_G=str(G), SDUMC_WIDE_STAGGER=str(st))
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)
