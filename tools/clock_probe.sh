#!/bin/bash
# Samples GPU clock / power (rocm-smi) while sustained kernels run: MFMA-only chain, GEMM skeleton on zero-filled and on
# random operands, and the library's 4096^3 fp32 GEMM.
poll() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' '; echo; sleep 0.45; done; }
echo "== MFMA-only chain (4 waves/SIMD), sustained ~4 s"
./tools/micro/mfma_chain sustain & P=$!; sleep 1.0; poll 5; wait $P
echo "== GEMM skeleton: zeros 4 s, random 4 s, glds random 4 s"
./tools/micro/mfma_gemm_like2 sustain & P=$!; sleep 1.0; poll 24; wait $P
