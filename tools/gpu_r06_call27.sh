#!/bin/bash
# KS2 v2 A/B: barrier at the step start, the next set's reads spread over the step; BP = 0 (product lib) and BP = 4 (side lib)
mkdir -p gpurun_out
{
echo "# LECO_CONV_KS2=0"; LECO_CONV_KS2=0 python tools/ablate_conv.py --case
echo "# LECO_CONV_KS2=1 BP=0"; LECO_CONV_KS2=1 python tools/ablate_conv.py --case
echo "# LECO_CONV_KS2=1 BP=4"; LECO_HIP_LIB=tools/_ablate/libleco_ks2bp4.so LECO_CONV_KS2=1 python tools/ablate_conv.py --case
} > gpurun_out/r06_c27_conv.txt 2>&1
LECO_CONV_KS2=1 python -m pytest tests/test_kernels.py -q -m gpu -k conv > gpurun_out/r06_c27_tests.log 2>&1
tail -3 gpurun_out/r06_c27_tests.log; cat gpurun_out/r06_c27_conv.txt
