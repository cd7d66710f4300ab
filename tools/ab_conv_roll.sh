#!/bin/bash
# [r5] conv_roll64_kernel vs conv_igemm_kernel: parity tests at the BASELINE layer shapes, then the within-process A/B of the 13 layers
# (option conv_roll = 0 / 1, interleaved rounds) at batch 78 and 10.   usage: gpurun -- bash tools/ab_conv_roll.sh [tag]
TAG=${1:-roll}
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_conv_fuzz_gpu.py -x -q -m gpu -k "conv" > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
python tools/bench_conv_ab.py 78 7 0,1 conv_roll > gpurun_out/${TAG}_ab78.txt 2>&1
cat gpurun_out/${TAG}_ab78.txt
python tools/bench_conv_ab.py 10 7 0,1 conv_roll > gpurun_out/${TAG}_ab10.txt 2>&1
head -5 gpurun_out/${TAG}_ab10.txt
