#!/bin/bash
# ablation builds of conv_roll.hip (tools/ab_build_one.sh <name> conv_roll.hip -DIM2IM_CROLL_ABL=..) on one layer, interleaved twice
shape=${1:-78,320,320,64,64}; kind=${2:-dgrad}; shift; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "new" ]; then python tools/conv_one.py $shape $kind 20 | sed "s/^/new        /"
    elif [ "$v" = "old" ]; then IM2IM_CONV_ROLL=0 python tools/conv_one.py $shape $kind 20 | sed "s/^/old        /"
    elif [ "$v" = "one" ]; then IM2IM_CONV_ROLL=2 python tools/conv_one.py $shape $kind 20 | sed "s/^/1wg\/cu     /"
    else IM2IM_LIB=$PWD/im2im_uq_amd/lib/libim2im_uq_$v.so python tools/conv_one.py $shape $kind 20 | sed "s/^/$v  /"; fi
  done
done
