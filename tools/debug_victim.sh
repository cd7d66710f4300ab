#!/bin/bash
# two aggressor processes (bf16 training steps in a loop) + a series of victim runs; output: gpurun_out/$1/victim.txt
out=gpurun_out/${1:-victim}; mkdir -p $out
secs=${2:-150}
python tools/debug_victim.py aggressor $secs > $out/aggr0.txt 2>&1 &
A0=$!
python tools/debug_victim.py aggressor $secs > $out/aggr1.txt 2>&1 &
A1=$!
sleep 25      # the aggressors import torch and reach their loop
{
  python tools/debug_victim.py victim 150
  IM2IM_SMALLCONV_VALU=128 VICTIM_ONLY="first-conv" python tools/debug_victim.py victim 150
  IM2IM_SMALLCONV_VALU=8 VICTIM_ONLY="first-conv" python tools/debug_victim.py victim 150
} 2>&1 | grep -E "^\[victim|Error|error" > $out/victim.txt
wait $A0 $A1
grep -h aggressor $out/aggr0.txt $out/aggr1.txt >> $out/victim.txt
# the same victims with the GPU to themselves
python tools/debug_victim.py victim 150 2>&1 | grep -E "^\[victim" | sed 's/^/[alone] /' >> $out/victim.txt
cat $out/victim.txt
