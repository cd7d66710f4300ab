#!/bin/bash
# the in-place bisect of smallconv_wgrad_vec_kernel's multiply-add forms (csrc/smallconv.hip, template parameter FM): two aggressor
# processes run bf16 training steps, the victim launches the first conv's weight gradient 100 times per form on fixed inputs and counts
# the launches whose bits differ from the first.  Needs a library with the forms: IM2IM_BUILD_EXPERIMENTAL=1 python -m im2im_uq_amd.build
# (the default library holds form 1 only; FORMS=default runs with it).  usage: bash tools/debug_victim2.sh OUTDIR [SECONDS]; FORMS="default 0 1 ..." selects.
out=gpurun_out/${1:-victim2}; mkdir -p $out
secs=${2:-200}
python tools/debug_victim.py aggressor $secs > $out/aggr0.txt 2>&1 &
A0=$!
python tools/debug_victim.py aggressor $secs > $out/aggr1.txt 2>&1 &
A1=$!
sleep 25
{
  for d in ${FORMS:-default 0 1 2 3 4 5}; do
    if [ "$d" = default ]; then VICTIM_ONLY="first-conv" python tools/debug_victim.py victim 100 2>&1 | sed "s/^\[victim/[victim form=default/"
    else IM2IM_SWG_DBG=$d VICTIM_ONLY="first-conv" python tools/debug_victim.py victim 100 2>&1 | sed "s/^\[victim/[victim form=$d/"; fi
  done
} 2>&1 | grep -E "^\[victim|Error|error" > $out/victim.txt
wait $A0 $A1
grep -h aggressor $out/aggr0.txt $out/aggr1.txt >> $out/victim.txt
cat $out/victim.txt
