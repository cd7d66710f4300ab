#!/bin/bash
# pkfma_probe alone, then beside two processes running bf16 training steps.  usage: bash tools/hwprobe/pkfma_probe.sh OUTDIR [launches] [iters]
out=gpurun_out/${1:-pkfma}; mkdir -p $out
n=${2:-100}; it=${3:-8}
echo "== alone" > $out/pkfma.txt
tools/hwprobe/pkfma_probe $n $it >> $out/pkfma.txt 2>&1
python tools/debug_victim.py aggressor 100 > $out/aggr0.txt 2>&1 &
A0=$!
python tools/debug_victim.py aggressor 100 > $out/aggr1.txt 2>&1 &
A1=$!
sleep 25
echo "== beside two processes running bf16 training steps" >> $out/pkfma.txt
tools/hwprobe/pkfma_probe $n $it >> $out/pkfma.txt 2>&1
tools/hwprobe/pkfma_probe $n $it >> $out/pkfma.txt 2>&1
wait $A0 $A1
grep -h aggressor $out/aggr0.txt $out/aggr1.txt >> $out/pkfma.txt
cat $out/pkfma.txt
