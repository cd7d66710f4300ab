for v in base swg2 swg4 base swg2 swg4; do
  if [ $v = base ]; then unset IM2IM_LIB; else export IM2IM_LIB=$PWD/im2im_uq_amd/lib/libim2im_uq_$v.so; fi
  echo "== $v $(bash tools/prof_quick.sh 2>&1 | grep 'wgrad_mfma\|l2s_mfma' | tr '\n' ' ' | cut -c1-200)"
done
