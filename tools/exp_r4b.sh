#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for cfg in "C5 2" "C4 30"; do
  echo "#### $cfg"
  tools/ab_variants.sh $cfg cur norounds noreads noringst norr
done
} 2>&1 | grep -v "^generic\|^checker\|^encode" > gpurun_out/exp_r4b.log
cat gpurun_out/exp_r4b.log
