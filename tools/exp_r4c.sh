#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for cfg in "C5 2" "C4 30" "C2 30"; do
  echo "#### $cfg"
  tools/ab_variants.sh $cfg "$@"
done
} 2>&1 | grep -v "^generic\|^checker\|^encode\|^streams" > gpurun_out/exp_r4c.log
cat gpurun_out/exp_r4c.log
