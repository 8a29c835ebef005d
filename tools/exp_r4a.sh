#!/bin/bash
# round-4 experiment A: occupancy sensitivity of the field-stream decoder + merged launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "C5 2" "C4 30"; do
  echo "#### $cfg"
  tools/ab_variants.sh $cfg cur split lds12 lds8 w20 w24
done
} > gpurun_out/exp_r4a.log 2>&1
tail -80 gpurun_out/exp_r4a.log
