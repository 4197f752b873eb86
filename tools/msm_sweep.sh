#!/bin/bash
# development aid: sweep the MSM planner knobs on one GPU.  usage: tools/msm_sweep.sh <log_n>
n=${1:-20}
for off in 3 4 5; do for K in 0 16 32 64 128 256; do
  echo "C_OFF=$off K=$K: $(LH_MSM_C_OFF=$off LH_MSM_K=$K python tools/big_run.py range $n | tail -1)"
done; done
for cmax in 18 20; do for K in 64 128 256; do
  echo "C_MAX=$cmax K=$K: $(LH_MSM_C_MAX=$cmax LH_MSM_K=$K python tools/big_run.py range $n | tail -1)"
done; done
