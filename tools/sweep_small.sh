#!/bin/bash
# development aid: knob sweep at small / medium sizes.  usage: tools/sweep_small.sh <log_n>
n=${1:-16}
run() { echo "$1: $(env $1 python tools/big_run.py range $n | tail -1 | cut -c1-42)"; }
run "LH_MSM_K=0"
for k in 4 8 16 32; do run "LH_MSM_K=$k"; done
for o in 2 3 4 5; do run "LH_MSM_C_OFF=$o"; done
for v in 16384 65536 262144; do run "LH_SC_LDS_MAX_ITEMS=$v"; done
for v in 32768 131072 524288; do run "LH_SC_TP_MAX_ITEMS=$v"; done
for s in 2 4 8; do run "LH_MSM_SEG=$s"; done
