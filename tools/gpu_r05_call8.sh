#!/bin/bash
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for mode in fused_mm fused_conv fused_bwd; do
  ( timeout 280 python -X faulthandler tools/graph_twice.py $mode sd15 2>&1 | grep -vE "Warn|warn|amdgpu.ids" | grep -E "ok|done|DONE|Fatal|Segmentation|File" | tail -12 ) > $O/${RN}_graph_twice_${mode}_sd15.txt
  echo "== graph_twice $mode sd15"; tail -10 $O/${RN}_graph_twice_${mode}_sd15.txt
done
