#!/bin/bash
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for mode in fused fused_release fused_keep; do
  ( timeout 280 python tools/graph_twice.py $mode sd15 2>&1 | grep -vE "Warn|warn|amdgpu.ids" | tail -12 ) > $O/${RN}_graph_twice_${mode}_sd15.txt
  echo "== graph_twice $mode sd15"; tail -8 $O/${RN}_graph_twice_${mode}_sd15.txt
done
