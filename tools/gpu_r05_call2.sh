#!/bin/bash
# Round-5 GPU call 2: the benchmark with the asynchronous step (pinned staging, fused glue launches), the launch-to-launch
# floor of a graph node, and the two-models-in-one-process capture crash at full size.
TAG=${1:-lease2}
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/gpu_r05_diag.sh $TAG short
( timeout 120 python tools/graph_node_floor.py 2000 2>&1 | tail -4 ) > $O/${RN}_graph_node_floor.txt
cat $O/${RN}_graph_node_floor.txt
for mode in plain shared; do
  ( timeout 200 python tools/graph_twice.py $mode sd15 2>&1 | grep -vE "Warn|warn" | tail -6 ) > $O/${RN}_graph_twice_${mode}_sd15.txt
  echo "== graph_twice $mode sd15 (rc $?)"; tail -4 $O/${RN}_graph_twice_${mode}_sd15.txt
done
