#!/bin/bash
# round 6, call 23: cost of one inter-workgroup hand-off (the N-split cluster's primitive, verdict item 6)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd tools/_scratch/cluster
( timeout 120 ./cluster_probe 2>&1 ) > $O/r06_c23_cluster_probe.txt; cat $O/r06_c23_cluster_probe.txt
