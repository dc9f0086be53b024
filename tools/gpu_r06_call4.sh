#!/bin/bash
# round 6, call 4: where does the dedup step turn non-finite (full size); micro-batch concurrency experiment
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 400 python tools/nan_probe.py --dedup --k 2 --steps 2 2>&1 | grep -v Warn | tail -30 ) > $O/r06_c4_nan_probe.txt
( timeout 400 python tools/exp_microbatch.py --k 20 2>&1 | grep -v Warn | tail -8 ) > $O/r06_c4_microbatch.txt
cat $O/r06_c4_nan_probe.txt; cat $O/r06_c4_microbatch.txt
