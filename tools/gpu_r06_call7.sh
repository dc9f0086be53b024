#!/bin/bash
# round 6, call 7: which launch of the dedup training forward depends on how it is issued (graph vs eager)
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python tools/graph_vs_eager.py --dedup --bench-like --plan train --which fwd_on 2>&1 | grep -v Warn | tail -12 ) > $O/r06_c7_a.txt; cat $O/r06_c7_a.txt
( timeout 300 python tools/graph_vs_eager.py --dedup --plan train --which fwd_on 2>&1 | grep -v Warn | tail -12 ) > $O/r06_c7_b.txt; cat $O/r06_c7_b.txt
( timeout 300 python tools/graph_vs_eager.py --dedup --bench-like --plan frozen --which fwd_off 2>&1 | grep -v Warn | tail -12 ) > $O/r06_c7_c.txt; cat $O/r06_c7_c.txt
