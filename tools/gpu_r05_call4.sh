#!/bin/bash
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for v in 0 1 2; do
  echo "== VAR $v"
  ( LECO_XGEMM_VAR=$v timeout 200 python -m pytest tests/test_kernels.py -q -m gpu -k "xgemm" 2>&1 | grep -E "passed|failed|FAILED" | tail -8 )
  ( LECO_XGEMM_VAR=$v timeout 200 python tools/bench_xgemm.py 2>&1 | grep -vE "Warn|warn|amdgpu.ids" | tail -9 ) | tee $O/${RN}_bench_xgemm_var$v.txt
done
