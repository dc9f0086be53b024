#!/bin/bash
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  ( env "$@" timeout 400 python -X faulthandler tests/test_fullsize.py sd15_512_bs2_rank4 sd15_512_bs2_rank4_enhance_g3 2>&1 | grep -E "^PASS|Segmentation|Fatal|Error|error" | head -6 ) > $O/${RN}_capture_twice_$name.txt
  echo "== $name: $(tr '\n' ' ' < $O/${RN}_capture_twice_$name.txt)"
}
run mode_global LECO_CAPTURE_MODE=global
run mode_relaxed LECO_CAPTURE_MODE=relaxed
run stream_lib LECO_CAPTURE_STREAM=lib
