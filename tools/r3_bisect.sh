#!/bin/bash
# which knob makes the 2448x2048 detector test fail under the split fit
for k in "X=0" "CK_SEQ_ALT=16" "CK_TAIL_WPS=128" "CK_SEQ_ALT=16 CK_TAIL_WPS=128"; do
  for rep in 1 2; do
    env CK_FIT_FLAT=2 $k timeout -k 10 300 python -m pytest tests/test_gpu_detect.py -x -q -m gpu -k "2448" 2>&1 | tail -n 1 | sed "s/^/$k rep $rep: /"
  done
done
