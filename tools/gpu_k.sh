#!/bin/bash
for sk in 0 1 2 3 4 5 6 7 8 9 10 "0,1,2,3" "4,5,6" "0,1,2,3,4,5,6,7"; do
  out=$(MPF_STATEFUL_SKIP=$sk MPF_FUZZ_OFFSET=300 timeout 120 python -m pytest tests/test_gpu_stateful.py -x -q -k "oracle[57]" 2>&1 | tail -1)
  echo "skip $sk: $out" | cut -c1-100
done
