#!/bin/bash
for sk in none 0 1 2 3 4 5 6 7 "0,1,2,3,4" "6,7"; do
  out=$(MPF_STATEFUL_SKIP=$sk MPF_FUZZ_OFFSET=700 timeout 120 python -m pytest tests/test_gpu_stateful.py -x -q -k "sequences_match_oracle and 25" 2>&1 | tail -1)
  echo "skip $sk: $out" | cut -c1-100
done
