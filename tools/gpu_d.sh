#!/bin/bash
export MPF_UFB_PROFILE=1
for o in "--opt split_cands=64" "--opt split_cands=100000"; do
python bench.py --steps 3 --warmup 1 --no-cpu --bootstrap-replicates 0 $o 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); u=d['ufboot_online']; print('$o', u['seconds'], u['roofline']['kernel_ms_total'], u['events'])"
grep ufboot /tmp/err.txt | tail -2
done
