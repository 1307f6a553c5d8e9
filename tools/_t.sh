timeout 900 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -3
for i in 1 2; do
timeout 300 python tools/climb_timing.py --workload C3 --start random --opt timing=1 < /dev/null 2>&1 | grep -v amdgpu | tail -1
done
timeout 300 python tools/climb_timing.py --workload C2 --start random --check < /dev/null 2>&1 | grep -v amdgpu | tail -2
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu < /dev/null 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['views'], d['host_ms_per_step'], d.get('bootstrap_wall_clock')['refinement_s'], d.get('ufboot_online')['seconds'])"
