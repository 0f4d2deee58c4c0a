# usage: ab_run.sh <mode> : prints ms_per_step of bench.py --mode <mode> twice
for i in 1 2; do timeout 300 python bench.py --mode $1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-box-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; done
