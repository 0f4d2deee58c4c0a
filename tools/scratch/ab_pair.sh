for i in 1 2; do for v in 1 0; do
DIMSUM_PAIR_IMAGES=$v timeout 300 python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-box-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$v', d['ms_per_step'])"
done; done
