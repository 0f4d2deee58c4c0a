python -m pytest tests/test_f16s_gpu.py tests/test_split3_gpu.py tests/test_gemm_gpu.py -q -m gpu --timeout 900 -x 2>&1 | tail -8
python -m pytest tests/test_model_gpu.py tests/test_xattn_gpu.py tests/test_sampler_gpu.py -q -m gpu --timeout 900 2>&1 | tail -8
for fs in 1 0; do DIMSUM_FORWARD_SCOPE=$fs python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-box-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scope', $fs, round(d['ms_per_step'],2), d['roofline']['avg_launch_ms'])"; done
