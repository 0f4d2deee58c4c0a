cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe > gpurun_out/check/train_prof.log 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
python3 tools/kstats.py $f | head -45 | cut -c1-170
tail -1 gpurun_out/check/train_prof.log | cut -c1-200
