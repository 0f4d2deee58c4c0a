# tools/profile_train.sh (GPU box): rocprofv3 kernel-trace + stats of the block_fwdbwd and train_step legs on one stream -> gpurun_out/now_{block,train}_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DIMSUM_BRANCH_STREAMS=0
for mode in block train; do
rm -rf /tmp/prof_$mode
extra=""; [ $mode = train ] && extra="--batch 64"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python3 bench.py --mode $mode $extra --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe > gpurun_out/prof_$mode.log 2>&1
f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/now_${mode}_kernel_stats.csv <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    w.writerow([c[:200] for c in r])
P
done
