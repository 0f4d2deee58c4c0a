cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "l2 --B 256 --L 256 --hd 64" "xl --B 64 --L 1024 --hd 72"; do
  set -- $shape; t=$1; shift
  rm -rf gpurun_out/xa_$t
  for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    rocprofv3 --pmc $set --kernel-trace -d gpurun_out/xa_$t/p$RANDOM --output-format csv -- python3 tools/bench_xattn.py --iters 3 --f16 "$@" > /dev/null 2>&1
  done
  python3 tools/pmc_csv.py gpurun_out/xa_$t xattn | grep -v "^#"
  find gpurun_out/xa_$t -type f -size +1M -delete
done
