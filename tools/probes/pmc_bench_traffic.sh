# HBM traffic of the bench's kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one caption batch,
# summarised by pmc_traffic_summary.py into gpurun_out/pmc_traffic/summary.json (copy into profiles/).
# usage: pmc_bench_traffic.sh ["bf16 2.6592" "split 4.6052" ...]   (default: both)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CZC_NORMAL_EXIT=1
[ $# -eq 0 ] && set -- "bf16 2.6592" "refine 4.6052"
NAMES=""
for MODE in "$@"; do
  P=${MODE% *}; S=${MODE#* }
  NAMES="$NAMES $P"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_traffic -o ${P}_$C -- \
      python bench.py --streams 1 --precision $P --logit-scale $S --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-alt --no-invariance \
      > gpurun_out/pmc_traffic_${P}_$C.log 2>&1
  done
done
python tools/probes/pmc_traffic_summary.py gpurun_out/pmc_traffic $NAMES > gpurun_out/pmc_traffic/summary.json
find gpurun_out/pmc_traffic -name "*.csv" -size +8M -delete   # the per-dispatch tables stay on the box
cat gpurun_out/pmc_traffic/summary.json | head -50
