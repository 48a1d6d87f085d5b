# HBM traffic of the bench's kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one caption batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CZC_NORMAL_EXIT=1
for MODE in "bf16 2.6592" "split 4.6052"; do
  set -- $MODE
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_traffic -o ${1}_$C -- \
      python bench.py --precision $1 --logit-scale $2 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-alt --no-invariance \
      > gpurun_out/pmc_traffic_${1}_$C.log 2>&1
  done
done
ls gpurun_out/pmc_traffic
