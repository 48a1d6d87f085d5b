# Round-6 evidence (run on the GPU box from the repo root): the GPU test suite, the driver's bench command, the other BASELINE
# shapes, the refine validation, rocprofv3 kernel stats of the bench in its modes, and the MFMA-op / HBM-traffic counters in
# their own passes.  Summaries land in gpurun_out/r06e/ (the ones to keep are copied into profiles/ as r06_*).
# usage: r05_evidence.sh [quick|kernels]   (quick: no test suite, no driver-length bench, no PMC passes; kernels: what a timing-only
# kernel change moves -- test suite, driver bench, default / single-image bench, kernel stats, MFMA counters -- and not the other
# BASELINE shapes, the refine validations or the HBM-traffic passes)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e
mkdir -p $O
if [ "$1" != "quick" ]; then
  python -m pytest tests -m gpu -q > $O/gpu_tests_summary.txt 2>&1
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
fi
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config 1 --steps 3 --no-alt > $O/bench_cfg1.json 2> $O/bench_cfg1.err
if [ "$1" != "kernels" ]; then
python bench.py --config 3 --total-images 256 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --gpus 8 --share-gpu --total-images 16 --steps 1 --warmup 0 --iters 1 --no-cpu-baseline --no-invariance --no-profile --no-alt > $O/bench_8ranks_shared_gpu.json 2> $O/bench_8ranks_shared_gpu.err
python bench.py --config 4 --total-images 64 --control both --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python tools/refine_validate.py 128 10 24 2000 > $O/refine_validate_128x10.jsonl 2> $O/refine_validate.err
if [ "$1" != "quick" ]; then GEN_SWEEPS=10 python tools/refine_validate.py 256 10 24 2000 > $O/refine_validate_256x10_fullcaptions.jsonl 2>> $O/refine_validate.err; fi
# (the further weight draws -- profiles/r06_refine_validate_draws_128x10.jsonl / _divergences_ -- were run on their own:
#  for d in "21 22 1" "31 32 1" "41 42 1" "51 52 1" "11 12 12" "61 62 6"; do set -- $d; BSEED=$1 CSEED=$2 OUTLIER=$3 EMB_SEED=$((3000+$1)) GEN_SWEEPS=10 GATES=400 python tools/refine_validate.py 128 10; done)
fi
export CZC_NORMAL_EXIT=1
COMMON="--no-cpu-baseline --no-alt --no-invariance"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_1s -o p -- python bench.py --streams 1 --steps 2 --warmup 1 $COMMON > $O/bf16_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_2s -o p -- python bench.py --steps 2 --warmup 1 $COMMON > $O/bf16_2s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/refine_1s -o p -- python bench.py --precision refine --logit-scale 4.6052 --streams 1 --steps 2 --warmup 1 $COMMON > $O/refine_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o p -- python bench.py --images 1 --steps 3 --warmup 1 --no-profile $COMMON > $O/b1.log 2>&1
for d in bf16_1s bf16_2s refine_1s b1; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
if [ "$1" != "quick" ]; then
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python bench.py --streams 1 --steps 1 --warmup 0 --no-profile $COMMON > $O/pmc_mfma.log 2>&1
  python tools/probes/pmc_mfma_summary.py $O/pmc_mfma > $O/pmc_mfma_summary.json 2> $O/pmc_mfma_summary.err
  if [ "$1" != "kernels" ]; then
    bash tools/probes/pmc_bench_traffic.sh > $O/pmc_traffic.log 2>&1
    cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary.json
  fi
fi
find $O gpurun_out/pmc_traffic -name "*kernel_trace.csv" -delete; find $O gpurun_out/pmc_traffic -name "*counter_collection.csv" -size +4M -delete
find $O gpurun_out/pmc_traffic -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
head -c 600 $O/bench_driver_cmd.json; echo; head -8 $O/bf16_1s_kernel_stats.csv | cut -c1-200; tail -3 $O/gpu_tests_summary.txt
