# Round-3 rocprofv3 evidence (run on the GPU box from the repo root): kernel stats of the bench in its modes, MFMA-op and HBM
# traffic counters in their own passes.  Summaries land in gpurun_out/r03p/ (copy the ones to keep into profiles/).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CZC_NORMAL_EXIT=1
O=gpurun_out/r03p
mkdir -p $O
COMMON="--no-cpu-baseline --no-alt --no-invariance"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_1s -o p -- python bench.py --streams 1 --steps 2 --warmup 1 $COMMON > $O/bf16_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_2s -o p -- python bench.py --steps 2 --warmup 1 --no-profile $COMMON > $O/bf16_2s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/refine_1s -o p -- python bench.py --precision refine --logit-scale 4.6052 --streams 1 --steps 2 --warmup 1 $COMMON > $O/refine_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -o p -- python bench.py --images 1 --steps 3 --warmup 1 --no-profile $COMMON > $O/b1.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE" | head -40 > $O/counters_available.txt
# MFMA ops of one caption batch (counter-derived FLOPs vs the engine's own count), own pass, no other tracing domain
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- python bench.py --streams 1 --steps 1 --warmup 0 --no-profile $COMMON > $O/pmc_mfma.log 2>&1
for d in bf16_1s bf16_2s refine_1s b1; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
python tools/probes/pmc_mfma_summary.py $O/pmc_mfma > $O/pmc_mfma_summary.json 2> $O/pmc_mfma_summary.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*.db" -delete
head -12 $O/bf16_1s_kernel_stats.csv | cut -c1-220; cat $O/pmc_mfma_summary.json | head -40; cat $O/counters_available.txt | head; tail -2 $O/*.log | cut -c1-300
