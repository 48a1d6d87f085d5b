cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
export CZC_NORMAL_EXIT=1
COMMON="--no-cpu-baseline --no-alt --no-invariance"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_1s -o p -- python bench.py --streams 1 --steps 2 --warmup 1 $COMMON > $O/bf16_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_2s -o p -- python bench.py --steps 2 --warmup 1 --no-profile $COMMON > $O/bf16_2s.log 2>&1
for d in bf16_1s bf16_2s; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
head -c 600 $O/bench_driver_cmd.json; echo; head -8 $O/bf16_1s_kernel_stats.csv | cut -c1-200; tail -1 $O/bf16_1s.log | cut -c1-200; tail -1 $O/bf16_2s.log | cut -c1-200
