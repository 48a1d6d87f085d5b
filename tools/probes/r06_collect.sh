# Copy the summaries of tools/probes/r06_evidence.sh (gpurun_out/r06e/) into profiles/ under their round-5 names, then regenerate
# the documentation blocks that quote them (tools/refresh_docs.py).  Run in the build container after the evidence run came back.
# usage: r06_collect.sh [kernels]   (after `r06_evidence.sh kernels`: only what that run produced)
set -e
S=gpurun_out/r06e
D=profiles
cp $S/bench_driver_cmd.json $D/r06_bench_driver_cmd.json
cp $S/bench_default.json $D/r06_bench_default.json
cp $S/bench_cfg1.json $D/r06_bench_cfg1.json
if [ "$1" != "kernels" ]; then
for c in cfg3 cfg4 8ranks_shared_gpu; do cp $S/bench_$c.json $D/r06_bench_$c.json; done
cp $S/refine_validate_128x10.jsonl $D/r06_refine_validate_128x10.jsonl
[ -s $S/refine_validate_256x10_fullcaptions.jsonl ] && cp $S/refine_validate_256x10_fullcaptions.jsonl $D/r06_refine_validate_256x10_fullcaptions.jsonl
cp $S/pmc_traffic_summary.json $D/r06_bench_gemm_traffic.json
fi
cp $S/bf16_1s_kernel_stats.csv $D/r06_bench_bf16_kernel_stats.csv
cp $S/bf16_2s_kernel_stats.csv $D/r06_bench_bf16_2streams_kernel_stats.csv
cp $S/refine_1s_kernel_stats.csv $D/r06_bench_refine_kernel_stats.csv
cp $S/b1_kernel_stats.csv $D/r06_bench_b1_kernel_stats.csv
cp $S/pmc_mfma_summary.json $D/r06_bench_mfma_counters.json
grep -v "amdgpu.ids" $S/gpu_tests_summary.txt > $D/r06_gpu_tests_summary.txt
python tools/refresh_docs.py
python tools/refresh_docs.py --check
