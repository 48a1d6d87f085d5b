set -x
mkdir -p gpurun_out/r03b
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:256,7:512,7:1024,7:2048,7:768,7:1536,7:3072,7:3584,7:2560 6 > gpurun_out/r03b/ab_fc2_ablate.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 3,3:0:1:64,3:0:1:8,7,7:0:1:64,7:0:1:8,7:0:1:32 6 > gpurun_out/r03b/ab_fc2_pad.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 2048 512 1 0 6,6:0:0:64,6:0:0:8 6 > gpurun_out/r03b/ab_fc1_pad.log 2>&1
CZC_PROBE_OUT=gpurun_out/r03b timeout 900 python tools/probes/fp16_error_probe.py full_scale100 full_cfg1 full_synth_b2 full_regular full_shuffle_k512 > gpurun_out/r03b/fp16_probe.jsonl 2> gpurun_out/r03b/fp16_probe.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03b/vendor -o vendor -- python /root/repo/tools/yardstick_hipblaslt.py 312000 > /root/repo/gpurun_out/r03b/yardstick.log 2>&1)
find gpurun_out/r03b/vendor -type f | head
for f in $(find gpurun_out/r03b/vendor -name "*kernel_stats.csv"); do head -12 $f; done
find gpurun_out/r03b/vendor -name "*kernel_trace.csv" -size +2M -delete
cat gpurun_out/r03b/ab_*.log; cat gpurun_out/r03b/fp16_probe.jsonl | cut -c1-400; tail -3 gpurun_out/r03b/fp16_probe.err
