set -x
mkdir -p gpurun_out/r03a
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm256_variants" 2>&1 | tail -15 > gpurun_out/r03a/test_variants.log
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 3,7,7:1,7:2,7:3 10 > gpurun_out/r03a/ab_fc2.log 2>&1
timeout 300 python tools/ab_gemm.py 156000 512 2048 0 1 3,7 8 >> gpurun_out/r03a/ab_fc2.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 3,7,7:2 8 > gpurun_out/r03a/ab_out.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 1536 512 0 0 3,7,6 6 > gpurun_out/r03a/ab_qkv.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 2048 512 1 0 3,7,6 6 > gpurun_out/r03a/ab_fc1.log 2>&1
timeout 400 python tools/probes/fp16_error_probe.py full_scale100 > gpurun_out/r03a/fp16_probe.jsonl 2> gpurun_out/r03a/fp16_probe.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03a/vendor -o vendor -- python /root/repo/tools/yardstick_hipblaslt.py 312000 > /root/repo/gpurun_out/r03a/yardstick.log 2>&1)
find gpurun_out/r03a/vendor -name "*stats*" | head
for f in $(find gpurun_out/r03a/vendor -name "*kernel_stats.csv"); do head -12 $f; done
find gpurun_out/r03a/vendor -name "*.db" -delete; find gpurun_out/r03a/vendor -name "*kernel_trace.csv" -size +2M -delete
cat gpurun_out/r03a/test_variants.log gpurun_out/r03a/ab_*.log; head -1 gpurun_out/r03a/fp16_probe.jsonl; tail -3 gpurun_out/r03a/fp16_probe.err
