set -x
mkdir -p gpurun_out/r03c
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r03c/pytest.log
timeout 400 python bench.py > gpurun_out/r03c/bench_default.json 2> gpurun_out/r03c/bench_default.err
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 3,7 8 > gpurun_out/r03c/ab_fc2.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 3,7 8 > gpurun_out/r03c/ab_out.log 2>&1
timeout 300 python tools/bench_gemm.py 312000 > gpurun_out/r03c/bench_gemm.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03c/vendor -o vendor -- python /root/repo/tools/yardstick_hipblaslt.py 312000 > /root/repo/gpurun_out/r03c/yardstick.log 2>&1)
for f in $(find gpurun_out/r03c/vendor -name "*kernel_stats.csv"); do head -12 $f; done
find gpurun_out/r03c/vendor -name "*kernel_trace.csv" -size +2M -delete
find gpurun_out/r03c/vendor -name "*.db" -delete
cat gpurun_out/r03c/pytest.log | tail -15; cat gpurun_out/r03c/bench_default.json | cut -c1-1500; tail -3 gpurun_out/r03c/bench_default.err; cat gpurun_out/r03c/ab_*.log gpurun_out/r03c/bench_gemm.log
