set -x
mkdir -p gpurun_out/r03j
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:20,7:44,7:88,7:176 8 > gpurun_out/r03j/ab_fc2_stagger.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 7,7:12,7:24 8 > gpurun_out/r03j/ab_out_stagger.log 2>&1
cat gpurun_out/r03j/ab_*.log
