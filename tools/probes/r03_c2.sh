set -x
mkdir -p gpurun_out/r03d
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "refine or legacy or precision_selected" 2>&1 | tail -40 > gpurun_out/r03d/pytest_refine.log
timeout 600 python bench.py --alt-split > gpurun_out/r03d/bench_default.json 2> gpurun_out/r03d/bench_default.err
export CZC_LIB_PATH=/root/repo/conzic_amd/lib/libconzic_hip_exp.so
timeout 300 python tools/ab_gemm.py 312000 512 2048 0 1 7,7:256,7:512,7:1024,7:2048,7:768,7:1536,7:3072,7:3584,7:2560 6 > gpurun_out/r03d/ab_fc2_ablate.log 2>&1
timeout 300 python tools/ab_gemm.py 312000 512 512 0 1 7,7:256,7:512,7:1024,7:2048,7:768 6 > gpurun_out/r03d/ab_out_ablate.log 2>&1
unset CZC_LIB_PATH
cat gpurun_out/r03d/pytest_refine.log | tail -30; cut -c1-600 gpurun_out/r03d/bench_default.json; tail -3 gpurun_out/r03d/bench_default.err; cat gpurun_out/r03d/ab_*.log
