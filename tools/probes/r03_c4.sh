set -x
mkdir -p gpurun_out/r03f
cd /root/repo
export TMPDIR=/tmp
timeout 900 python tools/refine_validate.py 128 10 12,8,6 4000,3000 0,1000,3000 > gpurun_out/r03f/refine_validate.jsonl 2> gpurun_out/r03f/refine_validate.err
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "free_running_half" 2>&1 | tail -30 > gpurun_out/r03f/pytest_div.log
timeout 300 python bench.py --images 1 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 > gpurun_out/r03f/bench_b1.json 2> gpurun_out/r03f/bench_b1.err
timeout 300 python bench.py --images 1 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 --opt test:wreg_min_m=1 --opt test:gemm256_min_m=1 --opt test:rowln_min_m=1 > gpurun_out/r03f/bench_b1_forced.json 2>> gpurun_out/r03f/bench_b1.err
timeout 300 python bench.py --images 1 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 --opt test:wreg_min_m=1 > gpurun_out/r03f/bench_b1_wreg.json 2>> gpurun_out/r03f/bench_b1.err
timeout 300 python bench.py --images 8 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 > gpurun_out/r03f/bench_b8.json 2>> gpurun_out/r03f/bench_b1.err
timeout 300 python bench.py --images 8 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 --opt test:wreg_min_m=1 --opt test:gemm256_min_m=1 --opt test:rowln_min_m=1 > gpurun_out/r03f/bench_b8_forced.json 2>> gpurun_out/r03f/bench_b1.err
cat gpurun_out/r03f/pytest_div.log; for f in gpurun_out/r03f/bench_b*.json; do echo $f; cut -c1-200 $f; done; tail -3 gpurun_out/r03f/bench_b1.err
