set -x
mkdir -p gpurun_out/r03e
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dropin_gpu.py tests/test_streams_gpu.py -x -q -m gpu -k "refine or legacy or precision_selected or scale100" 2>&1 | tail -40 > gpurun_out/r03e/pytest_refine.log
timeout 900 python tools/refine_validate.py 64 10 12,8,6,4,0 4000,6000,3000 > gpurun_out/r03e/refine_validate.jsonl 2> gpurun_out/r03e/refine_validate.err
timeout 600 python bench.py --precision refine --logit-scale 4.6052 --no-cpu-baseline --no-invariance > gpurun_out/r03e/bench_refine.json 2> gpurun_out/r03e/bench_refine.err
timeout 600 python bench.py --precision refine --logit-scale 4.6052 --no-cpu-baseline --no-invariance --opt refine_samples=6 > gpurun_out/r03e/bench_refine_s6.json 2>> gpurun_out/r03e/bench_refine.err
tail -30 gpurun_out/r03e/pytest_refine.log; cat gpurun_out/r03e/refine_validate.jsonl; tail -3 gpurun_out/r03e/refine_validate.err; cut -c1-300 gpurun_out/r03e/bench_refine.json; tail -3 gpurun_out/r03e/bench_refine.err
