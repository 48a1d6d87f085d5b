set -x
mkdir -p gpurun_out/r03n
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r03n/pytest_kernels.log
tail -3 gpurun_out/r03n/pytest_kernels.log
timeout 600 python bench.py --no-alt --no-cpu-baseline --no-invariance > gpurun_out/r03n/bench_bf16.json 2> gpurun_out/r03n/bench.err
timeout 300 python bench.py --images 1 --no-profile --no-alt --no-cpu-baseline --no-invariance --steps 3 > gpurun_out/r03n/bench_b1.json 2>> gpurun_out/r03n/bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r03n/bench_bf16.json", "gpurun_out/r03n/bench_b1.json"):
    d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("single_stream"), d["kernel_ms_one_step"])
PY
