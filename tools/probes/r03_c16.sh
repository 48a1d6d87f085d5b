set -x
mkdir -p gpurun_out/r03q
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r03q/pytest.log
tail -3 gpurun_out/r03q/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03q/smoke.log 2>&1; tail -6 gpurun_out/r03q/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03q/bench_driver_cmd.json 2> gpurun_out/r03q/bench_driver_cmd.err ) 2> gpurun_out/r03q/time_driver.txt
cat gpurun_out/r03q/time_driver.txt
( time timeout 600 python bench.py > gpurun_out/r03q/bench_default.json 2> gpurun_out/r03q/bench_default.err ) 2> gpurun_out/r03q/time_default.txt
cat gpurun_out/r03q/time_default.txt
python - <<'PY'
import json
for f in ("gpurun_out/r03q/bench_driver_cmd.json", "gpurun_out/r03q/bench_default.json"):
    d = json.load(open(f)); m = d["scale100_mode"]
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["single_stream"]["value"], "s100", m["value"], m["steps"], m["warmup"], d["batch_invariance"]["identical_token_frac"], d["cpu_baseline"]["value"])
PY
