set -x
mkdir -p gpurun_out/r03o
cd /root/repo
export TMPDIR=/tmp
C="--no-alt --no-cpu-baseline --no-invariance --no-profile --steps 3 --warmup 1"
for i in 1 2; do
timeout 300 python bench.py $C > gpurun_out/r03o/s2_$i.json 2>> gpurun_out/r03o/err.log
timeout 300 python bench.py $C --streams 3 > gpurun_out/r03o/s3_$i.json 2>> gpurun_out/r03o/err.log
timeout 300 python bench.py $C --opt fuse_ln=2 > gpurun_out/r03o/ln2_$i.json 2>> gpurun_out/r03o/err.log
done
timeout 300 python bench.py $C --precision refine --logit-scale 4.6052 > gpurun_out/r03o/refine_s2.json 2>> gpurun_out/r03o/err.log
timeout 300 python bench.py $C --precision refine --logit-scale 4.6052 --streams 3 > gpurun_out/r03o/refine_s3.json 2>> gpurun_out/r03o/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03o/*.json")):
    d = json.load(open(f)); print(f, d["value"], d["ms_per_step"])
PY
