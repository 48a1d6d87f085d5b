#!/bin/bash
# the refine engine with fp16 rows + folded LayerNorm in its SCREENING pass (option resid16 = 2): deviations, gate sweep, throughput
cd /root/repo; O=gpurun_out/r05h; mkdir -p $O
CZC_OPTS=resid16=2 GATES=400,600,800 python tools/refine_validate.py 128 10 12 2000 > $O/rv_r16_gates.jsonl 2> $O/rv_r16_gates.err
C="--precision refine --logit-scale 4.6052 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-invariance --no-profile"
for rep in 1 2; do
python bench.py $C > $O/b_base_$rep.json 2> $O/b_base.err
python bench.py $C --opt resid16=2 --opt refine_gate_x1e6=600 --opt refine_guard_x1e6=320 > $O/b_r16_$rep.json 2> $O/b_r16.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d.get('refine'))
    except Exception as e: print(f, 'ERR', e)
PY
