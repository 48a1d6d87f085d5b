#!/bin/bash
# the refine engine's czc_generate with its screening pass on fp16 rows (refine_rows16): the scale factor of gate / guard / theta
cd /root/repo; O=gpurun_out/r05h; mkdir -p $O
CZC_OPTS=refine_rows16_x1000=1750 GEN_SWEEPS=10 python tools/refine_validate.py 256 10 12 2000 > $O/rv256_f175.jsonl 2> $O/rv256_f175.err
C="--precision refine --logit-scale 4.6052 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-invariance --no-profile"
for rep in 1 2; do
python bench.py $C --opt refine_rows16=0 > $O/c_rows32_$rep.json 2> $O/c.err
python bench.py $C > $O/c_f150_$rep.json 2> $O/c.err
python bench.py $C --opt refine_rows16_x1000=1750 > $O/c_f175_$rep.json 2> $O/c.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/c_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('refine') or {}; print(f, d['value'], r.get('gated_frac'), r.get('re_encoded_frac'))
    except Exception as e: print(f, 'ERR', e)
PY
