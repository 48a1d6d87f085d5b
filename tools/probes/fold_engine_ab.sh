#!/bin/bash
# the LayerNorm fold inside the engine, both arms on one box: captions/s (two streams, 3 steps) and rocprofv3 kernel stats of a one-stream pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
C="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-invariance"
for rep in 1 2; do for arm in 0 1; do
  python bench.py $C --opt fold_ln=$arm > $O/b_fold${arm}_$rep.json 2> $O/b.err
done; done
export CZC_NORMAL_EXIT=1
for arm in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$arm -o p -- python bench.py --streams 1 --steps 2 --warmup 1 $C --opt fold_ln=$arm > $O/p$arm.log 2>&1
  f=$(find $O/p$arm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/fold${arm}_kernel_stats.csv
  find $O/p$arm -name "*kernel_trace.csv" -delete; find $O/p$arm -name "*.db" -delete
done
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r05t/b_fold*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], 'value', d['value'], 'frac', r['frac'], 'one-stream', r['single_stream_pass']['frac'], 'util', r['clip_text_mfma_util']['frac'], 'calib', (d.get('box_calibration') or {}).get('tflops'))
for arm in (0,1):
    rows=list(csv.DictReader(open(f'gpurun_out/r05t/fold{arm}_kernel_stats.csv')))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f'### fold_ln={arm}: GPU time {tot/1e6:.0f} ms')
    for r in rows[:7]:
        print(f"  {r['Name'].replace('void ','').replace('czc::','').replace('(anonymous namespace)::','')[:56]:56s} calls {r['Calls']:>6s}  mean {float(r['AverageNs'])/1e3:7.1f} us  {float(r['Percentage']):5.2f} %")
PY
