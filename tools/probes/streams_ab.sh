for s in 2 3 4 2 3; do python bench.py --streams $s --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-invariance --no-profile 2>/dev/null > gpurun_out/streams_$s.json; python -c "
import json; d=json.load(open('gpurun_out/streams_$s.json')); print('streams $s', d['value'])"; done
