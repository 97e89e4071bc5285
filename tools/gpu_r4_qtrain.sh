#!/bin/bash
O=gpurun_out/r4_qtrain; mkdir -p $O
timeout 900 python bench.py --model Q --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/q.json 2> $O/q.log; cut -c1-300 $O/q.json
timeout 900 python - > $O/qtrain.txt 2>&1 <<'PY'
import json, torch, bench
bench.log = lambda *a, **k: print(*a, **k)
out = bench.extra_gemnet_q(32, 32, 0, steps=5, warmup=2)
print(json.dumps(out))
PY
tail -c 2500 $O/qtrain.txt
