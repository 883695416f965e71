#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_fullsize_properties.py -x -q 2>&1 | tail -2
python3 tools/bench_query.py 8 | tail -1
python3 bench.py > gpurun_out/bench_arena.json 2> gpurun_out/bench_arena.err; tail -c 600 gpurun_out/bench_arena.err
python3 - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/bench_arena.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['samples_per_launch'] if 'samples_per_launch' in d['roofline'] else '')
print({k: d[k] for k in d if k.startswith('secondary')}.keys())
g=d.get('secondary_3dgs') or d.get('secondary') or {}
print(json.dumps(g)[:600])
P
