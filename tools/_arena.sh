#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_ngp_parity.py tests/test_gpu_garden_parity.py tests/test_gpu_fullsize_properties.py -x -q 2>&1 | tail -3
for i in 1 2; do python3 tools/bench_query.py 8 | tail -1; done
bash tools/kseq.sh 29 tools/bench_query.py 3 2>&1 | grep -B1 -A4 "k_scan_tiles"
