#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_gs_parity.py tests/test_gpu_fullsize_properties.py -x -q 2>&1 | tail -3
for i in 1 2; do python3 tools/gs_fwd_only.py 1000000 40 | tail -1; done
python3 tools/gs_fwd_only.py 6000000 10 | tail -1
python3 tools/gs_quick.py 2>&1 | tail -3
