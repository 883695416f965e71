#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_gs_parity.py tests/test_gpu_fused_training_ops.py tests/test_gpu_gs_lifecycle.py tests/test_gpu_graphs.py -x -q 2>&1 | tail -3
for i in 1 2; do python3 tools/gs_fwd_only.py 1000000 40 | tail -1; done
python3 tools/gs_quick.py 2>&1 | tail -4
bash tools/kseq.sh 13 tools/gs_fwd_only.py 1000000 20 2>&1 | tail -16
