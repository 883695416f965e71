#!/bin/bash
# tools/collect_kseq.sh ROUND -- run on the GPU box: kernel sequences (tools/kseq.sh) of every leg into gpurun_out/ROUND_kseq_*.txt (copy to profiles/ afterwards)
R=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out
bash $GRAFT_REPO_ROOT/tools/kseq.sh 30 tools/bench_query.py 3 > $O/${R}_kseq_ingp_frame.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/kseq.sh 28 tools/bench_train_fused.py 2200 20 0 0 0 1 > $O/${R}_kseq_ingp_fused_iteration.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/kseq.sh 32 tools/bench_train_fused.py 2200 20 1 0 0 1 > $O/${R}_kseq_ingp_fused_iteration_marched_ahead.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/kseq.sh 24 tools/gs_fwd_only.py 1000000 20 > $O/${R}_kseq_gs_forward.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/kseq.sh 44 tools/bench_gs_step.py 1000000 12 > $O/${R}_kseq_gs_step.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/kseq.sh 40 tools/bench_gs_step.py 1000000 12 fuse > $O/${R}_kseq_gs_step_rest_adam_in_backward.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/gs_stats.py 1000000 > $O/${R}_gs_stats.txt 2>&1
ls -la $O | grep ${R}_kseq
