#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_6.log 2>&1; tail -2 gpurun_out/r04_gputest_6.log
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -2 gpurun_out/collect.log
