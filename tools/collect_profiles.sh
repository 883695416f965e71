#!/bin/bash
# tools/collect_profiles.sh -- run on the GPU box (via gpurun): kernel-trace stats of the default bench command and PMC passes
# (separate runs, --kernel-trace only, as required on this pool) for the pipeline's kernels.  Outputs under gpurun_out/profiles_raw.
# NOTE: gpurun MERGES gpurun_out/ back into the local copy -- delete the local gpurun_out/profiles_raw before a new collection, otherwise
# tools/make_profile_summary.py averages the counters of old and new builds.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_raw
mkdir -p $O
# the driver's command line (bench.py --steps 20 --warmup 5, every leg): the per-kernel averages of the InstantNGP kernels must agree with the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_stats.log 2>&1
# the 3DGS kernels at ONE size (the bench command runs 1 M and 6 M Gaussians through the same kernel names)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gs_stats -- python3 $R/tools/bench_gs.py 1000000 20 > $O/gs_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gs6_stats -- python3 $R/tools/bench_gs.py 6000000 5 > $O/gs6_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 $R/tools/bench_train.py 2200 20 > $O/train_stats.log 2>&1
# the fused training iteration (nerficg_amd.ngp_trainer): 4 warm-up + 3 x 20 iterations, the next batch marched ahead
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused_stats -- python3 $R/tools/bench_train_fused.py 2200 20 1 0 0 1 > $O/fused_stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-32)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 $R/tools/bench_query.py 2 > $O/pmc_$tag.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcgs_$tag -- python3 $R/tools/bench_gs.py 1000000 2 > $O/pmcgs_$tag.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmctr_$tag -- python3 $R/tools/bench_train.py 2200 5 > $O/pmctr_$tag.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcfu_$tag -- python3 $R/tools/bench_train_fused.py 2200 3 0 0 0 1 > $O/pmcfu_$tag.log 2>&1
done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -path "*_stats/*" -name "*kernel_trace.csv" -delete; du -sh $O; ls -la $O | head -50; tail -5 $O/bench_stats.log
