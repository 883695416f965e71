cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcgs_$1
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o x -- python3 tools/exp_gs.py ${2:-1000000} 3 > $O/p$i.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        for key in ('k_render_bw','k_render','k_item_scatter','k_span_scatter','k_preprocess_bw','k_preprocess','k_radix_scatter'):
            if key+'(' in n or n.endswith(key) or ('::'+key+'(') in n:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value'])); break
for k,c in acc.items():
    print(k, {n: round(sum(v)/len(v)) for n,v in sorted(c.items())})
PY
