cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/ks_pmc -o ks -- python $R/profiles/kslots_bench.py > $R/gpurun_out/ks_pmc.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ks_pmc/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'step_fast64' not in k: continue
    agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    n = len(d['SQ_WAVES'])
    print(k, 'dispatches', n, {c: round(sum(v)/len(v)/1e6, 2) for c, v in d.items()})
PY
