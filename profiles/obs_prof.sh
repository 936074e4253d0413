cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_obs -o t -- python $GRAFT_REPO_ROOT/profiles/side_paths.py c2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_obs_pmc -o p -- python $GRAFT_REPO_ROOT/profiles/side_paths.py c2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_obs/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_obs_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "observe" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v[:130]) / 130, sum(v[-100:]) / 100, len(v))
PY
