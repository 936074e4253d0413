cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ro -o t -- python $GRAFT_REPO_ROOT/examples/rollout_sps.py --envs 4096 --slots 600 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ro/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
