#!/bin/bash
# The 1 -> 8 GPU scaling curve of the metric, one command (BASELINE.json: "agent-steps/sec ... at 1/2/4/8 GPU"):
#
#   bash profiles/scale.sh [N ...]          (default: 1 2 4 8; needs that many visible GPUs - fails loudly otherwise)
#
# For every N: `python bench.py --gpus N --lean` (self-launching: diral_amd/spawn.py starts one rank per GPU under
# torch.distributed.run, RCCL group over 127.0.0.1, weak scaling: 4096 envs per GPU, env offset = rank x 4096 of ONE
# seeded batch, the only collective one all-reduce of 7 doubles per report).  Prints agent-steps/s, the efficiency
# against N = 1 (value(N) / (N * value(1))), the slowest and fastest rank's own step and kernel time (`per_rank`) and
# the "RCCL process group up: backend=nccl world=N" line of rank 0.  JSON lines go to gpurun_out/scale_<N>.json.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
NS="$@"; [ -z "$NS" ] && NS="1 2 4 8"
VIS=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
for N in $NS; do
  if [ "$N" -gt "$VIS" ]; then echo "scale.sh: $N GPUs needed, $VIS visible" >&2; exit 2; fi
done
STEPS=${STEPS:-400}; WARM=${WARM:-50}
BASE=""
for N in $NS; do
  python bench.py --gpus $N --steps $STEPS --warmup $WARM --lean > gpurun_out/scale_$N.stdout 2> gpurun_out/scale_$N.log
  rc=$?
  if [ $rc -ne 0 ]; then echo "scale.sh: bench.py --gpus $N failed (rc $rc); see gpurun_out/scale_$N.log" >&2; tail -5 gpurun_out/scale_$N.log >&2; exit $rc; fi
  grep '^{' gpurun_out/scale_$N.stdout | tail -1 > gpurun_out/scale_$N.json
  python - "$N" "$BASE" <<'PY'
import json, sys
n = int(sys.argv[1]); base = float(sys.argv[2]) if sys.argv[2] else None
d = json.load(open("gpurun_out/scale_%d.json" % n))
assert d["n_gpus"] == n, d
v = d["value"]
eff = "" if base is None else "  efficiency vs N=1: %.3f" % (v / (n * base))
pr = d.get("per_rank")
prs = "" if not pr else "  per rank: step %.4f..%.4f ms, kernel %.4f..%.4f ms" % (
    pr["ms_per_step_min"], pr["ms_per_step_max"], pr["kernel_ms_min"], pr["kernel_ms_max"])
print("N=%d  %.4g agent-steps/s  %.4f ms/step%s%s" % (n, v, d["ms_per_step"], eff, prs))
PY
  [ -z "$BASE" ] && BASE=$(python -c "import json; print(json.load(open('gpurun_out/scale_$N.json'))['value'] / $N)")
  grep -h "RCCL process group up" gpurun_out/scale_$N.log gpurun_out/scale_$N.stdout 2>/dev/null | head -1
done
