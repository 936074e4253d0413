#!/bin/bash
# Round-3 first GPU session (through gpurun, from the repo root): parity suite, the bench line, the
# bench under torch.distributed.run with ONE rank (RCCL group creation log), PMC passes of the headline.
set -u
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r03a_pytest.txt
python bench.py 2> gpurun_out/r03a_bench.err | tail -1 > gpurun_out/r03a_bench.json
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --gpus 1 --steps 200 --warmup 20 --lean > gpurun_out/r03a_torchrun1.json 2> gpurun_out/r03a_torchrun1.log
python bench.py --gpus 2 > gpurun_out/r03a_gpus2.out 2>&1; echo "rc=$?" >> gpurun_out/r03a_gpus2.out
FULL_PMC=1 bash profiles/run_profile.sh r03a_c2 > /dev/null 2>&1
tail -5 gpurun_out/r03a_pytest.txt; cat gpurun_out/r03a_gpus2.out; tail -3 gpurun_out/r03a_torchrun1.log | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03a_bench.json").read())
print({k: d[k] for k in ("value", "ms_per_step")}, json.dumps(d["roofline"])[:1500])
print(json.dumps(d.get("also_measured"))[:3000])
PY
grep -A8 "PASS_NS\|steady" gpurun_out/prof_r03a_c2/summary.txt | head -40
