#!/bin/bash
# round-4 session T: graph vs eager launches of the final library (which one should bench.py default to?)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_t; mkdir -p gpurun_out
for i in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_graph${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  timeout 120 python bench.py --no-cpu-baseline --no-graph > gpurun_out/${tag}_reblur_ds_eager${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_reblur_ds_graph_driver_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-graph > gpurun_out/${tag}_reblur_ds_eager_driver_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_t_*_bench.json")):
    j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], j["launch"][:40], (j.get("whole_chain") or {}).get("sum_kernel_ms"))
PY
