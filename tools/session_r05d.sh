#!/bin/bash
# round-5 session D: the complete GPU suite (xdist + passive OpenMP + no rebuild on the box) with durations, then the bench lines after the exp2 / material-test changes
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_d; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=30 ) > gpurun_out/${tag}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log; tail -45 gpurun_out/${tag}_pytest_gpu.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2>> gpurun_out/${tag}_bench.err
for i in 1 2; do
  timeout 90 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  timeout 90 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_d_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
