#!/bin/bash
# round-5 session E: bench lines after the kernel-projection expansion and the constant ortho mode
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_e; mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2>> gpurun_out/${tag}_bench.err
for i in 1 2; do
  timeout 90 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
timeout 300 python -m pytest tests/test_numerics.py tests/test_sharded_cpp.py tests/test_reblur.py -m gpu -x -q > gpurun_out/${tag}_pytest_subset.log 2>&1; tail -3 gpurun_out/${tag}_pytest_subset.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_e_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()}, j.get("frame_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
