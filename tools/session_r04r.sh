#!/bin/bash
# round-4 session R: the XCD stripes as a staircase (moved on by one XCD every K tile rows) in EVERY pass: K = 1, 4, 16 against the product (HistoryFix passes rotated only)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_r; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local w=$1; local name=$2; shift; shift
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
for w in reblur_ds relax_ds_sh; do
  run $w product X=1
  for k in 1 4 16; do run $w stair$k NRD_HIP_LIBRARY=$V/stair$k/libNRD_hip.so; done
  run $w product2 X=1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_r_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
