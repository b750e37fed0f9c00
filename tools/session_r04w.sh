#!/bin/bash
# round-4 session W: per-kernel tile-row order (bottom-up in the REBLUR TA kernels and in the RELAX passes) against top-down everywhere
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_w; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
for i in 1 2; do
  for w in reblur_ds relax_ds_sh; do
    NRD_HIP_LIBRARY=$V/topdown/libNRD_hip.so timeout 60 python bench.py --workload $w --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_topdown${i}_bench.json 2>> gpurun_out/${tag}_bench.err
    timeout 60 python bench.py --workload $w --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_w_*_bench.json")):
    j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
PY
