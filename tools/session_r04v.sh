#!/bin/bash
# round-4 session V: tile rows walked bottom-up (the sky tiles of the scene dispatched last) against top-down
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_v; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
for i in 1 2; do
  timeout 60 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_topdown${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  NRD_HIP_LIBRARY=$V/rev/libNRD_hip.so timeout 60 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_bottomup${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
NRD_HIP_LIBRARY=$V/rev/libNRD_hip.so timeout 60 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_bottomup_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 60 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_topdown_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_v_*_bench.json")):
    j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
PY
