#!/bin/bash
# round-4 session K: RELAX TemporalAccumulation at 4 waves (128 VGPRs + 160 B scratch) and with the LDS window, against the product
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_k; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
run product X=1
run rta4 NRD_HIP_LIBRARY=$V/rta4/libNRD_hip.so
run window NRD_HIP_RELAX_TA_WINDOW=1
run rta4_window NRD_HIP_LIBRARY=$V/rta4/libNRD_hip.so NRD_HIP_RELAX_TA_WINDOW=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_k_relax_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
