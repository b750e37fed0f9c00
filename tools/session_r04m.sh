#!/bin/bash
# round-4 session M: REBLUR TemporalAccumulation with the curvature estimate's high-parallax tap requested in front of the surface-motion section
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_m; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_reblur_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
run old NRD_HIP_LIBRARY=$V/ta_old/libNRD_hip.so
run new X=1
run old2 NRD_HIP_LIBRARY=$V/ta_old/libNRD_hip.so
run new2 X=1
run old_nosky NRD_HIP_LIBRARY=$V/ta_old/libNRD_hip.so X=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_m_reblur_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_reblur.py tests/test_executor.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
