#!/bin/bash
# round-4 session P: RELAX HistoryFix with four pixels per thread in the early-out test
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_p; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local w=$1; local name=$2; shift; shift
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
run relax_ds_sh old NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so
run relax_ds_sh new X=1
run relax_ds_sh old2 NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so
run relax_ds_sh new2 X=1
run relax_ds new X=1
run relax_ds old NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_p_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_relax.py tests/test_dynamic_resolution.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
