#!/bin/bash
# round-4 session L: RELAX TemporalAccumulation with the virtual-motion geometry computed in front of the surface-motion taps (its depth quad requested early); C++ sharded test with measured motion
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_l; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
run old NRD_HIP_LIBRARY=$V/rta_old/libNRD_hip.so
run new X=1
run old2 NRD_HIP_LIBRARY=$V/rta_old/libNRD_hip.so
run new2 X=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_l_relax_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_sharded_cpp.py tests/test_relax.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
