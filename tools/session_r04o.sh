#!/bin/bash
# round-4 session O: HistoryFix reconstruction taps batched per stencil row (RELAX: guides / weights / signals in three phases; REBLUR: the row loop unrolled)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_o; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local w=$1; local name=$2; shift; shift
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
for w in relax_ds_sh reblur_ds; do
  run $w old NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so
  run $w new X=1
  run $w old2 NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so
  run $w new2 X=1
done
run relax_ds_sh hf4 NRD_HIP_LIBRARY=$V/hf4/libNRD_hip.so
# the driver's protocol: frames 5..24, where more pixels are young
env NRD_HIP_LIBRARY=$V/hf_old/libNRD_hip.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_old_driver_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_new_driver_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_o_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_relax.py tests/test_reblur.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
