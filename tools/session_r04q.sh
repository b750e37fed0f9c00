#!/bin/bash
# round-4 session Q: HistoryFix with the tile columns of each tile row moved on by one XCD stripe (RELAX: default on; REBLUR: variant)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_q; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local w=$1; local name=$2; shift; shift
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
run relax_ds_sh norot NRD_HIP_LIBRARY=$V/norot/libNRD_hip.so
run relax_ds_sh rot X=1
run relax_ds_sh norot2 NRD_HIP_LIBRARY=$V/norot/libNRD_hip.so
run relax_ds_sh rot2 X=1
run reblur_ds norot X=1
run reblur_ds rot NRD_HIP_LIBRARY=$V/rhf_rot/libNRD_hip.so
run reblur_ds norot2 X=1
run reblur_ds rot2 NRD_HIP_LIBRARY=$V/rhf_rot/libNRD_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_norot_driver_bench.json 2>> gpurun_out/${tag}_bench.err
env NRD_HIP_LIBRARY=$V/rhf_rot/libNRD_hip.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_rot_driver_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_q_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -x -q -k "baseline_size and RELAX" > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
