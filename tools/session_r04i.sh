#!/bin/bash
# round-4 session I: a-trous gathers on 4-byte inputs (viewZ / raw normal) instead of 16-byte guide texels; motion-rows tests
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_i; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_${name}_bench.json 2>> gpurun_out/${tag}_bench.err
}
run atr00 NRD_HIP_LIBRARY=$V/atr00/libNRD_hip.so
run atr10 NRD_HIP_LIBRARY=$V/atr10/libNRD_hip.so
run atr11 X=1
run atr11_bands0 NRD_HIP_ATROUS_BANDS=0
run atr00_bands0 NRD_HIP_LIBRARY=$V/atr00/libNRD_hip.so NRD_HIP_ATROUS_BANDS=0
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o trace -- python bench.py --workload relax_ds_sh --steps 20 --warmup 8 --no-cpu-baseline --no-graph > gpurun_out/${tag}_relax_ds_sh_trace_bench.json 2> gpurun_out/${tag}_relax_ds_sh_trace.err
python tools/rocprof_summary.py $(find /tmp/prof_i -name "*.db" | head -1) > gpurun_out/${tag}_relax_ds_sh_kernel_stats.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_i_relax_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], j["parity"]["max_rel_err"] if "parity" in j else None, {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
grep -i atrous gpurun_out/${tag}_relax_ds_sh_kernel_stats.txt | cut -c1-200
timeout 600 python -m pytest tests/test_motion_rows.py tests/test_relax.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
