#!/bin/bash
# round-4 session J: a-trous bands with undecoded 40-byte texels and the next band requested before the current one is filtered
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_j; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_${name}_bench.json 2>> gpurun_out/${tag}_bench.err; }
trace() { local name=$1; shift
    env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o trace -- python bench.py --workload relax_ds_sh --steps 20 --warmup 8 --no-cpu-baseline --no-graph > gpurun_out/${tag}_relax_${name}_trace_bench.json 2> gpurun_out/${tag}_relax_${name}_trace.err
    python tools/rocprof_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) > gpurun_out/${tag}_relax_${name}_kernel_stats.txt 2>&1; grep -i "atrous" gpurun_out/${tag}_relax_${name}_kernel_stats.txt | cut -c45-200; }
run old NRD_HIP_LIBRARY=$V/bands_old/libNRD_hip.so
run new X=1
run new_bands16 NRD_HIP_ATROUS_BANDS=16
trace new X=1
trace new_bands16 NRD_HIP_ATROUS_BANDS=16
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_j_relax_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_relax.py tests/test_full_parity.py -m gpu -x -q -k "relax or RELAX" > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
