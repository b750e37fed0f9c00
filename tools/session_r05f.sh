#!/bin/bash
# round-5 session F: A/B of the 4-waves-per-SIMD a-trous variant (NRD_WAVES_RELAX_ATROUS=4: step 2 at 128 VGPRs + 20 B of scratch)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_f; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
for i in 1 2; do
  NRD_HIP_LIBRARY=$V/atr4/libNRD_hip.so timeout 90 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_atr4_${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  timeout 90 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
rm -rf /tmp/prof_r; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o trace -- python bench.py --workload relax_ds_sh --steps 20 --warmup 8 --no-cpu-baseline --no-parity > /dev/null 2>> gpurun_out/${tag}_bench.err
python tools/rocprof_summary.py $(find /tmp/prof_r -name "*.db" | head -1) > gpurun_out/${tag}_relax_ds_sh_kernel_stats.txt 2>&1
rm -rf /tmp/prof_r; NRD_HIP_LIBRARY=$V/atr4/libNRD_hip.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o trace -- python bench.py --workload relax_ds_sh --steps 20 --warmup 8 --no-cpu-baseline --no-parity > /dev/null 2>> gpurun_out/${tag}_bench.err
python tools/rocprof_summary.py $(find /tmp/prof_r -name "*.db" | head -1) > gpurun_out/${tag}_relax_ds_sh_atr4_kernel_stats.txt 2>&1
head -16 gpurun_out/${tag}_relax_ds_sh_kernel_stats.txt | cut -c1-200; head -16 gpurun_out/${tag}_relax_ds_sh_atr4_kernel_stats.txt | cut -c1-200
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_f_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
