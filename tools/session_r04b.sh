#!/bin/bash
# round-4 session B: per-kernel issue floors (kernel traces of the uniform scene, product vs L1-resident build)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_b; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
for w in reblur_ds relax_ds_sh; do
  for lib in product l1; do
    rm -rf /tmp/prof_$w
    if [[ $lib == l1 ]]; then export NRD_HIP_LIBRARY=$V/l1/libNRD_hip.so; else unset NRD_HIP_LIBRARY; fi
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o trace -- python bench.py --workload $w --uniform --steps 20 --warmup 8 --no-cpu-baseline --no-parity --no-graph > gpurun_out/${tag}_${w}_uniform_${lib}_trace_bench.json 2> gpurun_out/${tag}_${w}_uniform_${lib}_trace.err
    python tools/rocprof_summary.py $(find /tmp/prof_$w -name "*.db" | head -1) > gpurun_out/${tag}_${w}_uniform_${lib}_kernel_stats.txt 2>&1
  done
done
unset NRD_HIP_LIBRARY
head -20 gpurun_out/${tag}_relax_ds_sh_uniform_product_kernel_stats.txt; head -20 gpurun_out/${tag}_relax_ds_sh_uniform_l1_kernel_stats.txt
