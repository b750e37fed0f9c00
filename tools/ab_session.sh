#!/bin/bash
# (development) kernel-trace A/B of library variants and run-time switches on the GPU box:
#   bash tools/ab_session.sh TAG WORKLOAD GREP NAME[:ENV=VAL,...][@VARIANT] ...     e.g.  bash tools/ab_session.sh r06_d relax_ds_sh Atrous base g32@g32 march:NRD_HIP_ATROUS_MARCH=16   (GREP: ONE word)
cd "$(dirname "$0")/.." 2>/dev/null; export TMPDIR=/tmp
tag=$1; w=$2; pat=$3; shift 3
for spec in "$@"; do
  name=${spec%%[:@]*}; envs=""; variant=""
  [[ "$spec" == *@* ]] && variant=${spec##*@} && spec=${spec%@*}
  [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=""; [[ -n "$variant" ]] && lib="NRD_HIP_LIBRARY=raytracingdenoiser_amd/lib/variants/$variant/libNRD_hip.so"
  rm -rf /tmp/prof_t
  env $envs $lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o trace -- python bench.py --workload $w --steps ${STEPS:-20} --warmup ${WARMUP:-8} --no-cpu-baseline --no-parity $BENCH_ARGS > gpurun_out/${tag}_${name}_trace_bench.json 2> gpurun_out/${tag}_${name}_trace.err
  python tools/rocprof_summary.py $(find /tmp/prof_t -name "*.db" | head -1) > gpurun_out/${tag}_${name}_kernel_stats.txt 2>&1
  echo "== $name: $(python -c "import json,sys; print(json.loads(open('gpurun_out/${tag}_${name}_trace_bench.json').read().strip().split(chr(10))[-1])['ms_per_step'])" 2>/dev/null) ms"
  grep -- "$pat" gpurun_out/${tag}_${name}_kernel_stats.txt < /dev/null | cut -c1-60,98-150
done
