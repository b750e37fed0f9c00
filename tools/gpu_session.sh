#!/bin/bash
# One GPU session (run on the GPU box through gpurun; everything lands in gpurun_out/<tag>_*). Parametrised replacement of the per-session scripts of
# round 2:  tools/gpu_session.sh TAG STEP [STEP ...]   with steps
#   smoke            __graft_entry__.smoke()
#   bench[:WORKLOAD] bench.py (default workload reblur_ds) -> <tag>_<workload>_bench.json       bench_nosky[:WORKLOAD]   the same with --no-sky
#   trace[:WORKLOAD] rocprofv3 --kernel-trace --stats of a short bench run -> <tag>_<workload>_kernel_stats.txt
#   pmc[:WORKLOAD]   tools/pmc_run.sh (one rocprofv3 --pmc pass per counter set)
#   pytest[:KEXPR]   python -m pytest tests -m gpu [-k KEXPR] -> <tag>_pytest_gpu.log
#   variant:NAME     bench.py with NRD_HIP_LIBRARY=raytracingdenoiser_amd/lib/variants/NAME/libNRD_hip.so (tools/build_variant.py)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=$1; shift; mkdir -p gpurun_out
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $kind in
    smoke) timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/${tag}_smoke.log ;;
    bench|bench_nosky)
      w=${arg:-reblur_ds}; extra=""; suffix=""; [[ $kind == bench_nosky ]] && extra="--no-sky --no-cpu-baseline" && suffix="_nosky"
      timeout 600 python bench.py --workload $w $extra > gpurun_out/${tag}_${w}${suffix}_bench.json 2> gpurun_out/${tag}_${w}${suffix}_bench.err; tail -1 gpurun_out/${tag}_${w}${suffix}_bench.json | cut -c1-400 ;;
    variant)
      timeout 300 env NRD_HIP_LIBRARY=raytracingdenoiser_amd/lib/variants/$arg/libNRD_hip.so python bench.py --no-cpu-baseline > gpurun_out/${tag}_variant_${arg}_bench.json 2>> gpurun_out/${tag}_variant.err
      tail -1 gpurun_out/${tag}_variant_${arg}_bench.json | cut -c1-200 ;;
    trace)
      w=${arg:-reblur_ds}; rm -rf /tmp/prof_$w
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o trace -- python bench.py --workload $w --steps 20 --warmup 8 --no-cpu-baseline --no-graph > gpurun_out/${tag}_${w}_trace_bench.json 2> gpurun_out/${tag}_${w}_trace.err
      python tools/rocprof_summary.py $(find /tmp/prof_$w -name "*.db" | head -1) > gpurun_out/${tag}_${w}_kernel_stats.txt 2>&1; head -14 gpurun_out/${tag}_${w}_kernel_stats.txt ;;
    pmc) w=${arg:-reblur_ds}; bash tools/pmc_run.sh ${tag}_${w} --workload $w --steps 8 --warmup 4 --no-cpu-baseline --no-graph ;;
    pytest)
      if [[ -n "$arg" ]]; then timeout 3000 python -m pytest tests -m gpu -x -q -k "$arg" > gpurun_out/${tag}_pytest_gpu.log 2>&1; else timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; fi
      echo "pytest exit $?" | tee -a gpurun_out/${tag}_pytest_gpu.log; tail -5 gpurun_out/${tag}_pytest_gpu.log ;;
    *) echo "unknown step $step" ;;
  esac
done
