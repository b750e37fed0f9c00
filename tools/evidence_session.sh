#!/bin/bash
# The evidence session of a round (run on the GPU box through gpurun): everything DESIGN.md and bench.py quote, at HEAD, into gpurun_out/<tag>_* (copy what is quoted to profiles/).
#   usage: bash tools/evidence_session.sh TAG [quick]
#   1. the complete -m gpu suite (timed, with durations) and smoke()
#   2. the bench lines: the driver's protocol (--steps 20 --warmup 5, with cpu_baseline + parity), SURVEY 8d's protocol, --no-sky, the other BASELINE configs
#   3. rocprofv3 --kernel-trace --stats of the same commands (per-kernel avg / min / max)
#   4. issue floors: bench.py --uniform with the product and with the L1-resident A/B build (tools/build_variant.py l1 -DNRD_EXPERIMENT_L1_RESIDENT=1 beforehand)
#   5. hardware counters, one rocprofv3 --pmc pass per set (tools/pmc_run.sh: FETCH_SIZE; WRITE_SIZE; SQ issue / waits; TCC hit / miss; LDS instructions + bank conflicts)
#   6. the N = 1 / 2 / 4 / 8 compute model of the halo scheme (tools/model_scaling.py: virtual ranks on one GPU -- MODELLED, no transfers)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r06_last}; quick=$2; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log; tail -4 gpurun_out/${tag}_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2> gpurun_out/${tag}_bench.err
timeout 400 python bench.py > gpurun_out/${tag}_reblur_ds_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 200 python bench.py --no-sky --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_nosky_bench.json 2>> gpurun_out/${tag}_bench.err
for w in relax_ds_sh reblur_diffuse sigma_shadow relax_ds; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err
done
trace() { # name, bench args...
  local name=$1; shift; rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o trace -- python bench.py "$@" --no-cpu-baseline --no-parity > gpurun_out/${tag}_${name}_trace_bench.json 2> gpurun_out/${tag}_${name}_trace.err
  python tools/rocprof_summary.py $(find /tmp/prof_t -name "*.db" | head -1) > gpurun_out/${tag}_${name}_kernel_stats.txt 2>&1
}
trace reblur_ds --steps 20 --warmup 5
trace reblur_ds_steady --steps 32 --warmup 32
trace relax_ds_sh --workload relax_ds_sh --steps 20 --warmup 8
trace sigma_shadow --workload sigma_shadow --steps 20 --warmup 8
if [[ -f $V/l1/libNRD_hip.so ]]; then
  for w in reblur_ds relax_ds_sh; do
    trace ${w}_uniform_product --workload $w --uniform --steps 20 --warmup 8
    NRD_HIP_LIBRARY=$V/l1/libNRD_hip.so trace ${w}_uniform_l1 --workload $w --uniform --steps 20 --warmup 8
  done
fi
if [[ -z "$quick" ]]; then
  for w in reblur_ds relax_ds_sh sigma_shadow; do
    bash tools/pmc_run.sh ${tag}_${w} --workload $w --steps 8 --warmup 4 --no-parity > /dev/null 2>&1
  done
  PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" bash tools/pmc_run.sh ${tag}_reblur_ds_nosky --workload reblur_ds --no-sky --steps 8 --warmup 4 --no-parity > /dev/null 2>&1
  for w in reblur_ds relax_ds_sh sigma_shadow; do
    timeout 900 python tools/model_scaling.py --workload $w > gpurun_out/${tag}_scaling_model_${w}.json 2> gpurun_out/${tag}_scaling_model_${w}.err
  done
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${tag}_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], j["value"], (j.get("roofline") or {}).get("frac"), (j.get("frame_ms") or {}).get("max"), (j.get("parity") or {}).get("max_rel_err"))
    except Exception as e:
        print(f, "unreadable", e)
PY
