#!/bin/bash
# End-of-round GPU session: the complete GPU suite, the bench lines of every workload, kernel traces, counters, the issue floors and the multi-GPU model -- all on ONE build.
# usage (through gpurun): bash tools/final_session.sh TAG      (build the L1-resident variant first: python tools/build_variant.py l1 -DNRD_EXPERIMENT_L1_RESIDENT=1)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=$1
bash tools/gpu_session.sh $tag smoke bench bench:relax_ds_sh bench:reblur_diffuse bench:sigma_shadow bench_nosky trace trace:relax_ds_sh
# the driver's protocol (frames 5..24 after the restart frame) beside the default 32 + 64 line
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2> gpurun_out/${tag}_reblur_ds_driver_protocol_bench.err
# counters of all three workloads on this build (FETCH / WRITE / SQ sets; never combined with other trace domains)
SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
PMC_SETS="$SETS" bash tools/pmc_run.sh ${tag}_reblur_ds --workload reblur_ds --steps 8 --warmup 4 --no-cpu-baseline --no-graph > /dev/null
PMC_SETS="$SETS" bash tools/pmc_run.sh ${tag}_reblur_ds_nosky --workload reblur_ds --no-sky --steps 8 --warmup 4 --no-cpu-baseline --no-graph > /dev/null
PMC_SETS="$SETS" bash tools/pmc_run.sh ${tag}_relax_ds_sh --workload relax_ds_sh --steps 8 --warmup 4 --no-cpu-baseline --no-graph > /dev/null
# issue floors: the product and the L1-resident build on the uniform scene
for w in reblur_ds relax_ds_sh; do
  for lib in product l1; do
    env=""; [[ $lib == l1 ]] && env="NRD_HIP_LIBRARY=raytracingdenoiser_amd/lib/variants/l1/libNRD_hip.so"
    rm -rf /tmp/prof_floor
    env $env timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_floor -o trace -- python bench.py --workload $w --uniform --steps 20 --warmup 8 --no-cpu-baseline --no-graph > gpurun_out/${tag}_${w}_uniform_${lib}_bench.json 2> gpurun_out/${tag}_${w}_uniform_${lib}.err
    python tools/rocprof_summary.py $(find /tmp/prof_floor -name "*.db" | head -1) > gpurun_out/${tag}_${w}_uniform_${lib}_kernel_stats.txt 2>&1
  done
done
timeout 600 python tools/model_scaling.py --workload reblur_ds > gpurun_out/${tag}_scaling_model_reblur_ds.json 2> gpurun_out/${tag}_scaling_model_reblur_ds.err
bash tools/gpu_session.sh $tag pytest
