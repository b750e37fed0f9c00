#!/bin/bash
# End-of-round GPU session: the complete GPU suite, the bench lines of every workload, kernel traces, counters and the multi-GPU model -- all on ONE build.
# usage (through gpurun): bash tools/final_session.sh TAG
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=$1
bash tools/gpu_session.sh $tag smoke bench bench:relax_ds_sh bench:reblur_diffuse bench:sigma_shadow bench_nosky trace trace:relax_ds_sh
bash tools/gpu_session.sh $tag pmc
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" bash tools/pmc_run.sh ${tag}_reblur_ds_nosky --workload reblur_ds --no-sky --steps 8 --warmup 4 --no-cpu-baseline --no-graph > /dev/null
for w in reblur_ds relax_ds_sh; do
  timeout 600 python tools/model_scaling.py --workload $w > gpurun_out/${tag}_scaling_model_${w}.json 2> gpurun_out/${tag}_scaling_model_${w}.err
done
timeout 600 python tools/model_scaling.py --workload reblur_ds --no-sky --balance 0 > gpurun_out/${tag}_scaling_model_reblur_ds_nosky_uniform.json 2>> gpurun_out/${tag}_scaling_model_reblur_ds.err
bash tools/gpu_session.sh $tag pytest
