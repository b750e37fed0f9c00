#!/bin/bash
# One GPU-box session: GPU test suite, bench lines, kernel-trace summaries and PMC passes for the two headline workloads.
# usage: tools/measure_round.sh <tag, e.g. r01_h>   (text-only results land in gpurun_out/)
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json | cut -c1-600
timeout 600 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_ds_sh_4k_bench.json 2>> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_relax_ds_sh_4k_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for wl in reblur_ds relax_ds_sh; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --workload $wl --steps 24 --warmup 8 --no-cpu-baseline > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
  db=$(find /tmp/kt -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $R/gpurun_out/${tag}_${wl}_kernel_stats.txt 2>&1
  head -12 $R/gpurun_out/${tag}_${wl}_kernel_stats.txt | cut -c1-200
done
cd $R
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" bash tools/pmc_run.sh ${tag}_reblur_ds --workload reblur_ds --steps 8 --warmup 4 > /dev/null 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" bash tools/pmc_run.sh ${tag}_relax_ds_sh --workload relax_ds_sh --steps 8 --warmup 4 > /dev/null 2>&1
ls -la gpurun_out | tail -20
