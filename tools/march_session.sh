#!/bin/bash
# (round 6 development) A/B of the marching a-trous kernel on the GPU box: bash tools/march_session.sh TAG [pmc]
cd "$(dirname "$0")/.." 2>/dev/null; export TMPDIR=/tmp
tag=$1
timeout 900 python -m pytest tests/test_relax.py -m gpu -x -q > gpurun_out/${tag}_pytest_relax.log 2>&1; tail -3 gpurun_out/${tag}_pytest_relax.log
trace() { local name=$1; shift; rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o trace -- python bench.py --workload relax_ds_sh --steps 20 --warmup 8 --no-cpu-baseline "$@" > gpurun_out/${tag}_${name}_trace_bench.json 2> gpurun_out/${tag}_${name}_trace.err
  python tools/rocprof_summary.py $(find /tmp/prof_t -name "*.db" | head -1) > gpurun_out/${tag}_${name}_kernel_stats.txt 2>&1; grep -i "atrous" gpurun_out/${tag}_${name}_kernel_stats.txt | cut -c1-150; tail -1 gpurun_out/${tag}_${name}_trace_bench.json | cut -c1-200; }
NRD_HIP_ATROUS_MARCH=0 trace off --no-parity
trace on
NRD_HIP_ATROUS_MARCH_SEG=4 trace seg4 --no-parity
NRD_HIP_ATROUS_MARCH_SEG=16 trace seg16 --no-parity
if [[ -n "$2" ]]; then
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" bash tools/pmc_run.sh ${tag}_march --workload relax_ds_sh --steps 6 --warmup 4 --no-parity | grep -i "kernel\|atrous" | cut -c1-400
fi
