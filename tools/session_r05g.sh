#!/bin/bash
# round-5 session G: SQ counters (set 3: executed VALU / waves / waits; set 5: LDS instructions, bank conflicts, VMEM) for the two big workloads at HEAD
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_g; mkdir -p gpurun_out
export PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
bash tools/pmc_run.sh ${tag}_reblur_ds --workload reblur_ds --steps 8 --warmup 4 --no-parity > /dev/null 2>&1
bash tools/pmc_run.sh ${tag}_relax_ds_sh --workload relax_ds_sh --steps 8 --warmup 4 --no-parity > /dev/null 2>&1
for f in gpurun_out/${tag}_*_pmc*.txt; do echo "== $f"; cut -c1-84,85-260 $f | head -14; done
