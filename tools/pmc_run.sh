#!/bin/bash
# Collects hardware counters for a bench.py workload, one rocprofv3 pass per counter set (PMC slots: SQ 8, TCC 4; FETCH_SIZE
# and WRITE_SIZE do not fit in one pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots"), and leaves only text summaries.
# usage: tools/pmc_run.sh <out-prefix> [bench.py args...]
prefix=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
DEFAULT_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
IFS=';' read -ra SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_pass
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_pass -- python $R/bench.py "$@" --no-cpu-baseline > /tmp/pmc_pass.log 2>&1 || tail -5 /tmp/pmc_pass.log
  db=$(find /tmp/pmc_pass -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db > $R/gpurun_out/${prefix}_pmc${i}.txt 2>&1
done
cat $R/gpurun_out/${prefix}_pmc*.txt | cut -c1-260
