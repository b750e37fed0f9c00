#!/bin/bash
# Round-2 GPU session C: L1 / LDS gather-rate probe; RELAX regression (exact build) after the a-trous / pre-pass rewrite; A/B of stripes and TA occupancy;
# SQ counters of the fast build.
tag=${1:-r02_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 120 tools/build/gather_bench > gpurun_out/${tag}_gather_bench.txt 2>&1; cat gpurun_out/${tag}_gather_bench.txt
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TCC|TD|SQ|GRBM|SPI)_[A-Za-z0-9_]+" | sort -u > $R/gpurun_out/${tag}_counters.txt); wc -l gpurun_out/${tag}_counters.txt
timeout 900 python -m pytest tests/test_relax.py tests/test_reblur.py tests/test_executor.py -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -4 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -q -k "exact_build_bit_exact and (RELAX or REBLUR_DIFFUSE_SPECULAR)" > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
tail -3 gpurun_out/${tag}_pytest_full_parity.log
B="python bench.py --no-cpu-baseline --steps 48 --warmup 16"
timeout 300 $B > gpurun_out/${tag}_bench_fast.json 2>> gpurun_out/${tag}_bench.err
for s in 0 2 5; do NRD_HIP_XCD_BANDS=$s timeout 300 $B > gpurun_out/${tag}_bench_fast_bands$s.json 2>> gpurun_out/${tag}_bench.err; done
NRD_HIP_TA_WAVES=3 timeout 300 $B > gpurun_out/${tag}_bench_fast_tawaves3.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast.json 2>> gpurun_out/${tag}_bench.err
for s in 0 3 5; do NRD_HIP_XCD_BANDS=$s timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast_bands$s.json 2>> gpurun_out/${tag}_bench.err; done
timeout 300 $B --workload relax_ds_sh --numerics exact > gpurun_out/${tag}_relax_bench_exact.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_c*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs',''),v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${tag}_bench.err
SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PMC_SETS="$SETS" bash tools/pmc_run.sh ${tag}_reblur_ds_sq --workload reblur_ds --steps 8 --warmup 4 > /dev/null 2>&1
PMC_SETS="$SETS" bash tools/pmc_run.sh ${tag}_relax_sq --workload relax_ds_sh --steps 6 --warmup 3 > /dev/null 2>&1
cat gpurun_out/${tag}_reblur_ds_sq_pmc1.txt | cut -c1-250 | head -9; cat gpurun_out/${tag}_reblur_ds_sq_pmc2.txt | cut -c1-250 | head -9
cat gpurun_out/${tag}_relax_sq_pmc1.txt | cut -c1-250 | head -9; cat gpurun_out/${tag}_relax_sq_pmc2.txt | cut -c1-250 | head -9
