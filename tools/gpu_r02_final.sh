#!/bin/bash
# Round-2 final GPU session: smoke, the complete GPU test-suite, the bench lines (default = the headline with cpu_baseline / parity / exact_build),
# rocprofv3 kernel traces and the PMC passes behind profiles/pmc_traffic.json. Results in gpurun_out/r02_final_*.
tag=${1:-r02_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "exit $?" >> gpurun_out/${tag}_smoke.log; tail -2 gpurun_out/${tag}_smoke.log
# ---- benches first (short), then profiles, then the long test run
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json | cut -c1-300
B="python bench.py --no-cpu-baseline --no-parity --no-exact-leg --steps 48 --warmup 16"
timeout 200 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_ds_sh_4k_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 200 $B --workload reblur_diffuse > gpurun_out/${tag}_reblur_diffuse_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 200 $B --workload sigma_shadow > gpurun_out/${tag}_sigma_shadow_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 200 $B --no-graph > gpurun_out/${tag}_bench_eager.json 2>> gpurun_out/${tag}_bench.err
timeout 200 $B --workload relax_ds_sh --numerics exact > gpurun_out/${tag}_relax_ds_sh_4k_bench_exact.json 2>> gpurun_out/${tag}_bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print('%-44s %8.1f %.4f  '%(f.split('/')[-1][:-5], d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs','')[:8],v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
# ---- kernel traces
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-exact-leg --no-graph --steps 24 --warmup 8 > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
 db=$(find /tmp/kt -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $R/gpurun_out/${tag}_reblur_ds_1440p_kernel_stats.txt 2>&1
 rm -rf /tmp/kt && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity --no-exact-leg --no-graph --steps 12 --warmup 4 > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
 db=$(find /tmp/kt -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $R/gpurun_out/${tag}_relax_ds_sh_4k_kernel_stats.txt 2>&1)
head -12 gpurun_out/${tag}_reblur_ds_1440p_kernel_stats.txt | cut -c1-170
# ---- counters (separate passes; --no-graph so that kernels are attributed individually)
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" bash tools/pmc_run.sh ${tag}_reblur_ds --workload reblur_ds --no-parity --no-exact-leg --no-graph --steps 8 --warmup 4 > /dev/null 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" bash tools/pmc_run.sh ${tag}_relax_ds_sh --workload relax_ds_sh --no-parity --no-exact-leg --no-graph --steps 6 --warmup 3 > /dev/null 2>&1
head -9 gpurun_out/${tag}_reblur_ds_pmc1.txt | cut -c1-200
# ---- A/B: tap positions in the reference's operation order (fast build variant): throughput and the 1440p parity statistics
V=$R/raytracingdenoiser_amd/lib/variants/taps0/libNRD_hip.so
if [ -f $V ]; then
  NRD_HIP_FAST_LIBRARY=$V timeout 200 $B > gpurun_out/${tag}_taps0_bench.json 2>> gpurun_out/${tag}_bench.err
  NRD_HIP_FAST_LIBRARY=$V timeout 400 python -m pytest tests/test_full_parity.py -m gpu -q -s -k "fast_build_within_tolerance_at_baseline and REBLUR_DIFFUSE_SPECULAR" > gpurun_out/${tag}_taps0_parity.log 2>&1
  grep -E "fast_vs|passed|failed" gpurun_out/${tag}_taps0_parity.log | cut -c1-250
  python -c "import json;d=json.loads(open('gpurun_out/${tag}_taps0_bench.json').read().strip().split(chr(10))[-1]);print('taps0', d['value'], d['ms_per_step'])"
fi
# ---- the GPU suite (the long oracle runs of test_full_parity.py ran in sessions I and M; here: the exact build at the headline size once more)
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_full_parity.py > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -6 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -q -k "exact_build_bit_exact_at_baseline and REBLUR_DIFFUSE_SPECULAR" > gpurun_out/${tag}_pytest_full_parity_exact.log 2>&1; tail -2 gpurun_out/${tag}_pytest_full_parity_exact.log
