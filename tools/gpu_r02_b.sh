#!/bin/bash
# Round-2 GPU session B: extended VALU price list; exactness regression of the restructured kernels (exact build); A/B of the XCD bands and the full-rect taps;
# PMC traffic with and without the bands.
tag=${1:-r02_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 120 tools/build/valu_bench > gpurun_out/${tag}_valu_bench.txt 2>&1; tail -16 gpurun_out/${tag}_valu_bench.txt
timeout 1500 python -m pytest tests/test_reblur.py tests/test_relax.py tests/test_sigma.py tests/test_sharding.py tests/test_executor.py tests/test_dynamic_resolution.py tests/test_full_size.py -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -6 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python -m pytest tests/test_full_parity.py -m gpu -q -s -k "exact_build_bit_exact or denoises" > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
grep -E "RMSE|passed|failed|Error" gpurun_out/${tag}_pytest_full_parity.log | cut -c1-300 | tail
B="python bench.py --no-cpu-baseline --steps 48 --warmup 16"
for num in fast exact; do
  timeout 300 $B --numerics $num > gpurun_out/${tag}_bench_${num}.json 2>> gpurun_out/${tag}_bench.err
  NRD_HIP_XCD_BANDS=0 timeout 300 $B --numerics $num > gpurun_out/${tag}_bench_${num}_nobands.json 2>> gpurun_out/${tag}_bench.err
  NRD_HIP_GENERIC_TAPS=1 timeout 300 $B --numerics $num > gpurun_out/${tag}_bench_${num}_generictaps.json 2>> gpurun_out/${tag}_bench.err
done
timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast.json 2>> gpurun_out/${tag}_bench.err
NRD_HIP_XCD_BANDS=0 timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast_nobands.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_b*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs',''),v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${tag}_bench.err
PMC_SETS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${tag}_reblur_ds_bands --workload reblur_ds --steps 8 --warmup 4 > /dev/null 2>&1
NRD_HIP_XCD_BANDS=0 PMC_SETS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${tag}_reblur_ds_nobands --workload reblur_ds --steps 8 --warmup 4 > /dev/null 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${tag}_relax_bands --workload relax_ds_sh --steps 6 --warmup 3 > /dev/null 2>&1
NRD_HIP_XCD_BANDS=0 PMC_SETS="FETCH_SIZE;WRITE_SIZE" bash tools/pmc_run.sh ${tag}_relax_nobands --workload relax_ds_sh --steps 6 --warmup 3 > /dev/null 2>&1
head -30 gpurun_out/${tag}_reblur_ds_bands_pmc1.txt | cut -c1-200; head -30 gpurun_out/${tag}_reblur_ds_nobands_pmc1.txt | cut -c1-200
