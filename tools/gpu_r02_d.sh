#!/bin/bash
# Round-2 GPU session D: LDS-staged Blur / PostBlur taps (correctness in the exact build, A/B timing), FR taps with 16 B + 4 B guides.
tag=${1:-r02_d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_reblur.py tests/test_relax.py tests/test_sharding.py tests/test_dynamic_resolution.py -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -4 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -q -k "exact_build_bit_exact and (REBLUR or RELAX_DIFFUSE_SPECULAR_SH_3840)" > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
tail -3 gpurun_out/${tag}_pytest_full_parity.log
B="python bench.py --no-cpu-baseline --steps 48 --warmup 16"
timeout 300 $B > gpurun_out/${tag}_bench_fast.json 2>> gpurun_out/${tag}_bench.err
NRD_HIP_LDS_TAPS=0 timeout 300 $B > gpurun_out/${tag}_bench_fast_nolds.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --numerics exact > gpurun_out/${tag}_bench_exact.json 2>> gpurun_out/${tag}_bench.err
NRD_HIP_LDS_TAPS=0 timeout 300 $B --numerics exact > gpurun_out/${tag}_bench_exact_nolds.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload reblur_diffuse > gpurun_out/${tag}_reblur_diffuse_bench_fast.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_d*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs',''),v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${tag}_bench.err
