#!/bin/bash
# Round-2 GPU session E: row-vector footprint loads in the temporal passes, (normal, viewZ) guide plane for the REBLUR taps: correctness (exact build) + A/B.
tag=${1:-r02_e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_reblur.py tests/test_relax.py tests/test_dynamic_resolution.py tests/test_executor.py -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -4 gpurun_out/${tag}_pytest_gpu.log
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -q -k "exact_build_bit_exact and (REBLUR_DIFFUSE_SPECULAR or RELAX_DIFFUSE_SPECULAR_SH)" > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
tail -3 gpurun_out/${tag}_pytest_full_parity.log
B="python bench.py --no-cpu-baseline --steps 48 --warmup 16"
timeout 300 $B > gpurun_out/${tag}_bench_fast.json 2>> gpurun_out/${tag}_bench.err
NRD_HIP_GUIDE_NZ=0 timeout 300 $B > gpurun_out/${tag}_bench_fast_noguidenz.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --numerics exact > gpurun_out/${tag}_bench_exact.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_relax_bench_fast.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload sigma_shadow > gpurun_out/${tag}_sigma_bench_fast.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_e*bench*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs',''),v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${tag}_bench.err
