#!/bin/bash
# Round-2 GPU session K: REBLUR temporal accumulation with batched requests (surface-motion batch, virtual-motion batch): parity of every REBLUR variant
# with the exact build, benches of the product build (now with per-file reassociation flags)
tag=${1:-r02_k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
B="python bench.py --no-cpu-baseline --no-parity --steps 48 --warmup 16"
timeout 300 $B > gpurun_out/${tag}_cur_reblur.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload reblur_diffuse > gpurun_out/${tag}_cur_reblur_diffuse.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_cur_relax.json 2>> gpurun_out/${tag}_bench.err
timeout 300 $B --numerics exact > gpurun_out/${tag}_exact_reblur.json 2>> gpurun_out/${tag}_bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${tag}_*_re*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print('%-34s %8.1f %.4f  '%(f.split('/')[-1][:-5], d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.split('_')[-1].replace('.cs','')[:8],v['avg_ms']) for k,v in d['passes'].items()))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/${tag}_bench.err
timeout 1200 python -m pytest tests/test_reblur.py tests/test_dynamic_resolution.py tests/test_executor.py -m gpu -q -x > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -3 gpurun_out/${tag}_pytest_gpu.log
