#!/bin/bash
# Round-2 GPU session G: A/B of compiler-level variants of the fast build (tools/build_variant.py): scheduler strategies, occupancy hints,
# a-trous taps with viewZ instead of the world-position texel. Usage: gpu_r02_g.sh tag variant...
tag=${1:-r02_g}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
B="python bench.py --no-cpu-baseline --no-parity --steps 48 --warmup 16"
for v in "$@"; do
  lib=$R/raytracingdenoiser_amd/lib/variants/$v/libNRD_hip.so
  NRD_HIP_FAST_LIBRARY=$lib timeout 300 $B > gpurun_out/${tag}_${v}_reblur.json 2>> gpurun_out/${tag}_bench.err
  NRD_HIP_FAST_LIBRARY=$lib timeout 300 $B --workload relax_ds_sh > gpurun_out/${tag}_${v}_relax.json 2>> gpurun_out/${tag}_bench.err
done
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
