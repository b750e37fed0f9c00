#!/bin/bash
# round-4 session A: LoadOrZero A/B, issue-floor experiment (L1-resident build on the uniform scene)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_a; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag smoke bench
V=raytracingdenoiser_amd/lib/variants
timeout 300 env NRD_HIP_LIBRARY=$V/legacy/libNRD_hip.so python bench.py --no-cpu-baseline > gpurun_out/${tag}_variant_legacy_bench.json 2> gpurun_out/${tag}_variant_legacy.err
for w in reblur_ds relax_ds_sh; do
  timeout 300 python bench.py --workload $w --uniform --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_uniform_bench.json 2> gpurun_out/${tag}_${w}_uniform.err
  timeout 300 env NRD_HIP_LIBRARY=$V/l1/libNRD_hip.so python bench.py --workload $w --uniform --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_uniform_l1_bench.json 2> gpurun_out/${tag}_${w}_uniform_l1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_a_*bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):v["avg_ms"] for k,v in j.get("passes",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
