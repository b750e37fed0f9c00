#!/bin/bash
# round-5 session B: (1) v_exp_f32 on negative arguments (tools/hw_exp_neg.hip: sign-magnitude hypothesis + the [-2, -1] deviation table), (2) the complete GPU suite under the new
# test configuration (passive OpenMP waits + 4 xdist workers) with durations, (3) the driver-protocol bench line at HEAD, (4) native v_min / v_max timing (not bit-identical: timing only)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_b; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
timeout 300 tools/build/hw_exp_neg gpurun_out > gpurun_out/${tag}_hw_exp_neg_report.txt 2>&1
python - <<'PY'
import zlib, os
p = "gpurun_out/hw_exp2neg.i8"
if os.path.exists(p):
    open(p + ".z", "wb").write(zlib.compress(open(p, "rb").read(), 9)); os.remove(p)
PY
cat gpurun_out/${tag}_hw_exp_neg_report.txt | cut -c1-300
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=30 ) > gpurun_out/${tag}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log; tail -45 gpurun_out/${tag}_pytest_gpu.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2>> gpurun_out/${tag}_bench.err
for i in 1 2; do
  NRD_HIP_LIBRARY=$V/minmax/libNRD_hip.so timeout 90 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_minmax${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  timeout 90 python bench.py --no-cpu-baseline --no-parity > gpurun_out/${tag}_reblur_ds_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
done
NRD_HIP_LIBRARY=$V/minmax/libNRD_hip.so timeout 90 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_minmax1_bench.json 2>> gpurun_out/${tag}_bench.err
timeout 90 python bench.py --workload relax_ds_sh --no-cpu-baseline --no-parity > gpurun_out/${tag}_relax_ds_sh_product1_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_b_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
