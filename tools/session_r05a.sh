#!/bin/bash
# round-5 session A (prepared at the end of round 4, when no GPU minutes were left): the complete GPU suite on the final library of round 4 (the strand-thickness fix and SIGMA's
# tile-classification grid were verified in emulation only, profiles/r04_final_emulated_gpu_suite.log), the bench lines, and what native v_min / v_max would buy (DESIGN.md section 8 item 1:
# priced statically at -1 % TA, -4 % spatial passes, -6 % TS; the variant is NOT bit-identical to the oracle -- timing only).
#   before: python tools/build_variant.py minmax -DNRD_NATIVE_MINMAX=1
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r05_a; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
bash tools/gpu_session.sh $tag pytest smoke bench bench:relax_ds_sh
for i in 1 2; do
  for w in reblur_ds relax_ds_sh; do
    NRD_HIP_LIBRARY=$V/minmax/libNRD_hip.so timeout 60 python bench.py --workload $w --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_minmax${i}_bench.json 2>> gpurun_out/${tag}_bench.err
    timeout 60 python bench.py --workload $w --no-cpu-baseline --no-parity > gpurun_out/${tag}_${w}_product${i}_bench.json 2>> gpurun_out/${tag}_bench.err
  done
done
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2>> gpurun_out/${tag}_bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_a_*_bench.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):round(v["avg_ms"],4) for k,v in j.get("passes",{}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
