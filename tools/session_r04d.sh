#!/bin/bash
# round-4 session D: banded LDS a-trous for steps 8 / 16 -- A/B against the global gathers, kernel trace, RELAX parity on the GPU
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_d; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag bench:relax_ds_sh trace:relax_ds_sh
NRD_HIP_ATROUS_BANDS=0 timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_ds_sh_nobands_bench.json 2> gpurun_out/${tag}_relax_ds_sh_nobands.err
python - <<'PY'
import json
for f in ("r04_d_relax_ds_sh_bench.json","r04_d_relax_ds_sh_nobands_bench.json"):
    j=json.loads(open("gpurun_out/"+f).read().strip().split("\n")[-1]); print(f, j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):v["avg_ms"] for k,v in j.get("passes",{}).items()})
PY
timeout 1500 python -m pytest tests/test_relax.py -m gpu -x -q > gpurun_out/${tag}_pytest_relax.log 2>&1; tail -3 gpurun_out/${tag}_pytest_relax.log
