#!/bin/bash
# round-4 session H: decode || classify in graph mode
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_h; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_bench.json 2> gpurun_out/${tag}_reblur_ds_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-graph > gpurun_out/${tag}_reblur_ds_eager_bench.json 2>> gpurun_out/${tag}_reblur_ds_bench.err
timeout 300 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_ds_sh_bench.json 2>> gpurun_out/${tag}_reblur_ds_bench.err
python - <<'PY'
import json
for f in ("r04_h_reblur_ds_bench.json","r04_h_reblur_ds_eager_bench.json","r04_h_relax_ds_sh_bench.json"):
    j=json.loads(open("gpurun_out/"+f).read().strip().split("\n")[-1]); print(f, j["ms_per_step"], j["launch"], {k.split("_")[-1].replace(".cs",""):v["avg_ms"] for k,v in j.get("passes",{}).items()})
PY
timeout 900 python -m pytest tests/test_executor.py tests/test_sigma.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; tail -3 gpurun_out/${tag}_pytest.log
