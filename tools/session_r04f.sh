#!/bin/bash
# round-4 session F: guide prefetch in front of the LDS fills (TS, HistoryFix, TA window kernel, RELAX TA / HistoryClamping), step-8 bands
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_f; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag smoke bench trace bench:relax_ds_sh trace:relax_ds_sh
python - <<'PY'
import json
for f in ("r04_f_reblur_ds_bench.json","r04_f_relax_ds_sh_bench.json"):
    j=json.loads(open("gpurun_out/"+f).read().strip().split("\n")[-1]); print(f, j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):v["avg_ms"] for k,v in j.get("passes",{}).items()})
PY
timeout 900 python -m pytest tests/test_reblur.py tests/test_executor.py -m gpu -x -q > gpurun_out/${tag}_pytest_reblur.log 2>&1; tail -3 gpurun_out/${tag}_pytest_reblur.log
