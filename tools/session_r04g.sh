#!/bin/bash
# round-4 session G: second batch of guide prefetches (centre signals in TS / HistoryFix, AtrousSmem, RELAX HistoryFix), raw-half denanify in the RELAX pre-pass
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_g; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag bench bench:relax_ds_sh
python - <<'PY'
import json
for f in ("r04_g_reblur_ds_bench.json","r04_g_relax_ds_sh_bench.json"):
    j=json.loads(open("gpurun_out/"+f).read().strip().split("\n")[-1]); print(f, j["ms_per_step"], {k.split("_")[-1].replace(".cs",""):v["avg_ms"] for k,v in j.get("passes",{}).items()}); print(json.dumps(j.get("parity"))[:900])
PY
