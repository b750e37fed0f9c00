#!/bin/bash
# Round-2 GPU session M: smoke() and the fast-build parity statistics with the final flags (reassociation, pixel-space taps, batched TA)
tag=${1:-r02_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "exit $?" >> gpurun_out/${tag}_smoke.log; tail -5 gpurun_out/${tag}_smoke.log
timeout 1500 python -m pytest tests/test_full_parity.py -m gpu -q -s -k "fast or denoises" > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
grep -E "passed|failed|Error" gpurun_out/${tag}_pytest_full_parity.log | tail -5
cp parity_report.jsonl gpurun_out/${tag}_parity_report.jsonl 2>/dev/null
