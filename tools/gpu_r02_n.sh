#!/bin/bash
# Round-2 GPU session N: the product build with the final default (tap positions in the reference's operation order): smoke, the headline bench line with
# cpu_baseline / parity / exact_build, the material-ID parity tests, the fast-build statistics at 1440p
tag=${1:-r02_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "exit $?" >> gpurun_out/${tag}_smoke.log; tail -3 gpurun_out/${tag}_smoke.log
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json | cut -c1-260
timeout 600 python -m pytest tests/test_reblur.py tests/test_relax.py -m gpu -q -k "material" > gpurun_out/${tag}_pytest_materials.log 2>&1; tail -3 gpurun_out/${tag}_pytest_materials.log
timeout 600 python -m pytest tests/test_full_parity.py -m gpu -q -s -k "fast_build_within_tolerance_at_baseline and REBLUR" > gpurun_out/${tag}_pytest_fast_parity.log 2>&1; grep -E "fast_vs|passed|failed" gpurun_out/${tag}_pytest_fast_parity.log | cut -c1-250
