#!/bin/bash
# Round-2 GPU session F: validation overlays, rectOrigin, executor tests after the arena / validation changes
tag=${1:-r02_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_validation.py tests/test_dynamic_resolution.py tests/test_executor.py -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -40 gpurun_out/${tag}_pytest_gpu.log
