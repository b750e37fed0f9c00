#!/bin/bash
# Round-2 GPU session J: TS without the MV-modification branch (tests), reassociation flags and the "no material test" experiment (benches)
tag=${1:-r02_j}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_reblur.py -m gpu -q -x -k "mv or basecolor or matches_oracle" > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -3 gpurun_out/${tag}_pytest_gpu.log
bash tools/gpu_r02_g.sh $tag "$@"
