#!/bin/bash
# round-4 session N: the final kernels once more after the RELAX TemporalAccumulation reorder -- full-size / long-run parity, bench lines, RELAX kernel trace
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_n; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag smoke bench:relax_ds_sh bench trace:relax_ds_sh
timeout 2400 python -m pytest tests/test_full_parity.py tests/test_full_size.py tests/test_motion_rows.py -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/${tag}_pytest_gpu.log; tail -4 gpurun_out/${tag}_pytest_gpu.log
