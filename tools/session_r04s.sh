#!/bin/bash
# round-4 session S (last): the final library -- headline bench line with parity, RELAX line, kernel traces, full-size parity, sharding on the GPU
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_s; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag bench trace
timeout 120 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_ds_sh_bench.json 2> gpurun_out/${tag}_relax_ds_sh_bench.err
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_driver_protocol_bench.json 2> /dev/null
timeout 120 python bench.py --no-sky --no-cpu-baseline > gpurun_out/${tag}_reblur_ds_nosky_bench.json 2> /dev/null
bash tools/gpu_session.sh $tag trace:relax_ds_sh
timeout 330 python -m pytest tests/test_full_parity.py -m gpu -x -q -k "baseline_size" > gpurun_out/${tag}_pytest_full_size.log 2>&1; tail -2 gpurun_out/${tag}_pytest_full_size.log
timeout 200 python -m pytest tests/test_sharding.py tests/test_sharded_cpp.py -m gpu -x -q > gpurun_out/${tag}_pytest_sharding.log 2>&1; tail -2 gpurun_out/${tag}_pytest_sharding.log
