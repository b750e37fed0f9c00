#!/bin/bash
# A short GPU-box session for the end of a round: bench lines + kernel trace of the headline workload, then the GPU tests that are not plain
# "HIP matches oracle" parity cases (tools/measure_round.sh is the complete session). usage: bash tools/final_run.sh   (results in gpurun_out/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=r01_r
timeout 120 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --workload reblur_ds --steps 24 --warmup 8 --no-cpu-baseline > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
db=$(find /tmp/kt -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $R/gpurun_out/${tag}_reblur_ds_1440p_kernel_stats.txt 2>&1; head -4 $R/gpurun_out/${tag}_reblur_ds_1440p_kernel_stats.txt | cut -c1-160
cd $R
timeout 60 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/${tag}_relax_ds_sh_4k_bench.json 2>> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_relax_ds_sh_4k_bench.json | cut -c1-160
timeout 200 python -m pytest tests/test_dynamic_resolution.py tests/test_full_size.py tests/test_sharded_cpp.py tests/test_reference.py tests/test_edge_sizes.py tests/test_integration_cpp.py tests/test_reblur.py tests/test_relax.py tests/test_sigma.py -m gpu -x -q -k "not (test_hip_matches_oracle and not checkerboard and not motion and not mv_mod)" > gpurun_out/${tag}_pytest_gpu_rest.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu_rest.log
