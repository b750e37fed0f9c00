#!/bin/bash
# round-4 session E: banded a-trous with the two-phase fill
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_e; mkdir -p gpurun_out
bash tools/gpu_session.sh $tag trace:relax_ds_sh
grep "RelaxAtrousKernel" gpurun_out/${tag}_relax_ds_sh_kernel_stats.txt | cut -c1-200
