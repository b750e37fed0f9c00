export TMPDIR=/tmp
bash tools/gpu_session.sh r03_j smoke bench bench:relax_ds_sh bench:reblur_diffuse trace trace:relax_ds_sh 2>&1 | grep -v "^\s*$" | cut -c1-200 | tail -40
NRD_HIP_TA_WINDOW=0 python bench.py --workload relax_ds_sh --no-cpu-baseline > gpurun_out/r03_j_relax_ds_sh_nowindow_bench.json 2>/dev/null; tail -1 gpurun_out/r03_j_relax_ds_sh_nowindow_bench.json | cut -c1-200
bash tools/gpu_session.sh r03_j pytest 2>&1 | tail -5
