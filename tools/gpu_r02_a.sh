#!/bin/bash
# Round-2 GPU session A: VALU price list, the whole GPU suite (both numerics builds), bench lines of the five BASELINE configs.
tag=${1:-r02_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
rm -f gpurun_out/parity_report.jsonl
timeout 120 tools/build/valu_bench > gpurun_out/${tag}_valu_bench.txt 2>&1; cat gpurun_out/${tag}_valu_bench.txt
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_full_parity.py > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_gpu.log
tail -15 gpurun_out/${tag}_pytest_gpu.log
timeout 2400 python -m pytest tests/test_full_parity.py -m gpu -q -s > gpurun_out/${tag}_pytest_full_parity.log 2>&1; echo "pytest exit $?" >> gpurun_out/${tag}_pytest_full_parity.log
grep -E "fast_vs|exact_vs|passed|failed|Error|error" gpurun_out/${tag}_pytest_full_parity.log | cut -c1-400 | tail -40
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.json | cut -c1-1500; tail -3 gpurun_out/${tag}_bench.err
timeout 600 python bench.py --numerics exact --no-cpu-baseline > gpurun_out/${tag}_bench_exact.json 2>> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench_exact.json | cut -c1-600
timeout 600 python bench.py --no-graph --no-cpu-baseline > gpurun_out/${tag}_bench_eager.json 2>> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench_eager.json | cut -c1-400
for wl in relax_ds_sh reblur_diffuse sigma_shadow; do
  timeout 900 python bench.py --workload $wl --cpu-frames 3 > gpurun_out/${tag}_${wl}_bench.json 2>> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_${wl}_bench.json | cut -c1-1200
done
tail -5 gpurun_out/${tag}_bench.err
