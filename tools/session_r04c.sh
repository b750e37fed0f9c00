#!/bin/bash
# round-4 session C: per-kernel floors on the uniform scene: product vs L1-resident build vs arithmetic-only build (kernel traces)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=r04_c; mkdir -p gpurun_out
V=raytracingdenoiser_amd/lib/variants
for w in reblur_ds relax_ds_sh; do
  for lib in product l1 alu; do
    rm -rf /tmp/prof_$w
    if [[ $lib == product ]]; then unset NRD_HIP_LIBRARY; else export NRD_HIP_LIBRARY=$V/$lib/libNRD_hip.so; fi
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o trace -- python bench.py --workload $w --uniform --steps 20 --warmup 8 --no-cpu-baseline --no-parity --no-graph > gpurun_out/${tag}_${w}_uniform_${lib}_trace_bench.json 2> gpurun_out/${tag}_${w}_uniform_${lib}_trace.err
    python tools/rocprof_summary.py $(find /tmp/prof_$w -name "*.db" | head -1) > gpurun_out/${tag}_${w}_uniform_${lib}_kernel_stats.txt 2>&1
  done
done
unset NRD_HIP_LIBRARY
python - <<'PY'
import re,glob
for w in ("reblur_ds","relax_ds_sh"):
    t={}
    for lib in ("product","l1","alu"):
        for line in open("gpurun_out/r04_c_%s_uniform_%s_kernel_stats.txt"%(w,lib)):
            m=re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%",line)
            if m:
                name=re.sub(r"nrdhip::|\(anonymous namespace\)::","",m.group(1)); name=re.sub(r"\(.*","",name)[:70]
                t.setdefault(name,{})[lib]=float(m.group(3))
    print(w)
    for k,v in sorted(t.items(),key=lambda kv:-kv[1].get("product",0)):
        if "product" in v: print("  %-72s product %8.1f  l1 %8.1f  alu %8.1f   ratio l1 %.2f alu %.2f"%(k,v["product"],v.get("l1",0),v.get("alu",0),v["product"]/max(v.get("l1",1e9),1e-9),v["product"]/max(v.get("alu",1e9),1e-9)))
PY
