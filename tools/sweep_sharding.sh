#!/bin/bash
# Verification sweep of the multi-GPU schemes (run on the GPU box through gpurun): every denoiser of the library and a set of settings variants, cut into 2 / 3 / 4 row strips at
# 1280 x 720, every rank in turn as a virtual rank against a full-frame run in lock-step (tools/model_scaling.py): halo exchange with re-cut strips, halo exchange with uniform
# strips, redundant halos + all-gather. One line per case: the ranks whose owned rows differ from the full-frame run (must be none) and the timed frames that ran unsharded.
#   usage: bash tools/sweep_sharding.sh [TAG]     -> gpurun_out/<TAG>_sweep_all.log
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-r06}; mkdir -p gpurun_out; out=gpurun_out/${tag}_sweep_all.log; : > $out
run() { # label, args...
  local label=$1; shift
  timeout 600 python tools/model_scaling.py --size 1280x720 --worlds 2,3,4 --frames 10 --warmup 6 "$@" > /tmp/sweep_case.json 2> /tmp/sweep_case.err
  python - "$label" >> $out <<'PY'
import json, sys
label = sys.argv[1]
try:
    m = json.load(open("/tmp/sweep_case.json"))
    bad = [(r["world"], r["rank"]) for r in m["ranks"] if r["owned_rows_bit_identical_to_full_frame_run"] is False]
    uns = sum(1 for r in m["ranks"] if r.get("timed_frames_run_unsharded"))
    print("%-110s ranks %2d  differ: %s  ranks with unsharded timed frames: %d" % (label, len(m["ranks"]), bad or "none", uns))
except Exception as e:  # noqa: BLE001
    print("%-110s FAILED: %s | %s" % (label, e, open("/tmp/sweep_case.err").read()[-600:].replace("\n", " / ")))
PY
}
DENOISERS="REBLUR_DIFFUSE REBLUR_SPECULAR REBLUR_DIFFUSE_SPECULAR REBLUR_DIFFUSE_SH REBLUR_SPECULAR_SH REBLUR_DIFFUSE_SPECULAR_SH REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION REBLUR_DIFFUSE_OCCLUSION REBLUR_SPECULAR_OCCLUSION REBLUR_DIFFUSE_SPECULAR_OCCLUSION SIGMA_SHADOW SIGMA_SHADOW_TRANSLUCENCY RELAX_DIFFUSE RELAX_DIFFUSE_SH RELAX_SPECULAR RELAX_SPECULAR_SH RELAX_DIFFUSE_SPECULAR RELAX_DIFFUSE_SPECULAR_SH"
for d in $DENOISERS; do
  run "$d halo balanced" --denoiser $d
  run "$d halo uniform" --denoiser $d --balance 0
  run "$d allgather" --denoiser $d --scheme allgather
done
for scheme in halo allgather; do
  for d in REBLUR_DIFFUSE_SPECULAR REBLUR_DIFFUSE_SPECULAR_OCCLUSION RELAX_DIFFUSE_SPECULAR; do
    for s in '{"hitDistanceReconstructionMode": 1}' '{"hitDistanceReconstructionMode": 2}' '{"enableAntiFirefly": true}' '{"checkerboardMode": 1}' '{"checkerboardMode": 2}'; do
      run "$d $scheme $s" --denoiser $d --scheme $scheme --settings "$s"
    done
  done
  run "REBLUR_DIFFUSE_SPECULAR $scheme performance mode" --denoiser REBLUR_DIFFUSE_SPECULAR --scheme $scheme --settings '{"enablePerformanceMode": true}'
  run "REBLUR_DIFFUSE_SPECULAR $scheme maxBlurRadius 60" --denoiser REBLUR_DIFFUSE_SPECULAR --scheme $scheme --settings '{"maxBlurRadius": 60.0}'
  run "REBLUR_DIFFUSE_SPECULAR $scheme no pre-pass" --denoiser REBLUR_DIFFUSE_SPECULAR --scheme $scheme --settings '{"diffusePrepassBlurRadius": 0.0, "specularPrepassBlurRadius": 0.0}'
  run "RELAX_DIFFUSE_SPECULAR_SH $scheme 8 a-trous iterations" --denoiser RELAX_DIFFUSE_SPECULAR_SH --scheme $scheme --settings '{"atrousIterationNum": 8}'
  run "RELAX_DIFFUSE_SPECULAR $scheme 2 a-trous iterations, no pre-pass" --denoiser RELAX_DIFFUSE_SPECULAR --scheme $scheme --settings '{"atrousIterationNum": 2, "diffusePrepassBlurRadius": 0.0, "specularPrepassBlurRadius": 0.0}'
  run "SIGMA_SHADOW $scheme no stabilisation" --denoiser SIGMA_SHADOW --scheme $scheme --settings '{"maxStabilizedFrameNum": 0}'
done
cat $out
grep -c "differ: none" $out; grep -vc "differ: none" $out
