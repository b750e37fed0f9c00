import sys
sys.path.insert(0, "tests")
import parity
sizes = [(144, 88), (108, 66), (126, 77), (144, 88), (90, 55)]
for name in ["SIGMA_SHADOW", "RELAX_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR", "REBLUR_DIFFUSE_SPECULAR_OCCLUSION", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW_TRANSLUCENCY", "REBLUR_DIFFUSE_SH"]:
    try:
        w = parity.run_parity(name, frames=6, verbose=True, resource=(160, 96), rect_sizes=sizes)
        print("==== %s worst %.3g" % (name, w), flush=True)
    except Exception as e:
        print("==== %s FAILED: %s" % (name, str(e)[:300]), flush=True)
