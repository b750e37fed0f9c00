"""Writes profiles/<tag>_ref_parity.jsonl: the per-pass comparison of the oracle with the reference's own shader text (oracle/_ref) for every denoiser, one
JSON line per (arithmetic, denoiser, pass, output plane). Two arithmetics of the hand-written oracle are held against the reference text, every pass on
identical inputs (tests/ref_parity.py):
  strict   liboracle_strict.so -- no contraction, true divisions: the restatement itself
  device   liboracle.so in device mode -- the arithmetic contract of the HIP library (the GPU's results are bit-identical to this oracle: tests -m gpu),
           i.e. "what the library computes" against "what the reference's shader says", one pass at a time, no recurrence
usage: python tools/ref_report.py TAG [denoiser ...]     (CPU only; needs oracle/_ref, i.e. /root/reference at build time)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity  # noqa: E402
import ref_parity  # noqa: E402


def main():
    tag = sys.argv[1]
    names = sys.argv[2:] or list(parity.DENOISERS)
    out = os.path.join(ROOT, "profiles", tag + "_ref_parity.jsonl")
    summary = []
    with open(out, "w") as fp:
        for mode in ("strict", "device"):
            for name in names:
                stats = ref_parity.run_per_pass(name, frames=3, sensitivity=True, strict=mode == "strict", ieee=mode == "strict")
                rows = stats.table()
                for r in rows:
                    fp.write(json.dumps(dict(r, arithmetic=mode, denoiser=name, frames=3, size=[192, 128])) + "\n")
                n = sum(r["texel_values"] for r in rows)
                worst = min(rows, key=lambda r: (r["within_1e-3_frac"], r["bit_exact_frac"]))  # the plane furthest from the north-star's tolerance, per component (both metrics, worst plane named)
                line = "%-7s %-40s outputs %3d  values %9d  bit-exact %.5f  min ok %.6f  min within 1e-3 %.6f  min within 1e-3 (vector) %.6f  outliers %d (sensitive %d)  worst plane: %s %s %s (bit-exact %.4f, per component %.6f, vector %.6f, max %.3g)" % (
                    mode, name, len(rows), n, sum(r["bit_exact_frac"] * r["texel_values"] for r in rows) / n, min(r["within_tol_frac"] for r in rows), min(r["within_1e-3_frac"] for r in rows),
                    min(r["within_1e-3_vec_frac"] for r in rows), sum(r["outliers"] for r in rows), sum(r["outliers_sensitive"] for r in rows),
                    worst["pass"], worst["output"], worst["format"], worst["bit_exact_frac"], worst["within_1e-3_frac"], worst["within_1e-3_vec_frac"], worst["max_err"])
                print(line, flush=True)
                summary.append(line)
    with open(os.path.join(ROOT, "profiles", tag + "_ref_parity_summary.txt"), "w") as fp:
        fp.write(__doc__.split("usage:")[0] + "\n3 frames at 192x128 (restart frame + 2 frames under camera motion), every pass of every frame on identical inputs.\n"
                 "ok = within 1e-5 relative or one unit in the last place of the stored format; vector = colour / direction channels relative to the texel's largest channel.\n\n" + "\n".join(summary) + "\n")


if __name__ == "__main__":
    main()
