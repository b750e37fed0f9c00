"""Per-pass comparison of the hand-written CPU oracle (oracle/*.cpp, IEEE mode) with the reference's own shaders compiled as C++ (oracle/_ref):
every dispatch of every frame runs through both ON IDENTICAL INPUTS (oracle.driver.ComparingExecutor), so a difference is the difference of one pass --
no recurrence, no accumulated drift. Used by tests/test_ref_parity.py and tools/ref_report.py."""
import numpy as np

import parity
from oracle import driver as oracle_driver
from raytracingdenoiser_amd import api

F = api.Format
FLOAT_FORMATS = (F.RGBA16_SFLOAT, F.R16_SFLOAT, F.R32_SFLOAT, F.RGBA32_SFLOAT)


def _raw(arr):
    a = np.ascontiguousarray(arr)
    return a.view(np.uint8).reshape(a.shape[0], -1)


def _split_packed(shader, fmt, a, b):
    """packed planes are compared field by field: [(suffix, kind, a_field, b_field)], kind in {"f16bits", "code", "exact"}"""
    if fmt == F.R32_UINT and shader.startswith("REBLUR_") and "TemporalAccumulation" in shader:  # PackData2, reference REBLUR_Common.hlsli:58-70
        ai, bi = a.astype(np.uint64), b.astype(np.uint64)
        return [(".occlusionBits", "exact", (ai & 0xFF).astype(np.float64), (bi & 0xFF).astype(np.float64)),
                (".virtualHistoryAmount", "code", ((ai >> 8) & 0xFF).astype(np.float64), ((bi >> 8) & 0xFF).astype(np.float64)),
                (".curvature", "f16bits", (ai >> 16).astype(np.float64), (bi >> 16).astype(np.float64))]
    return None


def embed_guides_at(frame, resource, origin):
    """a rect-sized generated frame inside resource-sized planes: guide inputs at `origin`, everything else (noisy inputs) at (0, 0) -- the layout the reference
    addresses with CommonSettings::rectOrigin (Common.hlsli:200-206 WithRectOrigin: guides only)"""
    import torch

    rw, rh = resource
    ox, oy = origin
    guides = ("mv", "normal_roughness", "viewz", "diff_confidence", "spec_confidence", "disocclusion_mix", "basecolor_metalness")
    out = {}
    for k, v in frame.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.dtype != torch.bool:
            big = torch.full([rh, rw] + list(v.shape[2:]), 33.0 if v.dtype.is_floating_point else 9, dtype=v.dtype, device=v.device)
            x0, y0 = (ox, oy) if k in guides else (0, 0)
            big[y0 : y0 + v.shape[0], x0 : x0 + v.shape[1]] = v
            v = big
        out[k] = v
    return out


class PassStats:
    """per (pass, output plane): texel counts by size of the difference between the two oracles"""

    def __init__(self, tol=1e-5):
        self.tol = tol
        self.rows = {}

    def add(self, shader, slot, fmt, width, mine, theirs, alts=()):
        a, b = parity.decode_plane(_raw(mine), fmt, width).astype(np.float64), parity.decode_plane(_raw(theirs), fmt, width).astype(np.float64)
        fields = _split_packed(shader, fmt, a, b)
        if fields:
            alt_fields = [_split_packed(shader, fmt, parity.decode_plane(_raw(alt), fmt, width).astype(np.float64), b) for alt in alts]
            for i, (suffix, kind, fa, fb) in enumerate(fields):
                self._add(shader, slot + suffix, fmt.name, kind, fa, fb, [af[i][2] for af in alt_fields])
            return
        kind = ("float" if fmt in (F.R32_SFLOAT, F.RGBA32_SFLOAT) else "f16" if fmt in (F.RGBA16_SFLOAT, F.R16_SFLOAT) else "exact" if fmt in (F.R16_UINT, F.R32_UINT, F.R8_UINT, F.R10_G10_B10_A2_UNORM)
                else "code16" if fmt in (F.R16_UNORM, F.RGBA16_SNORM) else "code")
        self._add(shader, slot, fmt.name, kind, a, b, [parity.decode_plane(_raw(alt), fmt, width).astype(np.float64) for alt in alts])

    def _add(self, shader, slot, fmt_name, kind, a, b, alts):
        if kind == "f16bits":  # fp16 bit patterns -> values
            a, b = a.astype(np.uint16).view(np.float16).astype(np.float64), b.astype(np.uint16).view(np.float16).astype(np.float64)
            alts = [c.astype(np.uint16).view(np.float16).astype(np.float64) for c in alts]
            kind = "f16"
        key = (shader, slot, fmt_name)
        r = self.rows.setdefault(key, {"n": 0, "exact": 0, "within_tol": 0, "within_1e3": 0, "within_1e3_vec": 0, "max": 0.0, "worst": None, "frames": 0, "outliers": 0, "outliers_sensitive": 0})
        is_float = kind in ("float", "f16")
        if is_float:
            both_nan = np.isnan(a) & np.isnan(b)
            err = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
            err = np.where(both_nan, 0.0, err)
            err = np.where(np.isnan(err), np.inf, err)
            # storage granularity: one unit in the last place of the stored format is as close as two correct implementations can be held
            if kind == "f16":
                ulp = np.maximum(np.abs(b), 6.1e-5) * 2.0 ** -10
                one_ulp = np.abs(a - b) <= ulp * 1.0001
            else:
                one_ulp = np.zeros(a.shape, bool)
            ok = (err <= self.tol) | one_ulp
        else:  # quantised codes (UNORM / SNORM): equal, or one code apart; packed bits and indices: equal
            err = np.abs(a - b)
            ok = err <= (1.0 if kind in ("code", "code16") else 0.0)
        out = ~ok if not is_float else (err > 1e-3) & ~ok
        if alts and np.any(out):
            # an outlier is "sensitive" when the oracle's own result at that texel moves by a comparable amount under a change of rounding alone
            moved = np.zeros(a.shape, bool)
            for c in alts:
                d_alt = np.abs(a - c) / np.maximum(np.abs(b), 1e-3) if is_float else np.abs(a - c)
                moved |= d_alt >= 0.25 * err
            r["outliers_sensitive"] += int(np.sum(out & moved))
        r["outliers"] += int(np.sum(out))
        r["n"] += a.size
        r["exact"] += int(np.sum(a == b) + (np.sum(np.isnan(a) & np.isnan(b)) if is_float else 0))
        r["within_tol"] += int(np.sum(ok))
        # 16-bit UNORM / SNORM codes resolve 1.5e-5 of full scale: the north-star's 1e-3 is measured on the value (floor: 1e-3 of full scale); 8-bit codes do not
        within = (err <= 1e-3) if is_float else ((np.abs(a - b) <= 1e-3 * np.maximum(np.abs(b), 65.535)) | ok) if kind == "code16" else ok
        r["within_1e3"] += int(np.sum(within))
        if is_float and a.ndim == 3 and a.shape[-1] == 4:
            # colour / direction texels: the first three channels relative to the texel's largest of them (a component that cancels to ~0 inherits the
            # absolute error of its neighbours: YCoCg chroma, SH1 direction), the fourth (hit distance, variance, ...) relative to itself
            scale = np.maximum(np.max(np.abs(b[..., :3]), axis=-1, keepdims=True), 1e-3)
            err_vec = np.concatenate([np.abs(a - b)[..., :3] / scale, err[..., 3:]], axis=-1)
            err_vec = np.where(np.isnan(err_vec), np.where(both_nan, 0.0, np.inf), err_vec)
            r["within_1e3_vec"] += int(np.sum(err_vec <= 1e-3))
        else:
            r["within_1e3_vec"] += int(np.sum(within))
        r["frames"] += 1
        m = float(np.max(np.where(np.isfinite(err), err, 1e30))) if err.size else 0.0
        if m >= r["max"]:
            r["max"] = m
            r["worst"] = [int(v) for v in np.unravel_index(int(np.argmax(err)), err.shape)] if err.size else None

    def table(self):
        out = []
        for (shader, slot, fmt), r in sorted(self.rows.items()):
            n = max(r["n"], 1)
            out.append({"pass": shader, "output": slot, "format": fmt, "texel_values": r["n"], "bit_exact_frac": r["exact"] / n, "within_tol_frac": r["within_tol"] / n,
                        "within_1e-3_frac": r["within_1e3"] / n, "within_1e-3_vec_frac": r["within_1e3_vec"] / n, "max_err": r["max"], "worst_at": r["worst"], "dispatches": r["frames"], "outliers": r["outliers"], "outliers_sensitive": r["outliers_sensitive"]})
        return out


def run_per_pass(name, width=192, height=128, frames=4, settings_overrides=None, cs_kw=None, extra_want=(), static_camera=False, tol=1e-5, ieee=True, verbose=False, promote_fp16=False, strict=True, sensitivity=True,
                 resource=None, rect_sizes=None, rect_origin=None):
    """Runs `frames` frames of denoiser `name` through the oracle and, pass by pass on identical inputs, through oracle/_ref. Returns PassStats.
    resource = (w, h) >= (width, height): dynamic resolution, the frame is the top-left rect of resource-sized planes; rect_sizes = [(w, h), ...]: the rect of frame f is
    rect_sizes[f % len] inside `resource` (tests/parity.py run_parity has the same two options). rect_origin = (ox, oy) with `resource`: CommonSettings::rectOrigin -- the guide
    inputs live at that offset inside their planes, everything else at (0, 0); compared with the reference's NRD_USE_VIEWPORT_OFFSET = 1 build (oracle/_ref/libnrdref_vo.so)."""
    stats = PassStats(tol)

    def on_pass(d, report):
        n_inputs = sum(1 for r in d.resources if r[0] == api.DescriptorType.TEXTURE)
        for res, fmt, w, mine, theirs, alts in report:
            slot = d.resources.index(res)
            label = ("out%d" % (slot - n_inputs) if res[0] == api.DescriptorType.STORAGE_TEXTURE else "in%d(modified)" % slot) + ":" + res[1].name + ("[%d]" % res[2] if "POOL" in res[1].name else "")
            stats.add(d.shader, label, fmt, w, mine, theirs, alts)

    prev = oracle_driver.set_ieee_mode(ieee)
    try:
        if rect_sizes:
            seq = [parity.synth.render_frame(*rect_sizes[f % len(rect_sizes)], f, static_camera=static_camera, want=tuple(parity.DENOISERS[name][1]) + tuple(extra_want)) for f in range(frames)]
        else:
            seq = parity.generate_sequence(name, width, height, frames, static_camera=static_camera, extra_want=extra_want, device="cpu")
        cs_kw = dict(cs_kw or {})
        if resource:
            seq = [embed_guides_at(fr, resource, rect_origin) if rect_origin else parity.embed_in_resource(fr, resource) for fr in seq]
            cs_kw.update(resourceSize=resource, resourceSizePrev=resource)
            if rect_origin:
                cs_kw.update(rectOrigin=rect_origin)
        rw, rh = resource or (width, height)
        run = parity.OracleRun(name, rw, rh, validation=bool((cs_kw or {}).get("enableValidation")))  # (the overlay plane OUT_VALIDATION is bound and compared like any output)
        cmp_ex = oracle_driver.ComparingExecutor(run.inst, rw, rh, api.FORMAT_BYTES, on_pass=on_pass, promote_fp16=promote_fp16, strict=strict, sensitivity=sensitivity,
                                                      ref_lib_path=oracle_driver.REF_VO_LIB_PATH if rect_origin else None)
        cmp_ex.user = run.ex.user  # the bound output planes
        if promote_fp16:  # the user's OUT_* planes double as scratch of the pass chain: promote the fp16 ones too
            for rt, (arr, fmt) in list(run.outs.items()):
                if fmt == F.RGBA16_SFLOAT:
                    big = np.zeros(arr.shape, np.float32)
                    run.outs[rt] = (big, F.RGBA32_SFLOAT)
                    cmp_ex.bind(rt, big, F.RGBA32_SFLOAT)
        run.ex = cmp_ex
        for f, frame in enumerate(seq):
            cam, cam_prev = frame["camera"], seq[max(f - 1, 0)]["camera"]
            if rect_sizes:
                width, height = rect_sizes[f % len(rect_sizes)]
                cs_kw.update(rectSize=(width, height), rectSizePrev=rect_sizes[max(f - 1, 0) % len(rect_sizes)])
            cs = parity.common_settings(cam, cam_prev, width, height, f, **cs_kw)
            parity.tag_checkerboard(frame, settings_overrides, f)
            run.step(frame, cs, parity.denoiser_settings(name, frame, settings_overrides))
            if verbose:
                print("frame", f, "done")
    finally:
        oracle_driver.set_ieee_mode(prev)
    return stats


def print_table(stats):
    for row in stats.table():
        print("%-52s %-34s %-20s n %9d  exact %.5f  ok %.6f  <=1e-3 %.6f  max %.3g at %s  outliers %d (sensitive %d)" % (row["pass"], row["output"], row["format"], row["texel_values"], row["bit_exact_frac"], row["within_tol_frac"],
                                                                                        row["within_1e-3_frac"], row["max_err"], row["worst_at"], row["outliers"], row["outliers_sensitive"]))


if __name__ == "__main__":
    import sys

    name = sys.argv[1] if len(sys.argv) > 1 else "REBLUR_DIFFUSE_SPECULAR"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print_table(run_per_pass(name, frames=frames, verbose=True, promote_fp16="--fp32" in sys.argv, strict="--contract" not in sys.argv))
