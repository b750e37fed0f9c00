"""include/NRD.hip.h -- the HIP counterpart of the reference's NRD.hlsli front-end / back-end functions (what an application's own
kernels include). tests/cpp/frontend_check.hip evaluates every function on a deterministic sample set: known answers on the host
(CPU test), device == host on the GPU. Here additionally: the normal/roughness words it packs equal the ones of the generator that
feeds all parity tests (raytracingdenoiser_amd/synth.py), so the header and the denoiser agree on the encoding."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from raytracingdenoiser_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "frontend_check.hip")
HDR = os.path.join(ROOT, "include", "NRD.hip.h")
EXE = os.path.join(ROOT, "tests", "cpp", "build", "frontend_check" + api.ENCODING_SUFFIX)  # (one harness per G-buffer encoding: tests/test_encodings.py)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        return
    cmd = [HIPCC, "-std=c++17", "-O2", "-ffp-contract=off", "--offload-arch=gfx950", "-DNRD_NORMAL_ENCODING=%d" % api.NORMAL_ENCODING, "-DNRD_ROUGHNESS_ENCODING=%d" % api.ROUGHNESS_ENCODING,
           "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _pcg(v):
    v = v.astype(np.uint64)
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w).astype(np.uint32)


def _u(i, k):
    return (_pcg(i * 64 + k) >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _expected_word_checksum(count):
    i = np.arange(count, dtype=np.uint64)
    v = np.stack([_u(i, 0) * np.float32(2) - np.float32(1), _u(i, 1) * np.float32(2) - np.float32(1), _u(i, 2) * np.float32(2) - np.float32(1)], -1)
    l = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2], dtype=np.float32)
    n = np.where((l < np.float32(0.05))[:, None], np.array([0, 0, 1], np.float32)[None], v / np.maximum(l, np.float32(1e-20))[:, None]).astype(np.float32)
    roughness = _u(i, 18)
    material = (_pcg(i + 77) & 3).astype(np.float32)
    words = np.ascontiguousarray(synth.pack_normal_roughness(torch.from_numpy(n), torch.from_numpy(roughness), torch.from_numpy(material)).numpy()).view(np.uint32).reshape(-1)
    c = 0  # (int16 [n, 4] texels of the 64-bit encodings: low word, then high word of every sample, as the harness folds them)
    for w in words.tolist():
        c = (c * 31 + w) & 0xFFFFFFFF
    return c


def test_frontend_header_known_answers_on_the_host():
    _build()
    r = subprocess.run([EXE, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"host OK: (\d+) samples, normal/roughness word checksum ([0-9a-f]{8})", r.stdout)
    assert m, r.stdout
    assert int(m.group(2), 16) == _expected_word_checksum(int(m.group(1)))


@pytest.mark.gpu
def test_frontend_header_device_matches_host():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frontend check OK" in r.stdout


# ---- the device results of the header against implementations that do NOT include it ---------------------------------------------
DUMP_FIELDS = [("N", 3), ("V", 3), ("radiance", 3), ("direction", 3), ("albedo", 3), ("Rf0", 3), ("roughness", 1), ("materialID_in", 1), ("hitDist", 1), ("viewZ", 1), ("Nw", 3),
               ("word", 1), ("unpackedNR", 4), ("reblurPacked", 4), ("reblurUnpacked", 4), ("sh0", 4), ("sh1", 4), ("relaxPacked", 4), ("relaxSh1", 4), ("dirOcc", 4), ("translucency", 4),
               ("normHitDist", 1), ("penumbra", 1), ("penumbraLocal", 1), ("shadow", 1), ("materialID", 1), ("diffFactor", 3), ("specFactor", 3), ("sgDiffuse", 3), ("sgSpecular", 3),
               ("shDiffuse", 3), ("shSpecular", 3), ("sgColor", 3), ("sgDir", 3), ("rejitter", 2), ("misc", 4)]


def _load_dump(path, count):
    words = sum(n for _, n in DUMP_FIELDS)
    raw = np.fromfile(path, dtype=np.uint32).reshape(count, words)
    out, k = {}, 0
    for name, n in DUMP_FIELDS:
        col = raw[:, k:k + n]
        out[name] = col.copy() if name == "word" else col.view(np.float32).copy()
        k += n
    return out


def _ulps(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia), np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


@pytest.mark.gpu
def test_frontend_header_device_results_match_the_oracle_and_a_float64_model():
    """131072 samples evaluated ON THE DEVICE by include/NRD.hip.h (tests/cpp/frontend_check --dump) against
      * the oracle's own restatements of the functions its passes consume (oracle/ml.h [nrd], through oracle_frontend): the R10G10B10A2 words bit for bit,
        the floats within a few ulp (the header is application-side IEEE code, the oracle the device contract: dot products fuse differently);
      * tests/frontend_model.py, a float64 numpy restatement of reference NRD.hlsli:300-1130 written without the header: packers, unpackers, material
        factors, SG / SH resolves, re-jitter.
    Neither expected-value side includes NRD.hip.h."""
    import ctypes as C

    import frontend_model as M
    from oracle import driver as oracle_driver

    _build()
    count = 131072
    path = os.path.join(os.path.dirname(EXE), "frontend_dump.bin")
    r = subprocess.run([EXE, os.environ.get("NRD_FRONTEND_DUMP_MODE", "--dump"), path, str(count)], capture_output=True, text=True, timeout=300)  # (--dump-host: debugging without a GPU)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _load_dump(path, count)
    os.remove(path)
    f64 = {k: v.astype(np.float64) for k, v in d.items() if k != "word"}
    scal = lambda k: f64[k][:, 0]

    # ---- vs the oracle (float32, the arithmetic of the passes; IEEE mode: the header knows nothing about v_rsq_f32 / v_exp_f32)
    lib = oracle_driver.load()
    prev = oracle_driver.set_ieee_mode(True)
    try:
        def ora(op, cols, nout):
            a = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.float32)
            o = np.zeros((count, nout), dtype=np.float32)
            lib.oracle_frontend(op, a.ctypes.data, a.shape[1], o.ctypes.data, nout, count)
            return o

        word = ora(0, [d["N"], d["roughness"], d["materialID_in"]], 1).view(np.uint32)[:, 0]
        assert np.array_equal(word, d["word"][:, 0]), "R10G10B10A2 words differ in %d of %d samples" % (int((word != d["word"][:, 0]).sum()), count)
        unpacked = ora(1, [d["word"].view(np.float32)], 5)
        assert _ulps(unpacked[:, :4], d["unpackedNR"]).max() <= 4 and np.array_equal(unpacked[:, 4], d["materialID"][:, 0])
        nhd = ora(2, [d["hitDist"], d["viewZ"], d["roughness"]], 1)
        assert np.max(np.abs(nhd - d["normHitDist"])) <= 2e-6
        packed = ora(3, [d["radiance"], d["normHitDist"]], 4)
        assert _ulps(packed, d["reblurPacked"]).max() <= 2
        unp = ora(4, [d["reblurPacked"]], 4)
        assert _ulps(unp, d["reblurUnpacked"]).max() <= 2
        assert np.array_equal(ora(5, [d["roughness"]], 1), d["shadow"])
    finally:
        oracle_driver.set_ieee_mode(prev)

    # ---- vs the float64 model of NRD.hlsli
    def close(got, want, rel, what, floor=1.0):  # |got - want| / max(|want|, floor): the inputs are O(1), sums and differences of them carry O(1) * 2^-24 of rounding
        err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), floor)
        assert np.all(np.isfinite(got)) and err.max() <= rel, "%s: max relative error %.3g (allowed %.3g) at sample %d" % (what, err.max(), rel, int(np.argmax(err.max(axis=-1) if err.ndim > 1 else err)))

    hdp = (3.0, 0.1, 20.0, -25.0)
    N, V, rad, dirn, rough = f64["N"], f64["V"], f64["radiance"], f64["direction"], scal("roughness")
    assert np.array_equal(M.store_r10g10b10a2(M.pack_normal_and_roughness(N, rough, scal("materialID_in"))), d["word"][:, 0]) or \
        int((M.store_r10g10b10a2(M.pack_normal_and_roughness(N, rough, scal("materialID_in"))) != d["word"][:, 0]).sum()) <= count // 2000  # float64 vs float32 at quantisation ties
    un, mat = M.unpack_normal_and_roughness(M.load_r10g10b10a2(d["word"][:, 0]))
    close(d["unpackedNR"], un, 2e-6, "NRD_FrontEnd_UnpackNormalAndRoughness")
    assert np.array_equal(mat, scal("materialID"))
    nhd = M.reblur_get_norm_hit_dist(scal("hitDist"), scal("viewZ"), hdp, rough)
    close(d["normHitDist"][:, 0], nhd, 5e-6, "REBLUR_FrontEnd_GetNormHitDist")
    nhd32 = scal("normHitDist")
    close(d["reblurPacked"], M.reblur_pack_radiance_and_norm_hit_dist(rad, nhd32), 2e-6, "REBLUR_FrontEnd_PackRadianceAndNormHitDist")
    close(d["reblurUnpacked"], M.reblur_unpack_radiance_and_norm_hit_dist(f64["reblurPacked"]), 2e-6, "REBLUR_BackEnd_UnpackRadianceAndNormHitDist")
    sh0, sh1 = M.reblur_pack_sh(rad, nhd32, dirn)
    close(d["sh0"], sh0, 2e-6, "REBLUR_FrontEnd_PackSh out0")
    close(d["sh1"], sh1, 2e-6, "REBLUR_FrontEnd_PackSh out1")
    r0, r1 = M.relax_pack_sh(rad, scal("hitDist"), dirn)
    close(d["relaxPacked"], r0, 1e-7, "RELAX_FrontEnd_PackSh out0")
    close(d["relaxSh1"], r1, 2e-6, "RELAX_FrontEnd_PackSh out1")
    close(d["dirOcc"], M.reblur_pack_directional_occlusion(dirn, nhd32), 2e-6, "REBLUR_FrontEnd_PackDirectionalOcclusion")
    occluder = np.where(np.arange(count) % 5 == 0, M.NRD_FP16_MAX, scal("hitDist"))
    close(d["penumbra"][:, 0], M.sigma_pack_penumbra(occluder, np.float64(np.float32(0.02))), 1e-6, "SIGMA_FrontEnd_PackPenumbra (directional)")
    close(d["penumbraLocal"][:, 0], M.sigma_pack_penumbra_local(scal("hitDist"), scal("hitDist") + 10.0, 0.5), 2e-6, "SIGMA_FrontEnd_PackPenumbra (local)")
    close(d["translucency"], M.sigma_pack_translucency(occluder, f64["albedo"]), 1e-7, "SIGMA_FrontEnd_PackTranslucency")
    close(d["shadow"][:, 0], rough * rough, 1e-6, "SIGMA_BackEnd_UnpackShadow")
    df, sf = M.material_factors(N, V, f64["albedo"], f64["Rf0"], rough)
    close(d["diffFactor"], df, 2e-5, "NRD_MaterialFactors diff")
    close(d["specFactor"], sf, 2e-5, "NRD_MaterialFactors spec")
    sg = M.unpack_sh(f64["sh0"], f64["sh1"])  # the SG the device resolved: from its own packed words (float32 inputs, float64 arithmetic)
    close(d["sgColor"], M.sg_extract_color(sg), 2e-6, "NRD_SG_ExtractColor")
    close(d["sgDir"], M.sg_extract_direction(sg), 2e-6, "NRD_SG_ExtractDirection")
    close(d["sgDiffuse"], M.sg_resolve_diffuse(sg, N), 1e-4, "NRD_SG_ResolveDiffuse")
    # NRD_SG_ResolveSpecular evaluates exp(d - a.sharpness - b.sharpness) with a.sharpness = 2 / roughness^4 / (4 |H.V|): in float32 (the shader's and the header's
    # arithmetic) the difference loses all its digits as the roughness goes to 0 -- inherent in NRD.hlsli:1003-1050, measured here on the host: max error 4e-5 for
    # roughness >= 0.4, 3e-4 for >= 0.2, 1.4e-2 for >= 0.1, O(1) below 0.05. Held to the float64 model where float32 can follow it; finite and non-negative everywhere.
    spec_want = M.sg_resolve_specular(sg, N, V, rough)
    close(d["sgSpecular"][rough >= 0.2], spec_want[rough >= 0.2], 1e-3, "NRD_SG_ResolveSpecular (roughness >= 0.2)")
    close(d["sgSpecular"][(rough >= 0.1) & (rough < 0.2)], spec_want[(rough >= 0.1) & (rough < 0.2)], 5e-2, "NRD_SG_ResolveSpecular (0.1 <= roughness < 0.2)")
    assert np.all(np.isfinite(d["sgSpecular"])) and d["sgSpecular"].min() >= 0.0
    close(d["shDiffuse"], M.sh_resolve_diffuse(sg, N), 1e-5, "NRD_SH_ResolveDiffuse")
    close(d["shSpecular"], M.sh_resolve_specular(sg, N, V, rough), 1e-4, "NRD_SH_ResolveSpecular")
    Z = scal("viewZ")
    Ze, Zw = (d["viewZ"][:, 0] * np.float32(1.001)).astype(np.float64), (d["viewZ"][:, 0] * np.float32(0.999)).astype(np.float64)
    rj = M.sg_rejitter(sg, sg, f64["Rf0"], V, rough, Z, Ze, Zw, Z, Z, N, N, f64["Nw"], N, N)
    near_threshold = np.abs(np.abs(Ze - Z) - M.NRD_REJITTER_VIEWZ_THRESHOLD * np.abs(Z) / (np.abs(M.dot(N, V)) * 0.95 + 0.05)) < 1e-5 * np.abs(Z)  # float32 decides the 4-neighbour test differently
    ok = ~near_threshold & (np.abs(M.dot(f64["Nw"], N)) > 1e-6)
    close(d["rejitter"][ok & (rough >= 0.1)], rj[ok & (rough >= 0.1)], 5e-4, "NRD_SG_ReJitter (roughness >= 0.1)")
    close(d["rejitter"][ok & (rough < 0.1)], rj[ok & (rough < 0.1)], 1e-2, "NRD_SG_ReJitter (roughness < 0.1: the GGX terms with m^2 -> 0 in float32)")


@pytest.mark.skipif(not __import__("oracle.driver", fromlist=["x"]).ref_available(), reason="oracle/_ref/libnrdref.so not built (needs /root/reference)")
def test_frontend_header_equals_the_reference_nrd_hlsli_text():
    """include/NRD.hip.h evaluated on the host (frontend_check --dump-host; the GPU test above holds device == host) against the REFERENCE'S OWN NRD.hlsli: a probe shader of this
    repository (oracle/ref/probes/NRD_FrontEndProbe.cs.hlsl) that only calls the reference's functions on the same 16 384 rows of inputs is compiled through the path of every reference
    entry (oracle/ref/hlsl2cpp.py -> oracle/_ref/libnrdref.so). NRD.hlsli does not use MathLib, so this comparison has no stand-in in it. Every packer, unpacker, the material factors,
    the SG colour / direction, the diffuse resolves: BIT FOR BIT. NRD_SG_ResolveSpecular: bit for bit for roughness >= 0.05 -- below, its SG inner product evaluates
    exp( d - sharpness_a - sharpness_b ) with sharpness = 2 / roughness^4 > 3e5 (NRD.hlsli:586, 1030), a cancellation in which one ulp of d moves the result by whole factors in ANY two
    fp32 evaluations; NRD_SH_ResolveSpecular and NRD_SG_ReJitter within 4e-6 of the texel's largest component, >= 98 % of the values bit for bit (a sum of products associates differently)."""
    import ctypes as C

    from oracle import driver as oracle_driver
    from raytracingdenoiser_amd import api

    _build()
    W = H = 128
    count = W * H
    path = os.path.join(os.path.dirname(EXE), "frontend_dump_host.bin")
    r = subprocess.run([EXE, "--dump-host", path, str(count)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _load_dump(path, count)
    os.remove(path)
    F = api.Format

    def tex(cols):
        a, k = np.zeros((H, W, 4), np.float32), 0
        for col in cols:
            col = np.asarray(col, np.float32).reshape(count, -1)
            a[..., k:k + col.shape[1]] = col.reshape(H, W, -1)
            k += col.shape[1]
        return a

    i = np.arange(count)
    ins = [tex([d["N"], d["roughness"]]), tex([d["V"], d["materialID_in"]]), tex([d["radiance"], d["hitDist"]]), tex([d["direction"], d["viewZ"]]),
           tex([d["albedo"], (i % 5 == 0).astype(np.float32)]), tex([d["Rf0"]]), tex([d["Nw"]])]
    word_in = np.ascontiguousarray(d["word"][:, 0].reshape(H, W).astype(np.uint32))
    word_out = np.zeros((H, W), np.uint32)
    outs = [np.zeros((H, W, 4), np.float32) for _ in range(20)]
    P = oracle_driver.OraclePlane
    planes = [P(a.ctypes.data, a.strides[0], int(F.RGBA32_SFLOAT), W, H) for a in ins]
    planes += [P(word_in.ctypes.data, word_in.strides[0], int(F.R10_G10_B10_A2_UNORM), W, H), P(word_out.ctypes.data, word_out.strides[0], int(F.R10_G10_B10_A2_UNORM), W, H)]
    planes += [P(a.ctypes.data, a.strides[0], int(F.RGBA32_SFLOAT), W, H) for a in outs]
    arr = (P * len(planes))(*planes)
    consts = np.array([3.0, 0.1, 20.0, -25.0], np.float32).tobytes() + np.array([W], np.uint32).tobytes() + bytes(12)
    buf = C.create_string_buffer(consts, len(consts))
    assert oracle_driver.load_ref().nrdref_dispatch(b"NRD_FrontEndProbe.cs", buf, len(consts), arr, len(planes), W // 8, H // 8) == 0

    assert np.array_equal(word_out.reshape(-1), d["word"][:, 0])  # NRD_FrontEnd_PackNormalAndRoughness through an R10G10B10A2_UNORM store
    exact = {"unpackedNR": 0, "reblurPacked": 2, "reblurUnpacked": 3, "sh0": 4, "sh1": 5, "relaxPacked": 6, "relaxSh1": 7, "dirOcc": 8, "translucency": 9, "specFactor": 11,
             "sgDiffuse": 12, "shDiffuse": 14, "sgColor": 16, "sgDir": 17,
             "misc": 19}  # REBLUR_GetHitDist, NRD_GetNormalizedStrandThickness, _NRD_SG_Integral, NRD_IsValidRadiance (every fifth row holds inf / NaN radiance)
    for name, k in exact.items():
        got = outs[k].reshape(count, 4)[:, : d[name].shape[1]]
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(d[name]).view(np.uint32)), name
    scalars = outs[1].reshape(count, 4)
    for k, name in enumerate(["normHitDist", "penumbra", "penumbraLocal", "shadow"]):
        assert np.array_equal(scalars[:, k].view(np.uint32), np.ascontiguousarray(d[name][:, 0]).view(np.uint32)), name
    dm = outs[10].reshape(count, 4)
    assert np.array_equal(dm[:, :3].view(np.uint32), np.ascontiguousarray(d["diffFactor"]).view(np.uint32)) and np.array_equal(dm[:, 3], d["materialID"][:, 0])
    well = d["roughness"][:, 0] >= 0.05
    assert well.sum() > 0.9 * count
    assert np.array_equal(outs[13].reshape(count, 4)[well][:, :3].view(np.uint32), np.ascontiguousarray(d["sgSpecular"][well]).view(np.uint32))
    for k, name, n in ((15, "shSpecular", 3), (18, "rejitter", 2)):  # relative to the texel's largest component: a chroma that cancels to ~1e-8 carries the rounding of the O(1) terms
        got, want = outs[k].reshape(count, 4)[well][:, :n].astype(np.float64), d[name][well].astype(np.float64)
        err = np.abs(got - want) / np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-3)
        assert err.max() <= 4e-6, (name, err.max())
        assert (got == want).mean() >= 0.98, name
