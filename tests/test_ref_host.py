"""The product's dispatch compiler (raytracingdenoiser_amd/csrc/host) against the reference's OWN host code.

oracle/_ref/libnrdhost.so is /root/reference/Source/{InstanceImpl,Wrapper,Reblur,Relax,Sigma,Reference,Timer}.cpp and Source/Denoisers/*.hpp compiled where they lie (oracle/ref/host/Makefile:
g++ on those files, no shader blobs) over a stand-in for the un-vendored MathLib (oracle/ref/host/ml.h, ml.hlsli -- "parity unpinned" for those two files alone). It exports the reference's C
entry points, so the very ctypes binding that drives the product (raytracingdenoiser_amd.api.Instance) drives the reference's nrd::Instance here, and both are fed the same settings, frame after
frame. Compared: the InstanceDesc (pipelines, pool formats and downsample factors) and, per frame, every DispatchDesc -- name, shader, grid, the resource list with its ping-pong state, the
constant buffer size and its bytes. Integers and bit patterns must be equal; floats are held to 1e-6 relative (in practice they are bit-identical: the two hosts compute in the same order).

Round 4 built this at the very end and it found four things at once (all fixed in the host tables): the size of the SIGMA and RELAX a-trous constant blocks (the reference reports sizeof of a
struct with 16-byte aligned members: 528 / 720, not the 516 / 712 bytes of fields), the pass-name prefix of REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION ("REBLUR_DirectionalOcclusion", sic), and the
order of the four history planes of RELAX_DIFFUSE_SPECULAR (the one variant that interleaves the signals). One difference is kept on purpose, see KNOWN below.
"""
import numpy as np
import pytest

import parity
from oracle import driver as oracle_driver
from raytracingdenoiser_amd import api

pytestmark = pytest.mark.skipif(not __import__("os").path.exists(oracle_driver.REF_HOST_LIB_PATH), reason="oracle/_ref/libnrdhost.so not built (needs /root/reference: make -C oracle/ref/host)")

# Reblur_DiffuseSpecularSh.hpp:61-85: the Transient enum lists 10 planes, the code below it adds 11 textures -- an RGBA16_SFLOAT full-resolution texture slipped in front of the tile
# plane, so the reference's Transient::TILES (index 9) is that texture and the R8_UNORM / 16 tile texture (index 10) is never bound. The product keeps the 10 planes of the enum with the
# tile plane at index 9: same indices in every dispatch, 29 MB less at 1440p; the InstanceDesc and the grid of the restart frame's clear of that plane differ accordingly.
KNOWN = {"REBLUR_DIFFUSE_SPECULAR_SH": ("transient pool", "Clear_Float.cs grid")}


def _compare(name, frames, settings_overrides=None, cs_kw=None, size=(96, 64), rect_sizes=None, extra_want=()):
    ref = oracle_driver.load_ref_host()
    den = parity.DENOISERS[name][0]
    a, b = api.Instance([(0, den)]), api.Instance([(0, den)], lib=ref)
    issues = []
    if a.pipelines != b.pipelines:
        issues.append("pipelines %s vs %s" % (a.pipelines, b.pipelines))
    if a.permanent_pool != b.permanent_pool:
        issues.append("permanent pool %s vs %s" % (a.permanent_pool, b.permanent_pool))
    if a.transient_pool != b.transient_pool:
        issues.append("transient pool %d vs %d planes" % (len(a.transient_pool), len(b.transient_pool)))
    W, H = size
    seq = parity.generate_sequence(name, W, H, frames, device="cpu", extra_want=extra_want)
    compared = 0
    for f in range(frames):
        kw = dict(cs_kw or {})
        rw, rh = rect_sizes[f % len(rect_sizes)] if rect_sizes else (W, H)
        if rect_sizes:
            prw, prh = rect_sizes[(f - 1) % len(rect_sizes)] if f else (rw, rh)
            kw.update(resourceSize=(W, H), resourceSizePrev=(W, H), rectSizePrev=(prw, prh))
        if callable(kw.get("per_frame")):
            kw.update(kw.pop("per_frame")(f))
        kw.pop("per_frame", None)
        cs = parity.common_settings(seq[f]["camera"], seq[max(f - 1, 0)]["camera"], rw, rh, f, **kw)
        st = parity.denoiser_settings(name, seq[f], settings_overrides)
        for inst in (a, b):
            assert inst.set_common_settings(cs) == api.Result.SUCCESS
            assert inst.set_denoiser_settings(0, st) == api.Result.SUCCESS
        (ra, da), (rb, db) = a.get_compute_dispatches(), b.get_compute_dispatches()
        if ra != rb or [d.shader for d in da] != [d.shader for d in db]:
            issues.append("frame %d: %s %s vs %s %s" % (f, ra.name, [d.shader for d in da], rb.name, [d.shader for d in db]))
            continue
        for x, y in zip(da, db):
            compared += 1
            for what, u, v in (("name", x.name, y.name), ("grid", x.grid, y.grid), ("constant size", len(x.constants), len(y.constants)), ("resources", x.resources, y.resources),
                               ("constants match previous", x.constants_match_previous, y.constants_match_previous)):
                if u != v:
                    issues.append("frame %d %s %s: %s vs %s" % (f, x.shader, what, u, v))
            if x.constants != y.constants and len(x.constants) == len(y.constants):
                ia, ib = np.frombuffer(x.constants, np.uint32), np.frombuffer(y.constants, np.uint32)
                fa, fb = np.frombuffer(x.constants, np.float32), np.frombuffer(y.constants, np.float32)
                ne = np.nonzero(ia != ib)[0]
                with np.errstate(all="ignore"):
                    rel = np.abs(fa[ne] - fb[ne]) / np.maximum(np.abs(fb[ne]), 1e-6)
                if not np.nanmax(rel) <= 1e-6:
                    issues.append("frame %d %s constants: dwords %s, product %s, reference %s" % (f, x.shader, ne[:8].tolist(), fa[ne][:4], fb[ne][:4]))
    known = KNOWN.get(name, ())  # each entry: words that all occur in the message of an expected difference
    left = [i for i in issues if not any(all(word in i for word in k.split()) for k in known)]
    assert not left, "\n".join(left[:12])
    assert compared >= frames * 2
    return issues


@pytest.mark.parametrize("name", list(parity.DENOISERS))
def test_default_settings_every_denoiser(name):
    issues = _compare(name, frames=4)
    if name in KNOWN:  # the list above stays honest
        assert any("transient pool" in i for i in issues)


@pytest.mark.parametrize("name, overrides, cs_kw", [
    ("REBLUR_DIFFUSE_SPECULAR", {"enablePerformanceMode": True, "enableAntiFirefly": True, "hitDistanceReconstructionMode": 1}, None),
    ("REBLUR_DIFFUSE_SPECULAR", {"maxStabilizedFrameNum": 0, "hitDistanceReconstructionMode": 2}, None),
    ("REBLUR_DIFFUSE_SPECULAR", {"checkerboardMode": 1}, None),
    ("REBLUR_DIFFUSE_SPECULAR", {"checkerboardMode": 2, "enablePerformanceMode": True}, None),
    ("REBLUR_DIFFUSE_SPECULAR_OCCLUSION", {"hitDistanceReconstructionMode": 2, "checkerboardMode": 1, "enablePerformanceMode": True}, None),
    ("REBLUR_DIFFUSE_SPECULAR_SH", {"enableAntiFirefly": True, "maxAccumulatedFrameNum": 5, "maxFastAccumulatedFrameNum": 2, "maxStabilizedFrameNum": 0}, None),
    ("REBLUR_DIFFUSE_DIRECTIONAL_OCCLUSION", {"enablePerformanceMode": True, "hitDistanceReconstructionMode": 1}, None),
    ("REBLUR_DIFFUSE_SPECULAR", None, dict(splitScreen=0.4)),
    ("REBLUR_DIFFUSE_SPECULAR", None, dict(enableValidation=True)),
    ("REBLUR_DIFFUSE_SPECULAR", None, dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / 96, 1.0 / 64, 1.0), isBaseColorMetalnessAvailable=True)),
    ("REBLUR_DIFFUSE_SPECULAR", None, dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True)),
    ("REBLUR_DIFFUSE_SPECULAR", None, dict(cameraJitter=(0.3, -0.2), cameraJitterPrev=(-0.1, 0.25), denoisingRange=20.0, disocclusionThreshold=0.003, rectOrigin=(0, 0))),
    ("REBLUR_DIFFUSE_SPECULAR", dict(minMaterialForDiffuse=0.0, minMaterialForSpecular=1.0), dict(strandMaterialID=1.0, cameraAttachedReflectionMaterialID=2.0, strandThickness=0.003)),
    ("RELAX_DIFFUSE_SPECULAR", {"enableAntiFirefly": True, "hitDistanceReconstructionMode": 1, "atrousIterationNum": 6}, None),
    ("RELAX_DIFFUSE_SPECULAR", {"checkerboardMode": 1, "enableRoughnessEdgeStopping": False, "hitDistanceReconstructionMode": 2}, None),
    ("RELAX_DIFFUSE_SPECULAR_SH", {"atrousIterationNum": 8, "enableAntiFirefly": True}, dict(isHistoryConfidenceAvailable=True, isDisocclusionThresholdMixAvailable=True)),
    ("RELAX_DIFFUSE_SPECULAR", {"atrousIterationNum": 2, "diffuseMaxAccumulatedFrameNum": 4, "specularMaxAccumulatedFrameNum": 6, "historyFixFrameNum": 1}, dict(splitScreen=0.3)),
    ("RELAX_DIFFUSE", None, dict(enableValidation=True)),
    ("RELAX_SPECULAR_SH", {"checkerboardMode": 2}, dict(isMotionVectorInWorldSpace=False, motionVectorScale=(1.0 / 96, 1.0 / 64, 0.0))),
    ("SIGMA_SHADOW", {"maxStabilizedFrameNum": 0}, None),
    ("SIGMA_SHADOW_TRANSLUCENCY", {"planeDistanceSensitivity": 0.05, "maxStabilizedFrameNum": 3}, dict(splitScreen=0.4)),
    ("SIGMA_SHADOW", None, dict(enableValidation=True)),
])
def test_option_matrix(name, overrides, cs_kw):
    _compare(name, frames=4, settings_overrides=overrides, cs_kw=cs_kw)


@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW_TRANSLUCENCY", "REBLUR_DIFFUSE_OCCLUSION"])
def test_dynamic_resolution_restarts_and_rect_origin(name):
    """the rect changes every frame inside 192x128 resources; the history is cleared / restarted in mid-sequence; a viewport origin"""
    _compare(name, frames=6, size=(192, 128), rect_sizes=[(192, 128), (144, 96), (96, 64)],
             cs_kw=dict(rectOrigin=(16, 8), per_frame=lambda f: dict(accumulationMode=(api.AccumulationMode.CLEAR_AND_RESTART if f == 3 else (api.AccumulationMode.RESTART if f == 4 else api.AccumulationMode.CONTINUE)))))


def test_mixed_instance_and_identifier_subsets():
    """three denoisers of three families in ONE instance (pools and pipelines are concatenated and de-duplicated), dispatches requested for subsets of the identifiers"""
    ref = oracle_driver.load_ref_host()
    members = [(7, "REBLUR_DIFFUSE_SPECULAR"), (3, "RELAX_DIFFUSE"), (11, "SIGMA_SHADOW")]
    dens = [(i, parity.DENOISERS[n][0]) for i, n in members]
    a, b = api.Instance(dens), api.Instance(dens, lib=ref)
    assert a.pipelines == b.pipelines and a.permanent_pool == b.permanent_pool and a.transient_pool == b.transient_pool
    seqs = {i: parity.generate_sequence(n, 96, 64, 3, device="cpu") for i, n in members}
    for f in range(3):
        cam, camp = seqs[7][f]["camera"], seqs[7][max(f - 1, 0)]["camera"]
        cs = parity.common_settings(cam, camp, 96, 64, f)
        for inst in (a, b):
            assert inst.set_common_settings(cs) == api.Result.SUCCESS
            for i, n in members:
                assert inst.set_denoiser_settings(i, parity.denoiser_settings(n, seqs[i][f], None)) == api.Result.SUCCESS
        for ids in ([7, 3, 11], [11], [3, 7]):
            (ra, da), (rb, db) = a.get_compute_dispatches(ids), b.get_compute_dispatches(ids)
            assert ra == rb
            assert [(d.identifier, d.shader, d.name, d.grid, d.resources, d.constants) for d in da] == [(d.identifier, d.shader, d.name, d.grid, d.resources, d.constants) for d in db], (f, ids)


def test_error_codes_and_library_desc():
    ref = oracle_driver.load_ref_host()
    mine = api.load_library()
    # library description: version, encodings, the list of denoisers
    la, lb = mine.GetLibraryDesc().contents, ref.GetLibraryDesc().contents
    for field, _ in api.LibraryDesc._fields_:
        u, v = getattr(la, field), getattr(lb, field)
        if field == "supportedDenoisers":
            assert [u[i] for i in range(la.supportedDenoisersNum)] == [v[i] for i in range(lb.supportedDenoisersNum)]
        elif not hasattr(u, "_fields_"):
            assert u == v, field
    # the reference's name table (Wrapper.cpp:58-95) is out of step with the enum for entries 3..15 (IN_DIFF_CONFIDENCE reads "IN_DIFF_RADIANCE_HITDIST", ...): the product returns
    # the enumerator's own name everywhere; outside that range the two agree
    for rt in api.ResourceType:
        k = int(rt)
        if rt.name == "MAX_NUM":
            continue
        assert mine.GetResourceTypeString(k) == rt.name.encode(), k
        if not 3 <= k <= 15:
            assert mine.GetResourceTypeString(k) == ref.GetResourceTypeString(k), k
    for k in range(len(parity.DENOISERS) + 2):
        assert mine.GetDenoiserString(k) == ref.GetDenoiserString(k), k

    import ctypes as C

    def create(lib, dens):
        descs = (api.DenoiserDesc * len(dens))(*[api.DenoiserDesc(i, int(d)) for i, d in dens])
        icd = api.InstanceCreationDesc()
        icd.denoisers, icd.denoisersNum = descs, len(dens)
        handle = C.c_void_p()
        r = api.Result(lib.CreateInstance(C.byref(icd), C.byref(handle)))
        if r == api.Result.SUCCESS:
            lib.DestroyInstance(handle)
        return r

    D = parity.DENOISERS
    for dens in ([(0, D["REBLUR_DIFFUSE"][0]), (0, D["SIGMA_SHADOW"][0])], [(0, 1000)], [(1, D["RELAX_DIFFUSE"][0]), (2, D["RELAX_DIFFUSE"][0])]):
        assert create(mine, dens) == create(ref, dens), dens
    a, b = api.Instance([(5, D["REBLUR_DIFFUSE"][0])]), api.Instance([(5, D["REBLUR_DIFFUSE"][0])], lib=ref)
    seq = parity.generate_sequence("REBLUR_DIFFUSE", 96, 64, 1, device="cpu")
    good = dict()
    bad = [dict(viewZScale=0.0), dict(denoisingRange=0.0), dict(disocclusionThreshold=0.0), dict(cameraJitter=(0.7, 0.0)), dict(resourceSize=(0, 64)),
           dict(isMotionVectorInWorldSpace=False, motionVectorScale=(0.0, 0.0, 0.0)), dict(strandMaterialID=0.0), dict(disocclusionThresholdAlternate=-1.0)]
    for kw in [good] + bad:
        cs = parity.common_settings(seq[0]["camera"], seq[0]["camera"], 96, 64, 0, **kw)
        assert a.set_common_settings(cs) == b.set_common_settings(cs), kw
    st = parity.denoiser_settings("REBLUR_DIFFUSE", seq[0], None)
    assert a.set_denoiser_settings(9, st) == b.set_denoiser_settings(9, st)  # unknown identifier
    assert a.get_compute_dispatches([9])[0] == b.get_compute_dispatches([9])[0]


def test_reference_accumulator_tables():
    ref = oracle_driver.load_ref_host()
    a, b = api.Instance([(0, api.Denoiser.REFERENCE)]), api.Instance([(0, api.Denoiser.REFERENCE)], lib=ref)
    assert a.pipelines == b.pipelines and a.permanent_pool == b.permanent_pool and a.transient_pool == b.transient_pool
    cam = parity.generate_sequence("REBLUR_DIFFUSE", 96, 64, 1, device="cpu")[0]["camera"]
    for f in range(5):
        cs = parity.common_settings(cam, cam, 80, 48, f, splitScreen=0.25 if f >= 3 else 0.0, resourceSize=(96, 64), resourceSizePrev=(96, 64), rectOrigin=(8, 4),
                                    accumulationMode=api.AccumulationMode.RESTART if f == 2 else api.AccumulationMode.CONTINUE)
        st = api.ReferenceSettings(maxAccumulatedFrameNum=3)
        for inst in (a, b):
            assert inst.set_common_settings(cs) == api.Result.SUCCESS
            assert inst.set_denoiser_settings(0, st) == api.Result.SUCCESS
        (ra, da), (rb, db) = a.get_compute_dispatches(), b.get_compute_dispatches()
        assert ra == rb
        assert [(d.shader, d.name, d.grid, d.resources, d.constants) for d in da] == [(d.shader, d.name, d.grid, d.resources, d.constants) for d in db], f


def _fuzz_struct(rng, obj, keep=()):
    """every scalar field of a ctypes settings struct set to a random value of its type (floats in [0, 2) with an occasional 0 / large value, counters 0..40, mode bytes 0..2, flags)"""
    import ctypes as C

    def value(t, name):
        if t is C.c_float:
            r = rng.random()
            return 0.0 if r < 0.05 else (float(rng.integers(1, 500)) if r < 0.1 else float(np.float32(rng.random() * 2.0)))
        if t is C.c_bool:
            return bool(rng.integers(0, 2))
        if t is C.c_ubyte:
            return int(rng.integers(0, 3))
        if t is C.c_uint:
            return int(rng.integers(0, 41))
        if t is C.c_ushort:
            return int(rng.integers(1, 300))
        raise TypeError((name, t))

    for name, t in obj._fields_:
        if name in keep:
            continue
        if hasattr(t, "_length_"):
            arr = getattr(obj, name)
            for i in range(t._length_):
                arr[i] = value(t._type_, name)
        else:
            setattr(obj, name, value(t, name))
    return obj


@pytest.mark.parametrize("name", list(parity.DENOISERS))
def test_fuzzed_settings_produce_the_same_dispatches(name):
    """40 frames of random denoiser settings and random common settings (valid cameras from the scene generator, everything else drawn at random, invalid combinations included: the two
    hosts must then agree on the error code): whatever a field does to the dispatch list or to a constant, it does the same in both"""
    ref = oracle_driver.load_ref_host()
    den = parity.DENOISERS[name][0]
    a, b = api.Instance([(0, den)]), api.Instance([(0, den)], lib=ref)
    rng = np.random.default_rng(sum(map(ord, name)))
    seq = parity.generate_sequence(name, 96, 64, 4, device="cpu")
    accepted = dispatches = 0
    for f in range(40):
        cam, camp = seq[f % 4]["camera"], seq[(f - 1) % 4]["camera"]
        cs = parity.common_settings(cam, camp, 96, 64, f)
        _fuzz_struct(rng, cs, keep=("viewToClipMatrix", "viewToClipMatrixPrev", "worldToViewMatrix", "worldToViewMatrixPrev", "worldPrevToWorldMatrix", "frameIndex", "timeDeltaBetweenFrames"))
        cs.frameIndex = f
        for k in range(2):  # rect inside the resource (the hosts do not check it; the shaders would read outside)
            cs.rectSize[k] = min(cs.rectSize[k], cs.resourceSize[k])
            cs.rectSizePrev[k] = min(cs.rectSizePrev[k], cs.resourceSizePrev[k])
            cs.cameraJitter[k], cs.cameraJitterPrev[k] = cs.cameraJitter[k] * 0.3 - 0.25, cs.cameraJitterPrev[k] * 0.3 - 0.25
        st = _fuzz_struct(rng, parity.denoiser_settings(name, seq[f % 4], None))
        ra, rb = a.set_common_settings(cs), b.set_common_settings(cs)
        assert ra == rb, (f, ra, rb)
        assert a.set_denoiser_settings(0, st) == b.set_denoiser_settings(0, st)
        if ra != api.Result.SUCCESS:
            continue
        accepted += 1
        (ra, da), (rb, db) = a.get_compute_dispatches(), b.get_compute_dispatches()
        assert ra == rb
        tile_clear = lambda d: name in KNOWN and d.shader == "Clear_Float.cs" and d.resources[0][1:] == (api.ResourceType.TRANSIENT_POOL, 9)  # (KNOWN: only its grid differs)
        assert [(d.shader, d.name, None if tile_clear(d) else d.grid, d.resources) for d in da] == [(d.shader, d.name, None if tile_clear(d) else d.grid, d.resources) for d in db], f
        for x, y in zip(da, db):
            dispatches += 1
            if x.constants != y.constants:
                assert len(x.constants) == len(y.constants)
                ia, ib = np.frombuffer(x.constants, np.uint32), np.frombuffer(y.constants, np.uint32)
                fa, fb = np.frombuffer(x.constants, np.float32), np.frombuffer(y.constants, np.float32)
                ne = np.nonzero(ia != ib)[0]
                with np.errstate(all="ignore"):
                    rel = np.abs(fa[ne] - fb[ne]) / np.maximum(np.abs(fb[ne]), 1e-6)
                assert np.nanmax(rel) <= 1e-6, (f, x.shader, ne[:8].tolist(), fa[ne][:4], fb[ne][:4])
    assert accepted >= 10 and dispatches >= 60, (accepted, dispatches)


@pytest.mark.skipif(not oracle_driver.ref_available(), reason="oracle/_ref/libnrdref.so not built")
@pytest.mark.parametrize("name", ["REBLUR_DIFFUSE_SPECULAR", "RELAX_DIFFUSE_SPECULAR_SH", "SIGMA_SHADOW", "REBLUR_DIFFUSE_SPECULAR_SH"])  # (the last: the KNOWN pool difference changes no output)
def test_the_complete_reference_host_and_shader_text_equals_the_products_host_driving_the_same_text(name):
    """The reference end to end on the CPU -- its own host code emitting the dispatches, its own shader text executing them (oracle/_ref/libnrdhost.so + libnrdref.so) -- against
    the product's host emitting the dispatches for the same shader text: every user output of every frame bit for bit. (That the product's KERNELS compute what that text computes is
    tests/test_ref_parity.py and the GPU suite; this closes the loop over the host.)"""
    w, h, frames = 96, 64, 4
    seq = parity.generate_sequence(name, w, h, frames, device="cpu")
    runs = []
    for lib in (None, oracle_driver.load_ref_host()):
        run = parity.OracleRun(name, w, h)
        if lib is not None:
            run.inst = api.Instance([(0, parity.DENOISERS[name][0])], lib=lib)
        ex = oracle_driver.RefExecutor(run.inst, w, h, api.FORMAT_BYTES)
        ex.user = run.ex.user
        run.ex = ex
        runs.append(run)
    for f, frame in enumerate(seq):
        cam, cam_prev = frame["camera"], seq[max(f - 1, 0)]["camera"]
        for run in runs:
            run.step(frame, parity.common_settings(cam, cam_prev, w, h, f), parity.denoiser_settings(name, frame, None))
        for rt in runs[0].outs:
            assert np.array_equal(np.asarray(runs[0].output(rt)), np.asarray(runs[1].output(rt)), equal_nan=True), (f, rt.name)


def _instance_desc_view(inst):
    d = inst.desc
    pipes = []
    for i in range(d.pipelinesNum):
        p = d.pipelines[i]
        pipes.append((p.shaderFileName, p.shaderEntryPointName, p.hasConstantData, p.computeShaderDXBC.size, p.computeShaderDXIL.size, p.computeShaderSPIRV.size,
                      [(p.resourceRanges[k].descriptorType, p.resourceRanges[k].baseRegisterIndex, p.resourceRanges[k].descriptorsNum) for k in range(p.resourceRangesNum)]))
    pool = d.descriptorPoolDesc
    return dict(constantBufferMaxDataSize=d.constantBufferMaxDataSize, constantBufferSpaceIndex=d.constantBufferSpaceIndex, constantBufferRegisterIndex=d.constantBufferRegisterIndex,
                samplers=[d.samplers[i] for i in range(d.samplersNum)], samplersSpaceIndex=d.samplersSpaceIndex, samplersBaseRegisterIndex=d.samplersBaseRegisterIndex,
                resourcesSpaceIndex=d.resourcesSpaceIndex, pipelines=pipes,
                descriptorPool=(pool.setsMaxNum, pool.constantBuffersMaxNum, pool.samplersMaxNum, pool.texturesMaxNum, pool.storageTexturesMaxNum))


@pytest.mark.parametrize("name", list(parity.DENOISERS) + ["REFERENCE", "MIXED"])
def test_the_whole_instance_desc(name):
    """every field of nrd::InstanceDesc an integration layer allocates from: register / space indices, samplers, per-pipeline resource ranges, entry-point names, the
    descriptor-pool budget, the largest constant block"""
    ref = oracle_driver.load_ref_host()
    if name == "MIXED":
        dens = [(1, parity.DENOISERS["REBLUR_DIFFUSE_SPECULAR_OCCLUSION"][0]), (2, parity.DENOISERS["RELAX_SPECULAR"][0]), (3, parity.DENOISERS["SIGMA_SHADOW_TRANSLUCENCY"][0]), (4, api.Denoiser.REFERENCE)]
    else:
        dens = [(0, api.Denoiser.REFERENCE if name == "REFERENCE" else parity.DENOISERS[name][0])]
    va, vb = _instance_desc_view(api.Instance(dens)), _instance_desc_view(api.Instance(dens, lib=ref))
    for key in va:
        if name in KNOWN and key == "descriptorPool":
            continue  # (one texture fewer in the transient pool: see KNOWN)
        assert va[key] == vb[key], (key, va[key], vb[key])


def test_reference_quirks_switch_reproduces_what_callers_of_the_reference_observe(monkeypatch):
    """NRD_HIP_REFERENCE_QUIRKS=1: where the library corrects the reference -- the GetResourceTypeString table and the transient pool of
    REBLUR_DIFFUSE_SPECULAR_SH -- it answers what the reference answers: zero differences over the frame sequence, the InstanceDesc and the strings; without the switch the
    two KNOWN differences are back (the other tests of this file)."""
    monkeypatch.setenv("NRD_HIP_REFERENCE_QUIRKS", "1")
    ref, mine = oracle_driver.load_ref_host(), api.load_library()
    for rt in api.ResourceType:
        if rt.name != "MAX_NUM":
            assert mine.GetResourceTypeString(int(rt)) == ref.GetResourceTypeString(int(rt)), rt
    name = "REBLUR_DIFFUSE_SPECULAR_SH"
    assert _compare(name, frames=4) == []
    dens = [(0, parity.DENOISERS[name][0])]
    va, vb = _instance_desc_view(api.Instance(dens)), _instance_desc_view(api.Instance(dens, lib=ref))
    assert va == vb
    a, b = api.Instance(dens), api.Instance(dens, lib=ref)
    assert a.transient_pool == b.transient_pool and len(a.transient_pool) == 11
    # a mixed instance: the alias is the SH denoiser's alone (the planes of the pool are shared between the denoisers of an instance)
    mixed = [(1, parity.DENOISERS[name][0]), (2, parity.DENOISERS["RELAX_SPECULAR"][0]), (3, parity.DENOISERS["REBLUR_DIFFUSE"][0])]
    a, b = api.Instance(mixed), api.Instance(mixed, lib=ref)
    assert a.transient_pool == b.transient_pool and a.permanent_pool == b.permanent_pool
    monkeypatch.delenv("NRD_HIP_REFERENCE_QUIRKS")
    assert mine.GetResourceTypeString(int(api.ResourceType.IN_DIFF_CONFIDENCE)) == b"IN_DIFF_CONFIDENCE"
    assert len(api.Instance(dens).transient_pool) == 10
