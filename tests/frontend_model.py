"""TEST INFRASTRUCTURE. A float64 numpy restatement of the reference's public front-end / back-end shader functions, written from
/root/reference/Shaders/Include/NRD.hlsli (line numbers cited per function) and from nothing else -- in particular not from include/NRD.hip.h, whose
device results tests/test_frontend_header.py holds against this model. Conventions: vectors are [..., 3] arrays, an NRD_SG is a dict with the fields of
NRD.hlsli:544-552. NRD_NORMAL_ENCODING = R10G10B10A2_UNORM, NRD_ROUGHNESS_ENCODING = LINEAR (the library defaults)."""
import numpy as np

NRD_FP16_MAX = 65504.0
NRD_PI = 3.14159265358979323846
NRD_EPS = 1e-6
NRD_INF = 1e6
NRD_REJITTER_VIEWZ_THRESHOLD = 0.01
NRD_ROUGHNESS_EPS = np.sqrt(np.sqrt(NRD_EPS))
NRD_MATERIAL_FACTOR_MIN_SCALE = 0.02
NRD_ROUGHNESS_FACTOR_MIN_SCALE = 0.1


def saturate(x):
    return np.clip(x, 0.0, 1.0)


def dot(a, b):
    return np.sum(a * b, axis=-1)


def length(a):
    return np.sqrt(dot(a, a))


def normalize(a):
    return a / length(a)[..., None]


def lerp(a, b, t):
    return a + (b - a) * t


def reflect(i, n):
    return i - 2.0 * dot(n, i)[..., None] * n


def step(edge, x):
    return (x >= edge).astype(np.float64)


# ---- NRD.hlsli:322-343 oct packing
def encode_unit_vector(v, signed=False):
    v = v / np.sum(np.abs(v), axis=-1)[..., None]
    oct_wrap = (1.0 - np.abs(v[..., [1, 0]])) * (step(0.0, v[..., :2]) * 2.0 - 1.0)
    xy = np.where((v[..., 2] >= 0.0)[..., None], v[..., :2], oct_wrap)
    return xy if signed else xy * 0.5 + 0.5


def decode_unit_vector(p, signed=False, do_normalize=False):
    p = p if signed else p * 2.0 - 1.0
    n = np.concatenate([p, (1.0 - np.abs(p[..., 0]) - np.abs(p[..., 1]))[..., None]], axis=-1)
    t = saturate(-n[..., 2])
    n[..., :2] -= t[..., None] * (step(0.0, n[..., :2]) * 2.0 - 1.0)
    return normalize(n) if do_normalize else n


# ---- NRD.hlsli:350-390 colour
def luminance(c):
    return dot(c, np.array([0.2126, 0.7152, 0.0722]))


def linear_to_ycocg(c):
    return np.stack([dot(c, np.array([0.25, 0.5, 0.25])), dot(c, np.array([0.5, 0.0, -0.5])), dot(c, np.array([-0.25, 0.5, -0.25]))], axis=-1)


def ycocg_to_linear(c):
    t = c[..., 0] - c[..., 2]
    return np.maximum(np.stack([t + c[..., 1], c[..., 0] + c[..., 2], t - c[..., 1]], axis=-1), 0.0)


def ycocg_to_linear_corrected(Y, Y0, cocg):
    Y = np.maximum(Y, 0.0)
    cocg = cocg * ((Y + NRD_EPS) / (Y0 + NRD_EPS))[..., None]
    return ycocg_to_linear(np.concatenate([Y[..., None], cocg], axis=-1))


# ---- NRD.hlsli:393-413 GGX dominant direction, magic curve
def specular_dominant_factor(NoV, roughness):
    a = 0.298475 * np.log(39.4115 - 39.0029 * roughness)
    return saturate(np.power(saturate(1.0 - NoV), 10.8649) * (1.0 - a) + a)


def specular_dominant_direction(N, V, f):
    R = reflect(-V, N)
    return normalize(lerp(N, R, f[..., None]))


def spec_magic_curve(roughness):
    return 1.0 - np.exp2(-30.0 * roughness * roughness)


# ---- NRD.hlsli:416-487 BRDF terms
def pow5(x):
    return np.power(saturate(1.0 - x), 5.0)


def fresnel_term(rf0, VoNH):
    return rf0 + (1.0 - rf0) * pow5(VoNH)


def distribution_term(roughness, NoH):
    m = roughness * roughness
    m2 = m * m
    t = (NoH * m2 - NoH) * NoH + 1.0
    a = m / t
    return a * a / NRD_PI


def geometry_term(roughness, NoL, NoV):
    m = roughness * roughness
    m2 = m * m
    a = NoL + np.sqrt(saturate((NoL - m2 * NoL) * NoL + m2))
    b = NoV + np.sqrt(saturate((NoV - m2 * NoV) * NoV + m2))
    return 1.0 / np.maximum(a * b, NRD_EPS)


def diffuse_term(roughness, NoL, NoV, VoH):
    m = roughness * roughness
    f = 2.0 * VoH * VoH * m - 0.5
    return (f * pow5(NoV) + 1.0) * (f * pow5(NoL) + 1.0) / NRD_PI


def compute_brdfs(Ld, Ls, N, V, rf0, roughness):
    NoV = np.abs(dot(N, V))
    H = normalize(Ld + V)
    NoL = saturate(dot(N, Ld))
    VoH = saturate(dot(V, H))
    x = (1.0 - fresnel_term(rf0, VoH)) * diffuse_term(roughness, NoL, NoV, VoH) * NoL
    H = normalize(Ls + V)
    H = normalize(lerp(N, H, roughness[..., None]))
    NoL = saturate(dot(N, Ls))
    NoH = saturate(dot(N, H))
    VoH = saturate(dot(V, H))
    y = fresnel_term(rf0, VoH) * distribution_term(roughness, NoH) * geometry_term(roughness, NoL, NoV) * NoL
    return np.stack([x, y], axis=-1)


def environment_term_rtg(Rf0, NoV, roughness):  # NRD.hlsli:490-517
    m = saturate(roughness * roughness)
    X = np.stack([np.ones_like(NoV), NoV, NoV * NoV, NoV * NoV * NoV], axis=-1)
    Y = np.stack([np.ones_like(m), m, m * m, m * m * m], axis=-1)
    M1 = np.array([[0.99044, -1.28514], [1.29678, -0.755907]])
    M2 = np.array([[1.0, 2.92338, 59.4188], [20.3225, -27.0302, 222.592], [121.563, 626.13, 316.627]])
    M3 = np.array([[0.0365463, 3.32707], [9.0632, -9.04756]])
    M4 = np.array([[1.0, 3.59685, -1.36772], [9.04401, -16.3174, 9.22949], [5.56589, 19.7886, -20.2123]])
    mul = lambda M, v: np.einsum("ij,...j->...i", M, v)
    bias = dot(mul(M1, X[..., [0, 1]]), Y[..., [0, 1]]) / np.maximum(dot(mul(M2, X[..., [0, 1, 3]]), Y[..., [0, 1, 3]]), NRD_EPS)
    scale = dot(mul(M3, X[..., [0, 1]]), Y[..., [0, 1]]) / np.maximum(dot(mul(M4, X[..., [0, 2, 3]]), Y[..., [0, 1, 3]]), NRD_EPS)
    return saturate(Rf0 * scale[..., None] + bias[..., None])


def hit_distance_normalization(viewZ, p, roughness):  # NRD.hlsli:520-523
    return (p[0] + np.abs(viewZ) * p[1]) * lerp(1.0, p[2], saturate(np.exp2(p[3] * roughness * roughness)))


# ---- NRD.hlsli:544-592 spherical gaussians
def sg_create(radiance, direction, norm_hit_dist):
    y = linear_to_ycocg(radiance)
    return {"c0": y[..., 0], "chroma": y[..., 1:], "c1": direction * y[..., :1], "normHitDist": norm_hit_dist, "sharpness": np.zeros_like(norm_hit_dist)}


def sg_extract_direction(sg):
    return sg["c1"] / np.maximum(length(sg["c1"]), NRD_EPS)[..., None]


def sg_integral_approx(sg):
    return 2.0 * NRD_PI * (sg["c0"] / sg["sharpness"])


def sg_inner_product(a, b):
    d = length(a["sharpness"][..., None] * sg_extract_direction(a) + b["sharpness"][..., None] * sg_extract_direction(b))
    c = np.exp(d - a["sharpness"] - b["sharpness"])
    c = c * (1.0 - np.exp(-2.0 * d))
    c = c / np.maximum(d, NRD_EPS)
    return NRD_PI * saturate(2.0 * c * a["c0"]) * b["c0"]


# ---- NRD.hlsli:597-687 front end, general
def unpack_normal_and_roughness(p):
    n = decode_unit_vector(p[..., :2], False, False)
    n = n / np.sqrt(dot(n, n) + 1e-9)[..., None]  # _NRD_SafeNormalize
    return np.concatenate([n, p[..., 2:3]], axis=-1), p[..., 3] * 3.0


def pack_normal_and_roughness(N, roughness, material_id):
    return np.concatenate([encode_unit_vector(N, False), roughness[..., None], saturate(material_id / 3.0)[..., None]], axis=-1)


def store_r10g10b10a2(unorm):
    q = lambda x, m: np.floor(saturate(x) * m + 0.5).astype(np.uint64)
    return (q(unorm[..., 0], 1023.0) | (q(unorm[..., 1], 1023.0) << 10) | (q(unorm[..., 2], 1023.0) << 20) | (q(unorm[..., 3], 3.0) << 30)).astype(np.uint32)


def load_r10g10b10a2(word):
    w = word.astype(np.uint64)
    return np.stack([(w & 1023) / 1023.0, ((w >> 10) & 1023) / 1023.0, ((w >> 20) & 1023) / 1023.0, (w >> 30) / 3.0], axis=-1)


def material_factors(N, V, albedo, Rf0, roughness):
    NoV = np.abs(dot(N, V))
    Fenv = environment_term_rtg(Rf0, NoV, roughness)
    diff = lerp(NRD_MATERIAL_FACTOR_MIN_SCALE, 1.0, (1.0 - Fenv) * albedo)
    spec = Fenv * lerp(NRD_ROUGHNESS_FACTOR_MIN_SCALE, 1.0, roughness)[..., None]
    return diff, lerp(NRD_MATERIAL_FACTOR_MIN_SCALE, 1.0, spec)


# ---- NRD.hlsli:722-856 front end, per denoiser (inputs are valid: `sanitize` only clamps)
def reblur_get_norm_hit_dist(hit_dist, viewZ, p, roughness):
    return saturate(hit_dist / hit_distance_normalization(viewZ, p, roughness))


def reblur_pack_radiance_and_norm_hit_dist(radiance, norm_hit_dist):
    return np.concatenate([linear_to_ycocg(np.clip(radiance, 0.0, NRD_FP16_MAX)), saturate(norm_hit_dist)[..., None]], axis=-1)


def reblur_pack_sh(radiance, norm_hit_dist, direction):
    sg = sg_create(np.clip(radiance, 0.0, NRD_FP16_MAX), np.clip(direction, -1.0, 1.0), saturate(norm_hit_dist))
    out0 = np.concatenate([sg["c0"][..., None], sg["chroma"], sg["normHitDist"][..., None]], axis=-1)
    return out0, np.concatenate([sg["c1"], sg["sharpness"][..., None]], axis=-1)


def reblur_pack_directional_occlusion(direction, norm_hit_dist):
    nhd = saturate(norm_hit_dist)
    sg = sg_create(np.repeat(nhd[..., None], 3, axis=-1), np.clip(direction, -1.0, 1.0), nhd)
    return np.concatenate([sg["c1"], sg["c0"][..., None]], axis=-1)


def relax_pack_sh(radiance, hit_dist, direction):
    radiance, hit_dist, direction = np.clip(radiance, 0.0, NRD_FP16_MAX), np.clip(hit_dist, 0.0, NRD_FP16_MAX), np.clip(direction, -1.0, 1.0)
    return np.concatenate([radiance, hit_dist[..., None]], axis=-1), np.concatenate([direction * luminance(radiance)[..., None], np.zeros_like(hit_dist)[..., None]], axis=-1)


def sigma_pack_penumbra(distance_to_occluder, tan_of_light_angular_radius):
    return np.where(distance_to_occluder >= NRD_FP16_MAX, NRD_FP16_MAX, np.minimum(distance_to_occluder * tan_of_light_angular_radius * 0.5, 32768.0))


def sigma_pack_penumbra_local(distance_to_occluder, distance_to_light, light_size):
    size = light_size * distance_to_occluder / np.maximum(distance_to_light - distance_to_occluder, NRD_EPS)
    return np.where(distance_to_occluder >= NRD_FP16_MAX, NRD_FP16_MAX, np.minimum(size * 0.5, 32768.0))


def sigma_pack_translucency(distance_to_occluder, translucency):
    return np.concatenate([(distance_to_occluder >= NRD_FP16_MAX).astype(np.float64)[..., None], saturate(translucency)], axis=-1)


# ---- NRD.hlsli:863-931 back end
def reblur_unpack_radiance_and_norm_hit_dist(data):
    return np.concatenate([ycocg_to_linear(data[..., :3]), data[..., 3:]], axis=-1)


def unpack_sh(sh0, sh1):
    return {"c0": sh0[..., 0], "chroma": sh0[..., 1:3], "normHitDist": sh0[..., 3], "c1": sh1[..., :3], "sharpness": sh1[..., 3]}


# ---- NRD.hlsli:937-1130 resolves
def sg_extract_color(sg):
    return ycocg_to_linear(np.concatenate([sg["c0"][..., None], sg["chroma"]], axis=-1))


def sg_resolve_diffuse(sg, N):
    sg = dict(sg, sharpness=np.full_like(sg["c0"], 4.0))
    c0 = 0.36
    c1 = 1.0 / (4.0 * c0)
    e = np.exp(-sg["sharpness"])
    e2 = e * e
    r = 1.0 / sg["sharpness"]
    scale = 1.0 + 2.0 * e2 - r
    bias = (e - e2) * r - e2
    NoL = dot(N, sg_extract_direction(sg))
    x = np.sqrt(saturate(1.0 - scale))
    x0 = c0 * NoL
    x1 = c1 * x
    n = x0 + x1
    y = np.where(np.abs(x0) <= x1, n * n / x, saturate(NoL))
    Y = (scale * y + bias) * sg_integral_approx(sg)
    return ycocg_to_linear_corrected(Y, sg["c0"], sg["chroma"])


def sg_resolve_specular(sg, N, V, roughness):
    roughness = np.maximum(roughness, NRD_ROUGHNESS_EPS)
    sg = dict(sg, sharpness=np.full_like(sg["c0"], 2.0))
    H = normalize(sg_extract_direction(sg) + V)
    H = normalize(lerp(N, H, roughness[..., None]))
    m = roughness * roughness
    m2 = m * m
    ndf_c0 = 1.0 / (NRD_PI * m2) * lerp(1.0, 0.75 * 2.0 * NRD_PI, m2)
    ndf_sharpness = 2.0 / np.maximum(m2, NRD_EPS)
    warped_c1 = reflect(-V, H)
    warped_sharpness = ndf_sharpness / np.maximum(4.0 * np.abs(dot(H, V)), NRD_EPS)
    NoV = np.abs(dot(N, V))
    NoL = saturate(dot(N, warped_c1))
    warped = {"c0": ndf_c0 * NoL * geometry_term(roughness, NoL, NoV), "c1": warped_c1, "sharpness": warped_sharpness}
    Y = sg_inner_product(warped, sg)
    return ycocg_to_linear_corrected(Y, sg["c0"], sg["chroma"])


def sg_rejitter(diff_sg, spec_sg, Rf0, V, roughness, Z, Ze, Zw, Zn, Zs, N, Ne, Nw, Nn, Ns):
    roughness = np.maximum(roughness, NRD_ROUGHNESS_EPS)
    rf0 = luminance(Rf0)
    Ld, Ls = sg_extract_direction(diff_sg), sg_extract_direction(spec_sg)
    Ls = normalize(lerp(V, Ls, spec_magic_curve(roughness)[..., None]))
    center = compute_brdfs(Ld, Ls, N, V, rf0, roughness)
    average = sum(compute_brdfs(Ld, Ls, n, V, rf0, roughness) for n in (Ne, Nn, Nw, Ns))
    NoV = np.abs(dot(N, V))
    z_threshold = NRD_REJITTER_VIEWZ_THRESHOLD * np.abs(Z) / (NoV * 0.95 + 0.05)
    valid = sum(((np.abs(z - Z) < z_threshold) & (dot(n, N) > 0.0)).astype(np.int64) for z, n in ((Ze, Ne), (Zn, Nn), (Zw, Nw), (Zs, Ns)))
    f = (center * 4.0 + NRD_EPS) / (average + NRD_EPS)
    return np.where((valid != 4)[..., None], 1.0, np.clip(f, 1.0 / NRD_PI, NRD_PI))


def sh_resolve_diffuse(sh, N):
    return ycocg_to_linear_corrected(dot(N, sh["c1"]) + 0.5 * sh["c0"], sh["c0"], sh["chroma"])


def sh_resolve_specular(sh, N, V, roughness):
    NoV = np.abs(dot(N, V))
    D = specular_dominant_direction(N, V, specular_dominant_factor(NoV, roughness))
    return ycocg_to_linear_corrected(dot(D, sh["c1"]) + 0.5 * sh["c0"], sh["c0"], sh["chroma"])
