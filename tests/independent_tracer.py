"""An INDEPENDENT restatement of the bounce loop, used only to cross-check oracle/pt_oracle.c (the oracle
and the device headers are siblings, so a shared misreading of the shaders would pass every GPU-vs-oracle test).

Written from the reference's shader text alone -- Shaders/Raytracing.hlsl:103-415 (DEFAULT permutation), Shaders/BxDF.hlsli:21-315,
SurfaceVectors.hlsli, HitInfo.hlsli:60-64,96-99, Camera.hlsli:27-41, Math.hlsli:7-15, ShadingHelpers.hlsli:11-30 -- plus SURVEY.md
Appendix A for the un-vendored MathLib bodies and DESIGN.md S3 for the build-defined ray-sphere hit.  It shares NO code with the
oracle or the kernels and deliberately differs from them in everything that is not the estimator itself:

  * double precision throughout, libm transcendentals (no fma placement, no polynomial sincos / pow / sky fits),
  * the literal formulas of the shader text (f and pdf evaluated separately and divided; textbook GGX D; HLSL lerp / pow),
  * brute-force closest hit, a flat procedural style (dicts and tuples) instead of the oracle's structs.

What has to agree with the oracle EXACTLY: the integer RNG stream (state after every bounce), every hit id, every lobe choice and
every termination reason; throughput, hit distance and radiance agree to rounding (1e-3 relative is asserted; typical 1e-6).
tests/test_independent_tracer.py compares per event against oracle_trace_pixel and keeps the traces as fixtures."""
import math

MISS = 0xFFFFFFFF
PI = math.pi


# ---------------------------------------------------------------- Rng::Hash (call sites Raytracing.hlsl:108,330,351)
def _hash(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def rng_seed(px, py, frame):
    seed = _hash((frame + 0x035F9F29) & 0xFFFFFFFF)
    v = ((px << 16) | py) & 0xFFFFFFFF
    return (seed ^ ((_hash(v) + 0x9E3779B9 + ((seed << 6) & 0xFFFFFFFF) + (seed >> 2)) & 0xFFFFFFFF)) & 0xFFFFFFFF


class Stream:
    def __init__(self, state):
        self.state = state

    def uint(self):
        self.state = _hash(self.state)
        return self.state

    def unit(self):
        """GetFloat = 2 - asfloat((u >> 9) | 0x3F800000): (0, 1], 23 bits"""
        mantissa = self.uint() >> 9
        return 2.0 - (1.0 + mantissa / 8388608.0)


# ---------------------------------------------------------------- small vector helpers (tuples of 3 floats)
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def mul(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def scale(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def neg(a): return (-a[0], -a[1], -a[2])
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def unit(a): return scale(a, 1.0 / math.sqrt(dot(a, a)))
def lum(c): return 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2]
def saturate(x): return 0.0 if not (x > 0.0) else (1.0 if x > 1.0 else x)
def finite3(a): return all(math.isfinite(x) for x in a)


def sign_ml(x):  # Math::Sign: Sign(0) = +1
    return 1.0 if x >= 0.0 else -1.0


# ---------------------------------------------------------------- MathLib
def get_basis(n):
    sz = sign_ml(n[2])
    a = 1.0 / (sz + n[2])
    ya = n[1] * a
    b = n[0] * ya
    c = n[0] * sz
    t = (c * n[0] * a - 1.0, sz * b, c)
    bt = (b, n[1] * ya - sz, n[1])
    return t, bt, n


def to_local(basis, v): return (dot(basis[0], v), dot(basis[1], v), dot(basis[2], v))           # RotateVector
def to_world(basis, v): return add(add(scale(basis[0], v[0]), scale(basis[1], v[1])), scale(basis[2], v[2]))  # RotateVectorInverse


def cosine_ray(u):
    phi = 2.0 * PI * u[0]
    ct = math.sqrt(saturate(u[1]))
    st = math.sqrt(saturate(1.0 - ct * ct))
    return (st * math.cos(phi), st * math.sin(phi), ct)


def vndf_ray(u, rough, vl):
    m = rough * rough
    vh = unit((m * vl[0], m * vl[1], vl[2]))
    phi = 2.0 * PI * u[0]
    z = (1.0 - u[1]) * (1.0 + vh[2]) - vh[2]
    s = math.sqrt(saturate(1.0 - z * z))
    nh = add((s * math.cos(phi), s * math.sin(phi), z), vh)
    return unit((m * nh[0], m * nh[1], max(nh[2], 1e-7)))


def ggx_d(rough, noh):
    m = rough * rough
    m2 = m * m
    t = (noh * m2 - noh) * noh + 1.0
    return m2 / (PI * t * t)


def vndf_pdf(vl, noh, rough):
    m = rough * rough
    nov = abs(vl[2])
    g1 = 2.0 * nov / (nov + math.sqrt(m * m + (1.0 - m * m) * nov * nov))
    return ggx_d(rough, noh) * g1 / (4.0 * nov)


def geometry_mod(rough, nol, nov):
    m = rough * rough
    m2 = m * m
    a = nov * math.sqrt(saturate((nol - m2 * nol) * nol + m2))
    b = nol * math.sqrt(saturate((nov - m2 * nov) * nov + m2))
    return 0.5 / (a + b)


def schlick(f0, voh):
    k = (1.0 - voh) ** 5
    return tuple(f + (1.0 - f) * k for f in f0)


def fresnel_dielectric(eta, von):
    sa2 = eta * eta * (1.0 - von * von)
    ca = math.sqrt(saturate(1.0 - sa2))
    rs = (eta * von - ca) / (eta * von + ca)
    rp = (eta * ca - von) / (eta * ca + von)
    return 0.5 * (rs * rs + rp * rp)


def burley(rough, nol, nov, voh):
    f = 2.0 * voh * voh * rough - 0.5
    return (1.0 + f * (1.0 - nol) ** 5) * (1.0 + f * (1.0 - nov) ** 5) / PI


def env_term_rtg(f0, nov, rough):
    m = rough * rough
    x = (1.0, nov, nov * nov, nov ** 3)
    y = (1.0, m, m * m, m ** 3)
    m1 = ((0.99044, -1.28514), (1.29678, -0.755907))
    m2 = ((1.0, 2.92338, 59.4188), (20.3225, -27.0302, 222.592), (121.563, 626.13, 316.627))
    m3 = ((0.0365463, 3.32707), (9.0632, -9.04756))
    m4 = ((1.0, 3.59685, -1.36772), (9.04401, -16.3174, 9.22949), (5.56589, 19.7886, -20.2123))
    xy, xyw, xzw, yxy, yxyw = (x[0], x[1]), (x[0], x[1], x[3]), (x[0], x[2], x[3]), (y[0], y[1]), (y[0], y[1], y[3])

    def bil(mat, a, b):  # dot(mat * a, b)
        return sum(sum(mat[r][c] * a[c] for c in range(len(a))) * b[r] for r in range(len(b)))

    bias = bil(m1, xy, yxy) / bil(m2, xyw, yxyw)
    sc = bil(m3, xy, yxy) / bil(m4, xzw, yxyw)
    return tuple(saturate(f * sc + bias) for f in f0)


def from_srgb(c):
    c = saturate(c)
    return c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4


def reflect(i, n): return sub(i, scale(n, 2.0 * dot(n, i)))


def refract(i, n, eta):
    ni = dot(n, i)
    k = 1.0 - eta * eta * (1.0 - ni * ni)
    if k < 0.0:
        return (0.0, 0.0, 0.0)
    return sub(scale(i, eta), scale(n, eta * ni + math.sqrt(k)))


# ---------------------------------------------------------------- geometry: analytic spheres, brute force
def hit_sphere(o, d, tmin, tmax, c, r):
    f = sub(o, c)
    bp = -dot(f, d)
    l = add(f, scale(d, bp))
    disc = r * r - dot(l, l)
    if not (disc >= 0.0):
        return None
    sq = math.sqrt(disc)
    q = bp + (sq if bp >= 0.0 else -sq)
    if q == 0.0:
        return None
    ta, tb = (dot(f, f) - r * r) / q, q
    t0, t1 = (ta, tb) if ta < tb else (tb, ta)
    t = t0 if t0 > tmin else t1
    return t if (t > tmin and t < tmax) else None


def surface_crossings(o, d, c, r):
    """both parameters at which the line meets the sphere, ascending (the textbook quadratic, unit d), or () -- the candidates
    of a non-opaque object, which the reference meets as the front and the back triangles of its mesh"""
    f = sub(o, c)
    b = dot(f, d)
    disc = b * b - (dot(f, f) - r * r)
    if not (disc >= 0.0):
        return ()
    sq = math.sqrt(disc)
    return (-b - sq, -b + sq)


def is_opaque(mat):
    """IsOpaque (ShadingHelpers.hlsli:105-115) without a base-colour map: BaseColor.a >= AlphaCutoff"""
    return mat["BaseColor"][3] >= mat["AlphaCutoff"]


def cast_ray(spheres, o, d, tmin, tmax, materials=None):
    """closest hit, ties to the lowest id (ascending scan, strict <); returns the HitInfo the loop needs or None.
    An object whose AlphaMode is not Opaque is non-opaque geometry (Scene.ixx:242-243): each of its candidates is committed only
    if IsOpaque accepts it (RaytracingHelpers.hlsli:19-43)."""
    best, best_id = tmax, MISS
    for i, (cx, cy, cz, r) in enumerate(spheres):
        if materials is not None and materials[i].get("AlphaMode", 0) != 0:
            t = None
            for candidate in surface_crossings(o, d, (cx, cy, cz), r):   # the nearer candidate first
                if candidate > tmin and is_opaque(materials[i]):
                    t = candidate
                    break
        else:
            t = hit_sphere(o, d, tmin, math.inf, (cx, cy, cz), r)
        if t is not None and t < best:
            best, best_id = t, i
    if best_id == MISS:
        return None
    cx, cy, cz, r = spheres[best_id]
    c = (cx, cy, cz)
    n = unit(sub(add(o, scale(d, best)), c))
    p = add(c, scale(n, r))
    front = dot(n, d) < 0.0                                      # HitInfo.hlsli:47
    return {"id": best_id, "t": best, "P": p, "N": n, "front": front, "Ns": n if front else neg(n),  # :60-64
            "offset": 2.0 ** -16 * max(abs(p[0]), abs(p[1]), abs(p[2]), r)}


def safe_origin(hit, L):                                         # HitInfo.hlsli:96-99
    return add(hit["P"], scale(hit["N"], hit["offset"] * sign_ml(dot(L, hit["N"]))))


# ---------------------------------------------------------------- BSDF (Shaders/BxDF.hlsli)
def bsdf_of(mat, front, primary):
    base = tuple(mat["BaseColor"][:3])
    metallic, rough, ior = mat["Metallic"], mat["Roughness"], mat["IOR"]
    trans = mat["Transmission"]
    if primary and not (metallic < 1.0):                          # Raytracing.hlsl:148
        trans = 0.0
    iori, ioro = (1.0, ior) if front else (ior, 1.0)              # BxDF.hlsli:57-63
    f0d = ((iori - ioro) / (iori + ioro)) ** 2
    return {"base": base, "metallic": metallic, "albedo": scale(base, 1.0 - metallic), "rough": max(2e-3, rough), "iori": iori, "ioro": ioro,
            "f0": tuple(f0d + (b - f0d) * metallic for b in base), "trans": trans}


def lobe_weights(b, sv, V):                                       # :184-196, :21-34
    nov = abs(dot(sv["Ns"], V))
    wt = b["trans"] * (1.0 - b["metallic"])
    wr = 1.0 - wt
    fe = env_term_rtg(b["f0"], nov, b["rough"])
    diffuse = lum(mul(b["albedo"], tuple(1.0 - f for f in fe)))
    specular = lum(fe)
    total = diffuse + specular
    pd = diffuse / total if total > 0.0 else 1.0
    if 0.0 < pd < 1.0:
        pd = min(max(pd, 0.05), 0.95)
    return (pd * wr, (1.0 - pd) * wr, wt)


def find_lobe(w, r):                                              # :198-212
    lobe, weight = 3, 0.0
    while True:
        lobe -= 1
        if not lobe > 0:
            break
        weight += w[lobe]
        if r < weight:
            break
    return lobe


def sample(b, sv, V, w, rnd):                                     # :214-226
    lobe = find_lobe(w, rnd[0])
    basis = sv["basis"]
    if lobe == 0:                                                 # :81-86
        L = to_world(basis, cosine_ray((rnd[1], rnd[2])))
        return dot(sv["FrontNg"], L) > 0.0, L, lobe
    H = to_world(basis, vndf_ray((rnd[1], rnd[2]), b["rough"], to_local(basis, V)))
    if lobe == 1:                                                 # :110-118
        L = reflect(neg(V), H)
        return dot(sv["FrontNg"], L) > 0.0, L, lobe
    voh, eta = abs(dot(V, H)), b["iori"] / b["ioro"]              # :148-170
    if eta * eta * (1.0 - voh * voh) > 1.0 or rnd[3] < fresnel_dielectric(eta, voh):
        L = reflect(neg(V), H)
    else:
        L = refract(neg(V), H, eta)
        if not finite3(L):
            L = neg(V)
    return True, L, lobe


def half_vector(b, sv, L, V, transmissive):                       # :228-245
    N = sv["FrontNg"]
    if transmissive and dot(N, L) < 0.0:
        H = unit(add(scale(L, b["ioro"]), scale(V, b["iori"])))
        return neg(H) if dot(N, H) < 0.0 else H
    return unit(add(L, V))


def pdf_of(b, sv, L, V, w, lobe):                                 # :287-299
    H = half_vector(b, sv, L, V, w[2] > 0.0)
    N = sv["Ns"]
    if lobe == 0:                                                 # :88-97
        return (abs(dot(N, L)) / PI if dot(sv["FrontNg"], L) > 0.0 else 0.0) * w[0]
    if lobe == 1:                                                 # :120-131
        if dot(sv["FrontNg"], L) > 0.0:
            return vndf_pdf(to_local(sv["basis"], V), abs(dot(N, H)), b["rough"]) * w[1]
        return 0.0
    return abs(dot(N, L)) * w[2]                                  # :172-177


def eval_of(b, sv, L, V, w, lobe):                                # :301-315
    H = half_vector(b, sv, L, V, w[2] > 0.0)
    N = sv["Ns"]
    if lobe == 2:                                                 # :179-182
        return scale(b["base"], abs(dot(N, L)) * w[2])
    wr = 1.0 - w[2]
    if not dot(sv["FrontNg"], L) > 0.0:
        return (0.0, 0.0, 0.0)
    nol, nov, voh = abs(dot(N, L)), abs(dot(N, V)), abs(dot(V, H))
    if lobe == 0:                                                 # :99-108
        return scale(b["albedo"], nol * burley(b["rough"], nol, nov, voh) * wr)
    noh = abs(dot(N, H))                                          # :133-146
    k = nol * ggx_d(b["rough"], noh) * geometry_mod(b["rough"], nol, nov)
    return scale(schlick(b["f0"], voh), k * wr)


# ---------------------------------------------------------------- environment (ShadingHelpers.hlsli:11-30, no texture)
def environment(env_color, d):
    if env_color[3] >= 0.0:
        return tuple(env_color[:3])
    t = (d[1] + 1.0) * 0.5
    return tuple(from_srgb(1.0 + (c - 1.0) * t) for c in (0.5, 0.7, 1.0))


# ---------------------------------------------------------------- the pixel (Raytracing.hlsl:103-415, GBufferGeneration.hlsl:117-232)
def trace_pixel(spheres, materials, env_color, cam, w, h, frame, bounces, spp, rr, threshold, px, py):
    """spheres: [(cx, cy, cz, r)], materials: [dict], cam: dict(Position, Right, Up, Forward, Near, Far, Jitter).
    Returns (rgb, events); event = dict(sample, bounce, id, t, L, T, rng, lobe, flag) -- flag as oracle_trace_pixel:
    0 continues, 1 primary miss, 2 bounce miss, 3 sample failed, 4 pdf 0, 5 f 0, 6 roulette, 7 throughput cut-off."""
    rng = Stream(rng_seed(px, py, frame))                         # :108
    u = (px + 0.5 + cam["Jitter"][0]) / w                          # Math.hlsli:7-10
    v = (py + 0.5 + cam["Jitter"][1]) / h
    ndc = (u * 2.0 - 1.0, v * -2.0 + 1.0)                          # :12-15
    d0 = unit(add(add(scale(cam["Right"], ndc[0]), scale(cam["Up"], ndc[1])), cam["Forward"]))  # Camera.hlsli:27-41
    inv_cos = 1.0 / dot(unit(cam["Forward"]), d0)
    o0 = tuple(cam["Position"])
    events = []
    primary = cast_ray(spheres, o0, d0, cam["Near"] * inv_cos, cam["Far"] * inv_cos, materials)
    if primary is None:                                           # GBufferGeneration.hlsl:223-227; Raytracing.hlsl:249-252
        events.append({"sample": 0, "bounce": 0, "id": MISS, "t": math.inf, "L": (0, 0, 0), "T": (1, 1, 1), "rng": rng.state, "lobe": -1, "flag": 1})
        return environment(env_color, d0), events
    radiance = (0.0, 0.0, 0.0)
    for s in range(spp):                                          # :191
        o, d, hit = o0, d0, primary
        T, sample_radiance, L, lobe = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), -1
        for bounce in range(bounces + 1):                         # :213
            if bounce:
                o, d = safe_origin(hit, L), L                     # :219-224
                hit = cast_ray(spheres, o, d, 0.0, math.inf, materials)
            ev = {"sample": s, "bounce": bounce, "id": MISS if hit is None else hit["id"], "t": math.inf if hit is None else hit["t"], "lobe": lobe}
            if hit is None:                                       # :242-259
                sample_radiance = add(sample_radiance, mul(T, environment(env_color, d)))
                events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 2})
                break
            mat = materials[hit["id"]]
            b = bsdf_of(mat, hit["front"], bounce == 0)
            emission = scale(tuple(mat["EmissiveColor"]), mat["EmissiveStrength"])
            sample_radiance = add(sample_radiance, mul(T, emission))  # :320
            sv = {"FrontNg": hit["N"] if hit["front"] else neg(hit["N"]), "Ns": hit["Ns"], "basis": get_basis(hit["Ns"])}  # SurfaceVectors.hlsli
            V = neg(d)
            w = lobe_weights(b, sv, V)
            rnd = (rng.unit(), rng.unit(), rng.unit(), rng.unit())   # :330
            ok, L, lobe = sample(b, sv, V, w, rnd)
            ev["lobe"] = lobe
            if not ok:
                events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 3}); break
            pdf = pdf_of(b, sv, L, V, w, lobe)
            if pdf == 0.0:
                events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 4}); break
            f = eval_of(b, sv, L, V, w, lobe)
            if f == (0.0, 0.0, 0.0):
                events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 5}); break
            T = mul(T, scale(f, 1.0 / pdf))                        # :346
            if rr and bounce > 3:                                  # :348-356
                p = max(T)
                if rng.unit() >= p:
                    events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 6}); break
                T = scale(T, 1.0 / p)
            if lum(T) <= threshold:                                # :361
                events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 7}); break
            events.append({**ev, "L": L, "T": T, "rng": rng.state, "flag": 0})
        radiance = add(radiance, sample_radiance)                 # :373
    radiance = scale(radiance, 1.0 / spp) if finite3(radiance) else (0.0, 0.0, 0.0)   # :378
    return radiance, events
