"""CPU prototype of the matrix-pipe phase 1 (kernels.h: mfmaCandidates): is the decision CONSERVATIVE, how many candidates?

Phase 1 of the all-triangles intersector only has to keep every triangle phase 2 (trace.h: intersectTriangle +
testLeafTriangle, fp32 Moeller-Trumbore) could accept.  The matrix-pipe form evaluates, per triangle, five LINEAR
forms of the ray -- the three edge functions U, V, W and det on (d, p x d), t det on (p, 1), p = o - centre -- as k-ordered
fp32 fma chains (what v_mfma_f32_32x32x2_f32 computes), each row scaled so that one tolerance serves all rows.
This script emulates both sides in numpy (fma through float64: exact product, one rounding short of a true fmaf) on
the Cornell box and on random triangle soups, and reports (a) phase-2 acceptances the phase-1 rule would lose (must
be 0), (b) candidates per ray against the VALU phase 1's.  The GPU test (tests/test_gpu_fuzz.py) is the real check;
this is where the tolerances were derived and tried first.
"""
import sys

import numpy as np

F = np.float32
U24 = 2.0 ** -24


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def xdot(a, b):
    return fma(a[..., 0], b[..., 0], fma(a[..., 1], b[..., 1], (a[..., 2] * b[..., 2]).astype(F)))


def xcross(a, b):
    return np.stack([
        fma(a[..., 1], b[..., 2], -(a[..., 2] * b[..., 1]).astype(F)),
        fma(a[..., 2], b[..., 0], -(a[..., 0] * b[..., 2]).astype(F)),
        fma(a[..., 0], b[..., 1], -(a[..., 1] * b[..., 0]).astype(F)),
    ], axis=-1)


def phase2(o, d, v0, e1, e2, tnear, tfar):
    """intersectTriangle + acceptance, rays [N,3] x triangles [T,3] -> accepted [N,T] (closest: tfar = 1e5)."""
    o = o[:, None, :]; d = d[:, None, :]
    v0 = v0[None]; e1 = e1[None]; e2 = e2[None]
    pvec = xcross(np.broadcast_to(d, (o.shape[0], v0.shape[1], 3)), np.broadcast_to(e2, (o.shape[0], v0.shape[1], 3)))
    det = xdot(np.broadcast_to(e1, pvec.shape), pvec)
    tvec = (o - v0).astype(F)
    us = xdot(tvec, pvec)
    qvec = xcross(tvec, np.broadcast_to(e1, tvec.shape))
    vs = xdot(np.broadcast_to(d, qvec.shape), qvec)
    front = (det > 0) & (us >= 0) & (vs >= 0) & ((us + vs).astype(F) <= det)
    back = (det < 0) & (us <= 0) & (vs <= 0) & ((us + vs).astype(F) >= det)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (F(1) / det).astype(F)
        t = (xdot(np.broadcast_to(e2, qvec.shape), qvec) * inv).astype(F)
    ok = (front | back) & (t > F(tnear)) & (t <= tfar[:, None])
    return ok


def build_rows(v0, e1, e2, centre, radius):
    """Row tables in float64 -> fp32: edge rows [T,4,6] on (d, q), t rows [T,4] on (p, 1)."""
    v0 = v0.astype(np.float64); e1 = e1.astype(np.float64); e2 = e2.astype(np.float64)
    a = v0 - centre
    na = np.linalg.norm(a, axis=1); n1 = np.linalg.norm(e1, axis=1); n2 = np.linalg.norm(e2, axis=1)
    tiny = 1e-30
    sU = np.maximum((radius + na) * n2, tiny)
    sV = np.maximum((radius + na) * n1, tiny)
    sW = np.maximum((radius + na) * (n1 + n2) + n1 * n2, tiny)
    sT = np.maximum((radius + na + 1.0) * n1 * n2, tiny)
    lU, mU = e2, np.cross(a, e2)
    lV, mV = -e1, np.cross(e1, a)
    mD = np.cross(e2, e1)
    lW, mW = -(lU + lV), mD - mU - mV
    n = np.cross(e1, e2)
    edge = np.zeros((v0.shape[0], 4, 6))
    edge[:, 0, :3] = mU / sU[:, None]; edge[:, 0, 3:] = lU / sU[:, None]
    edge[:, 1, :3] = mV / sV[:, None]; edge[:, 1, 3:] = lV / sV[:, None]
    edge[:, 2, :3] = mW / sW[:, None]; edge[:, 2, 3:] = lW / sW[:, None]
    edge[:, 3, :3] = mD / sT[:, None]
    trow = np.zeros((v0.shape[0], 4))
    trow[:, :3] = n / sT[:, None]
    trow[:, 3] = -np.einsum("ij,ij->i", a, n) / sT
    return edge.astype(F), trow.astype(F)


def chain(rows, vec):
    """k-ordered fma chain: rows [T,K] x vec [N,K] -> [N,T] (the MFMA's arithmetic)."""
    acc = np.zeros((vec.shape[0], rows.shape[0]), dtype=F)
    for k in range(rows.shape[1]):
        acc = fma(np.broadcast_to(rows[None, :, k], acc.shape), np.broadcast_to(vec[:, None, k], acc.shape), acc)
    return acc


GAMMA = F(24 * U24)      # edge rows
GAMMA_T = F(20 * U24)    # near test
TINY_DET = F(10 * U24)   # below this the sign of det' is not trusted: interval tests do not reject


def phase1_mfma(o, d, tfar, edge, trow, centre, tnear, far):
    p = (o - centre.astype(F)).astype(F)
    q = xcross(p, d)
    ray6 = np.concatenate([d, q], axis=1)
    ray4 = np.concatenate([p, np.ones((p.shape[0], 1), dtype=F)], axis=1)
    Uv = chain(edge[:, 0], ray6); Vv = chain(edge[:, 1], ray6); Wv = chain(edge[:, 2], ray6); Dv = chain(edge[:, 3], ray6)
    Tv = chain(trow, ray4)
    lo = np.minimum(np.minimum(Uv, Vv), Wv); hi = np.maximum(np.maximum(Uv, Vv), Wv)
    inside = (lo >= -GAMMA) | (hi <= GAMMA)
    sign = np.where(np.signbit(Dv), F(-1), F(1))
    tnl = F(0.5 * tnear)
    x = fma(np.full_like(Dv, -tnl), Dv, Tv)
    near = (x * sign) >= -GAMMA_T
    keep_t = near
    if far:
        tfh = np.minimum((tfar + F(0.5 * tnear) + F(1e-5) * np.abs(tfar)).astype(F), F(1e30))
        w = fma(np.broadcast_to(tfh[:, None], Dv.shape), Dv, -Tv)
        gamma_f = fma(tfh, np.full_like(tfh, F(10 * U24)), np.full_like(tfh, F(16 * U24)))
        keep_t = keep_t & ((w * sign) >= -gamma_f[:, None])
    keep_t = keep_t | (np.abs(Dv) <= TINY_DET)
    return inside & keep_t, (Uv, Vv, Wv, Dv, Tv)


def phase1_valu(o, d, v0, e1, e2, tnear, tfar, far):
    """the VALU phase 1 (kernels.h: smallCandidates) for comparison of candidate counts"""
    o = o[:, None, :]; d = d[:, None, :]
    shape = (o.shape[0], v0.shape[0], 3)
    pvec = xcross(np.broadcast_to(d, shape), np.broadcast_to(e2[None], shape))
    det = xdot(np.broadcast_to(e1[None], shape), pvec)
    tvec = (o - v0[None]).astype(F)
    us = xdot(tvec, pvec)
    qvec = xcross(tvec, np.broadcast_to(e1[None], shape))
    vs = xdot(np.broadcast_to(d, shape), qvec)
    ts = xdot(np.broadcast_to(e2[None], shape), qvec)
    with np.errstate(over="ignore", invalid="ignore"):
        a = us * det; b = vs * det; c = (det - (us + vs)) * det
        e = fma(np.full_like(det, F(-0.5 * tnear)), det, ts) * det
        worst = np.minimum(np.minimum(a, b), np.minimum(c, e))
        if far:
            tfh = (tfar + F(0.5 * tnear) + F(1e-5) * np.abs(tfar)).astype(F)
            g = fma(np.broadcast_to(tfh[:, None], det.shape), det, -ts) * det
            worst = np.minimum(worst, g)
    return ~(worst < 0)


def cornell():
    verts = []; faces = []
    for line in open("scenes/CornellBox-Original.obj"):
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "v":
            verts.append([float(x) for x in parts[1:4]])
        elif parts[0] == "f":
            idx = [int(x.split("/")[0]) for x in parts[1:]]
            idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
            for k in range(1, len(idx) - 1):
                faces.append([idx[0], idx[k], idx[k + 1]])
    verts = np.array(verts, dtype=F); faces = np.array(faces)
    return verts[faces[:, 0]], (verts[faces[:, 1]] - verts[faces[:, 0]]).astype(F), (verts[faces[:, 2]] - verts[faces[:, 0]]).astype(F)


def soup(rng, n):
    scale = 10.0 ** rng.uniform(-1.0, 2.0)
    centres = rng.normal(size=(n, 3)) * scale
    spans = scale * 10.0 ** rng.uniform(-2.0, -0.3, size=(n, 1, 1))
    corners = centres[:, None, :] + rng.normal(size=(n, 3, 3)) * spans
    corners[0:3, 2] = corners[0:3, 1]
    corners[3:6, 2] = 0.5 * (corners[3:6, 0] + corners[3:6, 1])
    corners[6:10] = corners[10:14]
    corners[14:18, 1] = corners[14:18, 0] + (corners[14:18, 1] - corners[14:18, 0]) * 1e-4
    corners = corners.astype(F)
    return corners[:, 0], (corners[:, 1] - corners[:, 0]).astype(F), (corners[:, 2] - corners[:, 0]).astype(F), scale


def rays_for(rng, v0, e1, e2, camera, n):
    """origins on random triangles (the path's vertices) or at the camera; unit directions; half aimed at triangle points"""
    tri = rng.integers(0, v0.shape[0], size=n)
    r1 = rng.random(n); r2 = rng.random(n)
    edge_case = rng.random(n) < 0.2        # a fifth on an edge or a corner of the triangle
    r2 = np.where(edge_case, 0.0, r2)
    r1 = np.where(edge_case & (rng.random(n) < 0.3), 0.0, r1)
    a = 1 - np.sqrt(r1); b = np.sqrt(r1) * (1 - r2)
    origin = v0[tri] + e1[tri] * a[:, None].astype(F) * 0 + (e1[tri].astype(np.float64) * b[:, None] + e2[tri].astype(np.float64) * (1 - a - b)[:, None])
    from_camera = rng.random(n) < 0.15
    origin = np.where(from_camera[:, None], camera[None], origin).astype(F)
    direction = rng.normal(size=(n, 3))
    # aim half of the rays at a point of a random triangle (often an edge / corner): the cases where rounding decides
    target_tri = rng.integers(0, v0.shape[0], size=n)
    s1 = rng.random(n); s2 = rng.random(n)
    on_edge = rng.random(n) < 0.5
    s2 = np.where(on_edge, 0.0, s2)
    s1 = np.where(on_edge & (rng.random(n) < 0.3), 1.0, s1)
    aa = 1 - np.sqrt(s1); bb = np.sqrt(s1) * (1 - s2)
    target = v0[target_tri].astype(np.float64) + e1[target_tri].astype(np.float64) * bb[:, None] + e2[target_tri].astype(np.float64) * (1 - aa - bb)[:, None]
    aimed = rng.random(n) < 0.5
    direction = np.where(aimed[:, None], target - origin, direction)
    length = np.linalg.norm(direction, axis=1, keepdims=True)
    direction = np.where(length > 0, direction / np.maximum(length, 1e-30), [[0.0, 0.0, 1.0]])
    dist = np.linalg.norm(target - origin, axis=1)
    tfar = np.where(aimed, dist - 1e-3, rng.choice([1e4, 3e38, 2.0], size=n))
    # in-plane rays: directions inside the plane of the origin's triangle (det ~ 0)
    grazing = rng.random(n) < 0.05
    inplane = e1[tri].astype(np.float64) * rng.normal(size=(n, 1)) + e2[tri].astype(np.float64) * rng.normal(size=(n, 1))
    l2 = np.linalg.norm(inplane, axis=1, keepdims=True)
    inplane = np.where(l2 > 0, inplane / np.maximum(l2, 1e-30), [[1.0, 0.0, 0.0]])
    direction = np.where(grazing[:, None], inplane, direction)
    return origin.astype(F), direction.astype(F), np.maximum(tfar, 0).astype(F)


def run(name, v0, e1, e2, camera, rng, n_rays, tnear=1e-3):
    pts = np.concatenate([v0, v0 + e1, v0 + e2, camera[None]]).astype(np.float64)
    lo = pts.min(axis=0); hi = pts.max(axis=0)
    centre = (0.5 * (lo + hi)).astype(F).astype(np.float64)
    radius = float(np.linalg.norm(pts - centre, axis=1).max()) * 1.0001
    edge, trow = build_rows(v0, e1, e2, centre, radius)
    o, d, tfar = rays_for(rng, v0, e1, e2, camera.astype(F), n_rays)
    out = {}
    for far in (False, True):
        tf = tfar if far else np.full_like(tfar, F(1e5))
        acc = phase2(o, d, v0, e1, e2, tnear, tf)
        mine, _ = phase1_mfma(o, d, tf, edge, trow, centre, tnear, far)
        valu = phase1_valu(o, d, v0, e1, e2, tnear, tf, far)
        lost = int((acc & ~mine).sum())
        lost_valu = int((acc & ~valu).sum())
        out[far] = (lost, lost_valu, acc.sum() / n_rays, mine.sum() / n_rays, valu.sum() / n_rays)
    print("%-22s tris %3d  closest: lost %d (valu %d) accepted/ray %.3f  candidates/ray mfma %.3f valu %.3f | any-hit: lost %d (valu %d) accepted %.3f  mfma %.3f valu %.3f"
          % (name, v0.shape[0], *out[False], *out[True]))
    return out[False][0] + out[True][0]


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    lost = 0
    v0, e1, e2 = cornell()
    lost += run("cornell", v0, e1, e2, np.array([0.0, 1.0, 6.8]), rng, n_rays)
    for k in range(12):
        n = int(rng.integers(36, 61))
        v0, e1, e2, scale = soup(rng, n)
        lost += run("soup %d scale %.3g" % (k, scale), v0, e1, e2, np.array([0.0, 0.3 * scale, 3.0 * scale]), rng, n_rays)
    print("LOST TOTAL", lost)
    return 1 if lost else 0


if __name__ == "__main__":
    sys.exit(main())
