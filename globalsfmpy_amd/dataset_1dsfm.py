"""1DSfM-format datasets on the Python side: a writer for synthetic scenes (list.txt, cc.txt, coords.txt, tracks.txt,
EGs.txt as thirdparty/TheiaSfM/src/theia/io/read_1dsfm.cc:93-372 reads them) and a numpy reader that flattens the matched
features of every epipolar geometry the way `CalcCovariance` does on the host (src/uncertainty.cpp:3-33,99-123).
Used by the tests and examples; the product path for real datasets is the C++ `GlobalSfMpy.CalcCovariance`."""
import os

import numpy as np

from . import synth

_S = np.diag([1.0, -1.0, -1.0])  # bundler <-> theia axes (read_1dsfm.cc:307-308)


def _look_at(center, target, rng, jitter):
    z = target - center
    z /= np.linalg.norm(z)
    x = np.cross([0.0, 1.0, 0.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])  # world -> camera
    dq = synth.quat_to_matrix(synth.aa_to_quat(jitter * rng.standard_normal((1, 3))))[0]
    return dq @ R


def _synthetic_scene(rng, total, n_points, p_visible, noise_px):
    """Cameras on an arc looking at a point cloud; every point seen by a random subset of the cameras."""
    ang = np.linspace(-0.9, 0.9, total) + 0.03 * rng.standard_normal(total)
    centers = np.c_[9.0 * np.sin(ang), 0.8 * rng.standard_normal(total), -9.0 * np.cos(ang)]
    R = np.stack([_look_at(centers[k], 0.3 * rng.standard_normal(3), rng, 0.05) for k in range(total)])
    focal = rng.uniform(900, 1500, total)
    pp = np.c_[rng.integers(500, 700, total), rng.integers(350, 450, total)].astype(np.float64)  # floats survive the %f header exactly
    X = np.c_[rng.uniform(-2.5, 2.5, n_points), rng.uniform(-2, 2, n_points), rng.uniform(-2, 2, n_points)]
    keypoints = [[] for _ in range(total)]
    tracks = []
    for p in range(n_points):
        seen = np.flatnonzero(rng.uniform(size=total) < p_visible)
        obs = []
        for k in seen:
            Xc = R[k] @ (X[p] - centers[k])
            if Xc[2] <= 0.5:
                continue
            uv = focal[k] * Xc[:2] / Xc[2] + pp[k] + noise_px * rng.standard_normal(2)
            obs.append((int(k), len(keypoints[k])))
            keypoints[k].append(uv)
        if len(obs) >= 2:
            tracks.append(obs)
    return centers, R, focal, pp, keypoints, tracks


def write_synthetic_dataset(path, n_cams=12, n_points=600, seed=0, p_visible=0.6, noise_px=0.5, min_common=15, exif_every=2,
                            rel_rot_noise=0.01, rel_pos_noise=0.02, outside_cc=1):
    """A synthetic scene in 1DSfM format.  Returns the ground truth.  `outside_cc` extra views are listed in
    list.txt/coords.txt but left out of cc.txt (the reader must drop them)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    os.makedirs(path, exist_ok=True)
    total = n_cams + outside_cc
    centers, R, focal, pp, keypoints, tracks = _synthetic_scene(rng, total, n_points, p_visible, noise_px)
    in_cc = np.arange(total) < n_cams
    with open(os.path.join(path, "cc.txt"), "w") as f:
        f.write("\n".join(str(k) for k in range(total) if in_cc[k]) + "\n")
    with open(os.path.join(path, "list.txt"), "w") as f:
        for k in range(total):
            f.write("images/img%04d.jpg" % k + (" 0 %.17g" % focal[k] if k % exif_every == 0 else "") + "\n")
    with open(os.path.join(path, "coords.txt"), "w") as f:
        for k in range(total):
            f.write("#index = %d, name = img%04d.jpg, keys = %d, px = %.1f, py = %.1f, focal = %.3f\n" % (k, k, len(keypoints[k]), pp[k, 0], pp[k, 1], focal[k]))
            for n, uv in enumerate(keypoints[k]):
                f.write("%d %.17g %.17g 0 0 128 128 128\n" % (n, uv[0], uv[1]))
    with open(os.path.join(path, "tracks.txt"), "w") as f:
        f.write("%d\n" % len(tracks))
        for t in tracks:
            f.write("%d %s\n" % (len(t), " ".join("%d %d" % o for o in t)))
    common = {}
    for t in tracks:
        for a in range(len(t)):
            for b in range(a + 1, len(t)):
                common[(t[a][0], t[b][0])] = common.get((t[a][0], t[b][0]), 0) + 1
    edges = sorted(k for k, c in common.items() if c >= min_common)
    with open(os.path.join(path, "EGs.txt"), "w") as f:
        for (i, j) in edges:
            noise = synth.quat_to_matrix(synth.aa_to_quat(rel_rot_noise * rng.standard_normal((1, 3))))[0]
            Rij = noise @ R[j] @ R[i].T                         # theia convention, camera i -> camera j
            pos = R[i] @ (centers[j] - centers[i])
            pos = pos / np.linalg.norm(pos) + rel_pos_noise * rng.standard_normal(3)
            Rf = _S @ Rij.T @ _S                                # the reader applies R' = S R_file^T S, t' = S t_file
            tf = _S @ pos
            f.write("%d %d %s %s\n" % (i, j, " ".join("%.17g" % v for v in Rf.ravel()), " ".join("%.17g" % v for v in tf)))
    gt_aa = synth.quat_to_aa(synth.matrix_to_quat(R[:n_cams]))
    return {"n_cams": n_cams, "rotations_aa": gt_aa, "centers": centers[:n_cams], "focal": focal, "principal_point": pp,
            "edges": [e for e in edges if e[0] < n_cams and e[1] < n_cams], "num_tracks": len(tracks)}


def read_edge_matches(path):
    """numpy restatement of the host's Read1DSFMTracks + CollectEdgeMatches: per edge of EGs.txt (inside cc.txt, sorted by
    key) the matched features of the common tracks, the intrinsics (EXIF focal else 1.2 px) and rotation_2 / position_2."""
    cc = set(int(v) for v in open(os.path.join(path, "cc.txt")).read().split())
    focal = {}
    for k, line in enumerate(l for l in open(os.path.join(path, "list.txt")).read().splitlines() if l.strip()):
        parts = line.split()
        focal[k] = float(parts[2]) if len(parts) >= 3 else 0.0
    pp, kps = {}, {}
    with open(os.path.join(path, "coords.txt")) as f:
        while True:
            head = f.readline()
            if not head:
                break
            if not head.strip():
                continue
            fields = dict(kv.strip().split(" = ") for kv in head[1:].split(","))
            k, n = int(fields["index"]), int(fields["keys"])
            rows = [f.readline().split() for _ in range(n)]
            if k in cc:
                pp[k] = (float(np.float32(fields["px"])), float(np.float32(fields["py"])))
                kps[k] = np.array([[float(r[1]), float(r[2])] for r in rows]).reshape(-1, 2)
    toks = open(os.path.join(path, "tracks.txt")).read().split()
    pos, tracks = 1, []
    for _ in range(int(toks[0])):
        n = int(toks[pos]); pos += 1
        t = [(int(toks[pos + 2 * a]), int(toks[pos + 2 * a + 1])) for a in range(n)]
        pos += 2 * n
        tracks.append([o for o in t if o[0] in cc])
    edges = {}
    for line in open(os.path.join(path, "EGs.txt")).read().splitlines():
        v = line.split()
        if len(v) < 14:
            continue
        i, j = int(v[0]), int(v[1])
        if i not in cc or j not in cc:
            continue
        Rf = np.array(v[2:11], dtype=np.float64).reshape(3, 3)
        edges[(min(i, j), max(i, j))] = (_S @ Rf.T @ _S, _S @ np.array(v[11:14], dtype=np.float64))
    keys = sorted(edges)
    slot = {k: e for e, k in enumerate(keys)}
    per_edge = [[] for _ in keys]
    for t in tracks:
        for a in range(len(t)):
            for b in range(a + 1, len(t)):
                lo, hi = (t[a], t[b]) if t[a][0] < t[b][0] else (t[b], t[a])
                e = slot.get((lo[0], hi[0]))
                if e is not None and lo[0] != hi[0]:
                    per_edge[e].append(np.r_[kps[lo[0]][lo[1]], kps[hi[0]][hi[1]]])
    ptr = np.zeros(len(keys) + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum([len(m) for m in per_edge])
    matches = np.vstack([np.array(m).reshape(-1, 4) for m in per_edge]) if keys else np.zeros((0, 4))
    intr = np.zeros((len(keys), 6))
    for e, (i, j) in enumerate(keys):
        for s, v in enumerate((i, j)):
            f = focal.get(v, 0.0) or 1.2 * pp[v][0]
            intr[e, 3 * s:3 * s + 3] = (f, pp[v][0], pp[v][1])
    rot = synth.quat_to_aa(synth.matrix_to_quat(np.array([edges[k][0] for k in keys]).reshape(-1, 3, 3)))
    trans = np.array([edges[k][1] for k in keys]).reshape(-1, 3)
    return {"edges": keys, "match_ptr": ptr, "matches": np.ascontiguousarray(matches), "intrinsics": intr, "rot": rot, "trans": trans}


# ---------------------------------------------------------------------------------------------------------------------
# COLMAP export: two_views.txt (scripts/read_colmap_database.py:52-133, read by src/read_colmap_posegraph.cpp:55-164)
# ---------------------------------------------------------------------------------------------------------------------
def _fake_jpeg(path, width, height):
    """The marker segments of a baseline JPEG up to the frame header: enough for a size probe, not a decodable image."""
    app0 = b"\xff\xe0" + (16).to_bytes(2, "big") + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    sof0 = b"\xff\xc0" + (17).to_bytes(2, "big") + b"\x08" + int(height).to_bytes(2, "big") + int(width).to_bytes(2, "big") + \
        b"\x03\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    with open(path, "wb") as f:
        f.write(b"\xff\xd8" + app0 + sof0 + b"\xff\xd9")


def _fake_png(path, width, height):
    import struct
    import zlib
    ihdr = struct.pack(">IIBBBBB", int(width), int(height), 8, 2, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 13) + b"IHDR" + ihdr + struct.pack(">I", zlib.crc32(b"IHDR" + ihdr)))


def write_synthetic_colmap_export(path, n_cams=10, n_points=500, seed=0, p_visible=0.6, noise_px=0.5, min_common=15,
                                  rel_rot_noise=0.01, rel_pos_noise=0.02, reversed_every=4):
    """A synthetic scene as the COLMAP branch stores it: <path>/two_views.txt and <path>/images/*.JPG (header-only files
    whose size gives the principal point).  The translation column holds TwoViewInfo::position_2 in the convention of the
    Sampson functor of src/uncertainty.cpp:51-81 (x2^T K2^-T R [t]x K1^-1 x1 = 0), which is how the reference consumes it.
    Every `reversed_every`-th pair is listed larger-view-first to exercise SwapCameras."""
    rng = np.random.Generator(np.random.PCG64(seed))
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    centers, R, focal, pp, keypoints, tracks = _synthetic_scene(rng, n_cams, n_points, p_visible, noise_px)
    names = ["DSC_%04d.JPG" % k for k in range(n_cams)]
    for k in range(n_cams):
        _fake_jpeg(os.path.join(path, "images", names[k]), 2 * pp[k, 0], 2 * pp[k, 1])
    pairs = {}
    for t in tracks:
        for a in range(len(t)):
            for b in range(a + 1, len(t)):
                pairs.setdefault((t[a][0], t[b][0]), []).append((t[a][1], t[b][1]))
    rows = []
    first_seen = []
    for n, ((i, j), m) in enumerate(sorted(pairs.items())):
        if len(m) < min_common:
            continue
        noise = synth.quat_to_matrix(synth.aa_to_quat(rel_rot_noise * rng.standard_normal((1, 3))))[0]
        Rij = noise @ R[j] @ R[i].T
        pos = R[i] @ (centers[j] - centers[i])
        pos = pos / np.linalg.norm(pos) + rel_pos_noise * rng.standard_normal(3)
        xi = np.array([keypoints[i][a] for a, _ in m]); xj = np.array([keypoints[j][b] for _, b in m])
        a, b, fa, fb, Rab, pab, xa, xb = i, j, focal[i], focal[j], Rij, pos, xi, xj
        if reversed_every and n % reversed_every == reversed_every - 1 and i in first_seen and j in first_seen:
            a, b, fa, fb, Rab, pab, xa, xb = j, i, focal[j], focal[i], Rij.T, -(Rij @ pos), xj, xi
        for v in (a, b):
            if v not in first_seen:
                first_seen.append(v)
        rot = synth.quat_to_aa(synth.matrix_to_quat(Rab[None]))[0]
        rows.append((a, b, fa, fb, rot, pab, xa, xb))
    with open(os.path.join(path, "two_views.txt"), "w") as f:
        f.write("# img_name1 image_name2 f1 f2 num_inlier rot[0] rot[1] rot[2] trans[0] trans[1] trans[2]\n# features1 [p0x p0y p1x p1y ...]\n# features2 [p0x p0y p1x p1y ...]\n")
        for a, b, fa, fb, rot, pab, xa, xb in rows:
            f.write("%s %s %.17g %.17g %d %s %s\n" % (names[a], names[b], fa, fb, len(xa), " ".join("%.17g" % v for v in rot), " ".join("%.17g" % v for v in pab)))
            f.write(" ".join("%.17g %.17g" % (p[0], p[1]) for p in xa) + " \n")
            f.write(" ".join("%.17g %.17g" % (p[0], p[1]) for p in xb) + " \n")
    # ground truth indexed by the view ids the reader will assign (first appearance)
    order = first_seen
    gt_aa = synth.quat_to_aa(synth.matrix_to_quat(R[order]))
    return {"n_cams": len(order), "rotations_aa": gt_aa, "names": [names[k] for k in order], "focal": focal[order],
            "principal_point": pp[order], "num_pairs": len(rows)}


def read_colmap_two_views(path, image_sizes):
    """numpy restatement of the host's ReadColmapTwoViews.  image_sizes: {file name: (width, height)}."""
    toks = iter(open(path).read().split("\n", 3)[3].split())
    ids, edges = {}, {}
    pp = {}

    def vid(name):
        if name not in ids:
            ids[name] = len(ids)
            w, h = image_sizes.get(name, (0, 0))
            pp[ids[name]] = (float(int(w) // 2), float(int(h) // 2))
        return ids[name]

    while True:
        try:
            n1 = next(toks)
        except StopIteration:
            break
        n2 = next(toks)
        f1, f2 = float(next(toks)), float(next(toks))
        num = int(next(toks))
        r = np.array([float(next(toks)) for _ in range(3)]); t = np.array([float(next(toks)) for _ in range(3)])
        a = np.array([float(next(toks)) for _ in range(2 * num)]).reshape(-1, 2)
        b = np.array([float(next(toks)) for _ in range(2 * num)]).reshape(-1, 2)
        i, j = vid(n1), vid(n2)
        if i > j:
            Rm = synth.quat_to_matrix(synth.aa_to_quat(r[None]))[0]
            i, j, f1, f2, r, t, a, b = j, i, f2, f1, -r, -(Rm @ t), b, a
        edges[(i, j)] = (f1, f2, r, t, np.c_[a, b])
    keys = sorted(edges)
    ptr = np.zeros(len(keys) + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum([len(edges[k][4]) for k in keys])
    intr = np.array([[edges[k][0], pp[k[0]][0], pp[k[0]][1], edges[k][1], pp[k[1]][0], pp[k[1]][1]] for k in keys]).reshape(-1, 6)
    return {"edges": keys, "names": sorted(ids, key=ids.get), "match_ptr": ptr, "matches": np.ascontiguousarray(np.vstack([edges[k][4] for k in keys])),
            "intrinsics": intr, "rot": np.array([edges[k][2] for k in keys]), "trans": np.array([edges[k][3] for k in keys])}
