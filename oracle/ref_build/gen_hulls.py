#!/usr/bin/env python3
"""Generates `<mesh>.hulls.npz` fixtures with the REFERENCE's vendored V-HACD (compiled where it
lies into oracle/_ref/vhacd_driver) and the reference's parameters/selection rule
(src/mesh.cpp:351-355, :394-396, :426-429).  stillleben_amd.hulls picks these files up, so the
GPU box settles the test meshes on exactly the hulls the reference would cook."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from stillleben_amd import _loaders  # noqa: E402

DRIVER = os.path.join(HERE, "..", "_ref", "vhacd_driver")


def vhacd(cm):
    """The reference's V-HACD procedure on a consolidated mesh: list of (vertices, triangles)."""
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<II", len(cm.positions), len(cm.indices) // 3))
            f.write(cm.positions.astype(np.float32).tobytes())
            f.write(cm.indices.astype(np.uint32).tobytes())
        subprocess.run([DRIVER, fin, fout], check=True)
        raw = open(fout, "rb").read()
    off = 0
    (n,) = struct.unpack_from("<I", raw, off); off += 4
    out = []
    for _ in range(n):
        nv, nt = struct.unpack_from("<II", raw, off); off += 16
        v = np.frombuffer(raw, np.float32, 3 * nv, off).reshape(nv, 3); off += 12 * nv
        t = np.frombuffer(raw, np.uint32, 3 * nt, off).reshape(nt, 3); off += 12 * nt
        out.append((v.copy(), t.astype(np.int32)))
    return out


def write_ycb_like(seed=0, target_verts=8192):
    """stillleben_amd/data/ycb_like_hulls_seed<seed>.npz: the V-HACD decompositions of the 21 synthetic YCB-like classes."""
    from concurrent.futures import ThreadPoolExecutor

    from stillleben_amd import hulls as H
    from stillleben_amd import synthetic

    cms = [synthetic.make_class_mesh(name, seed, target_verts, 64)[0] for name in synthetic.YCB_CLASSES]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(vhacd, cms))
    out = {}
    for name, cm, hs in zip(synthetic.YCB_CLASSES, cms, res):
        key = name + "/"
        out[key + "digests"] = np.array(H._mesh_digests(cm), dtype=np.uint64)
        out[key + "n"] = np.int32(len(hs))
        for i, (v, t) in enumerate(hs):
            out["%sv%d" % (key, i)], out["%st%d" % (key, i)] = v, t
    np.savez_compressed(synthetic.HULL_DATA % seed, **out)
    print(synthetic.HULL_DATA % seed, {n: len(h) for n, h in zip(synthetic.YCB_CLASSES, res)})


def run(mesh_path):
    cm = _loaders.load_any(mesh_path)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<II", len(cm.positions), len(cm.indices) // 3))
            f.write(cm.positions.astype(np.float32).tobytes())
            f.write(cm.indices.astype(np.uint32).tobytes())
        subprocess.run([DRIVER, fin, fout], check=True)
        raw = open(fout, "rb").read()
    off = 0
    (n,) = struct.unpack_from("<I", raw, off); off += 4
    out = {"n_hulls": np.int32(n)}
    for i in range(n):
        nv, nt = struct.unpack_from("<II", raw, off); off += 8
        (vol,) = struct.unpack_from("<d", raw, off); off += 8
        v = np.frombuffer(raw, np.float32, 3 * nv, off).reshape(nv, 3); off += 12 * nv
        t = np.frombuffer(raw, np.uint32, 3 * nt, off).reshape(nt, 3); off += 12 * nt
        out["v%d" % i], out["t%d" % i] = v.copy(), t.astype(np.int32)
    from stillleben_amd import hulls as H

    out["digests"] = np.array(H._mesh_digests(cm), dtype=np.uint64)   # the fixture is valid for THIS geometry only
    dst = mesh_path + ".hulls.npz"
    np.savez_compressed(dst, **out)
    print(mesh_path, "->", n, "hulls,", sum(len(out["v%d" % i]) for i in range(n)), "vertices,", os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    if "--ycb" in sys.argv:
        write_ycb_like()
        sys.exit(0)
    fx = os.path.join(ROOT, "tests", "fixtures")
    for p in sys.argv[1:] or [os.path.join(fx, "cube.glb"), os.path.join(fx, "stanford_bunny", "scene.gltf")]:
        run(p)
