#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so) — run in the build container only.

The reference itself cannot be built or imported here (DESIGN.md "oracle"), so these vectors are the oracle's own
outputs, frozen so that later edits of the oracle or of the HIP path cannot drift silently.  The externally pinned
values (SURVEY.md §8c probe: XORWOW KATs, world/octree counts, C1 PPM md5) are checked separately in
tests/test_oracle_pins.py.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import OracleScene, ppm_bytes, xorwow_stream  # noqa: E402


def rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform([-12, -0.2, -12], [13.5, 2.5, 12], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    o[: n // 3] = np.array([13, 2, 3], np.float32)
    tgt = rng.uniform([-11, 0, -11], [11, 0.4, 11], (n // 3, 3)).astype(np.float32)
    d[: n // 3] = tgt - o[: n // 3]
    d[-8:, 0] = 0.0
    d[-4:, 2] = 0.0
    return np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)


def main():
    out = {}
    for seed in (1984, 1985, 1984 + 400 * 225 - 1):
        init, u = xorwow_stream(seed, 16)
        out["xorwow_%d_state" % seed] = init[:6]
        out["xorwow_%d_uniform" % seed] = u
    np.savez_compressed(os.path.join(HERE, "xorwow.npz"), **out)

    # worlds + cameras + octrees (fp32 and fp16)
    out = {}
    for fp16 in (0, 1):
        for n, spl in ((22, 30), (500, 30), (10000, 32)):
            S = OracleScene(n, 1200, 800, fp16=bool(fp16), use_octree=True, spl=spl)
            g, m, k = S.spheres()
            tag = "%s_n%d" % ("fp16" if fp16 else "fp32", n)
            keep = slice(None) if n <= 500 else slice(0, n, 97)          # subsample the big world
            out[tag + "_geom"], out[tag + "_mat"], out[tag + "_kind"] = g[keep], m[keep], k[keep]
            out[tag + "_geom_sha"] = np.frombuffer(hashlib.sha256(g.tobytes() + m.tobytes() + k.tobytes()).digest(), np.uint8)
            out[tag + "_camera"] = S.camera()
            info = S.info()
            out[tag + "_info"] = np.array([info[x] for x in ("real", "world_draws", "node_count", "leaf_count", "leaf_entries", "dropped_full")], np.int64)
            t = S.octree()
            h = hashlib.sha256()
            for key in ("level", "box", "children", "counts", "indices"):
                h.update(np.ascontiguousarray(t[key]).tobytes())
            out[tag + "_octree_sha"] = np.frombuffer(h.digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "worlds.npz"), **out)

    # per-ray hit records
    out = {}
    for n, spl in ((22, 30), (500, 30), (10000, 32)):
        r = rays(1500, 7 + n)
        S = OracleScene(n, 1200, 800, use_octree=True, spl=spl)
        out["n%d_rays" % n] = r
        for mode, name in ((1, "list"), (2, "tree")):
            h = S.trace(r, mode)
            out["n%d_%s_sphere" % (n, name)] = h["sphere"]
            out["n%d_%s_t" % (n, name)] = h["t"]
            out["n%d_%s_p" % (n, name)] = h["p"]
            out["n%d_%s_normal" % (n, name)] = h["normal"]
    np.savez_compressed(os.path.join(HERE, "hits.npz"), **out)

    # small float framebuffers + final RNG states
    out = {}
    for name, n, nx, ny, ns, tree, spl, fp16 in (
        ("n22_list", 22, 64, 36, 4, False, 30, 0), ("n500_list", 500, 64, 36, 4, False, 30, 0),
        ("n500_tree", 500, 64, 36, 4, True, 30, 0), ("n10000_tree", 10000, 48, 32, 2, True, 32, 0),
        ("n9805_tree_fp16", 9805, 48, 32, 2, True, 32, 1), ("n500_list_fp16", 500, 48, 32, 2, False, 30, 1),
    ):
        S = OracleScene(n, nx, ny, fp16=bool(fp16), use_octree=tree, spl=spl)
        fb, st = S.render(ns, nthreads=8)
        out[name + "_fb"] = fb
        out[name + "_rng"] = st[:, :6]
        out[name + "_cfg"] = np.array([n, nx, ny, ns, int(tree), spl, fp16], np.int64)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)

    # C1 PPM
    S = OracleScene(22, 400, 225)
    fb, _ = S.render(4, nthreads=8)
    md5 = hashlib.md5(ppm_bytes(fb)).hexdigest()
    open(os.path.join(HERE, "c1_ppm.md5"), "w").write(md5 + "\n")
    print("C1 md5", md5)


if __name__ == "__main__":
    main()
