"""CPU: the builder of the EIGHT-WIDE tree (round 5; csrc/host_scene.cpp: build_wide8, dumped by tutu_host_wide8 without a GPU).
The tree is only ever WALKED -- hits are decided by the reference's triangle test and the reference's leaf box (BVH.hpp:145-167,
csrc/device_trace.h) -- so what has to hold is structural: every object is reachable, the ids of inner children are the node's
child_base + slot, the quantised box of every slot CONTAINS what lies beneath it (with the margin the kernel's arithmetic needs),
and the visiting-order tables are well formed.  The hits themselves are compared on the GPU (tests/test_hip_wide.py)."""
import numpy as np
import pytest

from conftest import golden_path  # noqa: F401  (keeps the suite's path set-up)


def _decode(nodes):
    f = nodes.view(np.float32)
    out = {
        "p": f[:, 0:3].astype(np.float64),
        "scale": np.stack([f[:, 3], f[:, 4], f[:, 5]], axis=1).astype(np.float64),
        "child_entry": nodes[:, 6],
        "meta": nodes[:, 7],
        "leaf": nodes[:, 20:28].view(np.int32),
    }
    planes = nodes[:, 8:20]  # qlo[3][2], qhi[3][2]
    q = np.zeros((len(nodes), 2, 3, 8), np.float64)  # [lo/hi][axis][slot]
    for side in range(2):
        for a in range(3):
            for w in range(2):
                word = planes[:, side * 6 + a * 2 + w]
                for b in range(4):
                    q[:, side, a, 4 * w + b] = (word >> (8 * b)) & 0xFF
    out["q"] = q
    return out


@pytest.fixture(scope="module")
def tr(built):
    import tuturenderer_amd

    tuturenderer_amd.load_library()
    return tuturenderer_amd


@pytest.mark.parametrize("name", ["veach", "bunny", "spheres"])
def test_eight_wide_tree_is_well_formed_and_conservative(tr, monkeypatch, name):
    from tuturenderer_amd import scenes

    monkeypatch.setenv("TUTU_SPLIT_MAX", "1")  # whole-triangle references: a leaf slot's box must contain the object's own box
    if name == "veach":
        sc = scenes.veach_room(64, 48, small_light=False)
    elif name == "bunny":
        sc = scenes.bunny_box(32, 32)
    else:
        from oracle.gen_golden import golden_scenes

        sc = golden_scenes()["cornell_spheres"][0]()
    nodes, info, boxes = tr.host_wide8(sc)
    d = _decode(nodes)
    n_ids, n_obj = len(nodes), len(boxes)
    assert info["n_nodes"] >= 1 and info["depth"] >= 1 and n_obj >= 3
    m = info["margin"]
    inner_mask = d["child_entry"] & 0xFF
    leaf_mask = d["meta"] & 0xFF
    base = (d["child_entry"] >> 11) << 3
    assert not (inner_mask & leaf_mask).any()
    # walk from the root: ids, reachability, depth; subtree boxes bottom-up
    seen_obj = np.zeros(n_obj, bool)
    visited = 0
    max_level = 0

    def slot_box(i, s):
        lo = d["p"][i] + d["q"][i, 0, :, s] * d["scale"][i]
        hi = d["p"][i] + d["q"][i, 1, :, s] * d["scale"][i]
        return lo, hi

    def walk(i, level):
        nonlocal visited, max_level
        visited += 1
        max_level = max(max_level, level)
        assert (inner_mask[i] | leaf_mask[i]) != 0, "a reachable node without children"
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        for s in range(8):
            if (leaf_mask[i] >> s) & 1:
                ref = int(d["leaf"][i, s])
                assert ref < 0
                li = (~ref) & ~0x40000000
                assert 0 <= li < n_obj
                seen_obj[li] = True
                blo, bhi = boxes[li, 0:3].astype(np.float64), boxes[li, 4:7].astype(np.float64)
            elif (inner_mask[i] >> s) & 1:
                c = int(base[i]) + s
                assert 0 < c < n_ids
                blo, bhi = walk(c, level + 1)
            else:
                continue
            lo, hi = slot_box(i, s)
            # the quantised box contains what lies beneath the slot, with the margin the kernel's arithmetic needs (host_scene.cpp)
            assert (lo <= blo - m * 0.999).all() and (hi >= bhi + m * 0.999).all(), (i, s, lo, blo, hi, bhi)
            lo_all, hi_all = np.minimum(lo_all, blo), np.maximum(hi_all, bhi)
        return lo_all, hi_all

    import sys

    sys.setrecursionlimit(10000)
    walk(0, 1)
    assert visited == info["n_nodes"] and max_level == info["depth"]
    assert seen_obj.all(), "an object no leaf slot refers to"
    # holes are empty records; the visiting-order tables: three bits per sign octant, opposite octants visit in opposite order
    used = (inner_mask | leaf_mask) != 0
    assert int(used.sum()) == info["n_nodes"] and not nodes[~used].any()
    table = d["meta"][used] >> 8
    for oct_ in range(8):
        a = (table >> (3 * oct_)) & 7
        b = (table >> (3 * (oct_ ^ 7))) & 7
        assert ((a ^ b) == 7).all()
    assert (((table >> 0) & 7) == 0).all()  # a ray that travels up every axis visits the slots in their own order


def test_eight_wide_tree_with_clipped_references_still_reaches_every_object(tr):
    """the default build of the veach room clips its sliver triangles: several leaf slots may name one object, none is lost"""
    from tuturenderer_amd import scenes

    nodes, info, boxes = tr.host_wide8(scenes.veach_room(64, 48, small_light=False))
    d = _decode(nodes)
    leaf_mask = d["meta"] & 0xFF
    refs = [(~int(d["leaf"][i, s])) & ~0x40000000 for i in range(len(nodes)) for s in range(8) if (leaf_mask[i] >> s) & 1]
    assert len(refs) > len(boxes) and set(refs) == set(range(len(boxes)))
