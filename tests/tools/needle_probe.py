"""TEST TOOL (GPU box): the adversarial needle scene of tests/tools/needle_study.py through the HIP library in its tree forms,
against the CPU restatement's unpruned recursion.  Prints how many rays differ per form."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from needle_study import needle_scene, grazing_rays
from oracle.pyoracle import Oracle
import tuturenderer_amd as tr

sc = needle_scene()
sets = {"grazing": grazing_rays(sc, 400_000), "oblique": grazing_rays(sc, 200_000, cmin=1e-2, cmax=1.0)}
P = Oracle("port")
S = P.scene(sc)
want = {k: S.closest(*v)[:3] for k, v in sets.items()}
S.close()
forms = {"default": {}, "whole_triangle_boxes": {"TUTU_SPLIT_MAX": "1"}, "reference_tree_pruned": {"TUTU_NO_SAH": "1"}, "binary": {"TUTU_WIDE": "0"}}
if len(sys.argv) > 1:
    forms["exact"] = {"TUTU_EXACT": "1"}
for form, env in forms.items():
    for k in ("TUTU_SPLIT_MAX", "TUTU_NO_SAH", "TUTU_WIDE", "TUTU_EXACT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with tr.Context(sc) as ctx:
        opt = ctx.options()
        for k, (O, D) in sets.items():
            hit, t, tri = want[k]
            h = ctx.trace_closest(O, D)
            wt = np.where(hit == 1, tri, -1)
            bad_tri = int((h["tri"] != wt).sum())
            both = (h["tri"] == wt) & (wt >= 0)
            bad_t = int((h["t"][both].view(np.uint32) != t[both].view(np.uint32)).sum())
            print(json.dumps({"form": form, "rays": k, "n": len(O), "hits": int((wt >= 0).sum()), "different_object": bad_tri, "same_object_different_t": bad_t,
                              "n_refs": opt["n_refs"], "wide": opt["wide_tree"]}), flush=True)
