#!/usr/bin/env python3
"""Where a frame goes beyond its kernels' own times (round 5, review item 7): from a reduced rocprofv3 kernel trace
(name,start_ns,end_ns,queue,stream -- profiles/sessions/r05_s9.sh writes one) and the bench line's exclusive per-kernel times:
    python profiles/frame_attribution.py profiles/r05_s9_c2_timeline.csv profiles/r05_a_bench_line_c2.json  ->  JSON
frame = the launches between the last two k_finalize kernels.  Per kernel class: launches, summed span in the overlapped frame, the
same launches' exclusive time (one pass in flight), and the stretch factor between the two; per stream: busy time and gaps; how long
n kernels were in flight."""
import collections
import csv
import json
import sys


def short(n):
    for k, v in (("k_trace_flat<false", "k_trace_closest"), ("k_trace_flat<true", "k_trace_any"), ("k_trace_wide8<false", "k_trace_closest"),
                 ("k_trace_wide8<true", "k_trace_any"), ("k_trace_wide<false", "k_trace_closest"), ("k_trace_wide<true", "k_trace_any"),
                 ("k_shade<0", "k_shade_depth0"), ("k_shade<5", "k_shade_connect_only"), ("k_shade<", "k_shade"), ("k_list_", "lists"),
                 ("k_resolve", "resolve"), ("k_finalize", "finalize"), ("k_primary", "primary")):
        if k in n:
            return v
    return "other"


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["start_ns"]), int(r["end_ns"]), r["name"], r["queue"]) for r in rows)
fin = [i for i, k in enumerate(ks) if "k_finalize" in k[2]]
fr = ks[fin[-2] + 1: fin[-1] + 1]
t0, t1 = fr[0][0], max(k[1] for k in fr)
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
ex = line["roofline"]["exclusive_kernel_ms_per_step"]
excl = {"k_trace_closest": ex["k_trace_closest"], "k_trace_any": ex["k_trace_any"], "k_shade": ex["k_shade"], "k_shade_depth0": ex["k_shade_depth0"],
        "k_shade_connect_only": ex["k_shade_connect_only"]}
span, cnt = collections.Counter(), collections.Counter()
for s, e, n, q in fr:
    span[short(n)] += e - s
    cnt[short(n)] += 1
ev = sorted([(s, 1) for s, e, n, q in fr] + [(e, -1) for s, e, n, q in fr])
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last, cur = t, cur + d
byq = collections.defaultdict(list)
for s, e, n, q in fr:
    byq[q].append((s, e))
streams = {}
for q, l in byq.items():
    if len(l) < 4:
        continue
    streams[q] = {"launches": len(l), "busy_ms": sum(e - s for s, e in l) / 1e6,
                  "gaps_ms": sum(max(0, l[i + 1][0] - l[i][1]) for i in range(len(l) - 1)) / 1e6}
print(json.dumps({
    "frame_ms_under_the_tracer": (t1 - t0) / 1e6, "frame_ms_untraced": line["ms_per_step"], "launches": len(fr),
    "sum_of_exclusive_kernel_ms": sum(ex.values()), "frame_over_sum_of_exclusive": line["ms_per_step"] / sum(ex.values()),
    "ms_with_n_kernels_in_flight": {str(k): hist[k] / 1e6 for k in sorted(hist)},
    "per_class": {k: {"launches": cnt[k], "span_ms_overlapped": span[k] / 1e6, "exclusive_ms": excl.get(k),
                      "stretch": (span[k] / 1e6 / excl[k]) if excl.get(k) else None} for k in span},
    "streams": streams,
    "note": "in flight is not co-running: a persistent traversal grid holds every wave slot of a CU, so a shade launch of another pass gets in as "
            "its blocks drain -- the spans of the overlapped frame are 2-10 x the kernels' own times, and the frame is 0.80-0.85 x the sum of those"},
    indent=1))
