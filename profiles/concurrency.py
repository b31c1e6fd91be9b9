#!/usr/bin/env python3
"""How many kernels are in flight during one frame of the default bench command (four passes on four streams), from a
rocprofv3 --kernel-trace CSV:   python profiles/concurrency.py <kernel_trace.csv>  ->  JSON
(frame = the launches between the last two k_finalize kernels; concurrency histogram in ms, per-kernel-class span sums)."""
import collections
import csv
import json
import sys


def short(n):
    for k, v in (("k_trace<true, false", "k_trace_closest"), ("k_trace<true, true", "k_trace_any"), ("k_trace<false, false", "k_trace_closest"),
                 ("k_trace<false, true", "k_trace_any"), ("k_trace_wide<false", "k_trace_closest"), ("k_trace_wide<true", "k_trace_any"), ("k_trace_wide8<false", "k_trace_closest"), ("k_trace_wide8<true", "k_trace_any"),
                 ("k_trace_flat<false", "k_trace_closest"), ("k_trace_flat<true", "k_trace_any"), ("k_shade<0", "k_shade_depth0"), ("k_shade<5", "k_shade_connect_only"), ("k_shade<", "k_shade"),
                 ("k_list_", "lists"), ("k_resolve", "resolve"), ("k_finalize", "finalize"), ("k_primary", "primary")):
        if k in n:
            return v
    return "other"


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
fin = [i for i, k in enumerate(ks) if "k_finalize" in k[2]]
fr = ks[fin[-2] + 1: fin[-1] + 1]
t0, t1 = fr[0][0], max(k[1] for k in fr)
ev = sorted([(s, 1) for s, e, n in fr] + [(e, -1) for s, e, n in fr])
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last, cur = t, cur + d
span, cnt = collections.Counter(), collections.Counter()
for s, e, n in fr:
    span[short(n)] += e - s
    cnt[short(n)] += 1
print(json.dumps({"frame_ms": (t1 - t0) / 1e6, "launches": len(fr),
                  "ms_with_n_kernels_in_flight": {str(k): hist[k] / 1e6 for k in sorted(hist)},
                  "sum_of_kernel_spans_ms": sum(span.values()) / 1e6,
                  "per_class": {k: {"launches": cnt[k], "span_ms": span[k] / 1e6, "avg_us": span[k] / cnt[k] / 1e3} for k in span}}, indent=1))
