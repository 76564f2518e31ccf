#!/usr/bin/env python3
"""How much do the solves of several sweep lanes overlap on the GPU?  From a rocprofv3 --kernel-trace CSV of
`bench.py --sweep --lanes N --steps 1 --warmup 0`:  lane_overlap.py <kernel_trace.csv> [<t_skip_fraction>]

Prints the wall span of the kernels, the time at least one kernel runs, the sum of the kernel durations, their ratio (the
average number of kernels in flight while the GPU is busy), the share of the span with 0 / 1 / 2 / 3 / 4+ kernels in
flight, and the same per hardware queue (Queue_Id) -- HIP streams beyond GPU_MAX_HW_QUEUES share a queue and serialise."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
perq = collections.defaultdict(lambda: [0, 0])
for r in rows:
    t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((t0, 1))
    ev.append((t1, -1))
    q = r.get("Queue_Id", "?")
    perq[q][0] += 1
    perq[q][1] += t1 - t0
ev.sort()
span = ev[-1][0] - ev[0][0]
hist = collections.defaultdict(int)
cur, last = 0, ev[0][0]
for t, d in ev:
    hist[min(cur, 4)] += t - last
    cur += d
    last = t
busy = span - hist[0]
total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms ({100 * busy / span:.0f} %)  sum of durations {total / 1e6:.1f} ms"
      f"  average in flight while busy {total / max(busy, 1):.2f}")
print("share of the span with k kernels in flight: " + "  ".join(f"{k}{'+' if k == 4 else ''}: {100 * hist[k] / span:.1f} %" for k in range(5)))
print("hardware queues: " + "  ".join(f"q{q}: {n} kernels, {t / 1e6:.1f} ms" for q, (n, t) in sorted(perq.items())))
