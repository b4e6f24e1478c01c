#!/usr/bin/env python
"""Prints the dispatches of one frame (between the 5th and 6th k_compact) of a timeline CSV: kernel, workgroups, busy, gap, slot."""
import csv
import sys

rows = list(csv.DictReader([l for l in open(sys.argv[1]) if not l.startswith("#")]))
idx = [i for i, r in enumerate(rows) if r["kernel"].startswith("k_compact")]
a, b = idx[4] + 1, idx[5] + 1
tot = 0.0
for i in range(a, b):
    r = rows[i]
    tot += float(r["slot_us"])
    print("%3d %-44s wg %5s busy %6s gap %5s slot %6s" % (i - a, r["kernel"][:44], r["workgroups"], r["busy_us"], r["gap_before_us"], r["slot_us"]))
print("frame: %d dispatches, %.1f us" % (b - a, tot))
