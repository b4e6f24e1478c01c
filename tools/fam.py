#!/usr/bin/env python
"""Prints the per-family busy / slot sums of a timeline JSON written by tools/timeline.py (one line per kernel family)."""
import json
import sys

for path in sys.argv[1:]:
    j = json.load(open(path))
    for leg in ("fwd", "train"):
        x = j[leg]
        print("{} {}: product {:.4f} ms, {} dispatches, busy {:.0f} us, gaps {:.0f} us".format(
            path, leg, x["product_ms_per_step"], x["dispatches"], x["busy_us"], x["gap_us"]))
        for k, v in sorted(x["families"].items(), key=lambda kv: -kv[1]["slot_us"])[:int(18)]:
            print("   {:26s} n={:5d} slot {:8.1f} busy {:8.1f} (avg busy {:6.2f})".format(k, v["launches"], v["slot_us"], v["busy_us"], v["busy_us"] / v["launches"]))
