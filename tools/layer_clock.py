"""Per-layer in-kernel time of the dense launches of one forward pass (device wall clock, first-workgroup-start to
last-workgroup-end, written by the kernels themselves): `SQAIR_PROF_DUMP=file python bench.py ...` writes the per-launch CSV,
this script aggregates it by (layer id, rows)."""
import collections
import csv
import sys

NAMES = ("IENC0 IENC1 PREDISC PRIOR_GRU1 PRIOR_GRU2 PRIOR_LIN TAU1 WB2 MASK2 GENC0 GENC1 WHAT_LOC WHAT_HEAD PRE PROP_RNN PROP_T1 "
         "PROP_T2 PROP_T3 PROP_GRU1 PROP_GRU2 PROP_HEADS PROP_S1 LAT0 LAT1 PRED RNCOND DISC_RNN DISC_T1 DISC_T2 DISC_T3 DISC_S1 DEC0 "
         "DEC1 DEC2 PROP_RNN2 DISC_RNN2").split()


def main(path, min_rows=0):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(int(r["layer"]), int(r["M"]))].append((int(r["end"]) - int(r["start"])) / 100.0)
    tot = 0.0
    for (lid, m), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        tot += sum(v)
        if m >= min_rows:
            print("%-11s M=%-6d n=%-3d mean %6.2f us  sum %7.1f us" % (NAMES[lid] if lid < len(NAMES) else lid, m, len(v), sum(v) / len(v), sum(v)))
    print("total in-kernel %.1f us over %d launches" % (tot, sum(len(v) for v in agg.values())))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
