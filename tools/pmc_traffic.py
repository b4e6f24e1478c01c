"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_hbm_traffic.json.
Run on the GPU box (counters in their own passes, --kernel-trace only; tools/profile_round.sh does all of this):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $REPO/bench.py --steps 3 --warmup 1 --train-steps 3 --no-cpu-baseline --no-graph
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $REPO/bench.py ... (same)
  python $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $REPO/gpurun_out/rNN_hbm_traffic.json
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md section HBM).
Kernels are kept per template instantiation: `dominant` is the instantiation with the most launches among the dense kernels
(k_linear<4, 1, ...>: the 160-row slot layers), the one bench.py's roofline is about."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    from sqair_amd._capi import build_id
    fa, wa = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fa) | set(wa)):
        nf, f = fa.get(k, [0, 0.0])
        nw, w = wa.get(k, [0, 0.0])
        if not k.startswith("k_"):
            continue
        out[k] = dict(launches=max(nf, nw, 1), fetch_kb_raw=f / max(nf, 1), write_kb=w / max(nw, 1),
                      hbm_bytes_per_launch=(2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0)
    dense = {k: v for k, v in out.items() if k.startswith(("k_linear<", "k_linear_what<", "k_linear_rows", "k_linear_mt"))}
    tot_n = sum(v["launches"] for v in dense.values())
    dom = max(dense, key=lambda k: dense[k]["launches"]) if dense else None
    fam = collections.defaultdict(lambda: [0, 0.0])
    for k, v in out.items():
        from sqair_amd.timeline import family as _family   # (the bench's family names, incl. the frame-size aliases)
        f = _family(k)
        fam[f][0] += v["launches"]
        fam[f][1] += v["hbm_bytes_per_launch"] * v["launches"]
    blob = dict(build_id=build_id(),
                family_bytes_per_launch={f: b / max(n, 1) for f, (n, b) in sorted(fam.items())},
                dominant=dom, dominant_bytes_per_launch=dense[dom]["hbm_bytes_per_launch"] if dom else None,
                dense_family_bytes_per_launch=sum(v["hbm_bytes_per_launch"] * v["launches"] for v in dense.values()) / max(tot_n, 1),
                method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (python bench.py --steps 3 --warmup 1 "
                       "--train-steps 3 --no-cpu-baseline --no-graph); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 reports half the bytes of 16 B/lane coalesced reads), WRITE_SIZE uncalibrated; "
                       "`dominant` = the dense instantiation with the most launches; dense_family = launch-weighted mean over all "
                       "forward dense kernels (k_linear<...>, k_linear_rows, k_linear_mt; not the backward's k_linear_dx)",
                per_kernel=out)
    json.dump(blob, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"]) for k, v in out.items()}, indent=0))


if __name__ == "__main__":
    main()
