"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV: takes the LAST full training
step (delimited by the first msda_fwd launch of consecutive steps) so MIOpen's first-use
auto-tuning launches do not pollute the statistics.

    python tools/step_breakdown.py <kernel_trace.csv> [msda launches per step] > profiles/...txt
"""
import collections
import csv
import sys


def main(path, per_step=4, top=70):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "msda_fwd" in r["Kernel_Name"]][::per_step]
    sel = rows[marks[-2]:marks[-1]]
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += d
        agg[r["Kernel_Name"]][1] += 1
    busy = sum(v[0] for v in agg.values())
    print(f"last full step: wall {(t1 - t0) / 1e6:.2f} ms, kernel-busy {busy / 1e6:.2f} ms, {len(sel)} launches, {len(agg)} distinct kernels")
    print(f"{'ms':>9} {'%busy':>6} {'calls':>6} {'avg_us':>9}  kernel")
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{d / 1e6:9.3f} {100 * d / busy:6.1f} {c:6d} {d / c / 1e3:9.1f}  {n[:130]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4)
