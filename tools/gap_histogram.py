"""Idle time between consecutive kernels of the last full step of a rocprofv3 --kernel-trace CSV (steps delimited by msda_fwd)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "msda_fwd" in r["Kernel_Name"]][::4]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # which step, counted from the end (bench.py ends with 3 eager probe steps)
sel = rows[marks[-back - 1]:marks[-back]]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
gaps = []
end = int(sel[0]["End_Timestamp"])
for r in sel[1:]:
    s = int(r["Start_Timestamp"])
    gaps.append((max(0, s - end), r["Kernel_Name"][:60]))
    end = max(end, int(r["End_Timestamp"]))
tot = sum(g for g, _ in gaps)
print(f"step wall {(t1-t0)/1e6:.2f} ms, kernel-busy {busy/1e6:.2f} ms, {len(sel)} launches, idle between kernels {tot/1e6:.2f} ms")
h = collections.Counter()
for g, _ in gaps:
    b = 0 if g < 500 else 1 if g < 1000 else 2 if g < 2000 else 3 if g < 5000 else 4 if g < 20000 else 5
    h[b] += 1
names = ["<0.5us", "0.5-1us", "1-2us", "2-5us", "5-20us", ">20us"]
for b in range(6):
    print(f"  {names[b]:8s} {h[b]:5d} gaps  {sum(g for g, _ in gaps if (0 if g < 500 else 1 if g < 1000 else 2 if g < 2000 else 3 if g < 5000 else 4 if g < 20000 else 5) == b)/1e6:7.3f} ms")
for g, n in sorted(gaps, reverse=True)[:12]:
    print(f"  {g/1e3:8.1f} us before {n}")
