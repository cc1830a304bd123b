"""Per-kernel FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, KiB per dispatch) of the dfine:: kernels ->
JSON {kernel: {launches, fetch_KiB_raw, fetch_bytes (x2: gfx950 FETCH_SIZE counts 128-B requests as 64 B,
MI355X_MICROARCH.md §HBM), write_bytes, traffic_bytes}} averaged per launch."""
import collections, csv, glob, json, re, sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{counter}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"dfine::(\w+)", r["Kernel_Name"])
            if m and r["Counter_Name"] == counter:
                agg[m.group(1)][counter].append(float(r["Counter_Value"]))
res = {}
for k, v in sorted(agg.items()):
    fe, wr = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    if not fe or not wr:
        continue
    f_kib, w_kib = sum(fe) / len(fe), sum(wr) / len(wr)
    res[k] = {"launches": len(fe), "fetch_KiB_raw": round(f_kib, 1), "fetch_bytes": int(2 * f_kib * 1024),
              "write_bytes": int(w_kib * 1024), "traffic_bytes": int((2 * f_kib + w_kib) * 1024)}
print(json.dumps(res, indent=1))
