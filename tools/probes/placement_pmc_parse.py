import csv, glob, os, sys
d = sys.argv[1]
labels = [l.split()[1].rstrip(":") for l in open(sys.argv[2]) if l.startswith("SET ")]
ctr = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "caar_np4_kernel" in r["Kernel_Name"]:
            ctr.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "caar_np4_kernel" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(dur)
names = sorted({k for v in ctr.values() for k in v})
print("%-14s %10s " % ("set", "avg us") + " ".join("%22s" % n[:22] for n in names))
for i, lab in enumerate(labels):
    chunk = ids[i * 30 + 10: i * 30 + 30]
    if not chunk:
        break
    print("%-14s %10.1f " % (lab, sum(dur[c] for c in chunk) / len(chunk) / 1e3) +
          " ".join("%22.0f" % (sum(ctr.get(c, {}).get(n, 0) for c in chunk) / len(chunk)) for n in names))
