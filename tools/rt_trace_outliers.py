import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "dispatches; columns", list(rows[0].keys()))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
names = collections.Counter()
out = []
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > 2_000_000:
        out.append(((int(r["Start_Timestamp"]) - t0) / 1e6, d / 1e6, r["Kernel_Name"][:50], r.get("Queue_Id"), r.get("Stream_Id")))
out.sort()
print(len(out), "dispatches > 2 ms")
for o in out[:80]:
    print("t=%9.2f ms  dur=%6.2f ms  %s q=%s s=%s" % o)
# activity: total dispatches per 100 ms bin, and copyBuffer / fill activity late in the run
bins = collections.Counter()
late = collections.Counter()
for r in rows:
    b = int((int(r["Start_Timestamp"]) - t0) / 1e8)
    bins[b] += 1
    if "rocclr" in r["Kernel_Name"]:
        late[b] += 1
print("dispatches per 100 ms:", [bins[b] for b in range(max(bins) + 1)])
print("rocclr copies per 100 ms:", [late[b] for b in range(max(bins) + 1)])
