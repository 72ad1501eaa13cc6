import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last adj_march_box and print 14 kernels around it
idx = [i for i, r in enumerate(rows) if "adj_march_box" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2] - 6
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 16]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void fdtdhip::", ""))[:40]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name:40s} start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  queue {r.get('Queue_Id','?')}")
