import csv, glob, sys, collections
rows=[]
for p in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
d=collections.defaultdict(list)
for s,e,n in rows:
    key = "dshard_kernel" if "dshard_kernel" in n else "select" if "dshard_select" in n else "copy" if "copyBuffer" in n else "other"
    d[key].append((e-s)/1e3)
import statistics
for k,v in d.items():
    v2=sorted(v)
    print(k, len(v), "median %.1f p10 %.1f p90 %.1f max %.1f" % (statistics.median(v2), v2[len(v2)//10], v2[len(v2)*9//10], v2[-1]))
ds=[x for x in d["dshard_kernel"] if x>3]
small=[x for x in ds if x < 0.4*max(ds)]
big=[x for x in ds if x >= 0.4*max(ds)]
print("step non-flush n=%d mean %.1f ; flush n=%d mean %.1f" % (len(small), sum(small)/max(len(small),1), len(big), sum(big)/max(len(big),1)))
# gaps between consecutive kernels in the steady state
gaps=[rows[i+1][0]-rows[i][1] for i in range(len(rows)//2, len(rows)-1)]
print("median gap us", statistics.median(gaps)/1e3)
