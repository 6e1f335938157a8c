export TMPDIR=/tmp; R=$(pwd); cd /tmp
rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o p -- python3 $R/tools/kernel_rooflines.py --only sra > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, re
d=collections.defaultdict(list)
for f in glob.glob("/tmp/pk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        m=re.search(r"sra_\w+<[^>]*>", n)
        if m and ("x3" in n):
            d[(m.group(0), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    v.sort(); print(k, len(v), "median us", v[len(v)//2]/1e3)
PY
