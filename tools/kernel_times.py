import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"][:48]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    if "splat" in k or "light" in k: print(k, [x//1000 for x in v])
