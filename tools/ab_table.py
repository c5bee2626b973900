import sys, re, collections
# parse ab_libs.sh output -> table: op x lib (median over reps)
cur=None; data=collections.OrderedDict(); libs=[]
for l in sys.stdin:
    m=re.match(r"== (\S+) \(rep (\d+)\)",l)
    if m:
        cur=m.group(1).split("/")[-1]
        if cur not in libs: libs.append(cur)
        continue
    m=re.match(r"(.+?)\s+([\d.]+) us\s+([\d.]+) GB/s\s+([\d.]+)%",l)
    if m and cur and not l.startswith("  "):
        data.setdefault(m.group(1).strip()[:58],{}).setdefault(cur,[]).append(float(m.group(2)))
print(f"{'op':58s}"+"".join(f"{x[:14]:>15s}" for x in libs))
for op,d in data.items():
    print(f"{op:58s}"+"".join(f"{(sum(d.get(x,[0]))/max(1,len(d.get(x,[0])))):15.2f}" for x in libs))
