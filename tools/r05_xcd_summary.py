import sys
rows=[l.split() for l in open(sys.argv[1])]
by={}
for r in rows:
    x=int(r[1]); by.setdefault(x,[]).append((float(r[4]), float(r[8])))  # loop end, end
print("loop end by XCD:", " ".join("%.0f" % (sum(a for a,_ in by[x])/len(by[x])) for x in sorted(by)), " end:", " ".join("%.0f" % (sum(b for _,b in by[x])/len(by[x])) for x in sorted(by)))
