import re,collections,sys,statistics
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    m=re.match(r"\[(\w+)\] (\w+) .*: ([\d.]+) ms",l)
    if m: d[(m.group(1),m.group(2))].append(float(m.group(3)))
vs=sys.argv[2].split(',')
for s in ["P12","P23","Q23","Q12"]:
    print(s, "  ".join("%s med %.3f (%s)" % (v, statistics.median(d[(v,s)]), "/".join("%.3f"%x for x in d[(v,s)])) for v in vs))
