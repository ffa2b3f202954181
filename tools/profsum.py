import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2]) if len(sys.argv)>2 else 20
for r in rows[:n]: print(f'{r["Name"][:60]:60s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}')
