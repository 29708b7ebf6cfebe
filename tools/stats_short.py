import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=re.sub(r"\(.*","",n)
    print("%-110s %6s %9.2f us  %5s%%"%(n[:110],r["Calls"],float(r["AverageNs"])/1000,r["Percentage"]))
