import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "scan_gate" in n or "colsum" in n or "scan_fwd_state" in n:
        m=re.search(r"scan_gate_(fwd|bwd)_k|colsum|scan_fwd_state", n)
        print((m.group(0) if m else n[:30]), re.findall(r"Li(\d+)E", n)[:6], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
