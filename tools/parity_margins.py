import json,sys
d=json.load(open('gpurun_out/parity_report.json'))
d=[r for r in d if r['bound']]
d.sort(key=lambda r:-r['achieved']/r['bound'])
print(sys.argv[1], " | ".join("%.0f%% %s/%s"%(100*r['achieved']/r['bound'], r['test'].split('[')[1][:18], r['quantity'][:16]) for r in d[:5]))
