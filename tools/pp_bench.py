import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print("%.1f M/s  %.4f ms/iter" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,1) for k,v in d["roofline"]["all_kernels_avg_launch_us"].items()})
