import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["batches_per_step"], round(d["value"]), round(d["ms_per_step"],2), {k:round(v.get("avg_launch_ms"),2) for k,v in sorted(d["roofline"]["per_kernel"].items(), key=lambda kv:-kv[1]["avg_launch_ms"])})
