import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), d["single_call_ms"], {k:v.get("avg_launch_ms") for k,v in d["roofline"]["per_kernel"].items()})
