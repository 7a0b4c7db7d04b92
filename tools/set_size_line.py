"""one line of a set-size / sets-in-flight sweep from a bench.py JSON line (a file, or stdin when the argument is a number = sets in flight)"""
import json, sys
arg = sys.argv[1] if len(sys.argv) > 1 else "1"
text = sys.stdin.read() if arg.isdigit() else open(arg).read()
d = json.loads(text.strip().splitlines()[-1])
cols = [d["config"]["batches_per_step"]] + ([d["config"].get("sets_in_flight", 1)] if arg.isdigit() else [])
print(*cols, round(d["value"]), round(d["ms_per_step"], 2), {k: round(v.get("avg_launch_ms"), 2) for k, v in sorted(d["roofline"]["per_kernel"].items(), key=lambda kv: -kv[1]["avg_launch_ms"])})
