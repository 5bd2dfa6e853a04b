"""One-line summary of a bench.py JSON line: python tools/bench_summary.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| steps", d["steps"], "| frac", r["frac"], "| issue", r.get("issue_frac"),
      "| mem GB/s", (r.get("memory_side") or {}).get("achieved"), "| kernel", r.get("kernel"))
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print(" ", k, {a: b for a, b in v.items() if isinstance(b, (int, float, str)) and a not in ("workload", "what")})
