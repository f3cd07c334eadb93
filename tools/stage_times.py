import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("pictures/s", d["value"], "ms/step", d["ms_per_step"])
for k,v in d.get("kernels",{}).items():
    if "chain" in k or "resi" in k or "tr" in k: print(k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in ("ms","avg_ms","launch_ms")})
