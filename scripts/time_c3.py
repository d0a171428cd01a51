"""BASELINE config 3 (d = dy = 64, T = 10^4, one chain) and the mid sizes, as bench.py times them (extra_c3)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
r = bench.extra_c3(0)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
