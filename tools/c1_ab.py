"""config 1 (whole-loop kernel) at both batch sizes: ms per step; for A/B runs with SC_LIB_PATH"""
import json, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
torch.set_default_dtype(torch.float64)
out = {}
for n, steps in ((1000, 200), (100000, 50)):
    d = bench.config1(dev, n, steps)
    out[n] = {"whole_loop_ms_per_step": d["ms_per_step"], "stepwise": d["stepwise"]["ms_per_step"]}
print(json.dumps(out))
