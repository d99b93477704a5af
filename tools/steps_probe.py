import os, sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import libstevi_amd as sv
from helpers import parallax_pair
dev = torch.device("cuda:0")
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
sv.set_option(d_src, "census_winner_shortcut", 0)
MF = sv.matchingFunctions
def step():
    return sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, 256, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)
for _ in range(30): step()
torch.cuda.synchronize()
def region(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3
for K in (20, 50, 200, 1000):
    r = [region(K) for _ in range(9)]
    print(json.dumps({"K": K, "ms_per_step_median": round(float(np.median([a for a, _ in r])), 4), "min": round(min(a for a, _ in r), 4),
                      "host_issue_ms_per_step": round(float(np.median([b for _, b in r])), 4)}), flush=True)
# graph capture of K steps
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sv.set_option(d_src, "census_winner_shortcut", 0)
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): out = step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(json.dumps({"graph_of_20_steps_ms_per_step_median": round(float(np.median(ts)), 4), "min": round(min(ts), 4), "disp_checksum": int(out["disp"].sum().item())}))
except Exception as e:
    print("graph capture failed:", type(e).__name__, e)
