"""Diagnostic (library built with -DADVH_STAMPS): per-workgroup start / K-loop-end / end stamps of one fp32-class GEMM launch and
the CU each workgroup ran on -> how busy each CU slot is, the gaps between successive workgroups, how the epilogues line up."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, "xai-audio-deepfakes_amd")
os.environ["ADDVISOR_HIP_LIB"] = os.path.abspath("tools/experiments/libadvh_stamps.so")
import torch
from addvisor_hip import _lib, gemm as G
_lib.init()
dev = torch.device("cuda:0")
M, K, N = 38208, 768, 2304
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
a = torch.randn(M + 1024, K, generator=g)
fn = _lib.lib().advh_debug_wg_records
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
p = G.plan_linear(M, w, torch.zeros(N), device=dev, split=True)
A = G.split_planes(a).to(dev)
out = torch.empty((2, M, N), dtype=torch.float16, device=dev)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        p.run(A, out_h=out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    p.run(A, out_h=out)
e1.record()
torch.cuda.synchronize()
print(f"HIP events: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (20 back-to-back launches); tile {G.TILE_NAMES[p.tile]} sc {p.desc.sc}")
nwg = ((M + 127) // 128) * (N // 128)
buf = (ctypes.c_longlong * (4 * nwg))()
assert fn(buf, nwg) == 0
r = np.frombuffer(buf, dtype=np.int64).reshape(nwg, 4)
t0 = r[:, 0].min()
start, kend, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0      # us
hw = r[:, 3] & 0xffffffff
xcc = (r[:, 3] >> 32) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"{nwg} workgroups, launch spans {end.max():.1f} us; distinct CUs seen {len(np.unique(cuid))}; XCCs {sorted(np.unique(xcc))}")
dur = end - start
print(f"workgroup duration us: min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f};  K loop median {np.median(kend - start):.1f}, epilogue median {np.median(end - kend):.1f} p90 {np.percentile(end - kend, 90):.1f}")
busy = []
gaps = []
conc = []
for c in np.unique(cuid):
    idx = np.where(cuid == c)[0]
    busy.append(dur[idx].sum() / (2 * end.max()))
    conc.append(len(idx))
print(f"per CU: workgroups min {min(conc)} max {max(conc)}; slot occupancy (sum of durations / 2 slots / span) min {min(busy):.2f} mean {np.mean(busy):.2f} max {max(busy):.2f}")
# first-round starts and last ends
print(f"start times of the first 512 by start order: p50 {np.sort(start)[255]:.1f} us, p100 {np.sort(start)[511]:.1f} us; last start {start.max():.1f}; ends: p50 {np.median(end):.1f}")
# how many epilogues overlap at a time: sample on a 1 us grid
grid = np.arange(0, end.max(), 1.0)
ep = [(np.sum((kend <= t) & (end > t))) for t in grid]
kl = [(np.sum((start <= t) & (kend > t))) for t in grid]
print("workgroups in their epilogue, sampled every 1 us (first 120 us):", ep[:120])
print("workgroups in their K loop (first 120 us):", kl[:120])
print(f"mean concurrent K-loop workgroups {np.mean(kl):.0f} of 512 slots, mean in epilogue {np.mean(ep):.0f}")
