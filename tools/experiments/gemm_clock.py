"""Diagnostic: the clock the chip holds inside the fp32-class GEMM tile (library built with -DADVH_STAMPS): s_memtime (core clock)
and s_memrealtime (100 MHz) stamped around the K loop of one workgroup in the middle of the grid, after >= 2 s of back-to-back launches
on random data (MI355X_MICROARCH.md, DVFS give-back)."""
import ctypes, os, sys, time
sys.path.insert(0, "xai-audio-deepfakes_amd")
os.environ["ADDVISOR_HIP_LIB"] = os.path.abspath("tools/experiments/libadvh_stamps.so")
import torch
from addvisor_hip import _lib, gemm as G
_lib.init()
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (38208, 768, 2304)
g = torch.Generator().manual_seed(0)
w = torch.randn(N, K, generator=g) / K ** 0.5
a = torch.randn(M + 1024, K, generator=g)
fn = _lib.lib().advh_debug_gemm_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
p = G.plan_linear(M, w, torch.zeros(N), device=dev, split=True)
A = G.split_planes(a).to(dev)
out = torch.empty((2, M, N), dtype=torch.float16, device=dev)
t0 = time.time()
n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        p.run(A, out_h=out)
    torch.cuda.synchronize()
    n += 50
st = (ctypes.c_longlong * 8)()
assert fn(st) == 0
s = list(st)
cyc, ticks = s[2] - s[0], s[3] - s[1]
print(f"x3 128x128 M={M} K={K} N={N} after {n} launches: one workgroup's K loop = {cyc} core cycles in {ticks} x 10 ns -> clock {cyc / ticks * 100:.0f} MHz; "
      f"{K // 64} K-steps -> {cyc / (K // 64):.0f} cycles per K-step (96 MFMAs = 1 536 cycles of MFMA issue per wavefront)")
print(f"whole workgroup {s[5] - s[4]} cycles: prologue {s[0] - s[4]}, K loop {cyc}, fold + epilogue (to the last store retired) {s[5] - s[2]}")
print(f"  fold {s[6] - s[2]}, epilogue issue {s[7] - s[6]}, wait for the stores {s[5] - s[7]}")
fk = _lib.lib().advh_debug_kstep
fk.restype, fk.argtypes = ctypes.c_int, [ctypes.c_void_p]
ks = (ctypes.c_longlong * 8)()
assert fk(ks) == 0
k = list(ks)
print(f"K-step 6 of that workgroup, wavefront 0: issue 16 LDS-DMA {k[1] - k[0]}, wait for them {k[2] - k[1]}, barrier {k[3] - k[2]}, "
      f"32 ds_read_b128 + 96 MFMA {k[4] - k[3]}, closing barrier to the next step {k[5] - k[4]}; whole step {k[5] - k[0]} cycles")
