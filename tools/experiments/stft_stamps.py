"""Diagnostic: s_memtime stamps at the phase boundaries of one STFT / ISTFT workgroup (library built with -DADVH_STAMPS,
see csrc/stft.hip; ADDVISOR_HIP_LIB points at it).  s_memtime ticks at 100 MHz on gfx950."""
import ctypes, os, sys
sys.path.insert(0, "xai-audio-deepfakes_amd")
os.environ["ADDVISOR_HIP_LIB"] = os.path.abspath("tools/experiments/libadvh_stamps.so")
import torch
from addvisor_hip import _lib, ops, synthetic as syn
_lib.init()
dev = torch.device("cuda:0")
B, L = 64, 64000
w = syn.make_clips(B, L).to(dev)
mask = torch.rand(B, 512, 196, device=dev)
X, mag, _ = ops.stft_forward(w, L, want_phase=False)
for _ in range(20):
    ops.stft_forward(w, L, want_phase=False)
    ops.istft_masked_c64(X, mask, L, "log1p")
torch.cuda.synchronize()
st = (ctypes.c_longlong * 16)()
fn = _lib.lib().advh_debug_stamps
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
assert fn(st) == 0
s = list(st)
print("forward  (10 ns ticks): stage", s[1] - s[0], "fft", s[2] - s[1], "epilogue", s[3] - s[2], "total", s[3] - s[0])
print("inverse pass 0        : load", s[5] - s[4], "fft+ola", s[6] - s[5], "emit", s[7] - s[6], "total", s[7] - s[4])
