"""STFT / masked-ISTFT kernel timing at the BASELINE shape (B = 64 x 4 s) against their algorithmic HBM bytes
(1.89 MB / clip forward incl. X, 1.47 MB / clip per resynthesis)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xai-audio-deepfakes_amd"))
from addvisor_hip import _lib, ops, synthetic as syn
_lib.init()
dev = torch.device("cuda:0")
B, L = 64, 64000
fb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
_lib.check(_lib.lib().advh_set_option(b"stft_frames_per_workgroup", fb), "set_option")
w = syn.make_clips(B, L).to(dev)
mask = torch.rand(B, 512, 196, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


X, mag, ph = ops.stft_forward(w, L)
nbin = B * 513 * 199
cases = [
    ("forward X+|X|+phase", lambda: ops.stft_forward(w, L), B * L * 4 + nbin * 16),
    ("forward X+|X|", lambda: ops.stft_forward(w, L, want_phase=False), B * L * 4 + nbin * 12),
    ("forward |X|+phase", lambda: ops.stft_forward(w, L, want_complex=False), B * L * 4 + nbin * 8),
    ("inverse c64 in+out log1p", lambda: ops.istft_masked_c64(X, mask, L, "log1p"), nbin * 8 + mask.numel() * 4 + 2 * B * L * 4),
    ("inverse c64 in+out linear", lambda: ops.istft_masked_c64(X, mask, L, "linear"), nbin * 8 + mask.numel() * 4 + 2 * B * L * 4),
    ("inverse polar in+out log1p", lambda: ops.istft_masked(mag, ph, mask, L, "log1p"), nbin * 8 + mask.numel() * 4 + 2 * B * L * 4),
    ("inverse plain c64", lambda: ops.istft_complex(X, L), nbin * 8 + B * L * 4),
]
for name, fn, nbytes in cases:
    us = timeit(fn)
    print(f"FB={fb} {name:28s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:6.1f} MB (bytes actually moved once)", flush=True)
