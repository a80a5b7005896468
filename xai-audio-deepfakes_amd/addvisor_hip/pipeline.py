"""One explanation per clip, end to end on the GPU (the unit of work of SURVEY.md §8d):

    STFT -> classifier(clean) -> U-Net mask -> masked ISTFT (mask-in, mask-out) -> classifier x2
         -> LMAC metric accumulators

i.e. the loop body of ``run_addvisor_metrics`` (LMAC_metrics.py:117-157) with the D1-D5 resolutions of
SURVEY.md §2.3.  Utterances are independent, so a data set shards over ranks by clip index with no
data-path collective; the only exchange is one fixed-order gather of the per-clip probabilities
(``gather_probabilities``), after which every rank reduces the same vector in the same order and
obtains bit-identical metrics for any world size.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib, ops
from .embedder import HipEmbedder, default_precision
from .synthetic import EmbedderConfig
from .unet import HipUNet

METRIC_NAMES = ("faithfulness", "fidelity", "AD", "AI", "AG")


class ExplainPipeline:
    def __init__(self, emb_cfg: EmbedderConfig, emb_sd, coef, intercept, unet_sd, device,
                 audio_length: float = 4, sampling_rate: int = 16000, domain: str = "log1p",
                 hop: int = 322, win: int = 644, streams: int = 1, vocoder=None, precision: Optional[str] = None,
                 embedder: Optional[HipEmbedder] = None, unet: Optional[HipUNet] = None):
        """``precision``: "f32" (fp32-class split-format kernels: the reference's arithmetic class, the default) or "f16"
        (fp16 operands; 2-3x faster); ``None`` = ``ADDVISOR_PRECISION``.  ``vocoder``: an ``addvisor_hip.hifigan.HipHifigan``; when given, the mask-in / mask-out resyntheses are
        re-rendered by the vocoder (mel front end of hifigan.py:163-178 -> HiFi-GAN V1 -> crop to the clip length)
        before the classifier re-forward -- the "masked spectrogram -> vocoder -> classifier" variant of the path.
        ``embedder`` / ``unet``: already-built engines to share (the drop-in modules' process-wide singletons) instead of
        packing the weights again; the weight arguments are then ignored."""
        self.dev = device
        self.L = int(audio_length * sampling_rate)
        self.hop, self.win, self.domain = hop, win, domain
        self.precision = precision or (embedder.precision if embedder is not None else default_precision())
        # the vocoder runs at the path's precision: one arithmetic class per explanation (a generator built at another
        # precision is rebuilt from its own weights)
        self.vocoder = None if vocoder is None else vocoder.with_precision(self.precision)
        self.embedder = embedder if embedder is not None else HipEmbedder(emb_cfg, emb_sd, coef, intercept, device, precision=self.precision)
        self.unet = unet if unet is not None else HipUNet(unet_sd, device, precision=self.precision)
        if self.embedder.precision != self.precision or self.unet.precision != self.precision:
            raise ValueError("embedder, U-Net and pipeline must share one precision")
        # the 3B embedder batch can be split over several HIP streams: kernels of independent sub-batches then
        # fill each other's tail waves and launch gaps (utterances are independent)
        self.nstreams = max(1, streams)
        self._streams = [torch.cuda.Stream(device=device) for _ in range(self.nstreams - 1)]

    def explain(self, waves: torch.Tensor, keep: bool = False) -> Dict[str, torch.Tensor]:
        """``waves [B, n]`` fp32 on the GPU -> clean / mask-in / mask-out probabilities ``[B,1]`` and the mask."""
        L = self.L
        B, n = waves.shape
        # X and |X| only: the masked resynthesis scales X itself (advh_istft_masked_c64), so no phase is computed or read
        spec, mag, _ = ops.stft_forward(waves, L, self.hop, self.win, want_complex=True, want_phase=False)
        mask = self.unet.forward(mag)
        # clean clip, mask-in and mask-out resyntheses go through the embedder as ONE 3B batch
        allw = torch.empty((3 * B, L), dtype=torch.float32, device=waves.device)
        if n >= L:
            allw[:B].copy_(waves[:, :L])
        else:
            allw[:B, :n].copy_(waves)
            allw[:B, n:].zero_()
        rc = _lib.lib().advh_istft_masked_c64(
            torch.view_as_real(spec).data_ptr(), mask.data_ptr(), mask.shape[1], mask.shape[2],
            {"linear": 1, "log1p": 2}[self.domain], allw[B:].data_ptr(), allw[2 * B:].data_ptr(), L, B, mag.shape[2], L,
            self.hop, self.win, None, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "advh_istft_masked_c64")
        if self.vocoder is not None:
            voc = self.vocoder.decode_batch(ops.mel_spectrogram(allw[B:]))[:, 0]        # [2B, 256 * (1 + L // 256)]
            k = min(L, voc.shape[1])
            allw[B:, :k].copy_(voc[:, :k])
            allw[B:, k:].zero_()
        if self.nstreams == 1 or (3 * B) % self.nstreams:
            _, _, p3 = self.embedder.forward(allw, L, want_hidden=False)
        else:
            cur = torch.cuda.current_stream()
            per = 3 * B // self.nstreams
            parts = [None] * self.nstreams
            for i, st in enumerate([cur] + self._streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    parts[i] = self.embedder.forward(allw[i * per:(i + 1) * per], L, want_hidden=False, slot=i)[2]
            for st in self._streams:
                cur.wait_stream(st)
            p3 = torch.cat(parts, 0)
        p_clean, p2, both = p3[:B], p3[B:], allw[B:]
        out = dict(predictions=p_clean, theta_out=p2[:B], masked_predictions=p2[B:], mask=mask)
        if keep:
            out.update(mag=mag, spec=spec, wave_in=both[:B], wave_out=both[B:])
        return out

    # ------------------------------------------------------------------ HIP graph replay (launch-bound small batches)
    def capture(self, B: int) -> None:
        """Record ``explain`` for batch size ``B`` into a HIP graph (``torch.cuda.CUDAGraph`` = hipGraph on ROCm).  One
        explanation is ~150 kernel launches; for small batches (the interactive single-clip use of the reference's
        streamlit study) the step is launch-bound and a single graph launch replaces the Python launch loop.  Every kernel
        takes caller-owned buffers and a stream and never synchronises, so the whole step is capturable as it is."""
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        static_in = torch.zeros((B, self.L), dtype=torch.float32, device=self.dev)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                    # warm-up outside capture: workspaces, packed weights, LDS attributes
            for _ in range(2):
                self.explain(static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.explain(static_in)
        self._graphs[B] = (graph, static_in, out)

    def explain_graphed(self, waves: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Replay the captured step on ``waves [B, n]``; the returned tensors are the graph's static outputs (valid until
        the next replay -- clone what must be kept)."""
        B = waves.shape[0]
        if B not in getattr(self, "_graphs", {}):
            self.capture(B)
        graph, static_in, out = self._graphs[B]
        n = min(waves.shape[1], self.L)
        static_in[:, :n].copy_(waves[:, :n])
        if n < self.L:
            static_in[:, n:].zero_()
        graph.replay()
        return out

    def tune(self, B: int):
        """Pick the fastest GEMM tile per launch by measurement, on a throw-away batch (call once, outside any
        timed region; the results of this pass are discarded)."""
        from . import gemm as G
        G.TUNER.active = True
        try:
            self.explain(torch.zeros((B, self.L), dtype=torch.float32, device=self.dev).uniform_(-0.1, 0.1))
            torch.cuda.synchronize()
        finally:
            G.TUNER.active = False

    def flops(self, B: int) -> float:
        """Algorithmic FLOPs of one explain() call on B clips (3 embedder passes + U-Net)."""
        T = 1 + self.L // self.hop
        return self.embedder.flops(3 * B, self.L) + self.unet.flops(B, 512, (T // 4) * 4)


def lmac_metrics(predictions: torch.Tensor, theta_out: torch.Tensor, masked_predictions: torch.Tensor,
                 per_clip: bool = False):
    """The five dataset means of LMAC_metrics.py:164-172 from ``[N,1]`` (or ``[N]``) GPU probabilities."""
    p = predictions.reshape(-1).contiguous().float()
    t = theta_out.reshape(-1).contiguous().float()
    o = masked_predictions.reshape(-1).contiguous().float()
    n = p.numel()
    if t.numel() != n or o.numel() != n or n == 0:
        raise ValueError("need three probability vectors of equal, non-zero length")
    sums = torch.empty(6, dtype=torch.float64, device=p.device)
    pc = torch.empty((5, n), dtype=torch.float32, device=p.device) if per_clip else None
    rc = _lib.lib().advh_lmac_metrics_accumulate(p.data_ptr(), t.data_ptr(), o.data_ptr(), n, sums.data_ptr(),
                                                 None if pc is None else pc.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "advh_lmac_metrics_accumulate")
    s = sums.cpu()
    res = {k: float(s[i] / s[5]) for i, k in enumerate(METRIC_NAMES)}
    return (res, pc) if per_clip else res


# ------------------------------------------------------------------------------------------ sharding
def shard_indices(n_total: int, rank: int, world: int) -> range:
    """Contiguous block partition of clip indices: rank r owns [r*ceil(N/W), ...)."""
    per = -(-n_total // world)
    return range(min(rank * per, n_total), min((rank + 1) * per, n_total))


def gather_probabilities(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """``local [n_r, 3]`` (p, theta, p_out of this rank's block) -> ``[n_total, 3]`` in clip order on every
    rank.  One all_gather of equal-sized, zero-padded blocks (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local                                             # no process group: single process (a one-rank group still runs the collective)
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":      # gloo has no GPU all_gather (one-GPU rehearsals): stage through the host
        local = local.cpu()
    pad = torch.zeros((per, 3), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat(parts, 0)[:n_total].to(dev)
