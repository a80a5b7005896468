"""Drop-in for the reference's ``addvisor`` module (addvisor.py:1-84): ``ConvBlock`` / ``UNet`` with the
reference's parameter names (so checkpoints load with ``load_state_dict``), forward on the HIP
implicit-GEMM kernels.  ``ADDvisor`` is the name LMAC_metrics.py:6 imports."""
import os
import warnings

import torch
import torch.nn as nn

from audioprocessor import AudioProcessor

audio_processor = AudioProcessor()


class ConvBlock(nn.Module):
    """addvisor.py:12-25 -- parameter container; the arithmetic runs inside ``UNet.forward``."""

    def __init__(self, in_ch, out_ch, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, kernel_size, stride=stride, padding=padding),
            nn.BatchNorm2d(out_ch),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_ch),
            nn.LeakyReLU(0.2, inplace=True),
        )

    def forward(self, x):
        """The reference's module forward (addvisor.py:24-25).  ``UNet.forward`` never calls it: the arithmetic of the
        whole network runs on the HIP kernels; it exists so code that holds a bare ConvBlock keeps working."""
        return self.block(x)


class _BatchStatEngine:
    """Inference with BATCH-statistics BatchNorm: what the reference's metrics script computes, because it never
    calls ``model.eval()`` (LMAC_metrics.py:18-26: ``nn.BatchNorm2d`` in train mode normalises with the statistics of the
    current batch and updates the running buffers, also under ``no_grad``).  Runs the training engine's forward
    (advh_bn_stats / advh_bn_coef / advh_bn_apply) on device copies of the module's tensors and writes the updated running
    statistics back into the module.  Same call shape as ``HipUNet`` (``forward(mag)``, ``precision``, ``flops``)."""

    def __init__(self, module: "UNet", dev):
        from addvisor_hip.unet_train import HipUNetTrain
        self.module, self.dev = module, dev
        self.tensors = {k: v.detach().to(dev, copy=True) for k, v in module.state_dict().items()}
        self.engine = HipUNetTrain(self.tensors, dev)
        self.precision = self.engine.precision

    def forward(self, mag, H: int = 512, W=None):
        mask = self.engine.forward(mag, H=H, W=W)
        with torch.no_grad():                              # the reference's buffers move with every batch it sees
            own = dict(self.module.named_buffers())
            for k, v in self.tensors.items():
                if k in own:
                    own[k].copy_(v)
        return mask

    def flops(self, B, H, W):
        ws = self.engine._workspace(B, H, W)
        return sum(L["fwd"].flops for L in ws["layers"] if "fwd" in L) + 2.0 * B * (H // 2) * W * 32 * 15 + 2.0 * B * H * W * 32


class UNet(nn.Module):
    def __init__(self, bn_mode=None):
        """``bn_mode`` (not in the reference's signature, ``UNet()``): how a module that is in ``train()`` mode normalises when
        gradients are disabled -- "eval" (default, or ``ADDVISOR_BN_MODE``): running statistics, BatchNorm folded into the
        convolutions (what ``streamlit_controlled_study.py:41`` does by calling ``.eval()``); "batch": batch statistics +
        running-buffer updates, i.e. the reference's metrics script exactly as written (LMAC_metrics.py:18-26 never calls
        ``.eval()``).  A module in ``eval()`` mode always uses its running statistics."""
        super().__init__()
        self.bn_mode = (bn_mode or os.environ.get("ADDVISOR_BN_MODE", "eval")).lower()
        if self.bn_mode not in ("eval", "batch"):
            raise ValueError("bn_mode must be 'eval' or 'batch'")
        self.e1 = ConvBlock(1, 32, kernel_size=(5, 3), stride=(2, 1), padding=(2, 1))
        self.e2 = ConvBlock(32, 64, kernel_size=(5, 3), stride=(2, 1), padding=(2, 1))
        self.e3 = ConvBlock(64, 128, stride=(2, 2))
        self.e4 = ConvBlock(128, 256, stride=(2, 2))
        self.bottleneck = nn.Sequential(
            nn.Conv2d(256, 512, kernel_size=3, padding=2, dilation=2), nn.BatchNorm2d(512), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(512, 512, kernel_size=3, padding=4, dilation=4), nn.BatchNorm2d(512), nn.LeakyReLU(0.2, inplace=True))
        self.up4 = nn.ConvTranspose2d(512, 256, kernel_size=(2, 2), stride=(2, 2))
        self.d4 = ConvBlock(384, 256)
        self.up3 = nn.ConvTranspose2d(256, 128, kernel_size=(2, 2), stride=(2, 2))
        self.d3 = ConvBlock(192, 128)
        self.up2 = nn.ConvTranspose2d(128, 64, kernel_size=(2, 1), stride=(2, 1))
        self.d2 = ConvBlock(96, 64)
        self.up1 = nn.ConvTranspose2d(64, 32, kernel_size=(2, 1), stride=(2, 1))
        self.d1 = ConvBlock(33, 32)
        self.mask_head = nn.Sequential(nn.Conv2d(32, 1, kernel_size=1), nn.Sigmoid())
        self._hip = None
        self._warned = False

    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in state_dict.items()}
        self._hip = None                                   # repack the fp16 weights on next forward
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _engine(self, dev):
        """The inference engine of the module's current state: eval-mode BatchNorm folded into the packed weights, or -- a
        module left in train() mode with ``bn_mode="batch"`` -- the batch-statistics forward."""
        batch = self.training and self.bn_mode == "batch"
        if self._hip is None or isinstance(self._hip, _BatchStatEngine) != batch:
            if batch:
                self._hip = _BatchStatEngine(self, dev)
            else:
                from addvisor_hip.unet import HipUNet
                self._hip = HipUNet({k: v.detach().cpu() for k, v in self.state_dict().items()}, dev)
        return self._hip

    def forward(self, x):
        """``x [B,1,F,T]`` (or ``[B,F,T]``) magnitude, F % 16 == 0 and T % 4 == 0 -> mask, same shape."""
        squeeze = x.dim() == 3
        x4 = x[:, None] if squeeze else x
        if x4.dim() != 4 or x4.shape[1] != 1:
            raise ValueError("expected [B,1,F,T]")
        B, _, Fq, Tq = x4.shape
        if Fq % 16 or Tq % 4:
            raise RuntimeError(f"U-Net skip connections need F % 16 == 0 and T % 4 == 0, got {Fq}x{Tq}")
        dev = torch.device("cuda")
        if self.training and torch.is_grad_enabled():
            # Training step (train_addvisor.py:364-378): the mask carries an autograd graph back to the parameters:
            # ONE autograd.Function over the HIP training kernels (addvisor_hip/unet_train.py: batch-statistics
            # BatchNorm, dgrad / wgrad as implicit GEMMs).  Parameters stay ordinary nn.Parameters, so torch optimisers
            # and DistributedDataParallel (RCCL gradient all-reduce) work unchanged.  There is no eager-torch path: the
            # parameters must live on the GPU.
            self._hip = None                                   # weights change: repack before the next inference forward
            pdev = self.mask_head[0].weight.device
            if pdev.type != "cuda":
                raise RuntimeError("UNet training runs on the HIP kernels only: move the module to the GPU (.to('cuda')); "
                                   "there is no CPU / eager-PyTorch path")
            mask = self._forward_hip_train(x4.to(pdev, torch.float32))
            return mask[:, 0] if squeeze else mask
        if self.training and self.bn_mode != "batch" and not self._warned:
            warnings.warn("UNet is in training mode but gradients are disabled; the HIP path uses eval-mode "
                          "BatchNorm (running statistics); UNet(bn_mode='batch') / ADDVISOR_BN_MODE=batch gives the "
                          "batch-statistics forward of the reference's metrics script, see SURVEY.md D5", stacklevel=2)
            self._warned = True
        mask = self._engine(dev).forward(x4[:, 0].to(dev, torch.float32).contiguous(), H=Fq, W=Tq)
        return mask if squeeze else mask[:, None]

    def _forward_hip_train(self, x):
        from addvisor_hip.unet_train import HipUNetTrain
        names = [k for k, _ in self.named_parameters()]
        tensors = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}     # live storage of params and buffers
        if getattr(self, "_train_engine", None) is None or self._train_engine.dev != x.device:
            self._train_engine = HipUNetTrain(tensors, x.device)
        self._train_engine.p = tensors
        return _UNetTrainFn.apply(self._train_engine, x, names, *[p for _, p in self.named_parameters()])


class _UNetTrainFn(torch.autograd.Function):
    """mask = UNet(x) in train() mode and its backward, both on the HIP kernels."""

    @staticmethod
    def forward(ctx, engine, x, names, *params):
        ctx.engine, ctx.names, ctx.shapes = engine, names, [p.shape for p in params]
        B, _, H, W = x.shape
        mask = engine.forward(x[:, 0].contiguous(), H=H, W=W)[:, None]
        ctx.generation = engine.generation               # the engine keeps ONE set of saved activations: stamp this forward
        return mask

    @staticmethod
    def backward(ctx, gmask):
        if ctx.generation != ctx.engine.generation:
            raise RuntimeError("UNet: backward of a training forward whose saved activations were overwritten by a later "
                               "training forward of the same module (the HIP engine keeps one set); call backward() "
                               "before the next forward, or accumulate gradients step by step")
        grads = ctx.engine.backward(gmask[:, 0])
        return (None, None, None) + tuple(grads[k].reshape(s) for k, s in zip(ctx.names, ctx.shapes))


ADDvisor = UNet
