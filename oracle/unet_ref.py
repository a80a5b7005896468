"""Oracle: the ADDvisor U-Net mask decoder.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Functional restatement of addvisor.py:12-84 over a state dict with the reference's names.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(x, sd, p, train_stats: bool):
    """nn.BatchNorm2d, eps 1e-5.  ``train_stats`` reproduces LMAC_metrics.py as written
    (no model.eval(): batch statistics); default is eval (running stats)."""
    return F.batch_norm(x, sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone(),
                        sd[p + ".weight"], sd[p + ".bias"], training=train_stats, momentum=0.1, eps=1e-5)


def conv_block(x, sd, name, stride=(1, 1), padding=(1, 1), bn_batch=False):
    """addvisor.py:12-25 -- Conv2d -> BN -> LeakyReLU(0.2) -> Conv2d 3x3 p1 -> BN -> LeakyReLU(0.2)."""
    x = F.conv2d(x, sd[f"{name}.block.0.weight"], sd[f"{name}.block.0.bias"], stride=stride, padding=padding)
    x = F.leaky_relu(_bn(x, sd, f"{name}.block.1", bn_batch), 0.2)
    x = F.conv2d(x, sd[f"{name}.block.3.weight"], sd[f"{name}.block.3.bias"], padding=1)
    return F.leaky_relu(_bn(x, sd, f"{name}.block.4", bn_batch), 0.2)


def unet_forward(x: torch.Tensor, sd, bn_batch: bool = False, return_logits: bool = False) -> torch.Tensor:
    """addvisor.py:62-84.  ``x [B,1,F,T]`` with F % 16 == 0, T % 4 == 0 -> mask ``[B,1,F,T]``."""
    x1 = conv_block(x, sd, "e1", stride=(2, 1), padding=(2, 1), bn_batch=bn_batch)      # :31
    x2 = conv_block(x1, sd, "e2", stride=(2, 1), padding=(2, 1), bn_batch=bn_batch)     # :32
    x3 = conv_block(x2, sd, "e3", stride=(2, 2), bn_batch=bn_batch)                     # :33
    x4 = conv_block(x3, sd, "e4", stride=(2, 2), bn_batch=bn_batch)                     # :34
    b = F.conv2d(x4, sd["bottleneck.0.weight"], sd["bottleneck.0.bias"], padding=2, dilation=2)   # :36-43
    b = F.leaky_relu(_bn(b, sd, "bottleneck.1", bn_batch), 0.2)
    b = F.conv2d(b, sd["bottleneck.3.weight"], sd["bottleneck.3.bias"], padding=4, dilation=4)
    b = F.leaky_relu(_bn(b, sd, "bottleneck.4", bn_batch), 0.2)
    y4 = F.conv_transpose2d(b, sd["up4.weight"], sd["up4.bias"], stride=(2, 2))         # :45
    y4 = conv_block(torch.cat([y4, x3], 1), sd, "d4", bn_batch=bn_batch)
    y3 = F.conv_transpose2d(y4, sd["up3.weight"], sd["up3.bias"], stride=(2, 2))        # :48
    y3 = conv_block(torch.cat([y3, x2], 1), sd, "d3", bn_batch=bn_batch)
    y2 = F.conv_transpose2d(y3, sd["up2.weight"], sd["up2.bias"], stride=(2, 1))        # :51
    y2 = conv_block(torch.cat([y2, x1], 1), sd, "d2", bn_batch=bn_batch)
    y1 = F.conv_transpose2d(y2, sd["up1.weight"], sd["up1.bias"], stride=(2, 1))        # :54
    y1 = conv_block(torch.cat([y1, x], 1), sd, "d1", bn_batch=bn_batch)
    logits = F.conv2d(y1, sd["mask_head.0.weight"], sd["mask_head.0.bias"])             # :57-60
    return logits if return_logits else torch.sigmoid(logits)


def crop_for_unet(mag: torch.Tensor):
    """feed ``mag[:, :512, :4*floor(T/4)]`` (the reference has no runnable shape contract)."""
    B, Fq, T = mag.shape
    Fm = (min(Fq, 512) // 16) * 16
    Tm = (T // 4) * 4
    return mag[:, None, :Fm, :Tm]
