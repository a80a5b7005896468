"""GPU parity of the LMAC loss BACKWARD (loss_function.py:36-66 under
train_addvisor.py:374-378) against torch autograd run through the CPU oracle.

Stated tolerance: in the default fp32-class mode (split-format operands and gradients, the reference's fp32 autograd class)
gradients are compared by max |err| / max |ref| <= 1e-4 and cosine similarity > 0.999999; with ADDVISOR_PRECISION=f16 (fp16
GEMM operands and gradients) 3e-2 / 0.999; the ISTFT adjoint alone is fp32: 1e-4."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from addvisor_hip import ops, runtime, synthetic as syn
from oracle import lmac_ref, signal_ref, wav2vec2_ref

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def spec(B, L, seed, audio_length):
    w = syn.make_clips(B, L, seed=seed)
    _, mag, ph = signal_ref.compute_stft(w, audio_length=audio_length)
    return w, mag.contiguous(), ph.contiguous()


@pytest.mark.parametrize("domain", ["linear", "log1p"])
@pytest.mark.parametrize("crop", [False, True])
def test_istft_masked_backward(gpu_device, domain, crop):
    """advh_istft_masked_bwd = the vector-Jacobian product of (mask -> masked ISTFT) for both branches."""
    B, L = 2, 16000
    _, mag, ph = spec(B, L, 61, 1)
    T = mag.shape[2]
    Fm, Tm = (512, 4 * (T // 4)) if crop else (513, T)
    gen = torch.Generator().manual_seed(62)
    mask = torch.rand(B, Fm, Tm, generator=gen)
    r_in, r_out = torch.randn(B, L, generator=gen), torch.randn(B, L, generator=gen)
    with torch.enable_grad():
        m = mask.clone().requires_grad_(True)
        rel, irr = signal_ref.apply_mask(signal_ref.embed_mask(m, 513, T), mag, ph, domain)
        w_in = signal_ref.compute_invert_stft(rel, audio_length=1)
        w_out = signal_ref.compute_invert_stft(irr, audio_length=1)
        g_in_ref, = torch.autograd.grad((w_in * r_in).sum(), m, retain_graph=True)
        g_out_ref, = torch.autograd.grad((w_out * r_out).sum(), m)
    d = gpu_device
    g_in = ops.istft_masked_bwd(r_in.to(d), mag.to(d), ph.to(d), mask.to(d), 0, domain=domain)
    g_out = ops.istft_masked_bwd(r_out.to(d), mag.to(d), ph.to(d), mask.to(d), 1, domain=domain)
    e_in, e_out = relerr(g_in.cpu(), g_in_ref), relerr(g_out.cpu(), g_out_ref)
    print(f"istft_masked_bwd {domain} crop={crop}: rel err in {e_in:.2e} out {e_out:.2e}")
    assert e_in < 1e-4 and e_out < 1e-4


@pytest.fixture
def tiny_runtime():
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    runtime.reset()
    yield
    os.environ.pop("ADDVISOR_EMBEDDER", None)
    runtime.reset()


def test_lmac_loss_backward_matches_oracle_autograd(gpu_device, tiny_runtime):
    """total.backward() through the drop-in LMACLoss: d total / d xhat and d total / d w_raw vs autograd on the
    oracle restatement of loss_function.py:32-66 (full 513 x T mask, the only shape the reference code runs)."""
    import loss_function
    B, L = 2, 80000
    w, mag, ph = spec(B, L, 71, 5)
    T = mag.shape[2]
    cfg, sd = runtime.embedder_config_and_weights()
    clf = runtime.classifier()
    coef, icpt = torch.as_tensor(clf.coef_, dtype=torch.float32), float(np.asarray(clf.intercept_).reshape(-1)[0])
    gen = torch.Generator().manual_seed(72)
    xhat0 = torch.rand(B, 1, 513, T, generator=gen)
    cp = torch.rand(B, 1, generator=gen)
    with torch.enable_grad():
        xr = xhat0.clone().requires_grad_(True)
        wr = torch.tensor([3.0, 0.5, 3.0], requires_grad=True)
        tot_ref, losses_ref, _ = lmac_ref.lmac_loss(xr, mag, ph, cp, wr, sd, cfg, coef.reshape(1, -1), icpt)
        gx_ref, gw_ref = torch.autograd.grad(tot_ref, [xr, wr])

        loss = loss_function.LMACLoss()
        xh = xhat0.clone().to(gpu_device).requires_grad_(True)
        total, losses, _ = loss.loss_function(xh, mag, ph, cp)
        total.backward()
    f32 = runtime.hip_embedder_grad().precision == "f32"
    assert (losses.detach().cpu() - losses_ref.detach()).abs().max().item() < (1e-5 if f32 else 1e-2)
    gx = xh.grad.cpu()
    err = relerr(gx, gx_ref)
    cos = F.cosine_similarity(gx.double().flatten(), gx_ref.double().flatten(), dim=0).item()
    print(f"d total / d xhat [{runtime.hip_embedder_grad().precision}]: max rel err {err:.3e}, cosine {cos:.8f}, |ref| max {gx_ref.abs().max():.3e}")
    assert (err < 1e-4 and cos > 0.999999) if f32 else (err < 3e-2 and cos > 0.999)
    assert torch.allclose(loss.w_raw.grad.cpu(), gw_ref, atol=1e-5 if f32 else 2e-3)
    # a descent step along the HIP gradient lowers the HIP loss (end-to-end sanity of sign and scale)
    with torch.no_grad():
        step = 0.05 / gx.abs().max()
        x2 = (xhat0 - step * gx.view_as(xhat0)).clamp(0, 1)
        t2, _, _ = loss.loss_function(x2.to(gpu_device), mag, ph, cp)
    assert t2.item() < total.item()


def test_training_loop_smoke(gpu_device, tiny_runtime):
    """train_addvisor.py:364-381 with the drop-in modules: torch U-Net (module API) + HIP loss backward + Adam."""
    import addvisor
    import loss_function
    B, L = 2, 80000
    _, mag, ph = spec(B, L, 81, 5)
    T4 = 4 * (mag.shape[2] // 4)
    cp = torch.tensor([[0.9], [0.2]])
    torch.manual_seed(0)
    with torch.enable_grad():
        net = addvisor.UNet().to(gpu_device)
        net.train()
        loss = loss_function.LMACLoss().to(gpu_device)
        opt_m = torch.optim.Adam(net.parameters(), lr=3e-4)
        opt_w = torch.optim.Adam(loss.parameters(), lr=1e-4)
        vals = []
        for _ in range(3):
            mask = net(mag[:, :512, :T4].unsqueeze(1).to(gpu_device))
            total, terms, _ = loss.loss_function(mask, mag, ph, cp)
            opt_m.zero_grad(); opt_w.zero_grad()
            total.backward()
            opt_m.step(); opt_w.step()
            vals.append(total.item())
    assert all(np.isfinite(vals)) and any(p.grad is not None and p.grad.abs().sum() > 0 for p in net.parameters())
    print("loss per step", vals)


def _ddp_gpu_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["ADDVISOR_EMBEDDER"] = "tiny"
    dist.init_process_group("gloo", rank=rank, world_size=world)       # one GPU here: the exchange is rehearsed over gloo
    import addvisor
    import loss_function
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = addvisor.UNet().to(dev)
    net.train()
    ddp = DDP(net)
    loss = loss_function.LMACLoss().to(dev)
    w = syn.make_clips(2, 80000, seed=500 + rank)                      # different clips per rank (utterance sharding)
    _, mag, ph = loss_function.audio_processor.compute_stft(w)
    T4 = 4 * (mag.shape[2] // 4)
    with torch.enable_grad():
        mask = ddp(mag[:, :512, :T4].unsqueeze(1).contiguous())
        total, _, _ = loss.loss_function(mask, mag, ph, torch.tensor([[0.8], [0.3]]))
        total.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, bool(torch.isfinite(flat).all()) and bool(flat.abs().sum() > 0) and all(torch.equal(g, gathered[0]) for g in gathered)))
    dist.destroy_process_group()


def test_data_parallel_training_step_two_ranks(gpu_device):
    """Data-parallel training step (train_addvisor.py:410-412): two ranks, one process each, different utterances; the
    U-Net gradients produced by the HIP backward are averaged by DistributedDataParallel (RCCL on a multi-GPU node;
    gloo here because both ranks share the one GPU of the test box) and end up identical on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=600) for _ in procs]
    [p.join(120) for p in procs]
    assert all(ok for _, ok in res) and all(p.exitcode == 0 for p in procs)


def test_train_addvisor_module_end_to_end(gpu_device, tiny_runtime, tmp_path):
    """The reference's training driver as functions (train_addvisor.py:200-393): metadata -> AudioDataset -> DataLoader with
    collate_fn -> train_addvisor for two epochs on a four-file synthetic corpus; the checkpoint it writes loads into a
    fresh UNet and gives the same inference mask."""
    import addvisor
    import loss_function
    import train_addvisor as T
    from addvisor_hip.wavio import write_wav
    root = tmp_path / "wavs"
    root.mkdir()
    names = []
    for i in range(4):
        write_wav(root / f"clip{i}.wav", syn.make_clips(1, 70000 + 3000 * i, seed=600 + i)[0], 16000, encoding="pcm16")
        names.append(f"clip{i}.wav")
    meta = tmp_path / "meta.txt"
    meta.write_text("".join(f"{n},bonafide\n" for n in names))
    assert T.extract_wavs(str(meta)) == names and T.extract_wavs(str(meta), one_sample_index=2) == ["clip2.wav"] * 2
    ds = T.AudioDataset(None, None, T.audio_processor, gpu_device, save_paths_txt=str(meta), root=str(root))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=T.collate_fn)
    feats, mag, ph, logits = next(iter(loader))
    assert mag.shape == ph.shape == (2, 513, 249) and logits.shape == (2, 1) and feats.shape[:2] == (2, 249)
    torch.manual_seed(1)
    net = addvisor.UNet().to(gpu_device)
    loss = loss_function.LMACLoss().to(gpu_device)
    hist = T.train_addvisor(net, 2, loss, loader, str(tmp_path / "ckpt"), optimizer_model=torch.optim.Adam(net.parameters(), lr=3e-4))
    assert len(hist) == 2 and all(np.isfinite(h).all() for h in hist) and hist[1][0] < hist[0][0]
    ckpts = sorted(os.listdir(tmp_path / "ckpt"))
    assert len(ckpts) == 2 and ckpts[0].startswith("addvisor_epoch_1_loss_")
    fresh = addvisor.UNet().to(gpu_device)
    fresh.load_state_dict(torch.load(tmp_path / "ckpt" / ckpts[1], map_location="cpu"))
    net.eval(); fresh.eval()
    with torch.no_grad():
        x = T.crop_for_model(mag)
        assert torch.equal(net(x), fresh(x))


def test_stale_training_forward_is_refused(gpu_device):
    """The HIP training engine keeps ONE set of saved activations per module: backward of a forward that a later training
    forward has overwritten must raise instead of silently using the newer activations (round-1 advisor finding)."""
    import addvisor
    torch.manual_seed(0)
    with torch.enable_grad():
        net = addvisor.UNet().to(gpu_device)
        net.train()
        x = torch.rand(2, 1, 32, 8, device=gpu_device)
        first = net(x)
        second = net(x * 0.5)
        with pytest.raises(RuntimeError, match="saved activations were overwritten"):
            first.sum().backward()
        second.sum().backward()                                   # the latest forward is still valid
    assert any(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in net.parameters())
    with torch.enable_grad():
        with pytest.raises(RuntimeError, match="no CPU"):
            addvisor.UNet().train()(torch.rand(1, 1, 32, 8))      # parameters on the CPU: no eager-torch path any more
