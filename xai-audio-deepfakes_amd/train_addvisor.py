"""Drop-in for the reference's ``train_addvisor`` module (train_addvisor.py:1-420) as FUNCTIONS: importing it runs
nothing (the reference builds its data set, hands the model to ``accelerate`` and trains for 1000 epochs at import,
train_addvisor.py:396-420).  Dataset, collate function and training loop keep the reference's names and flow; every
tensor operation is the HIP path:

  collate_fn          STFT + embedder + pooled logreg for the batch               (train_addvisor.py:247-260)
  train_addvisor      mask = model(magnitude); LMAC loss; backward; two Adam steps (train_addvisor.py:345-393)
                      -- mask decoder forward/backward: addvisor_hip/unet_train.py; loss forward/backward through ISTFT
                      and the frozen embedder: addvisor_hip/lmac_loss.py

Deviations, both forced: D2 -- the U-Net only accepts F % 16 == 0 and T % 4 == 0, so the magnitude is
cropped to ``[:, :512, :4*floor(T/4)]`` before the model (the loss embeds the mask back with zeros outside); and
``extract_wavs`` returns every file instead of the debugging leftover ``[audio_files[22000]] * 2``
(train_addvisor.py:208) unless ``one_sample_index`` is given.  Plot helpers (matplotlib) are out of scope.
For multi-GPU training wrap the model in ``torch.nn.parallel.DistributedDataParallel`` (what ``accelerator.prepare`` does,
train_addvisor.py:410-412): the parameters are ordinary ``nn.Parameter``s and the HIP backward feeds DDP's all-reduce.
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset  # noqa: F401  (names the reference module exposes)

from addvisor import UNet  # noqa: F401
from audioprocessor import AudioProcessor
from loss_function import LMACLoss  # noqa: F401

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
audio_processor = AudioProcessor()


def extract_wavs(metadata, one_sample_index=None):
    """train_addvisor.py:200-210: first CSV field of every metadata line."""
    audio_files = []
    with open(metadata, "r") as f:
        for path in f:
            audio_files.append(path.strip().split(",")[0])
    if one_sample_index is not None:
        return [audio_files[one_sample_index], audio_files[one_sample_index]]
    return audio_files


class AudioDataset(Dataset):
    """train_addvisor.py:213-244; ``root`` replaces the hard-coded "LJSpeech_vocoded22K" folder."""

    def __init__(self, directory1, directory2, audio_processor, device,
                 save_paths_txt="metadata/ljspeech_manipulated_metadata.txt", root="LJSpeech_vocoded22K",
                 one_sample_index=None):
        self.file_paths = extract_wavs(save_paths_txt, one_sample_index)
        self.audio_processor = audio_processor
        self.device = device
        self.root = root

    def __len__(self):
        return len(self.file_paths)

    def __getitem__(self, idx):
        audio_path = self.file_paths[idx]
        waveform, sr = self.audio_processor.load_audio(os.path.join(self.root, audio_path))
        return waveform.to(self.device), audio_path


def collate_fn(batch):
    """train_addvisor.py:247-260 -> ``(features, magnitude, phase, yhat_logits)``."""
    waveforms, audio_paths = zip(*batch)
    waveforms = torch.stack(waveforms, dim=0)
    _, magnitude, phase = audio_processor.compute_stft(waveforms)
    features = audio_processor.extract_features(waveforms)
    yhat_logits, _ = audio_processor.classify(waveforms)            # mean over time + TorchLogReg, fused behind the embedder
    return features, magnitude, phase, yhat_logits


def crop_for_model(magnitude):
    """``[B, 513, T] -> [B, 1, 512, 4*floor(T/4)]``."""
    T4 = 4 * (magnitude.shape[-1] // 4)
    return magnitude[:, :512, :T4].unsqueeze(1).contiguous()


def train_addvisor(model, num_epochs, loss_fn, data_loader, save_path, optimizer_model=None, optimizer_w=None,
                   log_every=0):
    """train_addvisor.py:345-393.  Returns the per-epoch mean of ``(total, l_in, l_out, l1)``.  A checkpoint
    ``addvisor_epoch_<n>_loss_<x>.pth`` (plain ``state_dict``, what ``LMAC_metrics.py:21-26`` loads) is written per epoch
    when ``save_path`` is given."""
    if optimizer_model is None:
        optimizer_model = torch.optim.Adam(model.parameters(), lr=3e-5)             # train_addvisor.py:104
    if optimizer_w is None:
        optimizer_w = torch.optim.Adam([loss_fn.w_raw], lr=1e-4)                    # train_addvisor.py:105
    history = []
    model.train()
    for epoch in range(num_epochs):
        tot = [0.0, 0.0, 0.0, 0.0]
        for i, batch in enumerate(data_loader):
            features, magnitude, phase, yhat_logits = batch
            magnitude, phase = magnitude.to(device), phase.to(device)
            with torch.enable_grad():
                mask = model(crop_for_model(magnitude))
                loss_value, individual_losses, weights = loss_fn.loss_function(mask, magnitude, phase, torch.sigmoid(yhat_logits))
                optimizer_model.zero_grad()
                optimizer_w.zero_grad()
                loss_value.backward()
            optimizer_model.step()
            optimizer_w.step()
            # train_addvisor.py:380-381 renormalises ``loss_fn.w.data``; ``w`` is a property computed from ``w_raw``, so the
            # assignment acts on a temporary and changes nothing -- reproduced by doing nothing
            vals = [loss_value.item()] + [v.item() for v in individual_losses]
            tot = [a + b for a, b in zip(tot, vals)]
            if log_every and i % log_every == 0:
                print(f"epoch {epoch + 1} batch {i}: loss {vals[0]:.4f} (l_in {vals[1]:.4f}, l_out {vals[2]:.4f}, l1 {vals[3]:.4f})")
        n = max(len(data_loader), 1)
        history.append(tuple(t / n for t in tot))
        if save_path:
            os.makedirs(save_path, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(save_path, f"addvisor_epoch_{epoch + 1}_loss_{history[-1][0]:.4f}.pth"))
    return history
