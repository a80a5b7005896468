"""Drop-in for the reference's ``LMAC_metrics`` module (LMAC_metrics.py:1-178): the six metric functions,
``extract_wavs`` / ``AudioDataset`` / ``collate_fn`` and ``run_addvisor_metrics``, with the explanation loop
running on the HIP pipeline.  Importing it loads no checkpoint; set ``ADDVISOR_UNET_CKPT`` to a ``.pth``
(``module.`` prefixes are stripped, LMAC_metrics.py:23-25) or the seeded synthetic U-Net is used."""
import os

import torch
from torch.nn import functional as F

from addvisor import ADDvisor
from addvisor_hip import ops as _ops, pipeline as _P, runtime as _rt
from audioprocessor import AudioProcessor
from classifier_embedder import TorchLogReg  # noqa: F401

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
audio_processor = AudioProcessor()
eps = 1e-10
_model = None


def get_model():
    global _model
    if _model is None:
        from addvisor_hip import synthetic as syn
        m = ADDvisor()
        path = os.environ.get("ADDVISOR_UNET_CKPT")
        m.load_state_dict(torch.load(path, map_location="cpu") if path else syn.unet_weights())
        _model = m.eval()
    return _model


def _vec(x):
    return x.reshape(-1).to(device, torch.float32).contiguous()


def _per_clip(theta_out, predictions, masked=None):
    """Rows of advh_lmac_metrics_accumulate's per-clip output: faithfulness, fidelity, AD, AI, AG."""
    p, t = _vec(predictions), _vec(theta_out)
    o = _vec(masked) if masked is not None else p
    return _P.lmac_metrics(p, t, o, per_clip=True)[1]


@torch.no_grad()
def compute_fidelity(theta_out, predictions, threshold=0.5):
    """LMAC_metrics.py:31-38 (threshold fixed at 0.5 like every call site)."""
    return _per_clip(theta_out, predictions)[1].view(*predictions.shape)


def get_score_for_predicted_class(p):
    """LMAC_metrics.py:43-45."""
    pred = (p > 0.5).float()
    return pred * p + (1 - pred) * (1 - p)


@torch.no_grad()
def compute_faithfulness(predictions, predictions_masked):
    """LMAC_metrics.py:48-52."""
    return _per_clip(predictions, predictions, predictions_masked)[0]


@torch.no_grad()
def compute_AD(theta_out, predictions):
    """LMAC_metrics.py:55-59."""
    return _per_clip(theta_out, predictions)[2]


@torch.no_grad()
def compute_AI(theta_out, predictions):
    """LMAC_metrics.py:62-66."""
    return _per_clip(theta_out, predictions)[3]


@torch.no_grad()
def compute_AG(theta_out, predictions):
    """LMAC_metrics.py:69-73."""
    return _per_clip(theta_out, predictions)[4]


def extract_wavs(metadata):
    """LMAC_metrics.py:76-81: first CSV field of every line."""
    audio_files = []
    with open(metadata, "r") as f:
        for path in f:
            audio_files.append(path.strip().split(",")[0])
    return audio_files


class AudioDataset(torch.utils.data.Dataset):
    """LMAC_metrics.py:84-106; ``root`` replaces the hard-coded "LJSpeech_vocoded" folder."""

    def __init__(self, directory1, directory2, audio_processor, device,
                 metadata="metadata/ljspeech_manipulated_metadata.txt", root="LJSpeech_vocoded"):
        self.file_paths = extract_wavs(metadata)
        self.audio_processor = audio_processor
        self.device = device
        self.root = root

    def __len__(self):
        return len(self.file_paths)

    def __getitem__(self, idx):
        path = self.file_paths[idx]
        waveform, _ = self.audio_processor.load_audio(os.path.join(self.root, path))
        return waveform.to(self.device), os.path.basename(path)


def collate_fn(batch):
    """LMAC_metrics.py:109-114: same return tuple; the STFT and the embedder run on the GPU stream."""
    waveforms, filenames = zip(*batch)
    waveforms = torch.stack(waveforms, dim=0)
    _, magnitude, phase = audio_processor.compute_stft(waveforms)
    features = audio_processor.extract_features(waveforms)
    return waveforms, magnitude, phase, features, filenames


def explain_batch(waveforms, magnitude, phase, domain="log1p"):
    """Loop body of run_addvisor_metrics (LMAC_metrics.py:125-157) for one batch; returns the three
    probability vectors ``(predictions, theta_out, masked_predictions)``, each ``[B,1]``."""
    ap = audio_processor
    L = int(ap.audio_length * ap.sampling_rate)
    emb = _rt.hip_embedder()
    _, _, probs_clean = emb.forward(waveforms.to(device, torch.float32), L, want_hidden=False)
    T = magnitude.shape[-1]
    mask = get_model()(magnitude[:, None, :512, :(T // 4) * 4])[:, 0]
    w_in, w_out = _ops.istft_masked(magnitude, phase, mask, L, domain=domain, hop=ap.hop_length, win=ap.win_length)
    _, _, probs_in = emb.forward(w_in, L, want_hidden=False)
    _, _, probs_out = emb.forward(w_out, L, want_hidden=False)
    return probs_clean, probs_in, probs_out


def run_addvisor_metrics(dir_path1, dir_path2, batch_size=4, dataset=None):
    """LMAC_metrics.py:117-172: prints the five means with two decimals.

    Under ``torch.distributed`` (one process per GPU, e.g. ``torchrun --nproc-per-node 8``: BASELINE config 4) every rank walks
    only its contiguous block of the data set (``pipeline.shard_indices``: utterances are independent, so there is no data-path
    collective), the per-clip probabilities are combined by ONE all_gather into clip order (``gather_probabilities``: RCCL over
    xGMI) and every rank reduces the same vector in the same order: the five numbers are bit-identical for any world size;
    rank 0 prints them."""
    from torch.utils.data import DataLoader, Subset
    import torch.distributed as dist
    dataset = dataset or AudioDataset(dir_path1, dir_path2, audio_processor, device)
    n_total = len(dataset)
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if sharded:
        dataset = Subset(dataset, list(_P.shard_indices(n_total, dist.get_rank(), dist.get_world_size())))
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, collate_fn=collate_fn)
    theta_out, predictions, masked_predictions = [], [], []
    for waveforms, magnitude, phase, features, filenames in loader:
        with torch.no_grad():
            p, t, o = explain_batch(waveforms, magnitude, phase)
            predictions.append(p), theta_out.append(t), masked_predictions.append(o)
    if predictions:
        local = torch.cat([torch.cat(predictions, 0), torch.cat(theta_out, 0), torch.cat(masked_predictions, 0)], 1)
    else:                                                   # a rank whose block is empty (more ranks than clips)
        local = torch.zeros((0, 3), dtype=torch.float32, device=device)
    if sharded:
        local = _P.gather_probabilities(local.contiguous(), n_total)
    predictions, theta_out, masked_predictions = (local[:, i:i + 1].contiguous() for i in range(3))
    m = _P.lmac_metrics(predictions, theta_out, masked_predictions)
    if sharded and dist.get_rank() != 0:
        return m
    print(f"faithfulness : {m['faithfulness']:.2f}")
    print(f"fidelity: {m['fidelity']:.2f}")
    print(f"average drop : {m['AD']:.2f}")
    print(f"average increase: {m['AI']:.2f}")
    print(f"average gain : {m['AG']:.2f}")
    return m
