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
        # The reference never calls .eval() (LMAC_metrics.py:18-26), so its BatchNorm layers use batch statistics; that is the
        # ADDVISOR_BN_MODE=batch switch.  Default: eval-mode BatchNorm, as the streamlit study does.
        _model = m.train() if m.bn_mode == "batch" else m.eval()
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


class LazyTensor:
    """A tensor that is computed on first use.  ``collate_fn`` (LMAC_metrics.py:109-114) returns the STFT magnitude / phase
    and the wav2vec2 features of every batch; the reference's loop reads none of the values it does not need, and the fused
    HIP path recomputes nothing from them -- so they are produced only if somebody actually touches them (attribute access,
    indexing, arithmetic, any ``torch.*`` call).  SURVEY.md D12: collate does no device work of its own."""
    __slots__ = ("_fn", "_v")

    def __init__(self, fn):
        self._fn, self._v = fn, None

    @property
    def materialized(self) -> bool:
        return self._v is not None

    def materialize(self) -> torch.Tensor:
        if self._v is None:
            self._v, self._fn = self._fn(), None
        return self._v

    def __getattr__(self, name):
        return getattr(self.materialize(), name)

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return f"LazyTensor({'pending' if self._v is None else repr(self._v)})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a.materialize() if isinstance(a, LazyTensor) else a
        return func(*[un(a) for a in args], **{k: un(v) for k, v in (kwargs or {}).items()})


def _binop(name):
    def op(self, *a):
        return getattr(self.materialize(), name)(*[x.materialize() if isinstance(x, LazyTensor) else x for x in a])
    op.__name__ = name
    return op


for _n in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "matmul", "neg", "abs", "lt", "le", "gt", "ge", "eq", "ne"):
    setattr(LazyTensor, f"__{_n}__", _binop(f"__{_n}__"))


def _tensor(x):
    return x.materialize() if isinstance(x, LazyTensor) else x


def collate_fn(batch):
    """LMAC_metrics.py:109-114: same return tuple ``(waveforms, magnitude, phase, features, filenames)``.  The reference runs
    the STFT and a full embedder pass here, on the loader's thread; this collate only stacks the clips -- magnitude, phase
    and features are ``LazyTensor``s, computed on first use (``run_addvisor_metrics`` below never needs them: the fused step
    works from the waveforms)."""
    waveforms, filenames = zip(*batch)
    waveforms = torch.stack(waveforms, dim=0)
    stft = LazyTensor(lambda: audio_processor.compute_stft(waveforms))          # one STFT serves magnitude and phase
    magnitude = LazyTensor(lambda: stft.materialize()[1])
    phase = LazyTensor(lambda: stft.materialize()[2])
    features = LazyTensor(lambda: audio_processor.extract_features(waveforms))
    return waveforms, magnitude, phase, features, filenames


_pipes = {}


def _pipeline(domain="log1p"):
    """The fused explanation step (addvisor_hip.pipeline.ExplainPipeline) over the process-wide embedder and this module's
    U-Net: STFT -> (X, |X|) -> mask -> complex masked ISTFT x2 (no phase, no atan2 / sincos) -> ONE 3B-clip embedder pass."""
    ap = audio_processor
    unet = get_model()._engine(device)
    emb = _rt.hip_embedder()
    key = (float(ap.audio_length), ap.sampling_rate, ap.hop_length, ap.win_length, domain, id(unet), id(emb))
    if key not in _pipes:
        _pipes.clear()                                                            # a reloaded checkpoint / new geometry replaces the old plan
        _pipes[key] = _P.ExplainPipeline(None, None, None, None, None, device, audio_length=ap.audio_length, sampling_rate=ap.sampling_rate,
                                         domain=domain, hop=ap.hop_length, win=ap.win_length, embedder=emb, unet=unet)
    return _pipes[key]


def explain_batch(waveforms, magnitude=None, phase=None, domain="log1p"):
    """Loop body of run_addvisor_metrics (LMAC_metrics.py:125-157) for one batch; returns the three
    probability vectors ``(predictions, theta_out, masked_predictions)``, each ``[B,1]``.

    With the collate's lazy ``magnitude`` / ``phase`` (or None) the batch takes the fused step: 3 embedder passes per clip in
    one 3B batch, the masked resynthesis straight from the complex spectrogram.  Explicit spectrogram TENSORS (a caller that
    edited them) are honoured: mask and resynthesis are computed from exactly those."""
    ap = audio_processor
    L = int(ap.audio_length * ap.sampling_rate)
    w = waveforms.to(device, torch.float32)
    explicit = lambda t: t is not None and not (isinstance(t, LazyTensor) and not t.materialized)
    if not explicit(magnitude) and not explicit(phase):
        out = _pipeline(domain).explain(w)
        return out["predictions"], out["theta_out"], out["masked_predictions"]
    magnitude, phase = _tensor(magnitude), _tensor(phase)
    emb = _rt.hip_embedder()
    T = magnitude.shape[-1]
    mask = get_model()(magnitude[:, None, :512, :(T // 4) * 4])[:, 0]
    w_in, w_out = _ops.istft_masked(magnitude, phase, mask, L, domain=domain, hop=ap.hop_length, win=ap.win_length)
    B = w.shape[0]
    allw = torch.zeros((3 * B, L), dtype=torch.float32, device=device)
    n = min(L, w.shape[1])
    allw[:B, :n] = w[:, :n]
    allw[B:2 * B], allw[2 * B:] = w_in, w_out
    _, _, p3 = emb.forward(allw, L, want_hidden=False)                            # clean, mask-in, mask-out: one 3B batch
    return p3[:B], p3[B:2 * B], p3[2 * B:]


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
