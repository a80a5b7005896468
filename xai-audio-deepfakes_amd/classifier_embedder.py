"""Drop-in for the reference's ``classifier_embedder`` module (classifier_embedder.py:1-63):
same names (``classifier``, ``processor``, ``wav2vec2``, ``TorchLogReg``, ``zero_mean_unit_var_norm``),
MI355X kernels underneath, and NO import-time fetch: see ``addvisor_hip.runtime`` for where weights
come from."""
import torch
import torch.nn as nn

from addvisor_hip import runtime as _rt


class _Lazy:
    def __init__(self, factory):
        object.__setattr__(self, "_f", factory)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_f")(), name)


classifier = _Lazy(_rt.classifier)          # .coef_ (1,H), .intercept_ (1,)   (classifier_embedder.py:12)
processor = None                            # the HF feature extractor is never used by the hot path (:13)


class _HiddenStates:
    """``output.hidden_states``: index k runs the first k encoder layers on the GPU."""

    def __init__(self, x):
        self._x = x

    def __getitem__(self, k):
        emb = _rt.hip_embedder()
        if k != emb.cfg.layer_index:
            raise IndexError(f"this embedder is built for hidden_states[{emb.cfg.layer_index}] "
                             "(set ADDVISOR_LAYER_INDEX to change it)")
        hid, _, _ = emb.forward(self._x, normalize=False)
        return hid


class _Output:
    def __init__(self, x):
        self.hidden_states = _HiddenStates(x)


class _Wav2Vec2:
    """``wav2vec2(input_values, output_hidden_states=True).hidden_states[9]`` (audioprocessor.py:76-77).
    ``input_values`` is the already normalised waveform, as in the reference."""

    def __call__(self, input_values, output_hidden_states=True):
        x = input_values
        if x.dim() == 1:
            x = x[None]
        return _Output(x.to(_rt.device(), torch.float32).contiguous())

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    @property
    def config(self):
        return _rt.embedder_config_and_weights()[0]


wav2vec2 = _Wav2Vec2()


class TorchLogReg(nn.Module):
    """classifier_embedder.py:21-38: Linear(H, 1) initialised from the sklearn model; (logits, probs)."""

    def __init__(self):
        super().__init__()
        clf = _rt.classifier()
        self.linear = nn.Linear(clf.coef_.shape[1], 1)
        self.linear.weight = nn.Parameter(torch.tensor(clf.coef_, dtype=torch.float32), requires_grad=False)
        self.linear.bias = nn.Parameter(torch.tensor(clf.intercept_, dtype=torch.float32), requires_grad=False)

    def forward(self, x):
        logits = self.linear(x)
        return logits, torch.sigmoid(logits)


def zero_mean_unit_var_norm(input_values):
    """classifier_embedder.py:59-63 (kept as tensor ops so autograd callers keep working; the fused HIP
    front end applies the same normalisation inside ``AudioProcessor.extract_features``)."""
    mean = input_values.mean(dim=-1, keepdim=True)
    std = input_values.std(dim=-1, keepdim=True)
    return (input_values - mean) / (std + 1e-7)
