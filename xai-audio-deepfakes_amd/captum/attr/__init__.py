"""``captum.attr``-compatible front ends (captum_saliency.py:3, 116-118, 131-135) over the HIP backward path.

``Method(model).attribute(inputs, target=None, ...)`` expects ``model`` to be a
``captum_saliency.Wav2vec2LogReg`` (or anything exposing ``.hip_attribution()``): the waveform -> logit
classifier whose frozen embedder runs on the GPU kernels.  Arbitrary ``nn.Module``s are not supported --
there is no autograd fallback."""
import torch


def _engine(model):
    if not hasattr(model, "hip_attribution"):
        raise TypeError("captum.attr (HIP build) only attributes captum_saliency.Wav2vec2LogReg models")
    return model.hip_attribution()


class _Method:
    def __init__(self, forward_func):
        self.model = forward_func

    @staticmethod
    def _check(inputs, target):
        if target is not None:
            raise NotImplementedError("the classifier has a single output; target must be None")
        if not torch.is_tensor(inputs) or inputs.dim() != 2:
            raise ValueError("inputs must be a [B, L] waveform tensor")


class Saliency(_Method):
    def attribute(self, inputs, target=None, abs=True, additional_forward_args=None):
        self._check(inputs, target)
        eng = _engine(self.model)
        return eng.saliency(inputs) if abs else eng.input_gradient(inputs)


class InputXGradient(_Method):
    def attribute(self, inputs, target=None, additional_forward_args=None):
        self._check(inputs, target)
        return _engine(self.model).input_x_gradient(inputs)


class IntegratedGradients(_Method):
    def __init__(self, forward_func, multiply_by_inputs=True):
        super().__init__(forward_func)
        if not multiply_by_inputs:
            raise NotImplementedError("multiply_by_inputs=False")

    def attribute(self, inputs, baselines=None, target=None, additional_forward_args=None, n_steps=50,
                  method="gausslegendre", internal_batch_size=None, return_convergence_delta=False):
        self._check(inputs, target)
        if torch.is_tensor(baselines):
            zero = not bool(baselines.any())
        else:
            zero = baselines is None or (isinstance(baselines, (int, float)) and baselines == 0)
        if not zero:
            raise NotImplementedError("only the zero baseline (Captum's default) is built")
        if method != "gausslegendre":
            raise NotImplementedError("only Captum's default 'gausslegendre' rule is built")
        return _engine(self.model).integrated_gradients(inputs, n_steps=n_steps, internal_batch_size=internal_batch_size)
