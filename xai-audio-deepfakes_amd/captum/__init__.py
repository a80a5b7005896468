"""Minimal ``captum`` stand-in so that ``from captum.attr import Saliency, InputXGradient,
IntegratedGradients`` (captum_saliency.py:3) resolves to the HIP attribution path when this directory is
first on ``sys.path``.  Only the three methods the reference names are provided."""
