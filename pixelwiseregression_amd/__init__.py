"""pixelwiseregression_amd -- MI355X-native hot path of IcarusWizard/PixelwiseRegression.

Public surface mirrors the reference's ``model.py`` / ``utils.py`` names for the hot path:
``PixelwiseRegression`` (model.py:154), ``recover_uvd`` (utils.py:332), ``uvd2xyz``
(datasets.py:100), ``save_model`` / ``load_model`` (utils.py:302-314), ``make_targets`` (the dense targets of
datasets.py:285-294 / :365-383, on the device), ``preprocess_batch`` (the crop / resize / augmentation pipeline of
datasets.py:243-299 for a batch of raw depth frames in HBM), ``StreamedInference`` (the evaluation loop of test.py:93-102 with two
batches in flight on two HIP streams).
"""
from .model import PixelwiseRegression  # noqa: F401
from .metric import recover_uvd, uvd2xyz, mean_joint_error, INTRINSICS  # noqa: F401
from .checkpoint import save_model, load_model  # noqa: F401
from .targets import make_targets  # noqa: F401
from .preprocess import preprocess_batch, draw_augmentation  # noqa: F401
from .serving import StreamedInference  # noqa: F401

__all__ = ["PixelwiseRegression", "recover_uvd", "uvd2xyz", "mean_joint_error", "INTRINSICS", "save_model", "load_model", "make_targets", "preprocess_batch",
           "draw_augmentation", "StreamedInference"]
