"""Drop-in `modeling` package: same factory, class names, constructor kwargs and state-dict keys as the
reference's modeling/ (modeling/__init__.py:6-10), with every operator running in libdm_hip kernels."""
from modeling.predictor import BasePredictor, GuidedPredictor
from modeling.unet import UNet

_MODELS = {"unet": UNet}


def get_model(name, **kwargs) -> BasePredictor:
    try:
        cls = _MODELS[name]
    except KeyError:
        raise ValueError(f"Unknown model name: {name}") from None
    return cls(**kwargs)
