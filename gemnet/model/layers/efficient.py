from gemnet_pytorch_amd.model.layers import EfficientInteractionDownProjection, EfficientInteractionBilinear  # noqa: F401
