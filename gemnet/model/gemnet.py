from gemnet_pytorch_amd.model.gemnet import GemNet  # noqa: F401
