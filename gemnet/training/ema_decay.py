from gemnet_pytorch_amd.training.ema_decay import ExponentialMovingAverage  # noqa: F401
