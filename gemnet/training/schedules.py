from gemnet_pytorch_amd.training.schedules import LinearWarmupExponentialDecay  # noqa: F401
