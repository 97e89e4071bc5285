from gemnet_pytorch_amd.model.scaling import AutomaticFit, AutoScaleFit, ScalingFactor  # noqa: F401
