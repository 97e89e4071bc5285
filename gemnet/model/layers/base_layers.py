from gemnet_pytorch_amd.model.layers import Dense, ResidualLayer  # noqa: F401
