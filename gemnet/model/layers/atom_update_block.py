from gemnet_pytorch_amd.model.layers import AtomUpdateBlock, OutputBlock  # noqa: F401
