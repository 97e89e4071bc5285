from gemnet_pytorch_amd.training.data_container import DataContainer  # noqa: F401
