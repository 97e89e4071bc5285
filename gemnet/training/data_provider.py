from gemnet_pytorch_amd.training.data_provider import DataProvider, collate  # noqa: F401
