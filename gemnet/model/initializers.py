from gemnet_pytorch_amd.model.initializers import he_orthogonal_init  # noqa: F401
