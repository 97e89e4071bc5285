from gemnet_pytorch_amd.model.layers import AtomEmbedding, EdgeEmbedding  # noqa: F401
