from gemnet_pytorch_amd.model.layers import BesselBasisLayer, SphericalBasisLayer, TensorBasisLayer  # noqa: F401
