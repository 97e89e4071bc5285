from gemnet_pytorch_amd.model.layers import InteractionBlock, InteractionBlockTripletsOnly, TripletInteraction, QuadrupletInteraction  # noqa: F401
