"""MI355X-native (gfx950) implementation of GemNet's InteractionBlock hot path.

Drop-in boundary: ``gemnet_pytorch_amd.model.gemnet.GemNet`` (also importable as
``gemnet.model.gemnet.GemNet``) keeps the reference's constructor/forward/predict API
(/root/reference/gemnet/model/gemnet.py:21-790).  All device compute goes through the
C-ABI library ``gemnet_pytorch_amd/csrc/libgemnet_hip.so`` (declared in
``include/gemnet_hip.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"
