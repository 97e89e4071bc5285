from gemnet_pytorch_amd.training.trainer import Trainer, MultiWrapper  # noqa: F401
from gemnet_pytorch_amd.training.schedules import ReduceLROnPlateau  # noqa: F401
