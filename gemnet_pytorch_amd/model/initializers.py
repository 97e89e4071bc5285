"""He-orthogonal initialisation (counterpart of gemnet/model/initializers.py:4-40): a random
(semi-)orthogonal matrix, standardised to zero mean / unit variance over the fan-in axes and
scaled by 1/sqrt(fan_in).  Init-time only; parity tests load explicit weights instead."""
import torch


def he_orthogonal_init(tensor: torch.Tensor) -> torch.Tensor:
    with torch.no_grad():
        torch.nn.init.orthogonal_(tensor)
        if tensor.dim() == 3:
            axes, fan_in = (0, 1), tensor.shape[0] * tensor.shape[1]
        else:
            axes, fan_in = (1,), tensor.shape[1]
        var, mean = torch.var_mean(tensor, dim=axes, unbiased=True, keepdim=True)
        tensor.copy_((tensor - mean) / (var + 1e-6) ** 0.5 * (1.0 / fan_in) ** 0.5)
    return tensor
