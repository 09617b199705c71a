"""Parameter-holding layers. Each subclasses the torch layer the reference instantiates, so initialisation,
parameter names/shapes and state-dict behaviour are identical; only `forward` is replaced by libdm_hip kernels."""
from typing import Optional

import torch
from torch import nn

from dm_hip import ops


class Conv2d(nn.Conv2d):
    """3x3 / 1x1 convolution as implicit GEMM on bf16 MFMA. Optional fused extras:
    `x2`  second tensor concatenated on the channel axis (never materialised),
    `res` residual added in the epilogue,
    `ups` nearest x2 upsampling of the input fused into the gather."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding)
        k = self.kernel_size[0]
        if self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1]:
            raise ValueError("Conv2d: square kernels and strides only")
        self._pad = (k - 1) // 2 if padding == "same" else int(self.padding[0])
        self._pack = ops.WeightPack()

    def forward(self, x, x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, ups: int = 0,
                bias_grad: bool = True, bias_twin=None, norm=None, box=None):
        """norm: a FusedNorm from GroupNorm.fused(...); the call then returns (y, precomputed-or-None), see
        ops.conv2d."""
        return ops.conv2d(x, self.weight, self.bias, self._pack, x2=x2, res=res, stride=self.stride[0],
                          pad=self._pad, ups=ups, bias_grad=bias_grad, bias_twin=bias_twin, norm=norm, box=box)


class Linear(nn.Linear):
    """Token-wise projection run as a 1x1 convolution over a pixel-major [B, C, H, W] tensor."""

    def __init__(self, in_features, out_features):
        super().__init__(in_features, out_features)
        self._pack = ops.WeightPack()

    def forward(self, x, res: Optional[torch.Tensor] = None):
        return ops.conv2d(x, self.weight, self.bias, self._pack, res=res)


class EmbLinear(nn.Linear):
    """fp32 linear of the embedding path ([B, E] inputs); `act_in` applies SiLU to the input inside the kernel."""

    def forward(self, x, act_in: bool = False):
        return ops.linear(x, self.weight, self.bias, act_in=act_in)


class GroupNorm(nn.GroupNorm):
    """GroupNorm with the following SiLU, an optional per-sample scale/shift or pre-add, and an optional
    second input tensor (channel concat) fused into one kernel."""

    def forward(self, x, act: bool = False, x2=None, scale_shift=None, pre_add=None, fork: bool = False,
                upstream_bias=None, precomputed=None, box=None):
        return ops.group_norm_act(x, self.weight, self.bias, self.num_groups, self.eps, act, x2=x2,
                                  scale_shift=scale_shift, pre_add=pre_add, fork=fork, upstream_bias=upstream_bias,
                                  precomputed=precomputed, box=box)

    def fused(self, act: bool = False, scale_shift=None, pre_add=None):
        """This layer as the epilogue of the convolution that produces its input (Conv2d's `norm`)."""
        return ops.FusedNorm(self.weight, self.bias, self.num_groups, self.eps, act, scale_shift=scale_shift,
                             pre_add=pre_add)
