"""Class-conditional UNet denoiser (reference: modeling/unet.py:13-166).

Constructor arguments, sub-module names and state-dict keys follow the reference; the image enters and leaves
as NCHW fp32, everything in between is pixel-major bf16 handled by libdm_hip kernels.
"""
from typing import Optional, Tuple

import torch
from torch import nn

from dm_hip import ops
from modeling.activation import get_activation, is_silu
from modeling.blocks import ConvBlock, MidBlock, ProjectedEmbeddings, get_down_block, get_up_block
from modeling.embeddings import MLPEmbedding, PeriodicEncoding
from modeling.layers import GroupNorm
from modeling.predictor import GuidedPredictor


class _EdgeConv(nn.Conv2d):
    """3x3 'same' conv with 1-4 channels on one side; parameters in the stock layout, kernels on the VALU."""

    def __init__(self, in_channels, out_channels, image_side: str):
        super().__init__(in_channels, out_channels, kernel_size=3, padding="same")
        self._image_side = image_side

    def forward(self, x):
        if self._image_side == "in":
            return ops.in_conv(x, self.weight, self.bias)
        return ops.out_conv(x, self.weight, self.bias)


class UNet(nn.Module, GuidedPredictor):
    def __init__(
            self,
            in_channels: int = 3,
            out_channels: int = 3,
            down_block_types: Tuple[str, ...] = ("UNetDownBlock",),
            up_block_types: Tuple[str, ...] = ("UNetUpBlock",),
            block_out_channels: Tuple[int, ...] = (64,),
            n_blocks_per_layer: int = 2,
            n_attention_heads: Optional[int] = None,
            n_dim_attention_head: Optional[int] = None,
            mid_attention: bool = True,
            norm_groups: int = 32,
            activation: str = "silu",
            time_encoding_dim: int = 64,
            embedding_dim: int = 64,
            embedding_mixing: str = "addition",
            n_class_embeddings: Optional[int] = None,
            dropout: float = 0.0,
            downsample_use_conv: bool = True,
            upsample_use_conv: bool = True,
    ):
        super().__init__()
        assert len(down_block_types) == len(up_block_types) == len(block_out_channels)
        n_levels = len(block_out_channels)
        widths = list(block_out_channels)

        self.in_conv = _EdgeConv(in_channels, widths[0], image_side="in")
        self.time_embedding = nn.Sequential(
            PeriodicEncoding(time_encoding_dim),
            MLPEmbedding(time_encoding_dim, embedding_dim),
        )
        if n_class_embeddings is not None:
            self.class_embedding = nn.Embedding(n_class_embeddings + 1, embedding_dim, padding_idx=n_class_embeddings)
            self.register_buffer("uncond_class", torch.tensor(n_class_embeddings))
            self._uncond_idx = int(n_class_embeddings)
        else:
            self.class_embedding = None

        shared = dict(n_heads=n_attention_heads, n_dim_head=n_dim_attention_head, norm_groups=norm_groups,
                      activation=activation, embedding_dim=embedding_dim, embedding_mixing=embedding_mixing,
                      dropout=dropout)

        down, prev = [], widths[0]
        for level, kind in enumerate(down_block_types):
            down.append(get_down_block(kind, in_channels=prev, out_channels=widths[level],
                                       n_blocks=n_blocks_per_layer, downsample=level < n_levels - 1,
                                       downsample_use_conv=downsample_use_conv, **shared))
            prev = widths[level]

        up, up_widths = [], widths[::-1] + [widths[0]]
        for level, kind in enumerate(up_block_types):
            up.append(get_up_block(kind, in_channels=prev, out_channels=up_widths[level],
                                   next_out_channels=up_widths[level + 1], n_blocks=n_blocks_per_layer + 1,
                                   upsample=level < n_levels - 1, upsample_use_conv=upsample_use_conv, **shared))
            prev = up_widths[level]

        self.down = nn.ModuleList(down)
        self.mid = MidBlock(channels=widths[-1], add_attention=mid_attention, **shared)
        self.up = nn.ModuleList(up)

        self.out_norm = GroupNorm(norm_groups, widths[0])
        self.out_act = get_activation(activation)
        self.out_conv = _EdgeConv(widths[0], out_channels, image_side="out")
        self._conv_blocks = [m for m in self.modules() if isinstance(m, ConvBlock)]  # plain list: not registered

    def forward(self, x: torch.Tensor, t: torch.Tensor, cls: Optional[torch.Tensor] = None,
                uncond_mask: Optional[torch.Tensor] = None):
        if x.shape[0] == 0:   # an empty batch (a ragged split's last shard): nothing to launch
            return x.new_zeros((0, self.out_conv.out_channels) + tuple(x.shape[2:]))
        if x.is_cuda and ops.EMB_SIDE_STREAM:
            # the embedding path beside the first convolutions: its first consumer is the first block's second norm
            # (one auxiliary stream per calling stream: callers that run several sub-batches on streams of their own
            # must not meet on a shared one)
            cur = torch.cuda.current_stream()
            side = ops.aux_stream(x.device, ("embedding", cur.cuda_stream))
            side.wait_stream(cur)
            for v in (t, cls, uncond_mask):
                if v is not None and v.is_cuda:
                    v.record_stream(side)
            with torch.cuda.stream(side):
                embedding = self._embed(t, cls, uncond_mask)
            embedding.produced_on(side)
        else:
            embedding = self._embed(t, cls, uncond_mask)

        h = self.in_conv(x)
        skips = (h,)
        for stage in self.down:
            h, stage_skips = stage(h, embedding=embedding)
            skips += stage_skips

        h = self.mid(h, embedding=embedding)

        for stage in self.up:
            n = len(stage.blocks)
            stage_skips, skips = skips[-n:], skips[:-n]
            h = stage(h, stage_skips, embedding=embedding)

        embedding.join()   # (a network without embedding consumers still leaves no stream unjoined)
        if is_silu(self.out_act):
            h = self.out_norm(h, act=True)
        else:
            h = self.out_act(self.out_norm(h, act=False))
        return self.out_conv(h)

    def _embed(self, t, cls, uncond_mask):
        embedding = self.time_embedding(t)
        if self.class_embedding is not None:
            # masked / missing labels map to the padding row, which is zero and never receives gradient
            embedding = ops.class_embed_add(embedding, self.class_embedding.weight, cls,
                                            uncond_mask if cls is not None else None, self._uncond_idx)
        # every ConvBlock projects the same SiLU(embedding): do all projections in one grouped launch
        return ProjectedEmbeddings(embedding, self._conv_blocks)

    def late_gradient_parameter_names(self):
        """Names of the parameters whose gradients appear only when backward is over, wherever they are registered:
        the embedding path. Every block's emb_proj is computed (and differentiated) in one grouped launch, which can
        run only after the LAST block of backward has produced its share, followed by the time MLP and the class
        table. A harness that ships gradients bucket by bucket during backward keeps these out of the early buckets."""
        late = [n for n, _ in self.time_embedding.named_parameters(prefix="time_embedding")]
        if self.class_embedding is not None:
            late.append("class_embedding.weight")
        for n, m in self.named_modules():
            if isinstance(m, ConvBlock) and m.emb_proj is not None:
                late += [f"{n}.emb_proj.{k}" for k, _ in m.emb_proj.named_parameters()]
        return late

    def pred_conditional(self, z, t, cond=None, uncond_mask=None):
        return self.forward(z, t, cls=cond, uncond_mask=uncond_mask)
