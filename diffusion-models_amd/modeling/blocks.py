"""Residual / resampling / stage blocks of the UNet (reference: modeling/blocks.py).

Same class names, constructor arguments, sub-module names and registration order as the reference, so state
dicts are interchangeable. Differences are internal: activations are bf16 pixel-major tensors, GroupNorm+SiLU
(+ scale/shift) is one kernel, the residual add rides in the conv epilogue, the skip concatenation and the
nearest-neighbour upsampling are fused into the consuming kernels' gathers instead of being materialised.
"""
from typing import Optional, Tuple

import torch
from torch import nn

from dm_hip import ops
from modeling.activation import get_activation, is_silu
from modeling.attention import Attention2D
from modeling.layers import Conv2d, EmbLinear, GroupNorm


class ProjectedEmbeddings:
    """All per-block embedding projections of one forward pass, computed in ONE grouped launch by the owner of the
    blocks (every emb_proj reads the same SiLU(embedding)), handed to the blocks in place of the raw embedding."""

    MAX_GROUP = 32

    def __init__(self, embedding: torch.Tensor, blocks):
        self._raw = embedding
        self._side = None
        self._untouched = True
        blocks = [b for b in blocks if b.emb_proj is not None and is_silu(b.nonlinearity)]
        self._proj = {}
        for i in range(0, len(blocks), self.MAX_GROUP):
            chunk = blocks[i:i + self.MAX_GROUP]
            outs = ops.linear_group(embedding, [(b.emb_proj.weight, b.emb_proj.bias) for b in chunk], act_in=True)
            self._proj.update({id(b): o for b, o in zip(chunk, outs)})

    def produced_on(self, stream):
        """The tensors were computed on `stream` (not the consumers' stream): the first consumer waits for it."""
        self._side = stream

    def join(self):
        if self._side is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._side)
            for t in [self._raw] + list(self._proj.values()):
                t.record_stream(cur)     # allocated on the other stream's pool, used on this one
            self._side = None

    def first_use(self):
        """True for the first consumer of a forward pass only. That block may have to wait for projections still being
        computed on another stream, so it starts its convolution first and normalises in a launch of its own; it does
        so whether or not the embedding path really runs on another stream, so that a pass gives the same bits either
        way (the separate GroupNorm launch and the convolution's epilogue add a sample's statistics in different
        orders)."""
        first, self._untouched = self._untouched, False
        return first

    @property
    def raw(self):
        self.join()
        return self._raw

    def for_block(self, block):
        self.join()
        return self._proj.get(id(block))


class ConvBlock(nn.Module):
    """GN-act-conv3x3, embedding mix, GN-act-(dropout)-conv3x3, plus (projected) residual
    (reference: modeling/blocks.py:10-75). `skip`, when given, is a second input concatenated after `x` on the
    channel axis (what the reference's up blocks pass as torch.cat([x, skip], 1))."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, norm_groups: int = 32,
                 activation: str = "swish", embedding_dim: Optional[int] = None,
                 embedding_mixing: str = "addition", dropout: float = 0.0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.embedding_mixing = embedding_mixing

        self.norm1 = GroupNorm(norm_groups, in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, padding="same")
        self.norm2 = GroupNorm(norm_groups, out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, padding="same")

        if embedding_dim is None:
            self.emb_proj = None
        elif embedding_mixing == "addition":
            self.emb_proj = EmbLinear(embedding_dim, out_channels)
        elif embedding_mixing == "scale_shift":
            self.emb_proj = EmbLinear(embedding_dim, out_channels * 2)
        else:
            raise ValueError(f'unknown embedding mixing "{embedding_mixing}"')

        self.nonlinearity = get_activation(activation)
        self.dropout = nn.Dropout(dropout)
        if in_channels == out_channels:
            self.res_proj = nn.Identity()
        else:
            self.res_proj = Conv2d(in_channels, out_channels, kernel_size=1)

    def _norm_act(self, norm, x, **kw):
        if is_silu(self.nonlinearity):
            return norm(x, act=True, **kw)
        return self.nonlinearity(norm(x, act=False, **kw))

    def _embedding_mix(self, embedding):
        """norm2's share of the embedding: {} | {"pre_add": e} | {"scale_shift": e}."""
        if self.emb_proj is None:
            return {}
        assert embedding is not None
        e = None
        if isinstance(embedding, ProjectedEmbeddings):
            e = embedding.for_block(self)
            embedding = embedding.raw
        if e is not None:
            pass
        elif is_silu(self.nonlinearity):
            e = self.emb_proj(embedding, act_in=True)
        else:
            e = self.emb_proj(self.nonlinearity(embedding))
        return {"pre_add" if self.embedding_mixing == "addition" else "scale_shift": e}

    def forward(self, x: torch.Tensor, embedding: Optional[torch.Tensor] = None,
                skip: Optional[torch.Tensor] = None):
        # x may also feed a skip connection: the gradient its late consumer leaves in the box joins norm1's backward
        box = ops.grad_box(x) if skip is None else None
        x = ops.as_nhwc(x)
        skip = ops.as_nhwc(ops.skip_tap(skip)) if skip is not None else None

        if is_silu(self.nonlinearity):
            # the residual branch reads x (and skip) through norm1's fork, so that its gradient is added inside
            # norm1's backward kernel
            h, x, skip = self.norm1(x, act=True, x2=skip, fork=True, box=box)
        else:
            h = self._norm_act(self.norm1, x, x2=skip, box=box)
        # conv1's bias gradient is the sum of norm2's input gradient, which norm2's backward has at hand. Where conv1's
        # output tiles hold whole samples (small images), norm2 runs in conv1's epilogue; it needs the embedding's
        # projection for that, so the first block of a pass, which may have to wait for projections still being computed
        # on another stream, starts its convolution first and normalises in a launch of its own.
        silu = is_silu(self.nonlinearity)
        if isinstance(embedding, ProjectedEmbeddings) and embedding.first_use():
            h, done = self.conv1(h, bias_grad=False), None
            mix = self._embedding_mix(embedding)
        else:
            mix = self._embedding_mix(embedding)
            h, done = self.conv1(h, bias_grad=False, norm=self.norm2.fused(act=silu, **mix))
        if h is None:    # no gradients: the fused launch wrote the normalised tensor only
            h = done[0] if silu else self.nonlinearity(done[0])
        else:
            h = self._norm_act(self.norm2, h, upstream_bias=self.conv1.bias, precomputed=done, **mix)
        if self.training and self.dropout.p > 0:
            h = self.dropout(h)

        if isinstance(self.res_proj, nn.Identity):
            res = x if skip is None else torch.cat([x, skip], dim=1)
            return self.conv2(h, res=res)
        # the 1x1 projection of the block's input rides in conv2's launch where the layer allows it (its input chunks
        # are multiplied into the same output tile behind the 3x3 taps: the projected residual never goes to memory)
        if self.conv2.stride[0] == 1 and self.res_proj.kernel_size == (1, 1):
            y = ops.conv2d_proj(h, self.conv2.weight, self.conv2.bias, self.conv2._pack, x, skip, self.res_proj.weight,
                                self.res_proj.bias, self.res_proj._pack)
            if y is not None:
                return y
        # res_proj's output and conv2's are the two terms of one sum: one launch of column sums over the block's
        # output gradient serves both biases
        twin = self.res_proj.bias if (self.res_proj.bias is not None and self.conv2.bias is not None) else None
        res = self.res_proj(x, x2=skip, bias_grad=twin is None)
        return self.conv2(h, res=res, bias_twin=twin)


class DownsampleBlock(nn.Module):
    """Stride-2 3x3 conv (pad 1) or 2x2 average pooling (reference: modeling/blocks.py:78-96)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, use_conv: bool = True):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if use_conv:
            self.conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1)
        else:
            assert in_channels == out_channels
            self.conv = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        if isinstance(self.conv, Conv2d):
            return self.conv(x, box=ops.grad_box(x, "conv"))   # (x usually feeds a skip connection as well)
        return ops.avgpool2x(x)


class UpsampleBlock(nn.Module):
    """Nearest x2 upsampling followed by a 3x3 conv; the upsampled tensor is never written: the conv gathers
    from the low-resolution source (reference: modeling/blocks.py:99-119)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, use_conv: bool = True):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if use_conv:
            self.conv = Conv2d(in_channels, out_channels, kernel_size=3, padding="same")
        else:
            assert in_channels == out_channels
            self.conv = None

    def forward(self, x):
        if self.conv is None:
            return ops.upsample2x(x)
        return self.conv(x, ups=1)


# ---------------------------------------------------------------------------------------------------------
# Stages. The eight public classes differ only in (a) attention after each ConvBlock, (b) whether a timestep
# embedding is consumed, (c) whether skips are produced/consumed; the shared plumbing lives in two bases whose
# sub-modules are registered in the reference's order: blocks, [attentions], downsample|upsample.
class _DownStage(nn.Module):
    _with_attention = False
    _with_embedding = True

    def __init__(self, in_channels: int, out_channels: int, n_blocks: int = 2, n_heads: Optional[int] = None,
                 n_dim_head: Optional[int] = None, norm_groups: int = 32, activation: str = "swish",
                 embedding_dim: Optional[int] = None, embedding_mixing: str = "addition", dropout: float = 0.0,
                 downsample: bool = True, downsample_use_conv: bool = True):
        super().__init__()
        emb = embedding_dim if self._with_embedding else None
        mixing = {"embedding_mixing": embedding_mixing} if self._with_embedding else {}
        self.blocks = nn.ModuleList(
            ConvBlock(in_channels if i == 0 else out_channels, out_channels=out_channels, norm_groups=norm_groups,
                      activation=activation, embedding_dim=emb, dropout=dropout, **mixing)
            for i in range(n_blocks))
        if self._with_attention:
            self.attentions = nn.ModuleList(
                Attention2D(out_channels, n_heads=n_heads, n_dim_head=n_dim_head, norm_groups=norm_groups,
                            dropout=dropout) for _ in range(n_blocks))
        self.downsample = DownsampleBlock(out_channels, out_channels, use_conv=downsample_use_conv) \
            if downsample else None

    def _run(self, x, embedding):
        outs = []
        for i, block in enumerate(self.blocks):
            x = block(x, embedding=embedding) if self._with_embedding else block(x)
            if self._with_attention:
                x = self.attentions[i](x)
            outs.append(x)
        if self.downsample is not None:
            x = self.downsample(x)
            outs.append(x)
        return x, tuple(outs)


class _UpStage(nn.Module):
    _with_attention = False
    _with_embedding = True   # embedding-consuming stages also consume skips

    def __init__(self, in_channels: int, out_channels: int, next_out_channels: Optional[int] = None,
                 n_blocks: int = 2, n_heads: Optional[int] = None, n_dim_head: Optional[int] = None,
                 norm_groups: int = 32, activation: str = "swish", embedding_dim: Optional[int] = None,
                 embedding_mixing: str = "addition", dropout: float = 0.0, upsample: bool = True,
                 upsample_use_conv: bool = True):
        super().__init__()
        blocks = []
        for i in range(n_blocks):
            cin = in_channels if i == 0 else out_channels
            if self._with_embedding:
                cin += out_channels if i < n_blocks - 1 else next_out_channels
                blocks.append(ConvBlock(cin, out_channels=out_channels, norm_groups=norm_groups,
                                        activation=activation, embedding_dim=embedding_dim,
                                        embedding_mixing=embedding_mixing, dropout=dropout))
            else:
                blocks.append(ConvBlock(cin, out_channels=out_channels, norm_groups=norm_groups,
                                        activation=activation, embedding_dim=None, dropout=dropout))
        self.blocks = nn.ModuleList(blocks)
        if self._with_attention:
            self.attentions = nn.ModuleList(
                Attention2D(out_channels, n_heads=n_heads, n_dim_head=n_dim_head, norm_groups=norm_groups,
                            dropout=dropout) for _ in range(n_blocks))
        self.upsample = UpsampleBlock(out_channels, out_channels, use_conv=upsample_use_conv) if upsample else None

    def _run(self, x, skips, embedding):
        if skips is not None:
            skips = tuple(reversed(skips))[:len(self.blocks)]
        for i, block in enumerate(self.blocks):
            if skips is not None:
                if i >= len(skips):
                    break  # zip() semantics of the reference: stop at the shorter sequence
                x = block(x, embedding=embedding, skip=skips[i])
            else:
                x = block(x)
            if self._with_attention:
                x = self.attentions[i](x)
        if self.upsample is not None:
            x = self.upsample(x)
        return x


class UNetDownBlock(_DownStage):
    def __init__(self, in_channels, out_channels, n_blocks=2, norm_groups=32, activation="swish",
                 embedding_dim=None, embedding_mixing="addition", dropout=0.0, downsample=True,
                 downsample_use_conv=True):
        super().__init__(in_channels, out_channels, n_blocks=n_blocks, norm_groups=norm_groups,
                         activation=activation, embedding_dim=embedding_dim, embedding_mixing=embedding_mixing,
                         dropout=dropout, downsample=downsample, downsample_use_conv=downsample_use_conv)

    def forward(self, x: torch.Tensor, embedding: Optional[torch.Tensor] = None):
        return self._run(x, embedding)


class UNetAttnDownBlock(_DownStage):
    _with_attention = True

    def forward(self, x: torch.Tensor, embedding: Optional[torch.Tensor] = None):
        return self._run(x, embedding)


class EncoderDownBlock(_DownStage):
    _with_embedding = False

    def __init__(self, in_channels, out_channels, n_blocks=2, norm_groups=32, activation="swish", dropout=0.0,
                 downsample=True, downsample_use_conv=True):
        super().__init__(in_channels, out_channels, n_blocks=n_blocks, norm_groups=norm_groups,
                         activation=activation, dropout=dropout, downsample=downsample,
                         downsample_use_conv=downsample_use_conv)

    def forward(self, x: torch.Tensor):
        return self._run(x, None)[0]


class EncoderAttnDownBlock(_DownStage):
    _with_attention = True
    _with_embedding = False

    def __init__(self, in_channels, out_channels, n_blocks=2, n_heads=None, n_dim_head=None, norm_groups=32,
                 activation="swish", dropout=0.0, downsample=True, downsample_use_conv=True):
        super().__init__(in_channels, out_channels, n_blocks=n_blocks, n_heads=n_heads, n_dim_head=n_dim_head,
                         norm_groups=norm_groups, activation=activation, dropout=dropout, downsample=downsample,
                         downsample_use_conv=downsample_use_conv)

    def forward(self, x: torch.Tensor):
        return self._run(x, None)[0]


class UNetUpBlock(_UpStage):
    def __init__(self, in_channels, out_channels, next_out_channels, n_blocks=2, norm_groups=32,
                 activation="swish", embedding_dim=None, embedding_mixing="addition", dropout=0.0, upsample=True,
                 upsample_use_conv=True):
        super().__init__(in_channels, out_channels, next_out_channels, n_blocks=n_blocks, norm_groups=norm_groups,
                         activation=activation, embedding_dim=embedding_dim, embedding_mixing=embedding_mixing,
                         dropout=dropout, upsample=upsample, upsample_use_conv=upsample_use_conv)

    def forward(self, x: torch.Tensor, skips: Tuple[torch.Tensor], embedding: Optional[torch.Tensor] = None):
        return self._run(x, skips, embedding)


class UNetAttnUpBlock(_UpStage):
    _with_attention = True

    def forward(self, x: torch.Tensor, skips: Tuple[torch.Tensor], embedding: Optional[torch.Tensor] = None):
        return self._run(x, skips, embedding)


class DecoderUpBlock(_UpStage):
    _with_embedding = False

    def __init__(self, in_channels, out_channels, n_blocks=2, norm_groups=32, activation="swish", dropout=0.0,
                 upsample=True, upsample_use_conv=True):
        super().__init__(in_channels, out_channels, None, n_blocks=n_blocks, norm_groups=norm_groups,
                         activation=activation, dropout=dropout, upsample=upsample,
                         upsample_use_conv=upsample_use_conv)

    def forward(self, x: torch.Tensor):
        return self._run(x, None, None)


class DecoderAttnUpBlock(_UpStage):
    _with_attention = True
    _with_embedding = False

    def __init__(self, in_channels, out_channels, n_blocks=2, n_heads=None, n_dim_head=None, norm_groups=32,
                 activation="swish", dropout=0.0, upsample=True, upsample_use_conv=True):
        super().__init__(in_channels, out_channels, None, n_blocks=n_blocks, n_heads=n_heads,
                         n_dim_head=n_dim_head, norm_groups=norm_groups, activation=activation, dropout=dropout,
                         upsample=upsample, upsample_use_conv=upsample_use_conv)

    def forward(self, x: torch.Tensor):
        return self._run(x, None, None)


class MidBlock(nn.Module):
    """block0, then n_blocks x ([attention] -> block) (reference: modeling/blocks.py:368-419)."""

    def __init__(self, channels: int, n_blocks: int = 1, n_heads: Optional[int] = None,
                 n_dim_head: Optional[int] = None, norm_groups: int = 32, activation: str = "swish",
                 embedding_dim: Optional[int] = None, embedding_mixing: str = "addition",
                 add_attention: bool = True, dropout: float = 0.0):
        super().__init__()

        def conv_block():
            return ConvBlock(channels, out_channels=channels, norm_groups=norm_groups, activation=activation,
                             embedding_dim=embedding_dim, embedding_mixing=embedding_mixing, dropout=dropout)

        blocks, attentions = [conv_block()], []
        for _ in range(n_blocks):
            blocks.append(conv_block())
            attentions.append(Attention2D(channels, n_heads=n_heads, n_dim_head=n_dim_head,
                                          norm_groups=norm_groups, dropout=dropout) if add_attention else None)
        self.blocks = nn.ModuleList(blocks)
        self.attentions = nn.ModuleList(attentions)

    def forward(self, x: torch.Tensor, embedding: Optional[torch.Tensor] = None):
        x = self.blocks[0](x, embedding=embedding)
        for block, attention in zip(self.blocks[1:], self.attentions):
            if attention is not None:
                x = attention(x)
            x = block(x, embedding=embedding)
        return x


_DOWN_TYPES = {
    "UNetDownBlock": (UNetDownBlock, False, True),
    "UNetAttnDownBlock": (UNetAttnDownBlock, True, True),
    "EncoderDownBlock": (EncoderDownBlock, False, False),
    "EncoderAttnDownBlock": (EncoderAttnDownBlock, True, False),
}
_UP_TYPES = {
    "UNetUpBlock": (UNetUpBlock, False, True),
    "UNetAttnUpBlock": (UNetAttnUpBlock, True, True),
    "DecoderUpBlock": (DecoderUpBlock, False, False),
    "DecoderAttnUpBlock": (DecoderAttnUpBlock, True, False),
}


def get_down_block(block_type: str, in_channels: int, out_channels: int, n_blocks: int = 2,
                   n_heads: Optional[int] = None, n_dim_head: Optional[int] = None, norm_groups: int = 32,
                   activation: str = "swish", embedding_dim: Optional[int] = None,
                   embedding_mixing: str = "addition", dropout: float = 0.0, downsample: bool = True,
                   downsample_use_conv: bool = True):
    if block_type not in _DOWN_TYPES:
        raise ValueError(f'unknown down block type "{block_type}"')
    cls, attn, emb = _DOWN_TYPES[block_type]
    kw = dict(in_channels=in_channels, out_channels=out_channels, n_blocks=n_blocks, norm_groups=norm_groups,
              activation=activation, dropout=dropout, downsample=downsample,
              downsample_use_conv=downsample_use_conv)
    if attn:
        kw.update(n_heads=n_heads, n_dim_head=n_dim_head)
    if emb:
        kw.update(embedding_dim=embedding_dim, embedding_mixing=embedding_mixing)
    return cls(**kw)


def get_up_block(block_type: str, in_channels: int, out_channels: int, next_out_channels: Optional[int] = None,
                 n_blocks: int = 2, n_heads: Optional[int] = None, n_dim_head: Optional[int] = None,
                 norm_groups: int = 32, activation: str = "swish", embedding_dim: Optional[int] = None,
                 embedding_mixing: str = "addition", dropout: float = 0.0, upsample: bool = True,
                 upsample_use_conv: bool = True):
    if block_type not in _UP_TYPES:
        raise ValueError(f'unknown up block type "{block_type}"')
    cls, attn, emb = _UP_TYPES[block_type]
    kw = dict(in_channels=in_channels, out_channels=out_channels, n_blocks=n_blocks, norm_groups=norm_groups,
              activation=activation, dropout=dropout, upsample=upsample, upsample_use_conv=upsample_use_conv)
    if attn:
        kw.update(n_heads=n_heads, n_dim_head=n_dim_head)
    if emb:
        kw.update(next_out_channels=next_out_channels, embedding_dim=embedding_dim,
                  embedding_mixing=embedding_mixing)
    return cls(**kw)
