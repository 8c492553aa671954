"""Hourglass building blocks with the reference's class names, constructor arguments and state_dict keys
(modules/util.py:45-203).  `nn.Conv3d` / `SynchronizedBatchNorm3d` objects are kept as *parameter holders* (same
initialisation order and RNG draws as the reference, 5-D weights for checkpoint compatibility); the arithmetic runs
in the gfx950 kernels of libmonkeynet_hip.so on folded NHWC activations ("act", see mnk/ops.py).

Every block has two entry points:
  forward(x)        public, reference signature: (B,C,D,H,W) in / out
  forward_act(...)  internal fast path on acts (no layout conversion), used by KPDetector / generator
"""
import torch
from torch import nn

from sync_batchnorm import SynchronizedBatchNorm3d as BatchNorm3d
from mnk import ops


def make_coordinate_grid(spatial_size, type):
    """Mesh grid [-1,1] x [-1,1] (h,w,2), x first (modules/util.py:26-42).  Tiny host-side helper kept for the
    public API (transfer.py / visualisation); the kernels generate the same coordinates on the fly."""
    h, w = spatial_size
    x = torch.arange(w).type(type)
    y = torch.arange(h).type(type)
    x = 2 * (x / (w - 1)) - 1
    y = 2 * (y / (h - 1)) - 1
    return torch.stack([x.view(1, w).expand(h, w), y.view(h, 1).expand(h, w)], dim=2)


def matrix_inverse(batch_of_matrix, eps=0):
    """2x2 inverse (modules/util.py:206-224); closed form (the reference's eps == 0 branch used an LU solve)."""
    a = batch_of_matrix[..., 0, 0].unsqueeze(-1)
    b = batch_of_matrix[..., 0, 1].unsqueeze(-1)
    c = batch_of_matrix[..., 1, 0].unsqueeze(-1)
    d = batch_of_matrix[..., 1, 1].unsqueeze(-1)
    det = a * d - b * c
    if eps != 0:
        det = det.max(torch.tensor(eps).type(det.type()))
    out = torch.cat([d, -b, -c, a], dim=-1) / det
    return out.view(batch_of_matrix.shape)


def matrix_det(batch_of_matrix):
    a, b = batch_of_matrix[..., 0, 0].unsqueeze(-1), batch_of_matrix[..., 0, 1].unsqueeze(-1)
    c, d = batch_of_matrix[..., 1, 0].unsqueeze(-1), batch_of_matrix[..., 1, 1].unsqueeze(-1)
    return a * d - b * c


def matrix_trace(batch_of_matrix):
    return batch_of_matrix[..., 0, 0].unsqueeze(-1) + batch_of_matrix[..., 1, 1].unsqueeze(-1)


def smallest_singular(batch_of_matrix):
    """Closed-form smallest singular value of 2x2 matrices, operation order of modules/util.py:244-255."""
    a = batch_of_matrix[..., 0, 0].unsqueeze(-1)
    b = batch_of_matrix[..., 0, 1].unsqueeze(-1)
    c = batch_of_matrix[..., 1, 0].unsqueeze(-1)
    d = batch_of_matrix[..., 1, 1].unsqueeze(-1)
    s1 = a ** 2 + b ** 2 + c ** 2 + d ** 2
    s2 = (a ** 2 + b ** 2 - c ** 2 - d ** 2) ** 2
    s2 = torch.sqrt(s2 + 4 * (a * c + b * d) ** 2)
    return torch.sqrt((s1 - s2) / 2)


def _width(block_expansion, max_features, level):
    """Channel count of hourglass level `level`: block_expansion * 2^level capped at max_features."""
    return min(max_features, block_expansion * (2 ** level))


_K133, _P011 = (1, 3, 3), (0, 1, 1)


def _require_plain_3x3(kernel_size, padding):
    if tuple(kernel_size) != (1, 3, 3) or tuple(padding) != (0, 1, 1):
        raise NotImplementedError("only the (1,3,3)/(0,1,1) convolutions used by every reference config are built "
                                  "(`temporal` is never set by the reference, SURVEY.md section 0)")


def _public(block, x, cin, cout_fn):
    """Run a block's act path behind the reference's 5-D signature."""
    b = x.shape[0]
    out, c = cout_fn(ops.to_act(x), cin)
    return ops.from_act(out, c, b)


class ResBlock3D(nn.Module):
    """Pre-activation residual block (modules/util.py:45-68)."""

    def __init__(self, in_features, kernel_size, padding):
        super(ResBlock3D, self).__init__()
        _require_plain_3x3(kernel_size, padding)
        self.conv1 = nn.Conv3d(in_channels=in_features, out_channels=in_features, kernel_size=kernel_size,
                               padding=padding)
        self.conv2 = nn.Conv3d(in_channels=in_features, out_channels=in_features, kernel_size=kernel_size,
                               padding=padding)
        self.norm1 = BatchNorm3d(in_features, affine=True)
        self.norm2 = BatchNorm3d(in_features, affine=True)
        self.in_features = in_features

    def forward_act(self, x, c, x_sums=None, want_stats=False):
        """x_sums: BatchNorm statistics of x from the producing conv epilogue; want_stats: also return those of the
        output (the next block's norm1 consumes them).  Returns (out, c[, out_sums])."""
        # skip=True: x comes back from the norm node, so that node alone consumes the block's input and adds the gradient of
        # `out += x` inside its own backward pass (ops.BNActSkipFn)
        out, x = ops.bn_act(x, c, self.norm1, relu=True, sums=x_sums, skip=True)
        out, s = ops.conv3x3(out, c, self.conv1.weight, self.conv1.bias, want_stats=self.norm2.training,
                             eval_bn=not self.norm2.training)
        out = ops.bn_act(out, c, self.norm2, relu=True, sums=s)
        out, s = ops.conv3x3(out, c, self.conv2.weight, self.conv2.bias, residual=x, want_stats=want_stats)
        return (out, c, s) if want_stats else (out, c)

    def forward(self, x):
        return _public(self, x, self.in_features, self.forward_act)


class UpBlock3D(nn.Module):
    """nearest x2 -> conv -> BN -> ReLU (modules/util.py:71-88); the up-sampling is a gather inside the conv."""

    def __init__(self, in_features, out_features, kernel_size=3, padding=1):
        super(UpBlock3D, self).__init__()
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features, kernel_size=kernel_size,
                              padding=padding)
        self.norm = BatchNorm3d(out_features, affine=True)
        self.in_features, self.out_features = in_features, out_features
        _require_plain_3x3(self.conv.kernel_size, self.conv.padding)

    def forward_act(self, x0, c0, x1=None, c1=0):
        out, s = ops.conv3x3(x0, c0, self.conv.weight, self.conv.bias, x1=x1, c1=c1, ups=True,
                             want_stats=self.norm.training, eval_bn=not self.norm.training)
        return ops.bn_act(out, self.out_features, self.norm, relu=True, sums=s), self.out_features

    def forward(self, x):
        return _public(self, x, self.in_features, self.forward_act)


class DownBlock3D(nn.Module):
    """conv -> BN -> ReLU -> avgpool(1,2,2) (modules/util.py:91-108); BN+ReLU+pool is one pass."""

    def __init__(self, in_features, out_features, kernel_size=3, padding=1):
        super(DownBlock3D, self).__init__()
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features, kernel_size=kernel_size,
                              padding=padding)
        self.norm = BatchNorm3d(out_features, affine=True)
        self.pool = nn.AvgPool3d(kernel_size=(1, 2, 2))
        self.in_features, self.out_features = in_features, out_features
        _require_plain_3x3(self.conv.kernel_size, self.conv.padding)

    def forward_act(self, x, c, skip=False):
        """skip: -> ((out, channels), x handed through) for an input that has a second consumer (Encoder.forward_act)."""
        if skip:
            out, s, x = ops.conv3x3(x, c, self.conv.weight, self.conv.bias, want_stats=self.norm.training, skip=True,
                                    eval_bn=not self.norm.training)
            return (ops.bn_act(out, self.out_features, self.norm, relu=True, pool=True, sums=s), self.out_features), x
        out, s = ops.conv3x3(x, c, self.conv.weight, self.conv.bias, want_stats=self.norm.training,
                             eval_bn=not self.norm.training)
        return ops.bn_act(out, self.out_features, self.norm, relu=True, pool=True, sums=s), self.out_features

    def forward(self, x):
        return _public(self, x, self.in_features, self.forward_act)


class SameBlock3D(nn.Module):
    """conv (grouped 1x1 in every use of the reference) -> BN -> ReLU (modules/util.py:111-126)."""

    def __init__(self, in_features, out_features, groups=None, kernel_size=3, padding=1):
        super(SameBlock3D, self).__init__()
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features, kernel_size=kernel_size,
                              padding=padding, groups=groups)
        self.norm = BatchNorm3d(out_features, affine=True)
        self.groups = groups
        self.in_features, self.out_features = in_features, out_features
        if tuple(self.conv.kernel_size) != (1, 1, 1) or in_features != out_features or not groups:
            raise NotImplementedError("SameBlock3D is built for the grouped (1,1,1) form of dense_motion_module.py:24-28")

    def forward_act(self, x, c):
        out = ops.GConv1x1Fn.apply(x, self.conv.weight, self.conv.bias, self.groups)
        return ops.bn_act(out, self.out_features, self.norm, relu=True), self.out_features

    def forward(self, x):
        return _public(self, x, self.in_features, self.forward_act)


class Encoder(nn.Module):
    """Hourglass encoder (modules/util.py:129-152): returns [x, d1, ..., dn]."""

    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256, temporal=False):
        super(Encoder, self).__init__()
        if temporal:
            raise NotImplementedError("temporal=True is never used by the reference")
        widths = [in_features] + [_width(block_expansion, max_features, lvl) for lvl in range(1, num_blocks + 1)]
        self.down_blocks = nn.ModuleList(
            [DownBlock3D(cin, cout, kernel_size=_K133, padding=_P011) for cin, cout in zip(widths[:-1], widths[1:])])
        self.in_features = in_features

    def forward_act(self, x, c):
        # every level but the deepest has two consumers: the next down block and whoever takes the returned list (the decoder's
        # skip connections, the generator's warps).  The down block hands its input through and the list holds THAT tensor, so
        # the block's convolution is the level's only consumer in the autograd graph and adds the other gradient in its own
        # data-gradient launch (ops.Conv3x3SkipFn)
        outs, cur = [], (x, c)
        for down_block in self.down_blocks:
            nxt, through = down_block.forward_act(*cur, skip=True)
            outs.append((through, cur[1]))
            cur = nxt
        outs.append(cur)
        return outs

    def forward(self, x):
        b = x.shape[0]
        outs = self.forward_act(ops.to_act(x), self.in_features)
        return [x] + [ops.from_act(a, c, b) for a, c in outs[1:]]


class Decoder(nn.Module):
    """Hourglass decoder (modules/util.py:155-189).  torch.cat([out, skip]) is never materialised: the next
    convolution reads its two sources directly."""

    def __init__(self, block_expansion, in_features, out_features, num_blocks=3, max_features=256, temporal=False,
                 additional_features_for_block=0, use_last_conv=True):
        super(Decoder, self).__init__()
        if temporal:
            raise NotImplementedError("temporal=True is never used by the reference")
        extra = additional_features_for_block
        up_blocks = []
        for level in reversed(range(num_blocks)):
            below = _width(block_expansion, max_features, level + 1)
            # the deepest block sees only the bottleneck; the others see [previous up-block | skip]
            cin = (below if level == num_blocks - 1 else 2 * below) + extra
            up_blocks.append(UpBlock3D(cin, _width(block_expansion, max_features, level), kernel_size=_K133,
                                       padding=_P011))
        self.up_blocks = nn.ModuleList(up_blocks)
        self.conv = None
        if use_last_conv:
            self.conv = nn.Conv3d(block_expansion + in_features + extra, out_features, kernel_size=_K133, padding=_P011)
        self.out_features = out_features

    def forward_act(self, skips):
        """skips: list of (act, channels), consumed from the end like the reference's x.pop()."""
        skips = list(skips)
        x0, c0 = skips.pop()
        x1, c1 = None, 0
        for up_block in self.up_blocks:
            x0, c0 = up_block.forward_act(x0, c0, x1, c1)
            x1, c1 = skips.pop()
        if self.conv is not None:
            out, _ = ops.conv3x3(x0, c0, self.conv.weight, self.conv.bias, x1=x1, c1=c1)
            return out, self.conv.out_channels
        return ops.Concat2Fn.apply(x0, c0, x1, c1), c0 + c1

    def forward(self, x):
        b = x[0].shape[0]
        out, c = self.forward_act([(ops.to_act(t), t.shape[1]) for t in x])
        del x[:]
        return ops.from_act(out, c, b)


class Hourglass(nn.Module):
    """modules/util.py:192-203."""

    def __init__(self, block_expansion, in_features, out_features, num_blocks=3, max_features=256, temporal=False, ):
        super(Hourglass, self).__init__()
        self.encoder = Encoder(block_expansion, in_features, num_blocks, max_features, temporal=temporal)
        self.decoder = Decoder(block_expansion, in_features, out_features, num_blocks, max_features, temporal=temporal)
        self.in_features = in_features

    def forward_act(self, x, c):
        return self.decoder.forward_act(self.encoder.forward_act(x, c))

    def forward(self, x):
        return _public(self, x, self.in_features, self.forward_act)
