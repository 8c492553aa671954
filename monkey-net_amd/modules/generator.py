"""MotionTransferGenerator (modules/generator.py): appearance encoder -> dense motion field -> warp every skip ->
concat key-point heat-maps -> decoder -> residual refinement -> 1x1 conv + sigmoid; all on the gfx950 kernels."""
import torch
from torch import nn

from modules.util import Encoder, Decoder, ResBlock3D
from modules.dense_motion_module import DenseMotionModule, IdentityDeformation
from modules.movement_embedding import MovementEmbeddingModule
from mnk import ops

_MODES = {'nearest': 0, 'trilinear': 1}


class MotionTransferGenerator(nn.Module):
    """Given key-points and a source frame, reconstruct the driving frame.  Returns the refined prediction and the
    purely warped source (generator.py:10-82)."""

    def __init__(self, num_channels, num_kp, kp_variance, block_expansion, max_features, num_blocks, num_refinement_blocks,
                 dense_motion_params=None, kp_embedding_params=None, interpolation_mode='nearest'):
        super(MotionTransferGenerator, self).__init__()
        self.appearance_encoder = Encoder(block_expansion, in_features=num_channels, max_features=max_features,
                                          num_blocks=num_blocks)
        if kp_embedding_params is not None:
            self.kp_embedding_module = MovementEmbeddingModule(num_kp=num_kp, kp_variance=kp_variance,
                                                               num_channels=num_channels, **kp_embedding_params)
            embedding_features = self.kp_embedding_module.out_channels
        else:
            self.kp_embedding_module = None
            embedding_features = 0
        if dense_motion_params is not None:
            self.dense_motion_module = DenseMotionModule(num_kp=num_kp, kp_variance=kp_variance,
                                                         num_channels=num_channels, **dense_motion_params)
        else:
            self.dense_motion_module = IdentityDeformation()
        self.video_decoder = Decoder(block_expansion=block_expansion, in_features=num_channels,
                                     out_features=num_channels, max_features=max_features, num_blocks=num_blocks,
                                     additional_features_for_block=embedding_features, use_last_conv=False)
        self.refinement_module = torch.nn.Sequential()
        in_features = block_expansion + num_channels + embedding_features
        for i in range(num_refinement_blocks):
            self.refinement_module.add_module('r' + str(i), ResBlock3D(in_features, kernel_size=(1, 3, 3),
                                                                       padding=(0, 1, 1)))
        self.refinement_module.add_module('conv-last', nn.Conv3d(in_features, num_channels, kernel_size=1, padding=0))
        self.interpolation_mode = interpolation_mode
        self.num_channels = num_channels
        self.refine_features = in_features

    def _mode(self):
        if self.interpolation_mode not in _MODES:
            raise NotImplementedError("interpolation_mode %r" % (self.interpolation_mode,))
        return _MODES[self.interpolation_mode]

    def deform_input(self, inp, deformations_absolute):
        """Public form of generator.py:51-58: inp (B,C,1,h,w), field (B,d,ho,wo,3) -> (B,C,d,h,w)."""
        b, c = inp.shape[:2]
        _, d, ho, wo, _ = deformations_absolute.shape
        if d != 1:
            raise NotImplementedError("the generator is only ever called with one driving frame (SURVEY.md app. A.15)")
        field = deformations_absolute[..., :2].reshape(b * d, ho, wo, 2).contiguous()
        out = ops.WarpSkipFn.apply(ops.to_act(inp), field, None, c, 0, self._mode())
        return ops.from_act(out, c, b)

    def forward(self, source_image, kp_driving, kp_source):
        b = source_image.shape[0]
        d = kp_driving['mean'].shape[1]
        if d != 1 or source_image.shape[2] != 1:
            raise NotImplementedError("the generator is only ever called with one source and one driving frame "
                                      "(train.py:38, reconstruction.py:15-17, transfer.py:72-74)")
        mode = self._mode()
        src_act = ops.to_act(source_image)
        # (the appearance encoder does not depend on the key points; running it as a second-stream branch next to the
        # dense-motion network was measured at 11.46 vs 11.44 ms per step -- every kernel fills the chip -- and removed)
        skips = self.appearance_encoder.forward_act(src_act, self.num_channels)
        field = self.dense_motion_module.field_act(source_image, kp_driving, kp_source)     # (B,hf,wf,2)
        emb, ke = None, 0
        if self.kp_embedding_module is not None:
            emb, ke = self.kp_embedding_module.forward_act(source_image, kp_driving, kp_source)
        # all warps of this forward as one autograd node: one shared field-gradient buffer (ops.WarpAllFn)
        specs = tuple((c, ke) for _, c in skips) + ((self.num_channels, 0),)
        outs = ops.WarpAllFn.apply(field, emb, mode, specs, *([a for a, _ in skips] + [src_act]))
        warped = [(o, c + ke) for o, (_, c) in zip(outs[:-1], skips)]
        deformed_img = outs[-1]
        video_deformed = ops.from_act(deformed_img, self.num_channels, b)
        out, c = self.video_decoder.forward_act(warped)
        last, sums = None, None
        blocks = list(self.refinement_module.named_children())
        for idx, (name, block) in enumerate(blocks):
            if name == 'conv-last':
                last = block
            else:     # hand the output statistics of each block to the next block's norm1 (fused in the conv epilogue)
                more = idx + 1 < len(blocks) and blocks[idx + 1][0] != 'conv-last' and block.training
                res = block.forward_act(out, c, x_sums=sums, want_stats=more)
                out, c = res[0], res[1]
                sums = res[2] if more else None
        video_prediction = ops.Conv1x1SigmoidFn.apply(out, last.weight, last.bias, c, b)
        return {"video_prediction": video_prediction, "video_deformed": video_deformed}
