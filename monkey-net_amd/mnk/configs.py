"""Model hyper-parameters of the reference's shipped experiments, restated as Python dicts for bench.py / smoke
tests (the reference reads the same values from config/<name>.yaml: taichi.yaml:10-49, moving-gif.yaml:17-58,
bair.yaml, shapes.yaml).  A user of the drop-in keeps using their own YAML files; nothing here is read by modules/*."""
import copy


def _cfg(num_kp, be, mx, nb_kp, nb_gen, nb_dm, *, kp_scale=1, dm_scale=1, emb_scale=None, clip=0.001, norm=100,
         use_difference=False, group_blocks=2, refinement=4, disc_be=32, disc_mx=256, interpolation='nearest',
         rec_weights=(10, 10, 10, 10, 1), rec_def=0, lr=2.0e-4):
    mask = {"use_heatmap": True, "use_deformed_source_image": True, "heatmap_type": "difference", "norm_const": norm}
    if use_difference:
        mask["use_difference"] = True
    dm = {"block_expansion": be, "max_features": mx, "num_blocks": nb_dm, "use_mask": True, "use_correction": True,
          "mask_embedding_params": mask, "num_group_blocks": group_blocks}
    if dm_scale != 1:
        dm["scale_factor"] = dm_scale
    kpe = {"use_heatmap": True, "norm_const": norm, "heatmap_type": "difference"}
    if emb_scale is not None:
        kpe["scale_factor"] = emb_scale
    kpd = {"temperature": 0.1, "block_expansion": be, "max_features": mx, "num_blocks": nb_kp}
    if kp_scale != 1:
        kpd["scale_factor"] = kp_scale
    if clip:
        kpd["clip_variance"] = clip
    gen = {"block_expansion": be, "max_features": mx, "num_blocks": nb_gen, "num_refinement_blocks": refinement,
           "dense_motion_params": dm, "kp_embedding_params": kpe}
    if interpolation != 'nearest':
        gen["interpolation_mode"] = interpolation
    return {
        "model_params": {
            "common_params": {"num_kp": num_kp, "kp_variance": "matrix", "num_channels": 3},
            "kp_detector_params": kpd, "generator_params": gen,
            "discriminator_params": {"kp_embedding_params": {"norm_const": norm}, "block_expansion": disc_be,
                                     "max_features": disc_mx, "num_blocks": 4}},
        "train_params": {"detach_kp_generator": False, "detach_kp_discriminator": True, "lr": lr,
                         "loss_weights": {"reconstruction": list(rec_weights), "reconstruction_deformed": rec_def,
                                          "generator_gan": 1, "discriminator_gan": 1}},
    }


CONFIGS = {
    # config/taichi.yaml: 64x64, 10 kp, everything at full resolution, 5-level hourglasses
    "taichi": _cfg(10, 32, 1024, 5, 5, 5),
    # config/moving-gif.yaml: 6-level generator, sub-networks at half resolution, (dx,dy) maps in the mask embedding
    "moving-gif": _cfg(10, 32, 1024, 5, 6, 5, kp_scale=0.5, dm_scale=0.5, emb_scale=0.5, use_difference=True),
    # config/shapes.yaml: 4 kp, narrow nets, no variance clipping, no group blocks
    "shapes": _cfg(4, 16, 128, 5, 5, 5, clip=None, norm=10, group_blocks=0, rec_weights=(10, 10, 10, 10, 1)),
    # config/bair.yaml (= nemo.yaml up to the normaliser): 512 features, 'sum'-normalised heat-maps, warped-frame loss
    "bair": _cfg(10, 32, 512, 5, 5, 5, norm="sum", rec_def=10),
    "nemo": _cfg(10, 32, 512, 5, 5, 5),
    # config/vox.yaml: 256x256, 7-level generator, sub-networks at quarter resolution, 'trilinear' field / embedding resize
    "vox": _cfg(10, 32, 1024, 5, 7, 5, kp_scale=0.25, dm_scale=0.25, emb_scale=0.25, interpolation="trilinear"),
}


def get(name):
    return copy.deepcopy(CONFIGS[name])
