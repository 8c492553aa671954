"""MI355X-native drop-in for the reference's `modules` package (hot path: util, keypoint_detector,
movement_embedding, dense_motion_module, generator; callers' helpers: discriminator, losses)."""
