"""Error metrics against the reference-held converged render of the Cornell box.

The reference ships exactly one rendered image, tools/cornell-gt.exr (400 x 400; committed here as the value
fixture tests/golden/cornell_gt_400.npz), and defines three error metrics over it in tools/error_reports.py:13-23:
    MSE  = sum((gt - test)^2) / (h w)          AE = sum(|gt - test|) / (h w)
    MRSE = sum((gt - test)^2 / (gt + 1e-5)) / (h w)
(sums over pixels AND channels, divided by the pixel count).  The image is the only artefact the reference holds
that pins the ESTIMATOR -- PathTracer::L / direct*, Scene::* -- rather than single functions: an unbiased
renderer's error against it falls like 1/spp until it reaches the image's own noise, and stays spatially white.
Plain numpy; used by tools/gt_compare.py, tools/run_configs.py and tests/test_gpu_gt.py.
"""
import os

import numpy as np

# The ground truth was NOT rendered with the camera scenes/cornell.json carries at this snapshot (origin z = 6.8,
# fov 19.5: its light would be 9 % larger): tools/gt_fit_camera.py fits (fov, camera distance, height) by least
# squares over the 150 000 pixels that do not see the light; four Nelder-Mead starts agree on fov 28.12, z 5.079
# (2 tan(fov / 2) = 0.50), where the per-pixel error reaches the ground truth's own noise (profiles/r2_gt_fit.log).
# Two fitted camera numbers against 480 000 pixel values: the shading, shadows and colour bleeding are not fitted.
GT_SCENE = "tests/golden/cornell_gt_scene.json"
# ... and with lastBounce 5: the image's energy per channel is 0.516 / 0.761 / 0.897 / 0.962 / 0.998 / 1.016 / 1.027 of the
# ground truth's at lastBounce 1 .. 7 (red; profiles/r2_gt_bounces.log), and only at 5 do all three channels agree (to 0.4 %).
GT_LAST_BOUNCE = 5

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cornell_gt_400.npz")


def load_gt():
    """(400, 400, 3) float32, row 0 = top scanline, columns in the CURRENT camera convention (the file is mirrored)."""
    rgb = np.load(FIXTURE)["rgb"].astype(np.float32)
    return np.ascontiguousarray(rgb[:, ::-1, :])


def mse(test, gt):
    return float(np.sum((gt - test) ** 2) / (gt.shape[0] * gt.shape[1]))


def ae(test, gt):
    return float(np.sum(np.abs(gt - test)) / (gt.shape[0] * gt.shape[1]))


def mrse(test, gt):
    return float(np.sum((gt - test) ** 2 / (gt + 1e-5)) / (gt.shape[0] * gt.shape[1]))


def to_display(sums, spp):
    """radiance sums (row 0 = bottom scanline) -> mean image with row 0 = top, as Image::set stores it (src/image.cpp:21-35)"""
    return (sums / np.float32(spp))[::-1].astype(np.float32)


def block_means(image, size):
    h, w, c = image.shape
    return image[: h - h % size, : w - w % size].reshape(h // size, size, w // size, size, c).mean(axis=(1, 3))


def compare(render, gt, max_spp, block=16):
    """render(seed, spp_begin, spp_count) -> radiance sums (H, W, 3), row 0 = bottom.  Renders 4^k spp up to max_spp
    (nested: every level extends the previous one) plus an independent second image at max_spp, and reports
      levels        MSE / AE / MRSE per level (and the MSE over the pixels that do not see the light directly)
      law_ratios    (MSE_k - MSE_k+1) / (MSE_k+1 - MSE_k+2): 4 for an error that falls like 1/spp (dim pixels)
      floor_mse_dim the level the error settles on (extrapolated 1/spp -> 0), to be compared with
      gt_noise_mse_dim   the ground truth's own noise, estimated from differences of horizontally adjacent pixels
                         of (gt - our most converged image), which cancels the image content
      block_*       |block mean of ours - of gt| / block mean of gt over block x block tiles (bias shows here, noise averages out)
      *_z_*         (gt - ours) in units of the ground truth's own noise, measured per block from adjacent-pixel differences"""
    levels = []
    spp = 1
    done = 0
    sums = np.zeros(gt.shape, dtype=np.float32)
    dim = gt.max(axis=2) < 4.0          # pixels that do not look into the light (Ke = 17, 12, 4)
    # grow the mask by one pixel: the light's edge pixels differ by sub-pixel camera conventions
    grown = dim.copy()
    grown[1:, :] &= dim[:-1, :]; grown[:-1, :] &= dim[1:, :]; grown[:, 1:] &= dim[:, :-1]; grown[:, :-1] &= dim[:, 1:]
    dim = grown
    images = {}
    while spp <= max_spp:
        sums = sums + render(1, done, spp - done)
        done = spp
        image = to_display(sums, spp)
        images[spp] = image
        levels.append({"spp": spp, "mse": mse(image, gt), "ae": ae(image, gt), "mrse": mrse(image, gt),
                       "mse_dim": float(np.sum(((gt - image) ** 2)[dim]) / dim.sum())})
        spp *= 4
    top = levels[-1]["spp"]
    first = images[top]
    second = to_display(render(2, 0, top), top)        # independent image: another seed
    ours = 0.5 * (first + second)                        # 2 x top samples per pixel
    variance_per_sample = 0.5 * (first - second) ** 2 * top   # per pixel and channel, very noisy: only used in sums
    dims = [row["mse_dim"] for row in levels]
    ratios = [(dims[k] - dims[k + 1]) / (dims[k + 1] - dims[k + 2]) for k in range(len(dims) - 2) if dims[k + 1] != dims[k + 2]]
    # error of level k = floor + c / spp_k: from the last two levels
    c = (dims[-2] - dims[-1]) / (1.0 / levels[-2]["spp"] - 1.0 / levels[-1]["spp"])
    floor = dims[-1] - c / levels[-1]["spp"]
    # noise of the ground truth itself, from adjacent-pixel differences of (gt - ours): content cancels, our noise is known
    difference = gt - ours
    pair = dim[:, 1:] & dim[:, :-1]
    adjacent = (difference[:, 1:] - difference[:, :-1])[pair]
    our_noise = float(np.sum(variance_per_sample[dim]) / dim.sum()) / (2 * top)
    gt_noise = float(np.sum(adjacent ** 2) / pair.sum()) / 2.0 - our_noise
    per_sample = float(np.sum(variance_per_sample[dim]) / dim.sum())
    gt_spp = per_sample / max(gt_noise, 1e-30)
    # bias: block means
    ours_blocks, gt_blocks, dim_blocks = block_means(ours, block), block_means(gt, block), block_means(dim[..., None].astype(np.float32), block)[..., 0]
    whole = dim_blocks == 1.0
    relative = (np.abs(ours_blocks - gt_blocks).sum(axis=2) / gt_blocks.sum(axis=2))[whole]
    # z-scores against the ground truth's OWN noise, measured block by block from adjacent-pixel differences of
    # (gt - ours): our image carries 2 x top samples per pixel, so what is left is the ground truth's noise
    dim3 = dim[..., None] & np.ones(3, dtype=bool)
    adjacent_all = np.zeros_like(difference)
    adjacent_all[:, 1:] = (difference[:, 1:] - difference[:, :-1]) ** 2 / 2.0
    local_variance = block_means(adjacent_all, block)                       # per block and channel
    sigma_blocks = np.sqrt(local_variance / (block * block))
    z_blocks = (block_means(difference, block) / np.maximum(sigma_blocks, 1e-12))[whole]
    smooth = np.repeat(np.repeat(local_variance, block, axis=0), block, axis=1)
    z_pixels = (difference / np.sqrt(np.maximum(smooth, 1e-18)))[dim3]
    return {
        "levels": levels, "law_ratios": [float(r) for r in ratios], "floor_mse_dim": float(floor), "gt_noise_mse_dim": float(3 * gt_noise),
        "variance_per_sample_dim": per_sample, "gt_equivalent_spp": float(gt_spp),
        "block_rel_p50": float(np.percentile(relative, 50)), "block_rel_p95": float(np.percentile(relative, 95)), "block_rel_max": float(relative.max()),
        "block_z_median": float(np.median(np.abs(z_blocks))), "block_z_p90": float(np.percentile(np.abs(z_blocks), 90)),
        "block_z_over_4": float(np.mean(np.abs(z_blocks) > 4.0)),
        "pixel_z_median": float(np.median(np.abs(z_pixels))), "pixel_z_p90": float(np.percentile(np.abs(z_pixels), 90)),
        "pixel_z_over_4": float(np.mean(np.abs(z_pixels) > 4.0)),
        "mean_rgb": ours.mean(axis=(0, 1)).tolist(), "gt_mean_rgb": gt.mean(axis=(0, 1)).tolist(),
        "energy_ratio": float(ours.sum() / gt.sum()),
    }


# ---- scenes/veach-ajar against the Tungsten render the reference ships (tests/golden/veach_ajar_tungsten_blocks.npz) ----

def veach_ajar_blocks(radiance, block=16):
    """Block means of a (H, W, 3) image whose row 0 is the BOTTOM scanline (the C ABI's convention), top row first like the EXR."""
    import numpy as np
    image = np.asarray(radiance, dtype=np.float64)[::-1]
    h, w = image.shape[:2]
    return image[:h - h % block, :w - w % block].reshape(h // block, block, w // block, block, 3).mean(axis=(1, 3))


def veach_ajar_compare(ours, theirs, mask):
    """Sanity-level comparison of two block-mean images outside `mask` (True = ignore): a third-party renderer, other
    sampler, other texture filtering -- energy and block structure, not per-pixel values."""
    import numpy as np
    keep = ~np.asarray(mask, dtype=bool)
    a = np.asarray(ours, dtype=np.float64)[keep].sum(axis=1)
    b = np.asarray(theirs, dtype=np.float64)[keep].sum(axis=1)
    relative = np.abs(a - b) / np.maximum(b, 1e-3)
    log_ratio = np.log((a + 1e-4) / (b + 1e-4))
    return {
        "blocks_compared": int(keep.sum()),
        "energy_ratio": float(a.sum() / b.sum()),
        "median_relative_difference": float(np.median(relative)),
        "p90_relative_difference": float(np.percentile(relative, 90)),
        "log_ratio_correlation": float(np.corrcoef(np.log(a + 1e-4), np.log(b + 1e-4))[0, 1]),
        "median_log_ratio": float(np.median(log_ratio)),
    }
