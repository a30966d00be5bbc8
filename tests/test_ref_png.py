"""The reference's own saved renders as fixtures (tests/golden/ref_png/).

The only data the reference holds for this path are PNG snapshots written by imshow() (main.cpp:19-42) next to the scenes:
`image<SPP>.png` is what the LAST run at that sample count left behind, suffixed copies are older experiments
(example-scenes-cg22/{test,veach-mis,staircase}/).  They were made with the reference's racy shared random engines, so they
pin nothing per sample — but their 16x16-block means, compared in linear space after the same 8-bit encode, pin the image
the committed code produces: geometry, the pixel grid (Q1/Q2), the estimator, and the light-sampling quirks Q3-Q5.

Which snapshot belongs to the committed code was established by ranking all of them against the parity-mode render
(tools/rank_ref_png.py):

  veach-mis/image10.png   matches parity mode to 1.3 % median block error, correlation 0.9992 — and NOT TRT_FLAG_FIXED_NEE
                          (23 %): the three-light bias of Q3 (every CDF draw spans light 1's area), Q4 and Q5 are in the
                          reference's own output, and the restatement reproduces them.
  staircase/image10.png   parity 7 % at 2 spp (noise of both renders), fixed-NEE 20 %: same conclusion on six lights and
                          three textures (texture orientation and the BGR order of pathTracing.cpp:24-25 included).
  staircase/image256.png  parity 9 %, fixed 22 %: a converged snapshot of the same estimator.
  veach-mis/image256.png  parity 61 %, fixed 36 %: an older light-selection experiment (its siblings image10-area /
                          -radiance / -avg / -num are named after them); kept as the negative control.
  test/image10.png, test/image10-0.png (`back`)   geometry and pixel grid exact (first/last lit row and column), block
                          structure correlated 0.987, but 11-22 % darker than the committed code renders this scene, the
                          ceiling (indirect light only) most: older revisions of the indirect term (without the 1 / P_RR of
                          pathTracing.cpp:84 the distance halves: test_back_snapshots_predate_...).  They pin Q1/Q2 and the
                          geometry, not the brightness.

Tolerances are stated per fixture below; the CPU tests use the oracle at a few spp, the -m gpu tests the HIP render
through the C-ABI at the snapshot's own sample count.
"""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import tinyraytracing_amd as T
from conftest import get_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_png")
SEEDS = {"back": T.SEED_BACK, "veach-mis": 0x5EED0002, "staircase": T.SEED_STAIRCASE}
BLOCK = 16


def _png(name):
    return np.asarray(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))


def _lin8(a):
    """Inverse of imshow()'s encode (uchar)(pow(x, 1/2.2f) * 255) at the centre of the truncation interval."""
    return ((a.astype(np.float64) + 0.5) / 255.0) ** 2.2


def _blocks(img, b=BLOCK):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def _compare(linear_img, png):
    """Both sides go through the reference's 8-bit encode (clamp at 1 included) and back to linear."""
    ob = _blocks(_lin8(T.tonemap(linear_img)))
    rb = _blocks(_lin8(png))
    rel = np.abs(ob - rb) / (0.02 + rb)
    lo = np.log(ob.sum(axis=2) + 0.01).ravel()
    lr = np.log(rb.sum(axis=2) + 0.01).ravel()
    return float(np.median(rel)), float(np.percentile(rel, 90)), float(np.corrcoef(lo, lr)[0, 1])


def _lit_extent(a8):
    rows = np.nonzero(a8.sum(axis=(1, 2)) > 0)[0]
    cols = np.nonzero(a8.sum(axis=(0, 2)) > 0)[0]
    return int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])


# fixture -> (scene, oracle spp of the CPU test, max median block error, max p90, min log-luminance correlation)
MATCHING = {
    "veach-mis_image10.png": ("veach-mis", 4, 0.03, 0.10, 0.997),    # measured at 4 spp: 0.013, 0.052, 0.9992
    "staircase_image10.png": ("staircase", 2, 0.11, 0.25, 0.965),    # measured at 2 spp: 0.070, 0.162, 0.978
    "staircase_image256.png": ("staircase", 2, 0.13, 0.28, 0.96),    # measured at 2 spp: 0.091, 0.187, 0.974
}
_render_cache = {}


def _oracle_render(scene_name, spp, flags=0):
    key = (scene_name, spp, flags)
    if key not in _render_cache:
        png = _png({"veach-mis": "veach-mis_image10.png", "staircase": "staircase_image10.png", "back": "back_image10.png"}[scene_name])
        h, w = png.shape[:2]
        s = get_scene(scene_name, w, h)
        _render_cache[key] = O.render(s.flat, T.make_params(w, h, spp, SEEDS[scene_name], flags=flags))[0]
    return _render_cache[key]


@pytest.mark.parametrize("fixture", sorted(MATCHING))
def test_parity_render_matches_the_references_own_snapshot(fixture):
    scene_name, spp, max_med, max_p90, min_corr = MATCHING[fixture]
    med, p90, corr = _compare(_oracle_render(scene_name, spp), _png(fixture))
    assert med <= max_med and p90 <= max_p90 and corr >= min_corr, (med, p90, corr)


@pytest.mark.parametrize("fixture", sorted(MATCHING))
def test_snapshots_carry_the_light_sampling_quirks(fixture):
    """Negative control: with Q3-Q5 opted out (TRT_FLAG_FIXED_NEE) the same scene is far from the reference's snapshot."""
    scene_name, spp, max_med, _, _ = MATCHING[fixture]
    med_parity, _, _ = _compare(_oracle_render(scene_name, spp), _png(fixture))
    med_fixed, _, _ = _compare(_oracle_render(scene_name, spp, T.TRT_FLAG_FIXED_NEE), _png(fixture))
    assert med_fixed >= 0.15 and med_fixed >= 2.0 * med_parity, (med_parity, med_fixed)  # measured 0.20-0.23 against 0.013-0.09


def test_veach_image256_is_an_older_light_selection_experiment():
    """veach-mis/image256.png does not come from the committed estimator (either flavour is > 30 % off): documented, not hidden."""
    png = _png("veach-mis_image256.png")
    med_parity, _, _ = _compare(_oracle_render("veach-mis", 4), png)
    med_fixed, _, _ = _compare(_oracle_render("veach-mis", 4, T.TRT_FLAG_FIXED_NEE), png)
    assert med_parity > 0.3 and med_fixed > 0.25, (med_parity, med_fixed)


@pytest.mark.parametrize("fixture", ["back_image10.png", "back_image10-0.png"])
def test_back_snapshots_pin_pixel_grid_and_geometry(fixture):
    """Q1 (rows shifted by one: y = (H - i)/(H - 1)) and Q2 decide which border rows/columns stay black; the reference's two
    snapshots of `back` and the parity render agree on them EXACTLY, and the un-quirked pixel grid does not."""
    png = _png(fixture)
    img = _oracle_render("back", 8)
    assert _lit_extent(T.tonemap(img)) == _lit_extent(png) == (19, 1001, 14, 1009)
    fixed = _oracle_render("back", 8, T.TRT_FLAG_FIXED_PIXELS)
    assert _lit_extent(T.tonemap(fixed)) != _lit_extent(png)
    med, p90, corr = _compare(img, png)
    assert corr >= 0.98, corr            # measured 0.987: same silhouettes block for block
    assert med <= 0.30, med              # measured 0.11 (image10-0) / 0.22 (image10): an older, darker indirect term


def test_back_snapshots_predate_the_russian_roulette_compensation(monkeypatch):
    """Why the `back` snapshots are darker than the committed code renders the scene: an estimator WITHOUT the 1 / P_RR of
    pathTracing.cpp:84 (an oracle experiment switch, never part of the parity path) halves the distance to both of them
    (median block error 11-22 % -> 5.6 %, correlation 0.987 -> 0.995) — they were written by an earlier revision of the indirect
    term.  The committed code's division is what veach-mis/image10.png and staircase/image10.png confirm to 1 % / 6 %."""
    s = get_scene("back", 1024, 1024)
    p = T.make_params(1024, 1024, 8, SEEDS["back"])
    committed = _oracle_render("back", 8)
    variant = O.render(s.flat, p, mode=O.MODE_ITERATIVE | O.MODE_EXPERIMENT_NO_RR_DIV)[0]  # an explicit mode bit, not an environment variable
    monkeypatch.setenv("ORACLE_EXPERIMENT_NO_RR_DIV", "1")  # ... which the library no longer reads
    tile = T.make_params(1024, 1024, 2, 3, tile=(500, 500, 516, 508))
    with_env = O.render(s.flat, tile)[0]
    monkeypatch.delenv("ORACLE_EXPERIMENT_NO_RR_DIV")
    assert np.array_equal(with_env, O.render(s.flat, tile)[0])
    for fixture in ("back_image10.png", "back_image10-0.png"):
        png = _png(fixture)
        med_c, _, corr_c = _compare(committed, png)
        med_v, _, corr_v = _compare(variant, png)
        assert med_v < 0.08 and corr_v > 0.993 and med_v < 0.7 * med_c and corr_v > corr_c, (fixture, med_c, med_v, corr_c, corr_v)


# ------------------------------------------------------------------------------------------------ HIP path
# fixture -> (samples per pixel of the snapshot = what the HIP render uses, max median block error, max p90, min correlation)
GPU_BOUNDS = {
    "veach-mis_image10.png": (10, 0.02, 0.07, 0.998),      # measured 0.0105, 0.045, 0.9994
    "staircase_image10.png": (10, 0.075, 0.25, 0.975),     # measured 0.056, 0.194, 0.985 (two 10-spp renders of a high-variance scene)
    "staircase_image256.png": (256, 0.075, 0.25, 0.975),   # measured 0.057, 0.199, 0.987
}


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", sorted(MATCHING))
def test_gpu_render_matches_the_references_own_snapshot(fixture, renderer_factory):
    """The HIP render through the C-ABI at the snapshot's native resolution AND sample count (the 8-bit encode clamps at 1
    after averaging: a 10-spp image loses more of its fireflies to the clamp than a converged one, so like is compared
    with like)."""
    scene_name = MATCHING[fixture][0]
    spp, max_med, max_p90, min_corr = GPU_BOUNDS[fixture]
    png = _png(fixture)
    h, w = png.shape[:2]
    s = get_scene(scene_name, w, h)
    r = renderer_factory(s)
    img, _ = r.render(T.make_params(w, h, spp, SEEDS[scene_name]))
    med, p90, corr = _compare(img, png)
    print(f"{fixture}: HIP {spp} spp vs snapshot: median block error {med:.4f}, p90 {p90:.4f}, correlation {corr:.4f}")
    assert med <= max_med and p90 <= max_p90 and corr >= min_corr, (med, p90, corr)
    fixed, _ = r.render(T.make_params(w, h, spp, SEEDS[scene_name], flags=T.TRT_FLAG_FIXED_NEE))
    med_fixed, _, _ = _compare(fixed, png)
    print(f"{fixture}: with TRT_FLAG_FIXED_NEE: median block error {med_fixed:.4f}")
    assert med_fixed >= 0.15 and med_fixed >= 2.0 * med, (med, med_fixed)


@pytest.mark.gpu
def test_gpu_back_pixel_grid_matches_the_snapshots(renderer_factory):
    s = get_scene("back", 1024, 1024)
    img, _ = renderer_factory(s).render(T.make_params(1024, 1024, 16, T.SEED_BACK))
    for fixture in ("back_image10.png", "back_image10-0.png"):
        png = _png(fixture)
        assert _lit_extent(T.tonemap(img)) == _lit_extent(png)
        assert _compare(img, png)[2] >= 0.98
