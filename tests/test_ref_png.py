"""The reference's own saved renders as fixtures (tests/golden/ref_png/).

The only data the reference holds for this path are PNG snapshots written by imshow() (main.cpp:19-42) next to the scenes:
`image<SPP>.png` is what the LAST run at that sample count left behind, suffixed copies are older experiments
(example-scenes-cg22/{test,veach-mis,staircase}/).  They were made with the reference's racy shared random engines, so they
pin nothing per sample — but their 16x16-block means, compared in linear space after the same 8-bit encode, pin the image
the committed code produces: geometry, the pixel grid (Q1/Q2), the estimator, and the light-sampling quirks Q3-Q5.

Which snapshot belongs to the committed code was established by ranking all of them against the parity-mode render
(tools/rank_ref_png.py):

  (Two sets of numbers appear below and in DESIGN.md §2; they differ because the renders do: the CPU tests of this file use the ORACLE at
  2-8 spp — 1.3 % / 7 % / 9 % —, the -m gpu tests and DESIGN.md's table the HIP path at the snapshot's own sample count — 1.05 % / 5.5 % / 5.7 %.)
  veach-mis/image10.png   matches parity mode to 1.3 % median block error, correlation 0.9992 (oracle, 4 spp; HIP at 10 spp: 1.05 %, on the
                          two-seed floor of 1.07 %) — and NOT TRT_FLAG_FIXED_NEE (23 %): the three-light bias of Q3 (every CDF draw spans
                          light 1's area), Q4 and Q5 are in the reference's own output, and the restatement reproduces them.
  staircase/image10.png   parity 7 % at 2 spp, 5.6 % at the snapshot's own 10 spp (two-seed floor 2.2 %); fixed-NEE 20 %: the same
                          conclusion on six lights and three textures (texture orientation and the BGR order of pathTracing.cpp:24-25
                          included).  The 3.4 points above the floor, the +7.3 % of the mean and its colour (1.09 / 1.06 / 1.03) are
                          EXPLAINED (round 4, test_staircase_snapshots_weight_specular_bounces_by_ks, profiles/r04_staircase_residual.txt):
                          the revision that wrote the staircase snapshots weighted a SPECULAR bounce by Ks, the committed source weights
                          it by the texel Kd (pathTracing.cpp:91-93, Q8).  With that one multiplication switched in the oracle (an explicit
                          experiment bit) the snapshot is matched ON the floor: mean 0.9999, channels within 0.1 %, 2.3 % median block error;
                          at 256 spp against image256.png: 0.9997, 0.69 %.  What it is NOT (each a test below): lost updates of the racy
                          accumulation (< 5e-5 of the mass), the three identically seeded engines (+-1.7 %), a missing 1 / P_RR (21 % too
                          DARK), a per-light, per-bounce or per-glass-interface factor (fits in the profiles file: none leaves less than 11 %).
  staircase/image256.png  parity 9 %, fixed 22 %: a converged snapshot of the same revision (5.7 % at 256 spp with the committed weighting,
                          0.69 % — its noise floor — with Ks).
  veach-mis/image256.png  parity 61 %, fixed 36 %: an older light-selection experiment (its siblings image10-area /
                          -radiance / -avg / -num are named after them); kept as the negative control.
  test/image10.png, test/image10-0.png (`back`)   geometry and pixel grid exact (first/last lit row and column), block
                          structure correlated 0.987, but 11-22 % darker than the committed code renders this scene, the
                          ceiling (indirect light only) most: older revisions of the indirect term (without the 1 / P_RR of
                          pathTracing.cpp:84 the distance halves: test_back_snapshots_predate_...).  They pin Q1/Q2 and the
                          geometry, not the brightness.  (Measured at the snapshots' 10 spp: 21.5 % / 11.3 % median block error against
                          a two-seed floor of 1.2 %; mean radiance of this build 1.26x / 1.11x the snapshots'.)

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
    # brightness: NOT pinned by these two files.  Measured 0.113 (image10-0) / 0.215 (image10) against a two-seed floor of 0.012
    # (tools/two_seed_floor.py) — systematic, and explained by the next test (snapshots written before the 1 / P_RR of
    # pathTracing.cpp:84 existed: without it this build is within 5.6 % of both).  The bound only says "that far and no further".
    assert med <= {"back_image10.png": 0.26, "back_image10-0.png": 0.15}[fixture], med


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


# ---- what the one-sided brightness residual is NOT: the two things the parity path does not take from the reference
EXPERIMENT_CASES = [("back", 256, 256), ("veach-mis", 320, 180), ("staircase", 320, 180)]


@pytest.mark.parametrize("name,w,h", EXPERIMENT_CASES, ids=[c[0] for c in EXPERIMENT_CASES])
def test_lost_updates_of_the_racy_accumulation_are_negligible(name, w, h):
    """Hypothesis (i): the reference adds color / SAMPLE into one shared `double image` from all its OpenMP threads without synchronisation
    (main.cpp:79-81,103-108: the team runs over the SAMPLE index, every thread sweeps the whole image), so overwritten updates could make
    its renders darker.  oracle_render_literal_experiment(ORACLE_EXP_RACY_ACCUM) does exactly that — a real data race, 8 threads in
    lock-step from pixel (0, 0) — with the per-sample counter streams, so the race-free sum is oracle_render_literal's image and the
    difference is the lost mass.  Measured here: 4.9e-5 (back), 7.6e-8 (veach-mis), 0 (staircase) of the image's energy, in 0-3 pixels
    — four orders of magnitude below back's 11-26 % and staircase's 7 %.  (A pixel's read-modify-write is three instructions; the
    path between two of them takes microseconds, and the threads drift apart within the first rows.)"""
    s = get_scene(name, w, h)
    p = T.make_params(w, h, 10, 12345)
    clean = O.render_literal(s.flat, p)[0].astype(np.float64)
    racy = O.render_literal_experiment(s.flat, p, O.EXP_RACY_ACCUM, threads=8).astype(np.float64)
    lost = (clean.sum() - racy.sum()) / clean.sum()
    assert -1e-6 <= lost <= 1e-3, lost


@pytest.mark.parametrize("name,w,h", EXPERIMENT_CASES, ids=[c[0] for c in EXPERIMENT_CASES])
def test_identically_seeded_engines_do_not_shift_the_brightness(name, w, h):
    """Hypothesis (ii): shade(), Sample() and nextRay() each own a static default_random_engine seeded with time(NULL) at its first call
    (pathTracing.cpp:33,113,149) — the same second, so the three produce IDENTICAL streams (light-point weights, lobe choice and
    direction angles are then correlated), RR()'s is default-seeded and main()'s was seeded before the scene was loaded.
    ORACLE_EXP_SHARED_ENGINES renders with five mt19937 engines (MSVC's default_random_engine) seeded that way, in the reference's
    loop order; ORACLE_EXP_INDEPENDENT_ENGINES is the control with three different seeds.  Mean radiance shared / independent,
    measured: 1.011 (back), 0.983 (veach-mis), 0.998 (staircase) — inside the spread two independent 10-spp renders of these sizes
    show, and on veach-mis, where this build matches the reference's snapshot to 1.0000 in the mean, in the direction that would
    spoil the match.  Not the cause of a one-sided 7-26 %."""
    s = get_scene(name, w, h)
    p = T.make_params(w, h, 10, 12345)
    shared = O.render_literal_experiment(s.flat, p, O.EXP_SHARED_ENGINES).astype(np.float64)
    indep = O.render_literal_experiment(s.flat, p, O.EXP_INDEPENDENT_ENGINES).astype(np.float64)
    counter = O.render_literal(s.flat, p)[0].astype(np.float64)
    r = shared.mean() / indep.mean()
    assert abs(r - 1.0) <= 0.035, r
    assert abs(indep.mean() / counter.mean() - 1.0) <= 0.035  # the control: another generator, the same picture


def test_staircase_snapshots_have_the_russian_roulette_compensation():
    """... and hypothesis (iii), the one that does explain `back` (next to this test's sibling above): on staircase the estimator WITHOUT
    1 / P_RR is 21 % too dark against image10.png (mean 0.79 of the snapshot's; the committed one 1.07) — the staircase snapshots were
    written with the division, so their 7 % are something else: the weight of SPECULAR bounces (test_staircase_snapshots_weight_specular_bounces_by_ks)."""
    png = _png("staircase_image10.png")
    s = get_scene("staircase", 640, 360)
    p = T.make_params(640, 360, 6, SEEDS["staircase"])
    committed = O.render(s.flat, p)[0]
    variant = O.render(s.flat, p, mode=O.MODE_ITERATIVE | O.MODE_EXPERIMENT_NO_RR_DIV)[0]
    ref_mean = _lin8(png).mean()
    r_c, r_v = _lin8(T.tonemap(committed)).mean() / ref_mean, _lin8(T.tonemap(variant)).mean() / ref_mean
    assert 1.0 <= r_c <= 1.15, r_c      # measured 1.07 at full size
    assert r_v <= 0.88, r_v             # measured 0.79


def test_staircase_snapshots_weight_specular_bounces_by_ks():
    """WHAT the staircase residual is (VERDICT r03 task 1; profiles/r04_staircase_residual.txt has the road to it).  pathTracing.cpp:91-93
    multiplies the light that comes back along a SPECULAR bounce by `Kd` (the texel; quirk Q8) — that is the committed source and what the
    parity path computes.  The revision that WROTE example-scenes-cg22/staircase/image*.png multiplied it by the material's `Ks` (the
    `m.Ks = m.Kd` that nextRay still carries at pathTracing.cpp:185 belongs to that weighting).  Three materials of this scene tell the two
    apart — FloorTiles (Ks 0.2 0.3 0.4 against a texel around 0.5-0.7: Kd returns more light, red most), Metal (Kd 0.2 grey, Ks 0 0.8 0.8: the
    strip by the door is TEAL in the snapshots and dark grey with Kd), Chrome (equal) — and no other shipped scene does (veach-mis: Kd = Ks on
    every glossy plate; back: no glossy material), which is why veach-mis sits on its noise floor either way.
    With TRT_FLAG_SPECULAR_KS (off by default: parity follows the committed source; the oracle, the literal oracle and the HIP path all take the switch) at image10.png's own
    10 spp the whole residual is gone: mean 1.0726 -> 0.9999 of the snapshot's, per channel 1.093 / 1.065 / 1.032 -> 1.000 / 1.000 / 1.000, median
    block error 5.6 % -> 2.3 % (the two-seed floor of this size is 2.2 %); at 256 spp against image256.png: 1.086 -> 0.9997, 5.7 % -> 0.69 %
    (full frame, measured once: 12 CPU-minutes).  Here: the lower 420 rows (the floor, the stairs, the strip) at 10 spp."""
    png = _png("staircase_image10.png")
    h, w = png.shape[:2]
    y0 = 300
    s = get_scene("staircase", w, h)
    p = T.make_params(w, h, 10, SEEDS["staircase"], tile=(0, y0, w, h))
    rb = _blocks(_lin8(png[y0:]))
    teal = (rb[..., 1] > 2.0 * rb[..., 0]) & (rb[..., 2] > 2.0 * rb[..., 0]) & (rb.sum(-1) > 0.05)  # the Metal strip as the snapshot shows it
    assert teal.sum() >= 1
    out = {}
    for name, flags in (("committed", 0), ("ks", T.TRT_FLAG_SPECULAR_KS)):   # (since round 4 a flag of the C-ABI too: the HIP path takes the same switch, GPU test below)
        p.flags = flags
        ob = _blocks(_lin8(T.tonemap(O.render(s.flat, p)[0])))
        ch = ob.mean((0, 1)) / rb.mean((0, 1))
        out[name] = (float(ob.mean() / rb.mean()), float(ch.max() / ch.min()), float(np.median(np.abs(ob - rb) / (0.02 + rb))), ob[teal].mean(0))
    (r_c, spread_c, med_c, teal_c), (r_k, spread_k, med_k, teal_k) = out["committed"], out["ks"]
    assert abs(r_k - 1.0) <= 0.01 and spread_k <= 1.01 and med_k <= 0.03, out["ks"]          # measured (full frame) 0.9999, 1.0007, 0.0233
    assert r_c >= 1.05 and spread_c >= 1.035 and med_c >= 1.8 * med_k, out["committed"]      # measured (full frame) 1.0726, 1.0588, 0.0564
    assert teal_k[1] > 3.0 * teal_k[0] and teal_k[2] > 1.5 * teal_k[0]                         # teal, as in the snapshot (0.037, 0.180, 0.080)
    assert teal_c[0] > teal_c[1] > teal_c[2]                                                   # the committed weighting: dim and warm


def test_the_glass_hypotheses_explain_none_of_the_staircase_residual():
    """VERDICT r03 task 1 (c): the three glass-path hypotheses for staircase's residual, each an explicit mode bit of the test oracle — (i) a mirror reflection instead of the
    fall-through to the opaque lobes when the Fresnel draw says "reflect" (pathTracing.cpp:173-194), (ii) no Tr on an emitter reached through a TRANSMISSION bounce (:95-96),
    (iii) no next-event estimation on Ni > 1 surfaces (:34-74) — judged like the Ks weight in the test above: an explanation brings the mean to 1, the three channels together
    AND the block error to its floor.  Measured on the lower 420 rows at 10 spp (mean / spread of the channel ratios / median block error; the test itself runs the
    three hypotheses on the lowest 260 rows — 1.133 / 1.101 / 10.2 %, 1.136 / 1.105 / 10.4 %, 0.990 / 1.160 / 7.2 % there — to keep the CPU suite short):
        committed            1.0934 / 1.076 / 7.1 %
        SPECULAR by Ks       0.9995 / 1.001 / 2.5 %   <- the explanation (two-seed floor 2.2 %)
        (i)  glass mirror    1.0944 / 1.078 / 7.2 %   nothing
        (ii) no Tr on emit.  1.0978 / 1.081 / 7.5 %   nothing, wrong way
        (iii) no NEE glass   0.9625 / 1.124 / 6.2 %   moves the MEAN past 1 (the balustrade's diffuse reflection is a tenth of these rows' light) but tears the channels
                                                      further apart and leaves the blocks as wrong as before: a different picture, not the snapshot's
    Only 17.7 % of the image's energy crosses a glass interface at all (profiles/r04_staircase_residual.txt)."""
    png = _png("staircase_image10.png")
    h, w = png.shape[:2]
    y0 = 460
    s = get_scene("staircase", w, h)
    p = T.make_params(w, h, 10, SEEDS["staircase"], tile=(0, y0, w, h))
    rb = _blocks(_lin8(png[y0:]))

    def stats(mode):
        ob = _blocks(_lin8(T.tonemap(O.render(s.flat, p, mode=O.MODE_ITERATIVE | mode)[0])))
        ch = ob.mean((0, 1)) / rb.mean((0, 1))
        return float(ob.mean() / rb.mean()), float(ch.max() / ch.min()), float(np.median(np.abs(ob - rb) / (0.02 + rb)))
    for name, bit in (("glass_mirror", O.MODE_EXPERIMENT_GLASS_MIRROR), ("no_tr_on_emitter", O.MODE_EXPERIMENT_NO_TR_ON_EMITTER), ("no_nee_on_glass", O.MODE_EXPERIMENT_NO_NEE_ON_GLASS)):
        mean, spread, med = stats(bit)
        print(f"{name}: mean {mean:.4f}, channel spread {spread:.4f}, median block error {med:.4f}")
        assert spread >= 1.05 and med >= 0.05, (name, mean, spread, med)   # the colour signature and the block error stay: not an explanation


# ------------------------------------------------------------------------------------------------ HIP path
# fixture -> (samples per pixel of the snapshot = what the HIP render uses, max median block error, max p90, min correlation)
GPU_BOUNDS = {
    "veach-mis_image10.png": (10, 0.02, 0.07, 0.998),      # measured 0.0105, 0.045, 0.9994
    # staircase: the committed estimator against snapshots of the Ks-weighting revision (file header): floor + the explained bias
    "staircase_image10.png": (10, 0.065, 0.25, 0.975),     # measured 0.055, 0.194, 0.985: a two-seed floor of 0.022 + 0.034 from Kd-for-Ks on the floor tiles
    "staircase_image256.png": (256, 0.065, 0.25, 0.975),   # measured 0.057, 0.199, 0.987: floor 0.007 + the same bias, converged
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
@pytest.mark.parametrize("fixture,spp,max_med", [("staircase_image10.png", 10, 0.03), ("staircase_image256.png", 256, 0.012)])
def test_gpu_render_with_specular_ks_matches_the_staircase_snapshots_on_their_noise_floor(fixture, spp, max_med, renderer_factory):
    """The product path end to end against the reference's own converged render: with TRT_FLAG_SPECULAR_KS — the one multiplication in which the revision behind the
    staircase snapshots differs from the committed source — the HIP render at the snapshot's size and sample count matches it to the two-seed noise floor: six lights, three
    JPEG textures, glass, the light-sampling quirks Q3-Q5, the pixel grid.  Measured (oracle): mean 0.9999 / 0.9997 of the snapshot's, every channel within 0.1 %, median
    block error 2.3 % (floor 2.2 %) at 10 spp and 0.69 % at 256 spp.  Without the flag: 5.6 % / 5.7 % and 7-9 % brighter (test above)."""
    png = _png(fixture)
    h, w = png.shape[:2]
    s = get_scene("staircase", w, h)
    r = renderer_factory(s)
    img, _ = r.render(T.make_params(w, h, spp, SEEDS["staircase"], flags=T.TRT_FLAG_SPECULAR_KS))
    ob, rb = _blocks(_lin8(T.tonemap(img))), _blocks(_lin8(png))
    ch = ob.mean((0, 1)) / rb.mean((0, 1))
    med = float(np.median(np.abs(ob - rb) / (0.02 + rb)))
    print(f"{fixture}: HIP {spp} spp with TRT_FLAG_SPECULAR_KS vs snapshot: mean {ob.mean() / rb.mean():.4f}, channels {np.round(ch, 4)}, median block error {med:.4f}")
    assert abs(ob.mean() / rb.mean() - 1.0) <= 0.005 and ch.max() / ch.min() <= 1.005 and med <= max_med, (ob.mean() / rb.mean(), ch, med)


@pytest.mark.gpu
def test_gpu_back_pixel_grid_matches_the_snapshots(renderer_factory):
    s = get_scene("back", 1024, 1024)
    img, _ = renderer_factory(s).render(T.make_params(1024, 1024, 16, T.SEED_BACK))
    for fixture in ("back_image10.png", "back_image10-0.png"):
        png = _png(fixture)
        assert _lit_extent(T.tonemap(img)) == _lit_extent(png)
        assert _compare(img, png)[2] >= 0.98
