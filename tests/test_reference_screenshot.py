"""The one output of the reference itself that ships with it: assets/screenshots/cornell.png (fixture made by
tools/make_fixtures.py).  It records no settings, frame count or camera pose - examples/cornell.rs puts the camera on
an orbit controller and an egui inspector in the window, and the short box in the picture has visibly been given
another material - so it cannot serve as a golden vector (the pin of the oracle is tests/test_wgsl_pin.py).  But it does test what the
oracle cannot test against itself: with ONE free parameter (the dolly distance of the orbit camera) the whole path
- glTF scene transform, pi/4 infinite reverse-Z projection, G-buffer, emitter strength 255*a*rgb, ReSTIR, denoiser,
Reinhard tone mapping, SMAA Tu4x + TAA, sRGB display encoding - must land on the reference's picture."""
import os

import numpy as np

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from conftest import ROOT
from oracle_lib import oracle_plugin


def _reference():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_screenshot_200x150.npz"))["rgb"].astype(np.float32) / 255.0


def _camera(z, w=400, h=300):
    return hk.Camera(hk.look_at_transform((0.0, 1.0, z), (0.0, 1.0, 0.0)), w, h)


def _box(a, k):  # mean over k x k blocks
    h, w = a.shape[0] // k * k, a.shape[1] // k * k
    return a[:h, :w].reshape(h // k, k, w // k, k, -1).mean(axis=(1, 3))


def test_silhouette_matches_after_fitting_the_dolly_distance():
    ref = _reference()
    ref_mask = ref.max(axis=2) > 0.06          # the example clears to black
    p = oracle_plugin()
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=0, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0, taa=hk.Taa.NONE)
    best = (0.0, None)
    for z in np.arange(3.4, 4.01, 0.05):
        p.render(_camera(float(z)), s, frame_number=1)
        mask = _box((p.engine.read(F.BUF_POSITION)[..., 3:4] > 0).astype(np.float32), 2)[..., 0] > 0.5
        iou = float((mask & ref_mask).sum() / (mask | ref_mask).sum())
        best = max(best, (iou, round(float(z), 2)))
    assert best[0] > 0.985, best               # 0.99 at z = 3.7; 0.81 at the example's start pose z = 4.0
    assert best[1] == 3.7


def test_picture_matches_the_reference_screenshot():
    ref = _reference()
    p = oracle_plugin()
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=1)   # HikariSettings::default(), as the example spawns it
    for n in range(1, 65):
        p.render(_camera(3.7), s, frame_number=n, antialias=True)
    img = np.clip(p.final_image(s)[..., :3], 0.0, 1.0)
    img = np.where(img <= 0.0031308, 12.92 * img, 1.055 * np.power(img, 1 / 2.4) - 0.055)   # the swap chain's sRGB encoding
    ours = _box(img, 2)                          # 400x300 -> 200x150
    inside = _box((p.engine.read(F.BUF_POSITION)[..., 3:4] > 0).astype(np.float32), 2)[..., 0] > 0.99
    # the short box was edited in the inspector before the shot (same base colour as the walls in the asset, dark grey
    # in the picture): leave its screen region out
    yy, xx = np.mgrid[0:150, 0:200]
    inside &= ~((xx > 92) & (xx < 140) & (yy > 92) & (yy < 140))
    bo, br, bm = _box(ours, 5), _box(ref, 5), _box(inside[..., None].astype(np.float32), 5)[..., 0] > 0.99
    corr = float(np.corrcoef(bo[bm].reshape(-1), br[bm].reshape(-1))[0, 1])
    mae = float(np.abs(bo[bm] - br[bm]).mean())
    assert corr > 0.96 and mae < 0.05, (corr, mae)     # measured: 0.977, 0.032
    # the two coloured walls, by name
    left, right = slice(60, 100), slice(40, 48)
    ol, rl = ours[left, right].mean(axis=(0, 1)), ref[left, right].mean(axis=(0, 1))
    assert ol[0] > 2 * ol[1] and rl[0] > 2 * rl[1] and np.abs(ol - rl).max() < 0.08, (ol, rl)
    orr, rr = ours[60:100, 152:160].mean(axis=(0, 1)), ref[60:100, 152:160].mean(axis=(0, 1))
    assert orr[1] > 1.4 * orr[0] and rr[1] > 1.4 * rr[0] and np.abs(orr - rr).max() < 0.08, (orr, rr)
