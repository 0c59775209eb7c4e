"""FidelityFX Super Resolution 1.0 (Upscale::Fsr1): EASU + RCAS, src/shaders/fsr/source.zip and
post_process.rs:503-534,1277-1308.  The oracle restates the GLSL through its gather4 structure; here a
second, independent numpy restatement indexes the 12 taps directly by pixel offset
(ffx_fsr1.h:315-331 tap diagram) and must agree bit for bit, plus the properties the algorithm guarantees."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from oracle_lib import oracle_plugin

S, U = hk.HikariSettings, hk.Upscale
f32 = np.float32


def f16(a):
    return a.view(np.float16).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, dtype=f32).view(np.uint32)


def lo_rcp(a):
    return (np.uint32(0x7ef07ebb) - bits(a)).view(f32)


def med_rcp(a):
    b = (np.uint32(0x7ef19fff) - bits(a)).view(f32)
    return b * (-b * a + f32(2.0))


def lo_rsq(a):
    return (np.uint32(0x5f347d74) - (bits(a) >> np.uint32(1))).view(f32)


def easu_numpy(img, ow, oh):
    """img: f32 [ih][iw][4] -> f32 [oh][ow][3]; every operation in f32, in the order of ffx_fsr1.h"""
    ih, iw = img.shape[:2]
    one = f32(1.0)
    cx, cy = f32(iw) * (one / f32(ow)), f32(ih) * (one / f32(oh))
    ox, oy = f32(0.5) * f32(iw) * (one / f32(ow)) - f32(0.5), f32(0.5) * f32(ih) * (one / f32(oh)) - f32(0.5)
    xs, ys = np.meshgrid(np.arange(ow, dtype=f32), np.arange(oh, dtype=f32))
    ppx, ppy = xs * cx + ox, ys * cy + oy
    fx, fy = np.floor(ppx), np.floor(ppy)
    ppx, ppy = ppx - fx, ppy - fy
    ix, iy = fx.astype(np.int64), fy.astype(np.int64)

    def tap(dx, dy):
        return img[np.clip(iy + dy, 0, ih - 1), np.clip(ix + dx, 0, iw - 1), :3]
    offs = dict(b=(0, -1), c=(1, -1), e=(-1, 0), f=(0, 0), g=(1, 0), h=(2, 0), i=(-1, 1), j=(0, 1), k=(1, 1), l=(2, 1), n=(0, 2), o=(1, 2))
    C = {k: tap(*v) for k, v in offs.items()}
    L = {k: v[..., 2] * f32(0.5) + (v[..., 0] * f32(0.5) + v[..., 1]) for k, v in C.items()}
    dirx, diry, ln = np.zeros_like(ppx), np.zeros_like(ppx), np.zeros_like(ppx)

    def sat(v):
        return np.minimum(np.maximum(v, f32(0.0)), f32(1.0))

    def accumulate(w, lA, lB, lC, lD, lE):
        nonlocal dirx, diry, ln
        lenx = lo_rcp(np.maximum(np.abs(lD - lC), np.abs(lC - lB)))
        dx = lD - lB
        dirx = dirx + dx * w
        lenx = sat(np.abs(dx) * lenx)
        ln = ln + (lenx * lenx) * w
        leny = lo_rcp(np.maximum(np.abs(lE - lC), np.abs(lC - lA)))
        dy = lE - lA
        diry = diry + dy * w
        leny = sat(np.abs(dy) * leny)
        ln = ln + (leny * leny) * w
    accumulate((one - ppx) * (one - ppy), L['b'], L['e'], L['f'], L['g'], L['j'])
    accumulate(ppx * (one - ppy), L['c'], L['f'], L['g'], L['h'], L['k'])
    accumulate((one - ppx) * ppy, L['f'], L['i'], L['j'], L['k'], L['n'])
    accumulate(ppx * ppy, L['g'], L['j'], L['k'], L['l'], L['o'])
    dirr = dirx * dirx + diry * diry
    zro = dirr < f32(1.0 / 32768.0)
    rs = np.where(zro, one, lo_rsq(dirr))
    dirx = np.where(zro, one, dirx) * rs
    diry = diry * rs
    ln = ln * f32(0.5)
    ln = ln * ln
    stretch = (dirx * dirx + diry * diry) * lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
    len2x, len2y = one + (stretch - one) * ln, one + f32(-0.5) * ln
    lob = f32(0.5) + f32((1.0 / 4.0 - 0.04) - 0.5) * ln
    clp = lo_rcp(lob)
    quad = np.stack([C['f'], C['g'], C['j'], C['k']])
    mn, mx = quad.min(0), quad.max(0)
    acc, wsum = np.zeros_like(C['f']), np.zeros_like(ppx)
    order = [('b', 0, -1), ('c', 1, -1), ('i', -1, 1), ('j', 0, 1), ('f', 0, 0), ('e', -1, 0), ('k', 1, 1), ('l', 2, 1), ('h', 2, 0), ('g', 1, 0),
             ('o', 1, 2), ('n', 0, 2)]
    for name, dx, dy in order:
        offx, offy = f32(dx) - ppx, f32(dy) - ppy
        vx = (offx * dirx + offy * diry) * len2x
        vy = (offx * (-diry) + offy * dirx) * len2y
        d2 = np.minimum(vx * vx + vy * vy, clp)
        wb = f32(2.0 / 5.0) * d2 + f32(-1.0)
        wa = lob * d2 + f32(-1.0)
        wb, wa = wb * wb, wa * wa
        wb = f32(25.0 / 16.0) * wb + f32(-(25.0 / 16.0 - 1.0))
        w = wb * wa
        acc = acc + C[name] * w[..., None]
        wsum = wsum + w
    return np.minimum(mx, np.maximum(mn, acc * (one / wsum)[..., None]))


def rcas_numpy(img, sharpness):
    h, w = img.shape[:2]
    p = np.zeros((h + 2, w + 2, 3), f32)          # texelFetch outside the image: zeros
    p[1:-1, 1:-1] = img[..., :3]
    b, d, e, f, hh = p[:-2, 1:-1], p[1:-1, :-2], p[1:-1, 1:-1], p[1:-1, 2:], p[2:, 1:-1]
    mn4 = np.minimum(np.minimum(b, np.minimum(d, f)), hh)
    mx4 = np.maximum(np.maximum(b, np.maximum(d, f)), hh)
    one = f32(1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        hit_min = np.minimum(mn4, e) * (one / (f32(4.0) * mx4))
        hit_max = (one - np.maximum(mx4, e)) * (one / (f32(4.0) * mn4 + f32(-4.0)))
    # GLSL max/min return the non-NaN operand (contract: minNum/maxNum)
    lobe3 = np.fmax(-hit_min, hit_max)
    lobe = np.fmax(f32(-0.1875), np.fmin(np.fmax(lobe3[..., 0], np.fmax(lobe3[..., 1], lobe3[..., 2])), f32(0.0))) * f32(2.0 ** -sharpness)
    rcp = med_rcp(f32(4.0) * lobe + one)
    lobe = lobe[..., None]
    return (lobe * b + lobe * d + lobe * hh + lobe * f + e) * rcp[..., None]


def _fsr_oracle(w, h, ratio, sharpness):
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    s = S(upscale=U.Fsr1(ratio, sharpness), taa=hk.Taa.NONE)
    cpu.render(hk.cornell_camera(w, h), s, frame_number=1, antialias=True)
    return cpu, s


def test_fsr_buffers_and_final_image():
    cpu, s = _fsr_oracle(90, 66, 1.5, 0.2)
    assert cpu.engine.buffer_info(F.BUF_TONE_MAPPED)[:2] == (60, 44)
    assert cpu.engine.buffer_info(F.BUF_UPSCALE_OUTPUT) == (90, 66, 8)       # post_process.rs:723: scale 1.0
    assert cpu.engine.buffer_info(F.BUF_UPSCALE_SHARPENED) == (90, 66, 8)
    img = cpu.final_image(s)
    assert img.shape == (66, 90, 4) and np.isfinite(img).all() and (img[..., 3] == 1.0).all()
    # the upscaled image still shows the scene: correlated with a nearest-neighbour blow-up of the tone-mapped frame
    tm = f16(cpu.engine.read(F.BUF_TONE_MAPPED))
    yy, xx = np.minimum((np.arange(66) / 1.5).astype(int), 43), np.minimum((np.arange(90) / 1.5).astype(int), 59)
    near = tm[yy][:, xx]
    assert np.corrcoef(near[..., :3].ravel(), img[..., :3].ravel())[0, 1] > 0.9


@pytest.mark.parametrize("size,ratio", [((90, 66), 1.5), ((64, 40), 2.0), ((50, 34), 1.0), ((77, 51), 1.3)])
def test_easu_matches_an_independent_numpy_restatement(size, ratio):
    cpu, s = _fsr_oracle(size[0], size[1], ratio, 0.0)
    iw, ih, _ = cpu.engine.buffer_info(F.BUF_TONE_MAPPED)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:ih, 0:iw]
    base = 0.5 + 0.4 * np.sin(xx * 0.31 + yy * 0.17)[..., None] * np.array([1.0, 0.7, -0.5, 0.0])     # smooth + edge + noise
    base[(xx + 2 * yy) % 23 < 9] *= 0.3
    img16 = (base + 0.03 * rng.standard_normal(base.shape)).clip(0, 1).astype(np.float16)
    cpu.engine.write(F.BUF_TONE_MAPPED, img16.view(np.uint16))
    cpu.engine.set_view_options(hk.Taa.NONE, F.UPSCALE_FSR1, 0.0)
    cpu.engine.pass_run(F.PASS_FSR_EASU)
    got = cpu.engine.read(F.BUF_UPSCALE_OUTPUT)
    want = easu_numpy(img16.astype(f32), size[0], size[1])
    assert (got[..., :3] == want.astype(np.float16).view(np.uint16)).all()
    assert (f16(got)[..., 3] == 1.0).all()
    # deringing: never outside the range of the nearest 2x2 input quad, hence of the whole input
    out = f16(got)[..., :3]
    assert out.min() >= img16[..., :3].astype(f32).min() and out.max() <= img16[..., :3].astype(f32).max()


def test_easu_keeps_a_flat_image_flat_and_rcas_a_flat_image_unchanged():
    cpu, s = _fsr_oracle(72, 48, 1.5, 0.0)
    iw, ih, _ = cpu.engine.buffer_info(F.BUF_TONE_MAPPED)
    flat = np.broadcast_to(np.array([0.25, 0.5, 0.75, 1.0], np.float16), (ih, iw, 4))
    cpu.engine.write(F.BUF_TONE_MAPPED, np.ascontiguousarray(flat).view(np.uint16))
    cpu.engine.pass_run(F.PASS_FSR_EASU)
    up = f16(cpu.engine.read(F.BUF_UPSCALE_OUTPUT))
    assert (up == np.array([0.25, 0.5, 0.75, 1.0], f32)).all()
    cpu.engine.pass_run(F.PASS_FSR_RCAS)
    sharp = f16(cpu.engine.read(F.BUF_UPSCALE_SHARPENED))
    # APrxMedRcpF1 is a 1-Newton-step reciprocal; the border sees zeros outside the image
    assert np.abs(sharp[1:-1, 1:-1] / np.array([0.25, 0.5, 0.75, 1.0], f32) - 1.0).max() < 5e-3


@pytest.mark.parametrize("sharpness", [0.0, 0.2, 1.0, 2.0])
def test_rcas_matches_numpy_and_sharpens(sharpness):
    cpu, s = _fsr_oracle(72, 48, 1.5, sharpness)
    up16 = cpu.engine.read(F.BUF_UPSCALE_OUTPUT)
    got = cpu.engine.read(F.BUF_UPSCALE_SHARPENED)
    want = rcas_numpy(f16(up16), sharpness)
    assert (got[..., :3] == want.astype(np.float16).view(np.uint16)).all()
    # sharpening: local contrast (mean |laplacian|) does not drop, and grows with lower `sharpness` stops
    def contrast(a):
        return np.abs(4 * a[1:-1, 1:-1] - a[:-2, 1:-1] - a[2:, 1:-1] - a[1:-1, :-2] - a[1:-1, 2:]).mean()
    assert contrast(f16(got)[..., :3]) >= contrast(f16(up16)[..., :3])


def test_fsr_passes_need_the_fsr_kind():
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    cpu.render(hk.cornell_camera(40, 24), S(), frame_number=1, antialias=True)
    with pytest.raises(hk.HikariError):
        cpu.engine.pass_run(F.PASS_FSR_EASU)


def test_upscale_band_plan():
    """Exchange E: the EASU input rows (taa_output with TAA, tone-mapped without) that the 12 taps of a band's window
    rows +-1 reach beyond its render band; nothing for SMAA Tu4x or a single band."""
    from bevy_hikari_amd.distributed import halo_plan

    fsr, fsr_no_taa = S(upscale=U.Fsr1(1.5, 0.2)).to_c(), S(upscale=U.Fsr1(1.5, 0.2), taa=hk.Taa.NONE).to_c()
    assert halo_plan(1920, 1080, 1.5, 0, 1, F.STAGE_UPSCALE, 5, fsr) == []
    assert halo_plan(1920, 1080, 2.0, 1, 4, F.STAGE_UPSCALE, 5, S().to_c()) == []
    # 1280x720 traced; band 1 of 4: window rows [270,540), EASU rows [269,541) -> f = floor((y + 0.5) / 1.5 - 0.5) in [179,359], taps f-1..f+2
    ops = halo_plan(1920, 1080, 1.5, 1, 4, F.STAGE_UPSCALE, 5, fsr)
    assert sorted((o.buffer, o.peer, o.row_begin, o.row_end, o.row_bytes) for o in ops) == [(F.BUF_TAA_OUTPUT, 0, 178, 180, 1280 * 8),
                                                                                             (F.BUF_TAA_OUTPUT, 2, 360, 362, 1280 * 8)]
    ops = halo_plan(1920, 1080, 1.5, 0, 2, F.STAGE_UPSCALE, 5, fsr_no_taa)
    assert [(o.buffer, o.peer, o.row_begin, o.row_end) for o in ops] == [(F.BUF_TONE_MAPPED, 1, 360, 362)]
