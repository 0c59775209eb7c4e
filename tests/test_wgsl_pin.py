"""The oracle pinned against the reference's OWN shader source.

tests/tests/tools/wgsl translates src/shaders/{light,denoise,tone_mapping,taa,smaa}.wgsl of the reference mechanically to Python and
tests/tools/wgsl_pin.py executes every compute entry point on the state the oracle has before the corresponding dispatch; what the
shader writes must equal what the oracle writes, byte for byte (reservoir buffers, render / variance / denoise / tone-mapped /
SMAA / TAA textures).  These tests run where the reference checkout is mounted (this container); on the GPU box they skip.
The translator itself is also tested on small WGSL programs that need no reference."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import wgsl_pin
from wgsl import runtime as R
from wgsl import translate
from wgsl import types as T

needs_reference = pytest.mark.skipif(not wgsl_pin.reference_available(), reason="/root/reference is not mounted here")


def run(src, fn, *args):
    ns = {"_R": R, "_T": T, "RESOURCES": {}, "WORKGROUP_VARS": {}, "ENTRY_POINTS": {}, "_ONCE": (0,)}
    exec(translate.translate(src), ns)
    return ns[fn](*args)


def test_translator_value_semantics_pointers_loops_and_integers():
    src = """
    struct Inner { v: vec3<f32>, n: u32, };
    struct Outer { a: Inner, w: f32, };
    fn bump(p: ptr<function, Outer>, acc: ptr<function, f32>, k: f32) {
        (*p).a.v.y = (*p).a.v.y + k;      // component of a vector inside nested structs, through a pointer
        (*p).w += k;                      // a FIELD called w, not a swizzle
        *acc += k * 2.0;                  // pointer to a scalar
    }
    fn f(n: u32) -> vec4<f32> {
        var o: Outer;                     // zero-initialised
        let copy = o;                     // value semantics: `copy` must not see the writes below
        var acc = 0.5;
        var sum = 0u;
        for (var i = 0u; i < n; i += 1u) {
            if i == 2u { continue; }
            if i == 5u { break; }
            bump(&o, &acc, f32(i));
            sum += i * i;
        }
        var q: Outer;
        q = o;
        q.a.n = 7u;
        return vec4<f32>(o.a.v.y + copy.a.v.y, o.w, acc, f32(sum + q.a.n + o.a.n));
    }
    fn ints(a: i32, b: i32) -> vec4<i32> { return vec4<i32>(a / b, a % b, a >> 1u, (a * 65536) * 65536); }
    fn hash(value: u32) -> u32 {
        var state = value;
        state = state ^ 2747636419u;
        state = state * 2654435769u;
        state = state ^ state >> 16u;
        return state * 2654435769u;
    }
    """
    out = run(src, "f", R.u32(10))
    assert [float(x) for x in out] == [0 + 1 + 3 + 4, 8.0, 0.5 + 16.0, float(0 + 1 + 9 + 16 + 7)]      # i = 0, 1, 3, 4
    assert [int(x) for x in run(src, "ints", R.i32(-7), R.i32(2))] == [-3, -1, -4, 0]                  # truncating division, wrapping multiply
    state = (12345 ^ 2747636419) * 2654435769 & 0xFFFFFFFF
    state ^= state >> 16
    assert int(run(src, "hash_", R.u32(12345))) == state * 2654435769 & 0xFFFFFFFF


def test_translator_rounds_every_f32_operation_once():
    src = "fn g(a: f32, b: f32, c: f32) -> f32 { return a * b + c; }\nfn d(a: vec3<f32>, b: vec3<f32>) -> f32 { return dot(a, b); }"
    a, b, c = np.float32(1.0000001), np.float32(1.0000001), np.float32(-1.0)
    assert run(src, "g", a, b, c) == np.float32(np.float32(a * b) + c)                     # no contraction
    va, vb = T.vec3f32(0.1, 0.2, 0.3), T.vec3f32(0.7, -0.4, 0.9)
    want = R.fma(va[2], vb[2], R.fma(va[1], vb[1], va[0] * vb[0]))                         # the contract's fma chain
    assert run(src, "d", va, vb) == want


def test_wgsl_memory_layout_matches_the_c_abi():
    """The struct layouts the translator derives from the reference's WGSL are the ones the C ABI declares (hikari_hip.h)."""
    if not wgsl_pin.reference_available():
        pytest.skip("/root/reference is not mounted here")
    m = wgsl_pin.module("light.wgsl", ("NO_TEXTURE",))
    size = lambda name: m.ns["S_" + name].TYPE.size
    assert (size("Frame"), size("View"), size("PreviousView"), size("Instance"), size("Material"), size("Emissive"), size("Node"), size("Vertex"),
            size("Primitive"), size("PackedReservoir")) == (256, 416, 128, 176, 80, 64, 32, 32, 48, 64)


@needs_reference
@pytest.mark.parametrize("case,size,frames,dispatches", [("cornell_b2", (16, 12), 2, 44), ("cornell_default_aa", (24, 16), 2, 48), ("cornell_fsr", (30, 20), 1, 24)])
def test_reference_shaders_reproduce_the_oracle(case, size, frames, dispatches):
    """cornell_b2: MULTIPLE_BOUNCES pipeline, both spatial passes, denoise x3 channels x4 levels, tone mapping.
    cornell_default_aa: HikariSettings::default() - single-bounce pipeline, ratio 2, SMAA Tu4x + extrapolation + TAA.
    cornell_fsr: Upscale::Fsr1 - TAA at the scaled size, then FSR 1.0 EASU + RCAS executed from the GLSL the reference ships in
    src/shaders/fsr/source.zip (tests/tools/glsl.py: cpp + a GLSL -> Python translation)."""
    results = wgsl_pin.run_case(case, size, frames)
    assert len(results) == dispatches
    assert [r for r in results if r["mismatch"]] == []


@needs_reference
def test_the_pin_notices_a_changed_constant():
    """Negative control: RAY_BIAS 0.02 -> 0.03 in the translated light.wgsl must break every dispatch that traces."""
    saved = dict(wgsl_pin._modules)
    wgsl_pin._modules.clear()
    try:
        def patch(m):
            m.ns["RAY_BIAS"] = np.float32(0.03)
        results = wgsl_pin.run_case("cornell_b2", (16, 12), 1, patch=patch)
    finally:
        wgsl_pin._modules.clear()
        wgsl_pin._modules.update(saved)
    bad = {r["pass"] for r in results if r["mismatch"]}
    assert {"direct_emissive", "indirect_lit_ambient"} <= bad


# ---------------------------------------------------------------- replay of committed shader-produced fixtures (no reference needed)
FIXTURES = {"cornell_b2": ((16, 12), 2), "cornell_default_aa": ((24, 16), 3), "yard_textured_aa": ((30, 22), 3), "yard_moving_camera": ((30, 22), 3),
            "cornell_fsr": ((36, 24), 2)}


def replay(plugin, case):
    """tests/golden/wgsl_<case>_*.npz holds what the REFERENCE'S SHADERS wrote in every dispatch of the sequence
    (tests/tools/wgsl_pin.py --write, run where the reference is mounted).  Drive `plugin` through the same dispatches and return the
    list of (dispatch, buffer) whose bytes differ from what the shader wrote."""
    import bevy_hikari_amd as hk

    size, frames = FIXTURES[case]
    data = np.load(os.path.join(ROOT, "tests", "golden", f"wgsl_{case}_{size[0]}x{size[1]}_f{frames}.npz"))
    by_dispatch = {}
    for key in data.files:
        by_dispatch.setdefault(int(key[1:4]), []).append(key)
    scene, cam_for, s, lights, antialias = wgsl_pin.CASES[case](size)
    plugin.set_scene(scene)
    e, index, bad = plugin.engine, [0], []
    real = e.pass_run

    def hooked(pass_id, arg=0, row_begin=0, row_end=0):
        index[0] += 1
        real(pass_id, arg, row_begin, row_end)
        for key in by_dispatch.get(index[0], ()):
            want = data[key]
            got = e.read(int(key.split("buf")[1])).view(np.uint8).reshape(-1)[:len(want)]
            if not (got == want).all():
                bad.append((index[0], key, int((got != want).sum())))

    e.pass_run = hooked
    for n in range(1, frames + 1):
        plugin.render(cam_for(n), s, lights=lights, frame_number=n, by_nodes=True, antialias=antialias)
    assert index[0] >= max(by_dispatch)
    return bad


@pytest.mark.parametrize("case", sorted(FIXTURES))
def test_oracle_equals_what_the_reference_shaders_wrote(case):
    from oracle_lib import oracle_plugin

    assert replay(oracle_plugin(), case) == []


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(FIXTURES))
def test_gpu_equals_what_the_reference_shaders_wrote(case):
    """The HIP path against the outputs of the reference's own WGSL, dispatch by dispatch - no oracle in between.  The moving-camera
    fixture resolves the reference's store race by highest invocation index (HK_CTX_DETERMINISTIC_SCATTER does the same)."""
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F

    assert replay(hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER), case) == []


def test_translator_block_scoping():
    src = """
    fn f(c: bool) -> vec3<f32> {
        let a = 1.0;
        var out = 0.0;
        if c {
            let a = 2.0;            // shadows, must not leak
            out = a;
        }
        for (var i = 0u; i < 2u; i += 1u) {
            let a = 5.0;
            out += a;
        }
        return vec3<f32>(a, out, 0.0);
    }
    """
    assert [float(x) for x in run(src, "f", True)] == [1.0, 12.0, 0.0]
    assert [float(x) for x in run(src, "f", False)] == [1.0, 10.0, 0.0]


@needs_reference
def test_the_reference_has_no_bounds_guard():
    """Documents the deviation of DESIGN section 6: the reference rounds its dispatch up to multiples of 8 and no entry point checks
    invocation_id.  At a render width of 27 (40 / 1.5) the invocations x = 27..31 of row y read a zero G-buffer texel, take the
    "background" branch and store reservoirs at x + 27 * y - the first five pixels of row y + 1.  Executing the reference's
    shaders with their full grid shows exactly that and nothing else; the oracle and the kernels run the guarded grid."""
    results = wgsl_pin.run_case("yard_textured_aa", (40, 28), 1, unguarded=True)
    light = [r for r in results if r["entry"] in ("direct_lit", "indirect_lit_ambient", "spatial_reuse")]
    assert light and all(r["mismatch"] for r in light[:2])                 # the sun and emissive dispatches clobber
    import re
    for r in light:
        for desc in r["mismatch"].values():
            m = re.search(r"first #\d+ \(x=(\d+), y=(\d+)\)", desc)
            assert m and int(m.group(1)) < 32 - 27, desc                   # only slots the out-of-range invocations alias into
    other = [r for r in results if r["entry"] not in ("direct_lit", "indirect_lit_ambient", "spatial_reuse")]
    assert all(not r["mismatch"] for r in other)                           # texture-only passes: out-of-range stores are discarded


def test_glsl_translator_out_parameters_ternary_and_chained_assignment():
    import glsl

    src = """
    void split(float x, out float lo, inout float acc){ lo = x < 0.5 ? x : 0.5; acc += x; }
    float f(float a){
        float lo, acc = 1.0;
        uvec4 c;
        c[2] = c[3] = 7u;
        split(a, lo, acc);
        split(a * 2.0, lo, acc);
        for(int i = 0; i < 3; i++){ if(i == 1) continue; acc += 1.0; }
        return lo + acc + float(c.z + c.w) + uintBitsToFloat(floatBitsToUint(0.25));
    }
    """
    tr = glsl.Translator(src)
    ns = {"_R": R, "_T": T, "_G": glsl.G, "_ONCE": (0,)}
    exec(tr.module(), ns)
    assert tr.skipped == []
    assert float(ns["f"](np.float32(0.2))) == pytest.approx(0.4 + (1.0 + 0.2 + 0.4 + 2.0) + 14.0 + 0.25)


@needs_reference
def test_prepass_fragment_formulas_match_the_ray_cast_gbuffer():
    """prepass.wgsl is a raster pipeline and the library ray-casts its G-buffer, so the two cannot be compared invocation by
    invocation - but the FORMULAS of the reference's fragment() can: fed with the world position the oracle stored for a pixel, the
    reference's own code must produce the oracle's velocity (un-jittered reprojection with view / previous_view), its id encoding
    (index + 0.5) and pass position / uv through.  Moving camera, so the velocity is not trivially zero."""
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_scene
    from oracle_lib import oracle_plugin
    from wgsl.engine import Module

    scene, sun = synthetic_scene(n_boxes=8, n_spheres=3, n_emitters=2, sphere_rings=5, sphere_segs=6)
    p = oracle_plugin()
    p.set_scene(scene)
    s = hk.HikariSettings(indirect_bounces=0, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0)
    cams = [hk.Camera(hk.look_at_transform((6.4 + 0.3 * n, 4.4, 8.0 - 0.2 * n), (0.0, 0.6, 0.0)), 40, 28) for n in (1, 2)]
    for n, cam in enumerate(cams, start=1):
        p.render(cam, s, lights=hk.lights_uniform(directional=sun), frame_number=n)
    e = p.engine
    position, velocity_uv, ids = e.read(F.BUF_POSITION), e.read(F.BUF_VELOCITY_UV), e.read(F.BUF_INSTANCE_MATERIAL)
    m = Module(wgsl_pin.SHADERS, "prepass.wgsl", ("TEMPORAL_ANTI_ALIASING", "SMAA_TU4X"))
    view, pview = cams[1].view_uniform(), cams[1].previous_view_uniform(cams[0])
    m.bind(view=wgsl_pin.as_bytes(view), previous_view=wgsl_pin.as_bytes(pview))
    fragment, VertexOutput, InstanceIndex = m.ns["fragment"], m.ns["S_VertexOutput"].TYPE, m.ns["S_InstanceIndex"].TYPE
    checked = 0
    for y in range(28):
        for x in range(40):
            if position[y, x, 3] < 1.19e-7:
                continue
            inp = VertexOutput.zero()
            inp.world_position = T.vec4f32(*position[y, x, :3], 1.0)
            inp.previous_world_position = inp.world_position          # static objects (prepass.wgsl:50 with previous_mesh.model == mesh.model)
            inp.uv = T.vec2f32(*velocity_uv[y, x, 2:])
            inp.clip_position = T.vec4f32(x + 0.5, y + 0.5, position[y, x, 3], 1.0)      # fragment-stage @builtin(position): (window xy, depth, 1/w)
            idx = InstanceIndex.zero()
            idx.instance, idx.material = R.u32(int(ids[y, x, 0])), R.u32(int(ids[y, x, 1]))
            m.ns["instance_index"] = idx
            out = fragment(inp)
            assert [np.float32(v) for v in out.velocity_uv] == list(velocity_uv[y, x]), (x, y)
            assert [np.float32(v) for v in out.position] == list(position[y, x]), (x, y)
            assert [np.float32(v) for v in out.instance_material] == list(ids[y, x]), (x, y)
            checked += 1
    assert checked > 300 and np.abs(velocity_uv[..., :2]).max() > 1e-3


@needs_reference
@pytest.mark.parametrize("smaa", [True, False])
def test_prepass_jitter_sequence_and_sign_match_the_reference(smaa):
    """The vertex stage shifts clip positions by 2 * frame_jitter() * texel_size * (1, -1) (prepass.wgsl:30-38,53-71).  The oracle
    shifts its primary rays instead; so the world point it stores for pixel (x, y), projected with the un-jittered view_proj and
    shifted by the REFERENCE's frame_jitter() for that frame, must land on the pixel centre - for the Halton index rule of both
    upscale kinds (frame >> 1 with SMAA Tu4x, frame otherwise)."""
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from oracle_lib import oracle_plugin
    from wgsl.engine import Module

    W, H = 48, 32
    p = oracle_plugin()
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=0, denoise=False, upscale=hk.Upscale.SmaaTu4x(1.0) if smaa else hk.Upscale.Fsr1(1.0, 0.0))
    cam = hk.cornell_camera(W, H)
    m = Module(wgsl_pin.SHADERS, "prepass.wgsl", ("TEMPORAL_ANTI_ALIASING",) + (("SMAA_TU4X",) if smaa else ()))
    vp = np.ctypeslib.as_array(cam.view_uniform().view_proj).reshape(4, 4).T.astype(np.float64)      # column-major -> math layout
    seen = set()
    for n in (1, 2, 3, 4, 7, 18, 33):
        p.render(cam, s, frame_number=n)
        m.bind(frame=wgsl_pin.as_bytes(hk.frame_uniform(s, n)))
        jitter = np.array([float(v) for v in m.ns["frame_jitter"]()], dtype=np.float64)
        seen.add(tuple(jitter))
        pos = p.engine.read(F.BUF_POSITION).astype(np.float64)
        ys, xs = np.nonzero(pos[..., 3] > 1.19e-7)
        clip = np.concatenate([pos[ys, xs, :3], np.ones((len(ys), 1))], axis=1) @ vp.T
        ndc = clip[:, :2] / clip[:, 3:4] + 2.0 * jitter * np.array([1.0 / W, 1.0 / H]) * np.array([1.0, -1.0])
        centre = np.stack([(xs + 0.5) / W * 2.0 - 1.0, 1.0 - (ys + 0.5) / H * 2.0], axis=1)
        err_px = np.abs(ndc - centre) * np.array([W, H]) / 2.0
        assert err_px.max() < 2e-3, (n, err_px.max())          # f32 ray / projection arithmetic, far below the jitter itself
    assert len(seen) >= 4
