"""bevy-hikari_amd - MI355X-native (HIP / gfx950) render path for bevy-hikari's deferred hybrid
path tracer, behind the C ABI of include/hikari_hip.h.

The directory name carries a hyphen (it mirrors the reference's crate name); import it as
`bevy_hikari_amd` (the sibling shim package at the repo root).
"""
from . import _ffi  # noqa: F401
from ._ffi import HikariError, api  # noqa: F401
from .plugin import (  # noqa: F401
    Camera,
    Engine,
    FrameCounter,
    HikariPlugin,
    HikariSettings,
    HikariUniversalSettings,
    LightNode,
    PostProcessNode,
    PrepassNode,
    SceneBuilder,
    SceneData,
    Taa,
    Upscale,
    cornell_camera,
    frame_uniform,
    graph,
    lights_uniform,
    load_cornell,
    load_noise,
    look_at_transform,
    standard_material,
)

__all__ = [n for n in dir() if not n.startswith("_")]
