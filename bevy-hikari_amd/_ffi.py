"""ctypes binding of the C ABI in include/hikari_hip.h (+ the test / measurement hooks of include/hikari_hip_debug.h).

This is the stub a host language binds (INTEGRATION.md shows the Rust `extern "C"` equivalent).
The structs are field-for-field the ones in the header; `Api` resolves every entry point of
libhikari_hip.so once and turns negative return codes into `HikariError`.  There is no fallback: if
the HIP library is missing or no GPU is present the calls fail loudly.

(`_SIGNATURES` is the part of the table the frame path itself uses; the test suite builds the table of
its CPU oracle from it - tests/oracle_lib.py - so that one driver class serves both.  Nothing in this
package knows about that library.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# HIKARI_HIP_LIB: development override for A/B-ing builds of the SAME library (tools/ab.sh)
LIB_PATH = os.environ.get("HIKARI_HIP_LIB", os.path.join(HERE, "libhikari_hip.so"))

HK_OK = 0
HK_E_INVALID, HK_E_NO_DEVICE, HK_E_HIP, HK_E_NOT_READY, HK_E_NOMEM, HK_E_UNSUPPORTED = -1, -2, -3, -4, -5, -6

# HkBuffer
BUF_POSITION, BUF_NORMAL, BUF_DEPTH_GRADIENT, BUF_INSTANCE_MATERIAL, BUF_VELOCITY_UV, BUF_ALBEDO = range(6)
BUF_VARIANCE0, BUF_RENDER0, BUF_RESERVOIR0 = 6, 9, 12
BUF_DENOISE_INTERNAL0, BUF_DENOISE_INTERNAL_VARIANCE, BUF_DENOISE_RENDER0, BUF_TONE_MAPPED = 22, 26, 27, 30
(BUF_PREVIOUS_POSITION, BUF_PREVIOUS_VELOCITY_UV, BUF_PREVIOUS_TONE_MAPPED, BUF_UPSCALE_OUTPUT, BUF_TAA_OUTPUT, BUF_PREVIOUS_TAA_OUTPUT,
 BUF_UPSCALE_SHARPENED) = range(31, 38)
BUF_PARKED_TO0, BUF_PARKED_RECORD0, BUF_COUNT = 38, 41, 44   # parked scatter stores, + channel (allocated on first use)
HISTORY_AUTO = 0xFFFF
# HkPass
(PASS_PREPASS, PASS_FULL_SCREEN_ALBEDO, PASS_DIRECT_LIT, PASS_DIRECT_EMISSIVE, PASS_INDIRECT, PASS_EMISSIVE_SPATIAL_REUSE,
 PASS_INDIRECT_SPATIAL_REUSE, PASS_DEMODULATION, PASS_DENOISE_L0, PASS_DENOISE_L1, PASS_DENOISE_L2, PASS_DENOISE_L3,
 PASS_TONE_MAPPING, PASS_SMAA_TU4X, PASS_SMAA_TU4X_EXTRAPOLATE, PASS_TAA_JASMINE, PASS_FSR_EASU, PASS_FSR_RCAS, PASS_COUNT) = range(19)
PASS_NAMES = ["prepass", "full_screen_albedo", "direct_lit", "direct_emissive", "indirect_lit_ambient", "emissive_spatial_reuse",
              "indirect_spatial_reuse", "demodulation", "denoise_l0", "denoise_l1", "denoise_l2", "denoise_l3", "tone_mapping",
              "smaa_tu4x", "smaa_tu4x_extrapolate", "taa_jasmine", "fsr_easu", "fsr_rcas"]
# HkStage
STAGE_TEMPORAL, STAGE_SPATIAL, STAGE_POST_PROCESS, STAGE_ANTIALIAS, STAGE_UPSCALE, STAGE_COUNT = range(6)
#: OR-ed into the flags of every Engine / MultiEngine this process creates.  0 in the product.  The GPU test suite sets it to
#: CTX_EXACT_TRAVERSAL (tests/conftest.py, through the environment so that spawned rank processes inherit it) because its bar is
#: bit equality with the oracle, which walks in the reference's order.
DEFAULT_CTX_FLAGS = int(os.environ.get("HIKARI_HIP_DEFAULT_CTX_FLAGS", "0"))
CTX_COUNT_RAYS, CTX_TIME_PASSES, CTX_PLAIN_DIVISION, CTX_SINGLE_STREAM, CTX_DETERMINISTIC_SCATTER, CTX_EXACT_TRAVERSAL = 1, 2, 4, 8, 16, 32
TREE_SAH, TREE_LBVH = 0, 1  # hk_rebuild_scene_trees
CTX_WAVEFRONT, CTX_FUSED_INDIRECT = 64, 128  # schedule of indirect_lit_ambient with >= 2 bounces (hikari_hip.h)
CTX_NO_WIDE_WALK = 256  # closest-hit walks of scenes beyond LDS keep the threaded skip-link walk (A/B switch)
CTX_COUNT_WALKS = 512   # the trace stages run the counting twin of their kernel (same schedule, same walks): hk_debug_read_wf_timeline
CTX_RACING_SCATTER = 1024   # hikari_hip.h HK_CTX_RACING_SCATTER: the reference's own race on previous_spatial (round 6: the default resolves it)
(DEBUG_OPT_SPATIAL_WINDOW, DEBUG_OPT_FRAME_PIPELINE, DEBUG_OPT_WF_TIMELINE, DEBUG_OPT_FLAT_WALK, DEBUG_OPT_FLAT_ORDERINGS, DEBUG_OPT_TRACE_UPDATE, DEBUG_OPT_POST_DEMODULATION, DEBUG_OPT_SIDE_JOIN, DEBUG_OPT_PERSISTENT_PATHS, DEBUG_OPT_MAIN_PRIORITY, DEBUG_OPT_PREPASS_PIPELINE) = range(11)  # hikari_hip_debug.h hk_debug_set_option
TIMING_TRACE_STAGES = 18  # hk_set_timing_mask bit / HkStats slot: every trace launch of the queue-based indirect pass
TRAVERSAL_WIDE = 0x100
FRAME_EXTERNAL_GBUFFER, FRAME_ANTIALIAS, FRAME_BALANCE_BANDS, FRAME_GATHER, FRAME_TIME_BAND = 1, 2, 4, 8, 16
TOPOLOGY_TRIANGLE_LIST, TOPOLOGY_TRIANGLE_STRIP = 0, 1
TAA_JASMINE, TAA_NONE = 0, 1
UPSCALE_FSR1, UPSCALE_SMAA_TU4X = 0, 1
NO_TEXTURE = 0xFFFFFFFF
ADDRESS_CLAMP_TO_EDGE, ADDRESS_REPEAT, ADDRESS_MIRROR_REPEAT = 0, 1, 2
TIMING_SLOTS = 24

f32, u32, u64 = C.c_float, C.c_uint32, C.c_uint64


class HkVertex(C.Structure):
    _fields_ = [("position", f32 * 3), ("u", f32), ("normal", f32 * 3), ("v", f32)]


class HkPrimitiveVertex(C.Structure):
    _fields_ = [("position", f32 * 3), ("index", u32)]


class HkPrimitive(C.Structure):
    _fields_ = [("vertices", HkPrimitiveVertex * 3)]


class HkNode(C.Structure):
    _fields_ = [("min", f32 * 3), ("entry_index", u32), ("max", f32 * 3), ("exit_index", u32)]


class HkMeshIndex(C.Structure):
    _fields_ = [("vertex", u32), ("primitive", u32), ("node_offset", u32), ("node_count", u32)]


class HkInstance(C.Structure):
    _fields_ = [("min", f32 * 3), ("material", u32), ("max", f32 * 3), ("node_index", u32), ("model", f32 * 16),
                ("inverse_transpose_model", f32 * 16), ("mesh", HkMeshIndex)]


class HkMaterial(C.Structure):
    _fields_ = [("base_color", f32 * 4), ("base_color_texture", u32), ("_pad0", u32 * 3), ("emissive", f32 * 4),
                ("emissive_texture", u32), ("perceptual_roughness", f32), ("metallic", f32), ("metallic_roughness_texture", u32),
                ("reflectance", f32), ("normal_map_texture", u32), ("occlusion_texture", u32), ("_pad1", u32)]


class HkAliasEntry(C.Structure):
    _fields_ = [("prob", f32), ("index", u32)]


class HkEmissive(C.Structure):
    _fields_ = [("emissive", f32 * 4), ("position", f32 * 3), ("radius", f32), ("instance", u32), ("_pad0", u32),
                ("alias_table", u32 * 2), ("surface_area", f32), ("node_index", u32), ("_pad1", u32 * 2)]


class HkFrame(C.Structure):
    _fields_ = [("kernel", (f32 * 4) * 3), ("halton", (f32 * 4) * 8), ("clear_color", f32 * 4), ("number", u32),
                ("direct_validate_interval", u32), ("emissive_validate_interval", u32), ("indirect_bounces", u32),
                ("temporal_reuse", u32), ("emissive_spatial_reuse", u32), ("indirect_spatial_reuse", u32),
                ("max_temporal_reuse_count", u32), ("max_spatial_reuse_count", u32), ("max_reservoir_lifetime", f32),
                ("solar_angle", f32), ("max_indirect_luminance", f32), ("upscale_ratio", f32), ("_pad", u32 * 3)]


class HkView(C.Structure):
    _fields_ = [("view_proj", f32 * 16), ("inverse_view_proj", f32 * 16), ("view", f32 * 16), ("inverse_view", f32 * 16),
                ("projection", f32 * 16), ("inverse_projection", f32 * 16), ("world_position", f32 * 3), ("_pad0", f32),
                ("viewport", f32 * 4)]


class HkPreviousView(C.Structure):
    _fields_ = [("view_proj", f32 * 16), ("inverse_view_proj", f32 * 16)]


class HkLights(C.Structure):
    _fields_ = [("directional_color", f32 * 4), ("direction_to_light", f32 * 3), ("n_directional_lights", u32),
                ("ambient_color", f32 * 4)]


class HkSettings(C.Structure):
    _fields_ = [("direct_validate_interval", u32), ("emissive_validate_interval", u32), ("max_temporal_reuse_count", u32),
                ("max_spatial_reuse_count", u32), ("max_reservoir_lifetime", f32), ("solar_angle", f32), ("indirect_bounces", u32),
                ("max_indirect_luminance", f32), ("clear_color", f32 * 4), ("temporal_reuse", u32), ("emissive_spatial_reuse", u32),
                ("indirect_spatial_reuse", u32), ("denoise", u32), ("taa", u32), ("upscale_kind", u32), ("upscale_ratio", f32),
                ("upscale_sharpness", f32)]


class HkImageDesc(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", u32), ("height", u32), ("is_srgb", u32), ("address_u", u32), ("address_v", u32),
                ("filter_linear", u32)]


class HkHaloOp(C.Structure):
    _fields_ = [("buffer", u32), ("peer", u32), ("row_begin", u32), ("row_end", u32), ("row_bytes", u64)]


class HkTransfer(C.Structure):
    _fields_ = [("buffer", u32), ("peer", u32), ("is_recv", u32), ("_pad", u32), ("offset", C.c_uint64), ("bytes", C.c_uint64)]


class HkMovedBox(C.Structure):
    _fields_ = [("min", f32 * 3), ("_pad0", f32), ("max", f32 * 3), ("_pad1", f32), ("previous_from_current", f32 * 16)]


class HkStats(C.Structure):
    _fields_ = [("rays_primary", u64), ("rays_tlas", u64), ("rays_blas", u64), ("frames", u64),
                ("pass_ms_total", C.c_double * TIMING_SLOTS), ("pass_launches", u64 * TIMING_SLOTS), ("last_frame_ms", f32),
                ("_pad", u32), ("scene_mesh_builds", u64), ("scene_instance_builds", u64),
                ("scene_async_instance_uploads", u64), ("scene_device_refits", u64), ("scene_device_tree_builds", u64),
                ("walk_node_steps", u64), ("walk_triangle_tests", u64), ("walk_instance_entries", u64), ("walk_closest_hits", u64), ("walk_top_node_steps", u64),
                ("wide_stack_lost", u64)]


assert C.sizeof(HkVertex) == 32 and C.sizeof(HkPrimitive) == 48 and C.sizeof(HkNode) == 32 and C.sizeof(HkInstance) == 176
assert C.sizeof(HkMaterial) == 80 and C.sizeof(HkEmissive) == 64 and C.sizeof(HkFrame) == 256 and C.sizeof(HkView) == 416
assert C.sizeof(HkPreviousView) == 128 and C.sizeof(HkAliasEntry) == 8


class HikariError(RuntimeError):
    def __init__(self, code, fn, message):
        super().__init__(f"{fn} failed with code {code}: {message}")
        self.code = code


P = C.POINTER
_vp = C.c_void_p

# name -> (argtypes); every function returns int unless listed in _NON_INT
_SIGNATURES = {
    "create": [C.c_int, u32, P(_vp)],
    "upload_meshes": [_vp, P(HkVertex), u32, P(HkPrimitive), u32, P(HkNode), u32],
    "upload_materials": [_vp, P(HkMaterial), u32],
    "upload_instances": [_vp, P(HkInstance), u32, P(HkNode), u32, P(HkEmissive), u32, P(HkNode), u32, P(HkAliasEntry), u32],
    "upload_previous_transforms": [_vp, P(f32), u32],
    "upload_noise": [_vp, _vp, C.c_size_t],
    "upload_textures": [_vp, P(HkImageDesc), u32],
    "resize": [_vp, u32, u32, f32],
    "set_view_options": [_vp, u32, u32, f32],
    "frame_begin": [_vp, P(HkFrame), P(HkView), P(HkPreviousView), P(HkLights)],
    "pass_run": [_vp, u32, u32, u32, u32],
    "frame_stage": [_vp, u32, P(HkSettings), u32],
    "frame_render": [_vp, P(HkFrame), P(HkView), P(HkPreviousView), P(HkLights), P(HkSettings), u32],
    "frame_wait": [_vp],
    "set_band": [_vp, u32, u32],
    "set_history_rows": [_vp, u32],
    "history_rows": [_vp, P(u32)],
    "scene_bounds": [_vp, P(f32), P(f32)],
    "set_band_bounds": [_vp, P(u32), u32],
    "row_costs": [_vp, P(u32), u32],
    "buffer_info": [_vp, u32, P(u32), P(u32), P(u32)],
    "read_buffer": [_vp, u32, _vp, C.c_size_t],
    "write_buffer": [_vp, u32, _vp, C.c_size_t],
    "device_ptr": [_vp, u32, P(_vp), P(C.c_size_t)],
    "get_stats": [_vp, P(HkStats)],
    "reset_stats": [_vp],
}
# include/hikari_hip_debug.h: test and measurement hooks (not part of the boundary a host binds)
_DEBUG = {
    "debug_math": [_vp, u32, P(f32), P(f32), P(f32), C.c_size_t],
    "debug_read_trees": [_vp, P(HkNode), u32, P(HkNode), u32],
    "debug_comm_loopback": [_vp, u32, u32, u32, u32, u32],
    "debug_read_wf_timeline": [_vp, P(C.c_uint64), u32],
    "debug_set_option": [_vp, u32, C.c_int64],
    "debug_comm_lanes": [_vp, P(u32)],
    "debug_multi_serial": [C.c_int],
    "debug_spatial_windowed_launches": [_vp, P(C.c_uint64)],
    "debug_main_stream_priority": [_vp, P(u32)],
    "measure_hbm": [_vp, C.c_size_t, u32, P(C.c_double), P(C.c_double)],
    "measure_valu": [_vp, u32, P(C.c_double)],
    "measure_gather": [_vp, C.c_size_t, u32, u32, u32, u32, P(C.c_double), P(C.c_double)],
}
# host logic, builders, multi-GPU and GPU-only entry points
_PRODUCT_ONLY = {
    "device_count": [P(C.c_int)],
    "settings_default": [P(HkSettings)],
    "frame_from_settings": [P(HkSettings), u32, P(HkFrame)],
    "scaled_size": [u32, u32, f32, P(u32), P(u32)],
    "scene_builder_create": [P(_vp)],
    "scene_builder_add_mesh": [_vp, P(f32), P(f32), P(f32), u32, P(u32), u32, u32, P(u32)],
    "scene_builder_add_material": [_vp, P(HkMaterial), P(u32)],
    "scene_builder_add_instance": [_vp, u32, u32, P(f32), P(u32)],
    "scene_builder_finish": [_vp],
    "scene_builder_finish_instances": [_vp],
    "scene_builder_remove_instance": [_vp, u32],
    "scene_builder_set_instance_material": [_vp, u32, u32],
    "scene_builder_set_instance_transform": [_vp, u32, P(f32)],
    "scene_builder_previous_transforms": [_vp, P(P(f32)), P(u32)],
    "scene_builder_vertices": [_vp, P(P(HkVertex)), P(u32)],
    "scene_builder_primitives": [_vp, P(P(HkPrimitive)), P(u32)],
    "scene_builder_asset_nodes": [_vp, P(P(HkNode)), P(u32)],
    "scene_builder_materials": [_vp, P(P(HkMaterial)), P(u32)],
    "scene_builder_instances": [_vp, P(P(HkInstance)), P(u32)],
    "scene_builder_instance_nodes": [_vp, P(P(HkNode)), P(u32)],
    "scene_builder_emissives": [_vp, P(P(HkEmissive)), P(u32)],
    "scene_builder_emissive_nodes": [_vp, P(P(HkNode)), P(u32)],
    "scene_builder_alias_table": [_vp, P(P(HkAliasEntry)), P(u32)],
    "upload_scene": [_vp, _vp],
    "upload_scene_instances": [_vp, _vp],
    "refit_scene_instances": [_vp, _vp, P(u32)],
    "rebuild_scene_trees": [_vp, u32],
    "update_scene_instances": [_vp, _vp, u32],
    "band_rows": [u32, u32, u32, P(u32), P(u32)],
    "balanced_band_bounds": [P(u32), u32, u32, u32, u32, u32, f32, P(u32)],
    "balance_bands": [_vp, u32, P(u32), u32],
    "rebalanced_band_bounds": [P(u32), P(f32), u32, u32, P(f32), u32, u32, f32, P(u32)],
    "band_migration_plan": [u32, u32, f32, P(u32), P(u32), u32, u32, u32, P(HkSettings), P(HkHaloOp), P(u32)],
    "band_migration_schedule": [u32, u32, f32, P(u32), P(u32), u32, u32, u32, P(HkSettings), P(HkTransfer), P(u32)],
    "migrate_bands": [_vp, P(u32), u32, u32, P(HkSettings)],
    "band_time_ms": [_vp, P(f32)],
    "multi_migrate_bands": [_vp, P(u32), u32, u32, P(HkSettings)],
    "band_gather_schedule": [u32, u32, f32, u32, P(u32), u32, u32, u32, u32, P(HkTransfer), P(u32)],
    "comm_gather": [_vp, u32, u32],
    "multi_gather": [_vp, u32, u32],
    "get_band_bounds": [_vp, P(u32), u32],
    "get_band": [_vp, P(u32), P(u32)],
    "band_plan_bounds": [u32, u32, f32, P(u32), u32, u32, u32, u32, P(HkSettings), P(HkHaloOp), P(u32)],
    "band_schedule_bounds": [u32, u32, f32, P(u32), u32, u32, u32, u32, P(HkSettings), P(HkTransfer), P(u32)],
    "band_plan": [_vp, u32, P(HkSettings), P(HkHaloOp), P(u32)],
    "band_plan_for": [u32, u32, f32, u32, u32, u32, u32, P(HkSettings), P(HkHaloOp), P(u32)],
    "stream": [_vp, P(_vp)],
    "set_stream": [_vp, _vp],
    "set_timing_mask": [_vp, u32],
    "indirect_schedule": [_vp, P(u32)],
    "traversal_mode": [_vp, P(u32), P(u32)],
    "bvh_rethread": [P(HkNode), u32, u32, P(HkNode)],
    "band_schedule": [u32, u32, f32, u32, u32, u32, u32, P(HkSettings), P(HkTransfer), P(u32)],
    "comm_unique_id": [P(C.c_uint8)],
    "comm_available": [_vp],
    "comm_init": [_vp, u32, u32, P(C.c_uint8)],
    "comm_destroy": [_vp],
    "comm_set_history_rows": [_vp, u32],
    "history_rows_bound": [P(HkView), P(HkPreviousView), u32, P(f32), P(f32), P(HkMovedBox), u32, P(u32)],
    "comm_exchange": [_vp, u32, P(HkSettings)],
    "multi_create": [u32, P(C.c_int), u32, P(_vp)],
    "multi_context": [_vp, u32, P(_vp)],
    "multi_upload_scene": [_vp, _vp],
    "multi_upload_scene_instances": [_vp, _vp],
    "multi_refit_scene_instances": [_vp, _vp, P(u32)],
    "multi_rebuild_scene_trees": [_vp, u32],
    "multi_update_scene_instances": [_vp, _vp, u32],
    "multi_set_band_bounds": [_vp, P(u32), u32],
    "multi_upload_textures": [_vp, P(HkImageDesc), u32],
    "multi_upload_noise": [_vp, _vp, C.c_size_t],
    "multi_resize": [_vp, u32, u32, f32],
    "multi_set_history_rows": [_vp, u32],
    "multi_frame_render": [_vp, P(HkFrame), P(HkView), P(HkPreviousView), P(HkLights), P(HkSettings), u32],
    "multi_wait": [_vp],
    "multi_read_buffer": [_vp, u32, _vp, C.c_size_t],
}
_VOID = {"destroy": [_vp], "scene_builder_destroy": [_vp], "multi_destroy": [_vp]}

#: every symbol include/hikari_hip.h declares / include/hikari_hip_debug.h declares (checked by tests/test_abi.py)
DECLARED_SYMBOLS = sorted(["hk_" + n for n in list(_SIGNATURES) + list(_PRODUCT_ONLY) + list(_VOID)] + ["hk_abi_version", "hk_last_error", "hk_final_buffer", "hk_build_info"])
DECLARED_DEBUG_SYMBOLS = sorted("hk_" + n for n in _DEBUG)


class Api:
    """Resolved entry points of libhikari_hip.so."""

    prefix = "hk_"

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the product path.")
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self._fns = {}
        for table, restype in ((_SIGNATURES, C.c_int), (_PRODUCT_ONLY, C.c_int), (_DEBUG, C.c_int), (_VOID, None)):
            for name, argtypes in table.items():
                fn = getattr(self.dll, "hk_" + name)
                fn.argtypes, fn.restype = argtypes, restype
                self._fns[name] = fn
        self._last_error = self.dll.hk_last_error
        self._last_error.restype = C.c_char_p
        self._abi = self.dll.hk_abi_version
        self._abi.restype = u32
        self._build_info = self.dll.hk_build_info
        self._build_info.restype = C.c_char_p
        self._final_buffer = self.dll.hk_final_buffer
        self._final_buffer.argtypes, self._final_buffer.restype = [P(HkSettings), u32], u32

    def final_buffer(self, settings_c, frame_flags=0):
        """hk_final_buffer: the HkBuffer id of the image a frame rendered with these settings / flags presents."""
        return int(self._final_buffer(C.byref(settings_c), frame_flags))

    def abi_version(self):
        return int(self._abi())

    def build_info(self):
        """hk_build_info: the hash of the sources, the compiler and flags this binary was built with, and when."""
        return self._build_info().decode()

    def last_error(self):
        msg = self._last_error()
        return msg.decode() if msg else ""

    def raw(self, name):
        return self._fns[name]

    def call(self, name, *args):
        rc = self._fns[name](*args)
        if rc is not None and rc != HK_OK:
            raise HikariError(rc, self.prefix + name, self.last_error())
        return rc


_API = None


def api():
    """The product library (loaded once)."""
    global _API
    if _API is None:
        _API = Api(LIB_PATH)
    return _API
