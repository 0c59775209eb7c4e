//! `src/hip.rs` of bevy-hikari with the MI355X library behind its nodes (feature `hip`).
//!
//! NOT COMPILED in the repository that ships it (no Rust toolchain there) - see README.md next to this file.
//!
//! The reference records three node runs per view and frame - `PrepassNode` (raster G-buffer, prepass.rs:556-640), `LightNode` (five
//! compute dispatches, light.rs:581-702) and `PostProcessNode` (demodulation, four a-trous levels, tone mapping, TAA / upscaling,
//! post_process.rs:1131-1311).  With the library the same frame is ONE call, `hk_frame_render`, issued by `HipFrameNode` in the slot of
//! `graph::node::LIGHT`; the other two nodes stay in the graph as no-ops so that `OverlayNode`'s edges do not move.
use std::ffi::CStr;
use std::ptr;

use bevy::{
    pbr::ExtractedDirectionalLight,
    prelude::*,
    render::{
        camera::ExtractedCamera,
        render_graph::{Node, NodeRunError, RenderGraphContext, SlotInfo, SlotType},
        renderer::RenderContext,
        view::ExtractedView,
        RenderApp, RenderStage,
    },
};
use hikari_hip_sys as hk;

use bvh::bounding_hierarchy::BHShape;

use crate::{
    mesh_material::{
        GpuAliasEntry, GpuEmissive, GpuInstance, GpuNode, GpuPrimitiveCompact, GpuStandardMaterial, GpuVertexCompact, InstanceRenderAssets, MaterialRenderAssets,
        MeshMaterialSystems, MeshRenderAssets,
    },
    transform::GlobalTransformQueue,
    view::FrameUniform,
    HikariSettings, Taa, Upscale,
};

/// `hk_*` return codes: 0 or a negative `HK_E_*` (hikari_hip.h:48-56); the text is the library's thread-local description.
#[derive(Debug)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}

fn check(code: i32) -> Result<(), HipError> {
    if code == 0 {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(hk::hk_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { code, message })
}

/// The library's context: one per render world (one GPU).  Freed with the render world.
#[derive(Resource)]
pub struct HipContext {
    ctx: *mut hk::HkCtx,
    size: (u32, u32, f32),
    noise_uploaded: bool,
}
// the library serialises on the context's own HIP stream; bevy runs the render graph on one thread per frame
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

impl HipContext {
    pub fn new(device: i32) -> Result<Self, HipError> {
        assert_eq!(unsafe { hk::hk_abi_version() }, hk::HK_ABI_VERSION, "libhikari_hip.so and hikari-hip-sys disagree about the ABI");
        let mut ctx = ptr::null_mut();
        check(unsafe { hk::hk_create(device, 0, &mut ctx) })?;
        Ok(Self { ctx, size: (0, 0, 0.0), noise_uploaded: false })
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { hk::hk_destroy(self.ctx) }
    }
}

// The reference's GPU records and the library's follow the same WGSL structs (mesh_material_types.wgsl; hikari_hip.h:60-141,
// tests/test_abi.py pins the sizes against the reference's WGSL text) - but the Rust structs are not `repr(C)` (encase serialises them
// field by field), so they are converted field by field here too, never reinterpreted.
fn vertex(v: &GpuVertexCompact) -> hk::HkVertex {
    hk::HkVertex { position: v.position.to_array(), u: v.u, normal: v.normal.to_array(), v: v.v }
}
fn primitive(p: &GpuPrimitiveCompact) -> hk::HkPrimitive {
    let corner = |k: usize| hk::HkPrimitiveVertex { position: p.vertices[k].position.to_array(), index: p.vertices[k].index };
    hk::HkPrimitive { vertices: [corner(0), corner(1), corner(2)] }
}
fn node(n: &GpuNode) -> hk::HkNode {
    hk::HkNode { min: n.min.to_array(), entry_index: n.entry_index, max: n.max.to_array(), exit_index: n.exit_index }
}
fn instance(i: &GpuInstance) -> hk::HkInstance {
    hk::HkInstance {
        min: i.min.to_array(),
        material: i.material,
        max: i.max.to_array(),
        node_index: i.bh_node_index() as u32,
        model: i.transform.to_cols_array(),
        inverse_transpose_model: i.inverse_transpose_model.to_cols_array(),
        mesh: hk::HkMeshIndex { vertex: i.mesh.vertex, primitive: i.mesh.primitive, node_offset: i.mesh.node.x, node_count: i.mesh.node.y },
    }
}
fn material(m: &GpuStandardMaterial) -> hk::HkMaterial {
    hk::HkMaterial {
        base_color: m.base_color.to_array(),
        base_color_texture: m.base_color_texture,
        _pad0: [0; 3],
        emissive: m.emissive.to_array(),
        emissive_texture: m.emissive_texture,
        perceptual_roughness: m.perceptual_roughness,
        metallic: m.metallic,
        metallic_roughness_texture: m.metallic_roughness_texture,
        reflectance: m.reflectance,
        normal_map_texture: m.normal_map_texture,
        occlusion_texture: m.occlusion_texture,
        _pad1: 0,
    }
}
fn emissive(e: &GpuEmissive) -> hk::HkEmissive {
    hk::HkEmissive {
        emissive: e.emissive.to_array(),
        position: e.position.to_array(),
        radius: e.radius,
        instance: e.instance,
        _pad0: 0,
        alias_table: e.alias_table.to_array(),
        surface_area: e.surface_area,
        node_index: e.bh_node_index() as u32,
        _pad1: [0; 2],
    }
}
fn alias_entry(a: &GpuAliasEntry) -> hk::HkAliasEntry {
    hk::HkAliasEntry { prob: a.prob, index: a.index }
}

pub struct HipPlugin;

impl Plugin for HipPlugin {
    fn build(&self, app: &mut App) {
        let render_app = match app.get_sub_app_mut(RenderApp) {
            Ok(render_app) => render_app,
            Err(_) => return,
        };
        match HipContext::new(0) {
            Ok(context) => {
                render_app
                    .insert_resource(context)
                    // after the reference's own prepare systems have filled the asset vectors (mesh.rs:106-165, material.rs:139-202,
                    // instance.rs:430-437): they carry these labels (mesh.rs:28, material.rs:31, instance.rs:42)
                    .add_system_to_stage(RenderStage::Prepare, upload_noise)
                    .add_system_to_stage(RenderStage::Prepare, upload_meshes.after(MeshMaterialSystems::PrepareAssets))
                    .add_system_to_stage(RenderStage::Prepare, upload_materials.after(MeshMaterialSystems::PrepareAssets))
                    .add_system_to_stage(RenderStage::Prepare, upload_instances.after(MeshMaterialSystems::PrepareInstances));
            }
            Err(error) => error!("hikari-hip: {} ({})", error.message, error.code),
        }
    }
}

/// NoiseTextures (lib.rs:189-219,515-598): 16 tiles of 64x64 RGBA8, tile-major, once they have all decoded.  `HipNoiseImages` is
/// the one thing the maintainer adds on the main-world side: the decoded `Image::data` of the 16 handles, extracted once.
#[derive(Resource, Default, Clone, Deref)]
pub struct HipNoiseImages(pub Vec<Vec<u8>>);

fn upload_noise(mut context: ResMut<HipContext>, tiles: Option<Res<HipNoiseImages>>) {
    let tiles = match tiles {
        Some(tiles) if !context.noise_uploaded && tiles.len() == crate::NOISE_TEXTURE_COUNT => tiles,
        _ => return,
    };
    let rgba: Vec<u8> = tiles.iter().flat_map(|tile| tile.iter().copied()).collect();
    if check(unsafe { hk::hk_upload_noise(context.ctx, rgba.as_ptr(), rgba.len()) }).is_ok() {
        context.noise_uploaded = true;
    }
}

/// MeshRenderAssets::set + write_buffer (mesh.rs:43-64): the same three vectors.
fn upload_meshes(context: Res<HipContext>, assets: Res<MeshRenderAssets>) {
    if !assets.is_changed() {
        return;
    }
    let vertices: Vec<_> = assets.vertex_buffer.get().data.iter().map(vertex).collect();
    let primitives: Vec<_> = assets.primitive_buffer.get().data.iter().map(primitive).collect();
    let nodes: Vec<_> = assets.node_buffer.get().data.iter().map(node).collect();
    let code = unsafe {
        hk::hk_upload_meshes(context.ctx, vertices.as_ptr(), vertices.len() as u32, primitives.as_ptr(), primitives.len() as u32, nodes.as_ptr(), nodes.len() as u32)
    };
    if let Err(error) = check(code) {
        error!("hk_upload_meshes: {}", error.message);
    }
}

/// MaterialRenderAssets (material.rs:52,201-202).  Textures: `hk_upload_textures` with the images of `MaterialTextures::data` in id
/// order (material.rs:55-86), as `HkImageDesc`s over their `Image::data`.
fn upload_materials(context: Res<HipContext>, assets: Res<MaterialRenderAssets>) {
    if !assets.is_changed() {
        return;
    }
    let materials: Vec<_> = assets.0.get().data.iter().map(material).collect();
    let code = unsafe { hk::hk_upload_materials(context.ctx, materials.as_ptr(), materials.len() as u32) };
    if let Err(error) = check(code) {
        error!("hk_upload_materials: {}", error.message);
    }
}

/// InstanceRenderAssets::set + write_buffer (instance.rs:82-108), then the previous frame's model matrices in instance order
/// (PreviousMeshUniform, instance.rs:111-128): both every frame, as the reference extracts both every frame.
fn upload_instances(context: Res<HipContext>, assets: Res<InstanceRenderAssets>, previous: Option<Res<HipPreviousModels>>) {
    let instances: Vec<_> = assets.instance_buffer.get().data.iter().map(instance).collect();
    let nodes: Vec<_> = assets.instance_node_buffer.get().data.iter().map(node).collect();
    let emissives: Vec<_> = assets.emissive_buffer.get().data.iter().map(emissive).collect();
    let emissive_nodes: Vec<_> = assets.emissive_node_buffer.get().data.iter().map(node).collect();
    let alias: Vec<_> = assets.alias_table_buffer.get().data.iter().map(alias_entry).collect();
    let code = unsafe {
        hk::hk_upload_instances(
            context.ctx,
            instances.as_ptr(),
            instances.len() as u32,
            nodes.as_ptr(),
            nodes.len() as u32,
            emissives.as_ptr(),
            emissives.len() as u32,
            emissive_nodes.as_ptr(),
            emissive_nodes.len() as u32,
            alias.as_ptr(),
            alias.len() as u32,
        )
    };
    if let Err(error) = check(code) {
        error!("hk_upload_instances: {}", error.message);
        return;
    }
    if let Some(previous) = previous {
        if previous.len() == instances.len() {
            let models: Vec<f32> = previous.iter().flat_map(|model| model.to_cols_array()).collect();
            let _ = check(unsafe { hk::hk_upload_previous_transforms(context.ctx, models.as_ptr(), instances.len() as u32) });
        }
    }
}

/// `GlobalTransformQueue[1]` of every instance IN THE ORDER `prepare_instances` pushed the instances (instance.rs:268-437): that
/// system is the place that knows the order - the maintainer fills this resource there, next to `InstanceRenderAssets::set`.
#[derive(Resource, Default, Clone, Deref)]
pub struct HipPreviousModels(pub Vec<Mat4>);

fn settings_to_hk(settings: &HikariSettings) -> hk::HkSettings {
    let (upscale_kind, upscale_sharpness) = match settings.upscale {
        Upscale::Fsr1 { sharpness, .. } => (hk::HK_UPSCALE_FSR1, sharpness),
        Upscale::SmaaTu4x { .. } => (hk::HK_UPSCALE_SMAA_TU4X, 0.0),
    };
    hk::HkSettings {
        direct_validate_interval: settings.direct_validate_interval as u32,
        emissive_validate_interval: settings.emissive_validate_interval as u32,
        max_temporal_reuse_count: settings.max_temporal_reuse_count as u32,
        max_spatial_reuse_count: settings.max_spatial_reuse_count as u32,
        max_reservoir_lifetime: settings.max_reservoir_lifetime,
        solar_angle: settings.solar_angle,
        indirect_bounces: settings.indirect_bounces as u32,
        max_indirect_luminance: settings.max_indirect_luminance,
        clear_color: settings.clear_color.as_linear_rgba_f32(),
        temporal_reuse: settings.temporal_reuse as u32,
        emissive_spatial_reuse: settings.emissive_spatial_reuse as u32,
        indirect_spatial_reuse: settings.indirect_spatial_reuse as u32,
        denoise: settings.denoise as u32,
        taa: match settings.taa {
            Taa::Jasmine => hk::HK_TAA_JASMINE,
            Taa::None => hk::HK_TAA_NONE,
        },
        upscale_kind,
        upscale_ratio: settings.upscale.ratio(),
        upscale_sharpness,
    }
}

/// FrameUniform (view.rs:105-123) -> its std140 image (hikari_hip.h HkFrame: the mat3's columns padded to vec4).
fn frame_to_hk(frame: &FrameUniform) -> hk::HkFrame {
    let mut out: hk::HkFrame = unsafe { std::mem::zeroed() };
    for (column, axis) in [frame.kernel.x_axis, frame.kernel.y_axis, frame.kernel.z_axis].iter().enumerate() {
        out.kernel[column] = [axis.x, axis.y, axis.z, 0.0];
    }
    for (k, h) in frame.halton.iter().enumerate() {
        out.halton[k] = h.to_array();
    }
    out.clear_color = frame.clear_color.to_array();
    out.number = frame.number;
    out.direct_validate_interval = frame.direct_validate_interval;
    out.emissive_validate_interval = frame.emissive_validate_interval;
    out.indirect_bounces = frame.indirect_bounces;
    out.temporal_reuse = frame.temporal_reuse;
    out.emissive_spatial_reuse = frame.emissive_spatial_reuse;
    out.indirect_spatial_reuse = frame.indirect_spatial_reuse;
    out.max_temporal_reuse_count = frame.max_temporal_reuse_count;
    out.max_spatial_reuse_count = frame.max_spatial_reuse_count;
    out.max_reservoir_lifetime = frame.max_reservoir_lifetime;
    out.solar_angle = frame.solar_angle;
    out.max_indirect_luminance = frame.max_indirect_luminance;
    out.upscale_ratio = frame.upscale_ratio;
    out
}

/// bevy_pbr 0.9.1 `ViewUniform` (prepare_view_uniforms) from the extracted view; the previous one as view.rs:46-75 builds it.
fn views_to_hk(view: &ExtractedView, previous: &GlobalTransformQueue) -> (hk::HkView, hk::HkPreviousView) {
    let projection = view.projection;
    let inverse_projection = projection.inverse();
    let view_matrix = view.transform.compute_matrix();
    let inverse_view = view_matrix.inverse();
    let current = hk::HkView {
        view_proj: (projection * inverse_view).to_cols_array(),
        inverse_view_proj: (view_matrix * inverse_projection).to_cols_array(),
        view: view_matrix.to_cols_array(),
        inverse_view: inverse_view.to_cols_array(),
        projection: projection.to_cols_array(),
        inverse_projection: inverse_projection.to_cols_array(),
        world_position: view.transform.translation().to_array(),
        _pad0: 0.0,
        viewport: [view.viewport.x as f32, view.viewport.y as f32, view.viewport.z as f32, view.viewport.w as f32],
    };
    let previous_view = previous[1];
    let previous = hk::HkPreviousView {
        view_proj: (projection * previous_view.inverse()).to_cols_array(),
        inverse_view_proj: (previous_view * inverse_projection).to_cols_array(),
    };
    (current, previous)
}

/// In the slot of `graph::node::LIGHT`.  Where the reference's nodes return `Ok(())` for a missing resource (light.rs:606-617) the
/// library returns `HK_E_NOT_READY` and enqueues nothing: mapped to `Ok(())` here, every other error is logged once per frame.
pub struct HipFrameNode {
    query: QueryState<(&'static ExtractedCamera, &'static ExtractedView, &'static GlobalTransformQueue, &'static FrameUniform, &'static HikariSettings)>,
    lights: QueryState<&'static ExtractedDirectionalLight>,
}

impl HipFrameNode {
    pub const IN_VIEW: &'static str = "view";

    pub fn new(world: &mut World) -> Self {
        Self { query: world.query_filtered(), lights: world.query() }
    }
}

impl Node for HipFrameNode {
    fn input(&self) -> Vec<SlotInfo> {
        vec![SlotInfo::new(Self::IN_VIEW, SlotType::Entity)]
    }

    fn update(&mut self, world: &mut World) {
        self.query.update_archetypes(world);
        self.lights.update_archetypes(world);
    }

    fn run(&self, graph: &mut RenderGraphContext, _render_context: &mut RenderContext, world: &World) -> Result<(), NodeRunError> {
        let entity = graph.get_input_entity(Self::IN_VIEW)?;
        let (camera, view, previous, frame, settings) = match self.query.get_manual(world, entity) {
            Ok(query) => query,
            Err(_) => return Ok(()),
        };
        let size = match camera.physical_target_size {
            Some(size) => size,
            None => return Ok(()),
        };
        // the resource is only ever touched from the render graph's thread
        let context = world.resource::<HipContext>();
        let context = unsafe { &mut *(context as *const HipContext as *mut HipContext) };
        let wanted = (size.x, size.y, settings.upscale.ratio());
        if context.size != wanted {
            // prepare_light_textures, light.rs:342-363: reallocate and zero the reservoirs on a size change
            if check(unsafe { hk::hk_resize(context.ctx, size.x, size.y, wanted.2) }).is_err() {
                return Ok(());
            }
            context.size = wanted;
        }
        // GpuLights.directional_lights[0] and ambient_color (light.wgsl:611,832,847-855)
        let mut lights: hk::HkLights = unsafe { std::mem::zeroed() };
        if let Some(light) = self.lights.iter_manual(world).next() {
            // bevy_pbr 0.9.1 prepare_lights: colour x illuminance x the default camera's exposure, 1 / (2^ev100 x 1.2) with
            // ev100 = log2(4.0^2 x 250 x 100 / 100) - i.e. 1 / 4800 (bevy-hikari_amd/plugin.py lights_uniform does the same)
            let color = light.color.as_linear_rgba_f32();
            let intensity = light.illuminance / 4800.0;
            lights.directional_color = [color[0] * intensity, color[1] * intensity, color[2] * intensity, color[3] * intensity];
            lights.direction_to_light = light.transform.back().to_array();
            lights.n_directional_lights = 1;
        }
        if let Some(ambient) = world.get_resource::<AmbientLight>() {
            let color = ambient.color.as_linear_rgba_f32();
            lights.ambient_color = [color[0] * ambient.brightness, color[1] * ambient.brightness, color[2] * ambient.brightness, color[3] * ambient.brightness];
        }
        let (view, previous_view) = views_to_hk(view, previous);
        let code = unsafe { hk::hk_frame_render(context.ctx, &frame_to_hk(frame), &view, &previous_view, &lights, &settings_to_hk(settings), hk::HK_FRAME_ANTIALIAS) };
        match check(code) {
            Ok(()) => Ok(()),
            Err(error) if error.code == hk::HK_E_NOT_READY => Ok(()),
            Err(error) => {
                error!("hk_frame_render: {} ({})", error.message, error.code);
                Ok(())
            }
        }
    }
}

/// What `OverlayNode` blits (overlay.rs:226-231): the buffer that holds the frame's final image for these settings, as a device pointer
/// for an external-memory import, after the frame's work has been waited for.
pub fn final_image(context: &HipContext, settings: &HikariSettings) -> Result<(*mut std::ffi::c_void, usize), HipError> {
    let hk_settings = settings_to_hk(settings);
    let buffer = unsafe { hk::hk_final_buffer(&hk_settings, hk::HK_FRAME_ANTIALIAS) };
    check(unsafe { hk::hk_frame_wait(context.ctx) })?;
    let (mut pointer, mut bytes) = (ptr::null_mut(), 0usize);
    check(unsafe { hk::hk_device_ptr(context.ctx, buffer, &mut pointer, &mut bytes) })?;
    Ok((pointer, bytes))
}
