/*
 * slhip.h -- C ABI of libslhip.so, the MI355X (gfx950) scene-synthesis hot path.
 *
 * This is the drop-in boundary for the path stillleben implements with
 *   - PhysX  : Scene::simulateTableTopScene / Scene::simulate / Scene::checkCollisions /
 *              ManipulationSim::step       (reference src/scene.cpp:612-759, :903-925,
 *                                           src/manipulation_sim.cpp:83-93)
 *   - OpenGL : RenderPass::render           (reference src/render_pass.cpp:303-796 and
 *                                           src/shaders/render_shader.{vert,geom,frag})
 *   - CUDA   : generateSobelValidMask / dilateObjectMask (reference python/src/diff.cu,
 *              python/src/bridge_diff.cpp:13-157) and the pose backward of
 *              python/stillleben/diff.py:355-523
 *
 * Conventions
 *   - plain C, no exceptions, no torch types.  Every function returns 0 on success and a
 *     negative code on failure; slhip_last_error() returns a thread-local message.
 *   - every pointer named d_* is a DEVICE pointer into caller-owned memory (the Python host
 *     allocates it with torch); h_* are host pointers.  Nothing is allocated behind the
 *     caller's back except the opaque handles created by *_create.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All launches are
 *     asynchronous on that stream; nothing synchronises unless documented.
 *   - matrices are ROW-major float[16] (m[4*r+c]); the reference's Magnum matrices are
 *     column-major and its Python boundary transposes them (python/src/py_magnum.h:55-69),
 *     so row-major here == the tensors the reference's Python API hands out.
 *   - all structs are PODs with explicit sizes; arrays of structs are tightly packed.
 */
#ifndef SLHIP_H
#define SLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLHIP_ABI_VERSION 5   /* 2: slhip_render_scratch.d_shadow_tiles, slhip_render_scratch_bytes fills 7 sizes
                                 3: slhip_render_scratch.d_vattr is REQUIRED (the post-transform vertex cache) and `_pad` became
                                    shadow_lights; d_clip holds 9 float4 planes per vertex; slhip_settle_params grew to 116 bytes
                                    (list capacities instead of caps, pair_contact_budget, resume: the contact state of a settle
                                    outlives the call); slhip_settle_caps fills counts[8]
                                 4: slhip_settle_params.max_body_pairs_per_scene (120 bytes); slhip_settle_caps fills ten counts
                                 5: slhip_settle_params.stabilization_threshold (128 bytes); slhip_body.stab (304 bytes) carries the
                                    stabilisation state of a body, SLHIP_BODY_FROZEN; slhip_settle_solver_wave_lds, slhip_host_convex_hull,
                                    slhip_host_fill_holes                                                                      */
#define SLHIP_NUM_LIGHTS 3 /* reference include/stillleben/common.h:17 */

/* ---------------------------------------------------------------------------------------------
 * Render half
 * ------------------------------------------------------------------------------------------- */

/* Image-based lighting ('next' row f1 of SURVEY.md 8f): the textures LightMap::load builds
 * (reference src/light_map.cpp:360-606).  All cube maps are RGBA f32; faces in the OpenGL order
 * +X -X +Y -Y +Z -Z; level l of a cube holds 6 faces of (size >> l)^2 texels and starts
 * 4 * 6 * sum_{k<l} (size >> k)^2 floats into the buffer; rows of a face run along t.
 * Sampling rules (OpenGL leaves them to the implementation; these are ours, shared by the oracle):
 * bilinear within a level, texels beyond a face edge are taken from the face their direction points
 * into (seamless), explicit-LOD fetches blend the two nearest levels linearly, `texture()` without an
 * explicit LOD reads level 0.                                                                     */
typedef struct {
    float* d_env;            /* environment cube, env_levels levels (light_map.cpp:379-430)          */
    float* d_irradiance;     /* diffuse irradiance cube, 1 level      (:455-515)                     */
    float* d_prefilter;      /* GGX-prefiltered cube, pre_levels levels, roughness = l / (levels-1) (:517-573) */
    float* d_brdf_lut;       /* f32 [lut_size][lut_size][2]: scale, bias of F0 (:575-603)             */
    uint32_t env_size, env_levels;   /* reference: 512, 10 */
    uint32_t irr_size;               /* 32  */
    uint32_t pre_size, pre_levels;   /* 128, 5 */
    uint32_t lut_size;               /* 512 */
} slhip_light_map;

/* Mesh pool: structure-of-arrays vertex storage shared by every scene of a batch.
 * Replaces the 68-byte interleaved GL vertex buffer of consolidateMesh
 * (reference src/mesh_tools/consolidate.cpp:53-61).  The 1-based `vertexIndex` attribute of
 * the reference (consolidate.cpp:335) is implicit: vertexIndex = (vertex - vtx_base) + 1.   */
typedef struct {
    const float* d_pos;   /* float4[V]  x y z 1                                   */
    const float* d_nrm;   /* float4[V]  nx ny nz 0                                */
    const float* d_uv;    /* float2[V]                                            */
    const float* d_col;   /* float4[V]  vertex colour (default 1,1,1,1)            */
    const float* d_tan;   /* float4[V]  tangent xyz, bitangent sign (consolidate.cpp:275-279 writes w = 1;
                             missing tangents are computed, compute_tangents.cpp); may be NULL when no
                             draw has a normal texture                                          */
    const uint32_t* d_idx;/* u32[3T]    indices relative to the draw's vtx_base   */
    const uint8_t* d_tex; /* RGBA8 texel pool (all base-colour textures, mip 0)   */
    uint64_t n_vertices;
    uint64_t n_indices;
    uint64_t n_tex_bytes;
    const slhip_light_map* d_light_maps;  /* DEVICE array; may be NULL when no scene uses one */
    uint64_t n_light_maps;
} slhip_mesh_pool;

/* draw flags */
#define SLHIP_DRAW_HAS_BASE_TEX   1u  /* base colour texture bound (render_shader.cpp:430-433)   */
#define SLHIP_DRAW_VERTEX_COLORS  2u  /* mesh carries vertex colours; informational: the reference's
                                         fragment shader never reads the varying (render_shader.vert:86) */
#define SLHIP_DRAW_CASTS_SHADOW   4u  /* Object::castsShadows (render_pass.cpp:437)              */
#define SLHIP_DRAW_ALPHA_TEST     8u  /* texture has an alpha channel: cut-off in the z pass     */
#define SLHIP_DRAW_NO_VERTEX_ID  16u  /* mesh without the vertexIndex attribute (the background
                                         plane): vertex ids read 0 (render_pass.cpp:573-581)     */
/* further material textures of RenderShader::setMaterial (render_shader.cpp:395-415), all RGBA8 in the
 * texel pool with their mip chains, sampled through the sampler of the asset file (see below)          */
#define SLHIP_DRAW_HAS_NORMAL_TEX    32u  /* tangent-space normal map (render_shader.frag:262-266)      */
#define SLHIP_DRAW_HAS_MR_TEX        64u  /* roughness in G, metallic in B (frag:284-288)               */
#define SLHIP_DRAW_HAS_OCCLUSION_TEX 128u /* R scales the image-based lighting term (frag:292-294,393)  */
#define SLHIP_DRAW_HAS_EMISSIVE_TEX  256u /* sRGB, multiplies the emissive factor (frag:296-298)        */
#define SLHIP_DRAW_HAS_STICKER       512u /* projected decal (frag:248-256, object.cpp:494-513)         */

/* Texture sampler state (one byte per texture of a draw).  Every 2D texture is stored with its full mip
 * chain: level l has max(1, w >> l) x max(1, h >> l) texels and follows level l-1 directly; a level is the
 * 2x2 box filter of the previous one, rounded to nearest (glGenerateMipmap leaves the filter to the
 * implementation).  Level of detail per OpenGL 4.5 section 8.14: rho = max(|d(u,v)/dx|, |d(u,v)/dy|) in
 * texels with forward differences of the perspective-correct coordinates to the pixel's +x / +y
 * neighbours, lambda = log2(rho); lambda <= 0 uses the magnification filter on level 0.               */
#define SLHIP_SAMPLER_WRAP_S(m)   ((m) & 3u)          /* 0 repeat, 1 clamp to edge, 2 mirrored repeat */
#define SLHIP_SAMPLER_WRAP_T(m)   (((m) >> 2) & 3u)
#define SLHIP_SAMPLER_MAG_LINEAR  0x10u               /* else nearest */
#define SLHIP_SAMPLER_MIN_LINEAR  0x20u               /* else nearest */
#define SLHIP_SAMPLER_MIP(m)      (((m) >> 6) & 3u)   /* 0 base level only, 1 nearest level, 2 linear between levels */
#define SLHIP_SAMPLER_DEFAULT     (SLHIP_SAMPLER_MAG_LINEAR | SLHIP_SAMPLER_MIN_LINEAR | (2u << 6))   /* repeat, trilinear */

/* One drawable (sub-mesh of an object, or the background plane) of one scene.
 * Carries what RenderShader::setTransformations / setMaterial / setClassIndex /
 * setInstanceIndex upload as uniforms (reference src/shaders/render_shader.cpp:233-265,
 * :326-417) -- normal matrices and camera position are derived on the host exactly there.   */
typedef struct {
    float mesh_to_object[16];
    float object_to_world[16];
    float normal_to_world[12];   /* 3x3 row-major, rows padded to 4 floats */
    float base_color[4];
    float emissive[4];
    float alpha_cutoff, metallic, roughness;
    uint32_t scene;              /* index of the owning scene in the batch            */
    uint32_t class_index, instance_index, flags;
    uint32_t n_verts;            /* vertices of the mesh this draw indexes into       */
    uint32_t vtx_base;           /* first vertex in the pool                          */
    uint32_t idx_base;           /* first index in the pool                           */
    uint32_t n_tris;
    uint32_t prim_base;          /* id of triangle 0 in the scene's draw order        */
    uint32_t tex_offset;         /* byte offset of the RGBA8 base-colour texture      */
    uint32_t tex_w, tex_h;
    uint32_t clip_base;          /* first entry of this draw in the clip-position scratch */
    uint32_t normal_tex_offset, normal_tex_w, normal_tex_h;
    uint32_t mr_tex_offset, mr_tex_w, mr_tex_h;
    uint32_t occlusion_tex_offset, occlusion_tex_w, occlusion_tex_h;
    uint32_t emissive_tex_offset, emissive_tex_w, emissive_tex_h;
    uint32_t sticker_tex_offset, sticker_tex_w, sticker_tex_h;   /* rectangle texture, clamp to edge, row 0 = top of the image */
    uint8_t  tex_sampler[8];     /* SLHIP_SAMPLER_* of the base, normal, metallic-roughness, occlusion, emissive
                                    texture (mesh.cpp:656-663: the file's filters and wrapping, mipmaps generated) */
    uint32_t _pad[3];
    float sticker_projection[16];  /* Object::stickerViewProjection, row-major (object.cpp:494-513)       */
    float sticker_range[4];        /* min.x, min.y, max(1e-6, size.x), max(1e-6, size.y) (render_shader.cpp:426-437) */
} slhip_draw;                    /* 432 bytes */

/* Per-scene camera + lights (reference Scene::setCameraIntrinsics src/scene.cpp:222-253,
 * RenderShader::setManualLighting render_shader.cpp:298-316).                               */
typedef struct {
    float proj[16];
    float world_to_cam[16];
    float cam_position[4];
    float light_dir[SLHIP_NUM_LIGHTS][4];    /* world frame; zero = inactive          */
    float light_color[SLHIP_NUM_LIGHTS][4];
    float shadow_mat[SLHIP_NUM_LIGHTS][16];  /* world -> light clip (render_pass.cpp:131-211) */
    float ambient[4];
    float manual_exposure;                   /* <0: auto exposure (tone_map_shader.frag:110) */
    uint32_t draw_begin, draw_end;           /* range in the draw array               */
    uint32_t n_prims;                        /* total triangles of the scene          */
    uint32_t light_map;                      /* 1 + index into pool->d_light_maps, 0 = none: IBL term
                                                (render_shader.frag:375-394) + sky background
                                                (render_pass.cpp:647-661)                          */
    uint32_t bg_tex[3];                      /* background image (Scene::setBackgroundImage, render_pass.cpp:637-646):
                                                byte offset in the texel pool, width, height; width 0 = none.
                                                A rectangle texture: one level, row 0 = top of the image     */
} slhip_scene;                               /* 480 bytes */

/* A unit of raster work: `count` consecutive triangles of one draw (<= SLHIP_CHUNK_TRIS).
 * Built on the host when the draw list is assembled so that every workgroup has a
 * wave-uniform scene and draw (matrices live in SGPRs).                                    */
#define SLHIP_CHUNK_TRIS 256
typedef struct {
    uint32_t scene, draw, first_tri, count;
} slhip_chunk;

/* output selection mask for slhip_render */
#define SLHIP_OUT_RGB        0x01u
#define SLHIP_OUT_COORD      0x02u   /* objectCoordinates xyz + camera z in w               */
#define SLHIP_OUT_CLASS      0x04u
#define SLHIP_OUT_INSTANCE   0x08u
#define SLHIP_OUT_NORMALS    0x10u
#define SLHIP_OUT_VERTEX_IDX 0x20u
#define SLHIP_OUT_BARY       0x40u
#define SLHIP_OUT_CAM_COORD  0x80u
#define SLHIP_OUT_ALL        0xFFu
#define SLHIP_OUT_GT6        0x1Fu   /* BASELINE "6-channel GT": rgb, coord+depth, class, instance, normals */

#define SLHIP_RENDER_SSAO     0x100u /* RenderPass::ssaoEnabled (render_pass.h:150)          */
#define SLHIP_RENDER_SHADOWS  0x200u /* shadow pass + PCF (render_pass.cpp:408-460)          */
#define SLHIP_RENDER_SHADOW_RESET 0x400u /* d_shadow / d_shadow_tiles hold garbage (first use of the buffers, or a
                                            previous call failed half way): clear all of it first.  Without the flag
                                            the call RELIES on the invariant it maintains: on entry and on exit every
                                            shadow texel is 1.0 and every tile bit 0 -- a render marks the 64x64-texel
                                            tiles its casters may touch and resets exactly those after shading, instead
                                            of clearing 16.8 MB per scene and light on every call                     */
#define SLHIP_RENDER_KEEP_HDR 0x800u /* with SLHIP_RENDER_SSAO: also store the float image the tone map consumes (AO applied) in the
                                        second half of scratch d_hdr -- the fused apply + tone-map pass otherwise never writes it */

/* Result buffers, batch-major [B][H][W][C] -- the 8 colour attachments of
 * RenderPass::Result (reference include/stillleben/render_pass.h:48-78, formats
 * src/render_pass.cpp:347-365).  A NULL pointer == output not wanted.                       */
typedef struct {
    uint8_t*  d_rgb;          /* u8  [B,H,W,4]   tone-mapped, linear (tone_map_shader.frag:129-130) */
    float*    d_coord;        /* f32 [B,H,W,4]   object xyz, camera z                        */
    uint16_t* d_class;        /* u16 [B,H,W]                                                 */
    uint16_t* d_instance;     /* u16 [B,H,W]                                                 */
    float*    d_normals;      /* f32 [B,H,W,4]   camera-frame normal, w = n.v                */
    uint32_t* d_vertex_idx;   /* u32 [B,H,W,4]   1-based ids, 4th = 0                        */
    float*    d_bary;         /* f32 [B,H,W,4]   4th = 0                                     */
    float*    d_cam_coord;    /* f32 [B,H,W,4]   camera xyz, 1                               */
} slhip_render_out;

/* Scratch the caller provides (sizes from slhip_render_scratch_bytes).                      */
typedef struct {
    uint64_t* d_vis;          /* u64 [B,H,W] visibility keys (depth24<<32 | prim id)         */
    float*    d_hdr;          /* f32 [B,H,W,4] HDR colour (ssaoRGBInput / postprocessInput)  */
    float*    d_ao;           /* f32 [B,H,W] occlusion + f32 [B,H+2,W+2] camera-z plane + per 8 x 8 tile a
                                 float2 record and a skip byte (SSAO; slhip_render_scratch_bytes sizes it) */
    float*    d_shadow;       /* f32 [B,NUM_LIGHTS,S,S] shadow depth (only active lights)    */
    uint32_t* d_queue;        /* large-triangle work queue: [0]=count, then (prim,tile) pairs */
    float*    d_lum;          /* f32 [B,4] HDR sums for auto exposure                        */
    float*    d_clip;         /* f32 [1 + NUM_LIGHTS][n_clip_verts][4]: clip positions written by the
                                 MFMA vertex-transform kernel (plane 0: camera, 1..3: lights)  */
    uint32_t* d_shadow_tiles; /* u32 [B][NUM_LIGHTS][ceil(ceil(S/64)^2 / 32)]: touched-tile bits (see
                                 SLHIP_RENDER_SHADOW_RESET)                                     */
    uint32_t  queue_capacity; /* number of (prim,tile) pairs that fit                        */
    uint32_t  shadow_res;     /* S (reference: 2048, render_pass.cpp:271)                    */
    uint32_t  n_clip_verts;   /* sum of n_verts over the draws of the batch                  */
    uint32_t  shadow_lights;  /* light maps per scene in d_shadow: 0 = SLHIP_NUM_LIGHTS; 1 or 2 = d_shadow is
                                 [B, shadow_lights, S, S] (lights beyond the count cast no shadow)  */
    float*    d_vattr;        /* n_clip_verts x 80 B, 64-byte aligned: the rest of the vertex stage
                                 (render_shader.vert:57-95) once per vertex -- [n_clip_verts] records of 64 B: (object xyz,
                                 camera z), (world xyz, camera x), (world normal, camera y), window coordinates (x, y in
                                 1/256 px as i32, depth, 1/w; x = INT_MIN behind the near plane) -- followed by the window
                                 coordinates once more as a dense [n_clip_verts] x 16 B plane.  (The light planes of d_clip
                                 end up holding window coordinates of the shadow map instead of clip positions.)        */
} slhip_render_scratch;

/* Renders a batch of scenes.  Replaces RenderPass::render (src/render_pass.cpp:303-796):
 * shadow pass, main G-buffer pass, SSAO, tone map.  d_depth_peel (f32 [B,H,W,4], the
 * previous result's objectCoordinates; NULL = none) implements `depthBufferResult`.         */
int slhip_render(const slhip_mesh_pool* pool,
                 const slhip_scene* d_scenes, const slhip_draw* d_draws,
                 const slhip_chunk* d_chunks, uint32_t n_scenes, uint32_t n_draws, uint32_t n_chunks,
                 uint32_t width, uint32_t height, uint32_t flags,
                 const float* d_depth_peel,
                 const slhip_render_out* out, const slhip_render_scratch* scratch,
                 void* stream);

/* Optional per-phase timing of slhip_render with HIP events recorded on the render stream
 * (used by bench.py for the roofline figures).  Phases: 0 shadow raster, 1 shadow large
 * triangles, 2 visibility raster, 3 large triangles, 4 deferred shade, 5 SSAO, 6 SSAO apply,
 * 7 tone map.  slhip_render_timings synchronises on the last render and fills ms_out[8].    */
int slhip_timing_enable(int on);
int slhip_render_timings(float* ms_out);
/* The SSAO pass runs its 64 taps only where something can occlude: on the open background plane (every texel a tap can reach
 * belongs to the plane or to the cleared background, no tap leaves the image) the occlusion is 1 exactly, and whole 8 x 8 tiles are
 * written without the loop (viewports that are multiples of 32 x 16; slhip_render.hip k_ssao_mask).  This read-out of the last
 * slhip_render on `scratch` (same n_scenes, width, height) returns counts[0] = tiles, counts[1] = tiles skipped; synchronises.  */
int slhip_render_ssao_skipped(const slhip_render_scratch* scratch, uint32_t n_scenes, uint32_t width, uint32_t height,
                              uint64_t counts[2], void* stream);

/* Bytes of each scratch buffer for a batch (host helper, no GPU needed).  `hdr` is sized for TWO float4 planes per scene: plane 0
 * = the fragment shader's linear colour (always written when rgb is asked for), plane 1 = the same after ambient occlusion, the
 * image the tone map consumes -- written ONLY under SLHIP_RENDER_KEEP_HDR (the fused blur + tone-map pass does not materialise it
 * otherwise; a caller that does not keep it may pass half the size).  Colour parity bar of the path (tests/test_gpu_render.py): the
 * float image within 1e-3 relative of the CPU restatement; the 8-bit rgb within 1 LSB on all but 1e-4 of the values, never more
 * than 2 (the tone map's divisions and the blur's exponentials go through the hardware's rcp / exp2).                          */
int slhip_render_scratch_bytes(uint32_t n_scenes, uint32_t width, uint32_t height,
                               uint32_t shadow_res, uint32_t queue_capacity,
                               uint64_t bytes_out[7]);   /* vis, hdr, ao, shadow, queue, lum, shadow_tiles */

/* ---------------------------------------------------------------------------------------------
 * Settle half (replaces PhysX as driven by Scene::simulateTableTopScene, scene.cpp:612-759)
 * ------------------------------------------------------------------------------------------- */

/* Convex collision shape: a vertex cloud in the OBJECT frame (mesh pretransform incl. scale
 * already applied -- the counterpart of PxConvexMeshGeometry + PxMeshScale + shape local pose,
 * reference src/object.cpp:173-204).  <= 64 vertices (VHACD.h:235).                          */
#define SLHIP_MAX_HULL_VERTS 64
typedef struct {
    uint32_t vtx_begin;      /* first vertex in the hull vertex pool (float4 each)            */
    uint32_t vtx_count;
    uint32_t _pad[2];
    float sphere[4];         /* bounding sphere: centre (object frame), radius                */
    float aabb_center[4];    /* object-frame AABB (broadphase: |R| * half is its world box)   */
    float aabb_half[4];
} slhip_hull;                /* 64 bytes */

#define SLHIP_BODY_STATIC   1u   /* Object::isStatic -> eKINEMATIC (object.cpp:515-520)       */
#define SLHIP_BODY_ASLEEP   2u
#define SLHIP_BODY_FROZEN   4u   /* out: held in place by the stabilisation this step (PxSceneFlag::eENABLE_STABILIZATION) */

/* Rigid body state.  `pose` is the object pose exactly as sl.Object.pose() returns it.       */
typedef struct {
    float pose[16];
    float lin_vel[4];        /* velocity of the centre of mass, world frame                   */
    float ang_vel[4];
    float com[4];            /* centre of mass, object frame                                  */
    float inv_inertia[12];   /* inverse inertia about the COM in object axes, 3 rows padded   */
    float inv_mass;          /* 0 for static bodies                                           */
    float mu_s, mu_d, restitution;    /* material (context.cpp:250-252 / object.cpp:565-605)  */
    float bsphere[4];        /* bounding sphere of all hulls (object frame centre, radius)    */
    float bbox_center[4];    /* mesh bbox centre (object frame); w = bbox diagonal / 2        */
    float max_lin_vel;       /* Object::setLinearVelocityLimit (object.cpp:259-264)           */
    float separation;        /* out: min contact separation of the last step (scene.cpp:73-116) */
    float wake_counter;
    uint32_t flags;
    uint32_t hull_begin, hull_end;
    int32_t stuck_counter;
    uint32_t drive_flags;    /* bit 0: linear spring drive enabled; bits 1..3: rotation about the
                                joint x/y/z axis locked (ManipulationSim, manipulation_sim.cpp:28-93) */
    float drive_target[4];   /* world position the object origin is driven to                 */
    float drive_frame[4];    /* joint frame orientation (quaternion x y z w) = initial pose   */
    float drive_params[4];   /* stiffness, damping, force limit (manipulation_sim.cpp:52-55), unused */
    float stab[4];           /* state of the stabilisation (slhip_settle_params.stabilization_threshold), 0 in a fresh record:
                                [0] seconds the body has spent below its threshold (PhysX: PXD_FREEZE_INTERVAL - PxsRigidBody::freezeCount [ext]),
                                [1] 1 - the share of gravity the body feels (PhysX: 1 - accelScale [ext]), [2], [3] unused      */
} slhip_body;                /* 304 bytes */

/* One scene of the settle batch: bodies [body_begin, body_end).                              */
typedef struct {
    uint32_t body_begin, body_end;
    uint32_t has_plane;      /* 1: static table box, top face at z = plane_z (scene.cpp:629-663) */
    float plane_z;
} slhip_settle_scene;

/* Constants of the step (defaults == the reference's call sites).   */
typedef struct {
    float dt;                    /* 0.01  = 1/25/4  (scene.cpp:681-684)                      */
    uint32_t substeps;           /* 4                                                         */
    uint32_t frames;             /* 100   (scene.cpp:720)                                     */
    uint32_t pos_iters, vel_iters; /* 4, 4 (object.cpp:209)                                  */
    float gravity[3];            /* 0,0,-9.81 (scene.cpp:157,665)                             */
    float contact_offset;        /* 0.004 = 0.02 * tolerance length 0.2 [ext]                 */
    float rest_offset;           /* 0.0015 (object.cpp:201); the plane box has 0              */
    float bounce_threshold;      /* 2.0   = 0.2 * tolerance speed 10 [ext]                    */
    float sleep_threshold;       /* 5e-3  = 5e-5 * speed^2 [ext]                              */
    float wake_time;             /* 0.4 s [ext]                                               */
    float angular_damping;       /* 0.05 [ext]                                                */
    float max_angular_velocity;  /* 100 rad/s [ext]                                           */
    float plane_mu_s, plane_mu_d, plane_restitution; /* 0.5, 0.5, 0 (scene.cpp:645)          */
    float redrop_z;              /* -0.5 (scene.cpp:746)                                      */
    float stuck_separation;      /* -0.01 (scene.cpp:748)                                     */
    int32_t stuck_frames;        /* 10 = 0.4 s * 25 FPS (scene.cpp:750)                       */
    uint32_t tabletop;           /* 1: run the redrop logic of simulateTableTopScene          */
    /* LDS sizing hints for the kernel (0 = worst case / hull vertices stay in global memory):
       maxima over the scenes of the batch.  A scene that exceeds a non-zero hint (or has more
       than 1024 hulls in one body) is left untouched by the launch -- the kernel never writes
       outside the layout the hints sized.                                                     */
    uint32_t max_bodies_per_scene;
    uint32_t max_hull_verts_per_scene;   /* sum over a scene's bodies of their hull vertices  */
    uint32_t max_hulls_per_scene;
    /* Capacities of a scene's per-step lists in the scratch (PhysX allocates as it goes, scene.cpp:738-739; here the caller sizes
       the scratch): candidate hull pairs the broadphase may file, contacts the solver may take.  0 = SLHIP_DEFAULT_HULL_PAIRS /
       SLHIP_DEFAULT_CONTACTS (on the 20-object YCB-like workload: p99.99 of the pairs 1 000, most ever seen 1 551; contacts 477).
       What a step offers beyond a capacity is dropped in list order AND COUNTED (slhip_settle_caps): a caller that finds a
       non-zero count settles again with larger capacities -- nothing is ever dropped silently.                                */
    uint32_t max_hull_pairs_per_scene;
    uint32_t max_contacts_per_scene;
    /* Compound manifold reduction (NOT in the reference: PhysX hands every convex pair's manifold to the solver).  0: the solver takes
       every point.  B >= 16: a body pair that touches through more than B hull pairs -- nested concave shapes: a mug in a bowl offers
       several hundred one-point manifolds, one Gauss-Seidel chain of that length -- keeps the B hull pairs with the deepest points
       (ties: list order); the others stay filed as manifolds (no impulse) and return when they are among the deepest.  Counted per
       (scene, step) by slhip_settle_caps.  On the 20-object workload a budget of 32 or 64 leaves the share of bodies at rest, the redrops
       and the deepest penetrations where they are without it.  Every host path of this repository passes 0.   */
    uint32_t pair_contact_budget;
    /* 0: the call starts from a cold contact state (it initialises the scratch).  N > 0: the call CONTINUES the N steps that earlier
       calls ran on the same d_scratch with the same scenes, bodies (same order, same hulls) and sizing hints -- the state PhysX
       keeps for the life of a PxScene (pair cache, persistent manifolds and their impulses, table contacts; scene.cpp:720-739,
       903-912, manipulation_sim.cpp:83-93) is taken from the scratch as the last call left it, d_bodies carries poses, velocities,
       wake counters and sleep flags.  k calls of one step give bit for bit what one call of k steps gives.                      */
    uint32_t resume;
    /* Capacity of a scene's list of touching BODY pairs (one solver group each, beside one group per body against the table).
       0: min(all body pairs, max_hull_pairs_per_scene, 12 x bodies + 64) -- enough for piles, where a body has a handful of
       neighbours.  A body pair beyond the capacity is dropped with its hull pairs AND COUNTED (slhip_settle_caps counts[8]):
       the caller settles again with a larger value, like for the other two lists.                                          */
    uint32_t max_body_pairs_per_scene;
    /* PxSceneFlag::eENABLE_STABILIZATION (scene.cpp:163).  The per-body stabilisation threshold, PhysX default 1e-5 * tolerance speed^2
       = 1e-3 with context.cpp:236-238's speed of 10 [ext].  A body in an island that rests on something static, whose frame energy
       (mass-normalised, from the velocities that moved it this step) is below min(10, touching body pairs) x threshold: both velocities
       are scaled by 1 - SLHIP_STAB_DAMPING dt, the share of gravity it feels goes to SLHIP_STAB_GRAVITY by a quarter of the
       distance per step (and back up by dt per step); after SLHIP_STAB_FREEZE_INTERVAL s of that, below
       SLHIP_STAB_FREEZE_TOLERANCE x threshold, it is frozen: the step's pose change is taken back (SLHIP_BODY_FROZEN).  0: off. */
    float stabilization_threshold;
    uint32_t _pad_params;
} slhip_settle_params;           /* 128 bytes */
#define SLHIP_STAB_DAMPING          0.5f   /* PXD_SLEEP_DAMPING [ext]   */
#define SLHIP_STAB_GRAVITY          0.9f   /* PXD_FREEZE_SCALE [ext]    */
#define SLHIP_STAB_FREEZE_INTERVAL  1.5f   /* PXD_FREEZE_INTERVAL [ext] */
#define SLHIP_STAB_FREEZE_TOLERANCE 0.25f  /* PXD_FREEZE_TOLERANCE [ext] */
#define SLHIP_STAB_MAX_INTERACTIONS 10

/* per-scene scratch (device), sized by slhip_settle_scratch_bytes */
#define SLHIP_MAX_BODIES     400  /* bodies per scene (the kernels keep a scene's working bodies in LDS: 152 B each -- and its body-
                                     pair groups: above ~300 bodies set max_body_pairs_per_scene so that both fit the 160 KB)       */
#define SLHIP_DEFAULT_HULL_PAIRS 2048 /* slhip_settle_params.max_hull_pairs_per_scene = 0 (at most 65535)                  */
#define SLHIP_DEFAULT_CONTACTS   1024 /* slhip_settle_params.max_contacts_per_scene = 0 (at most 65535)                    */
#define SLHIP_PAIR_CACHE_DENSE_HULLS 256 /* up to this many convex hulls per scene the pair cache (cached simplex + the way to the
                                          pair's persistent manifold) is a dense [hulls]^2 table, beyond it an open-addressing hash
                                          table keyed by the hull pair -- same contents, same results                         */

/* Steps every scene of the batch `frames * substeps` times without a host round trip -- a short sequence of kernel launches per
 * step over the whole batch (broadphase; GJK / portal refinement per hull pair; the face manifold of NEW contact pairs and of those that lost a point; persistent
 * manifolds, contact list and colouring; the warm-started 4 + 4 Gauss-Seidel sweeps, integration, sleeping), including the redrop
 * heuristic when params->tabletop.  Batches of up to 2048 scenes take ONE launch instead, in which a wave carries a scene through
 * all its steps (the same per-scene and per-pair functions: the results are the same bits; environment SLHIP_SETTLE_PERSISTENT=0/1
 * forces a form; the per-kernel timings below exist for the lockstep form only).  State that PhysX keeps from step to step lives in the scratch: the cached simplex, the
 * persistent contact manifold and its impulses per hull pair, the table contacts per body -- and stays valid after the call:
 * a later call with params->resume = (steps run so far) continues from it (Scene::simulate, ManipulationSim::step and the
 * frames of simulateTableTopScene with a visualisation callback step ONE long-lived PxScene in the reference).
 * d_bodies is updated in place (pose, velocities, separation).  Replaces the hot loop of
 * Scene::simulateTableTopScene (scene.cpp:720-756) and, with frames=substeps=1 and
 * tabletop=0, Scene::simulate(dt) (scene.cpp:903-912).                                       */
int slhip_settle(const slhip_settle_scene* d_scenes, uint32_t n_scenes,
                 slhip_body* d_bodies, const slhip_hull* d_hulls, const float* d_hull_verts,
                 const slhip_settle_params* params, void* d_scratch, uint64_t scratch_bytes,
                 void* stream);
/* Per-scene outcome of the last slhip_settle that used `d_scratch` (synchronises `stream`): h_status[i]
 * (may be NULL) = 0 when scene i was stepped, SLHIP_SETTLE_REFUSED_* when the kernel left it untouched
 * because it exceeds the sizing hints; *h_n_refused counts those.  Returns -2 (and sets the error
 * message) when any scene was refused -- a wrong hint must never pass silently.                       */
#define SLHIP_SETTLE_REFUSED_BODIES 1u   /* more bodies than max_bodies_per_scene (or than SLHIP_MAX_BODIES) */
#define SLHIP_SETTLE_REFUSED_HULLS  2u   /* more hulls than max_hulls_per_scene, or > 1024 in one body */
/* (below) What the capacities cost since the last cold start on `d_scratch` (synchronises `stream`; same `params` as those calls).
 * counts[0] (scene, step) pairs whose contacts went beyond what the scene's solver wave holds in LDS (swept from global
 * memory, nothing lost), [1] (scene, step) pairs in which contacts beyond max_contacts_per_scene were DROPPED, [2] (scene, step)
 * pairs in which hull pairs beyond max_hull_pairs_per_scene were DROPPED, [3] scenes with a non-zero [1] or [2], [4] scenes whose
 * contacts ever went beyond the LDS-resident part, [5] the most contacts and [6] the most hull pairs a step of any scene offered,
 * [7] (scene, step) pairs in which pair_contact_budget reduced some body pair's points, [8] (scene, step) pairs in which body
 * pairs beyond max_body_pairs_per_scene were DROPPED (such scenes count in [3] too), [9] the contacts the solver took, summed over
 * all (scene, step) pairs (bench.py's byte model: contacts per scene-step).
 * The reference has no caps (scene.cpp:738-739): [1] = [2] = [8] = 0 is the contract, a caller that sees otherwise re-sizes.     */
int slhip_settle_caps(const void* d_scratch, uint32_t n_scenes, const slhip_settle_params* params,
                      uint64_t counts[10], void* stream);
int slhip_settle_status(const void* d_scratch, uint32_t n_scenes, uint32_t* h_status, uint32_t* h_n_refused,
                        void* stream);
/* Optional live timing of the phases of a lockstep step (bench.py's roofline leg): HIP events on the launch's stream
 * around every phase of every 8th step.  slhip_settle_timings synchronises the recorded events and returns, since the last
 * call, the average duration [ms] and the number of timed launches of 0 k_w_begin (integrate, table contacts, broadphase),
 * 1 k_w_gjk_first + k_w_gjk_rest (the two passes of the main GJK), 2 k_w_manifold (face manifolds), 3 k_w_finish (contact list, groups, prep,
 * colouring, cost class), 4 k_w_solve.                                                                                  */
int slhip_settle_timing_enable(int on);
int slhip_settle_timings(float avg_ms_out[5], uint32_t launches_out[5]);
/* The same records step by step (profiles/rNN/solve_by_step.csv): slhip_settle_timing_every(k) times every k-th step (default 8);
 * slhip_settle_timings_by_step synchronises the recorded events and writes, for up to `capacity` timed steps since the last read-out,
 * ms_out[5 * i + kernel] and steps_out[i] (may be NULL) = the step's index within its settle call; *n_out = rows written.          */
int slhip_settle_timing_every(uint32_t every);
int slhip_settle_timings_by_step(float* ms_out, uint32_t* steps_out, uint32_t capacity, uint32_t* n_out);
/* LDS bytes a solver wave (k_w_solve) of the last slhip_settle call was launched with: 20 KB, more when the batch's largest scene
 * shape needs it, SLHIP_SOLVE_LDS_KB overrides (measurement read-out: bench.py states the configuration it measured).             */
int slhip_settle_solver_wave_lds(void);
/* scratch for n_scenes scenes: accumulators + the per-scene pair cache, sized from the hints in
 * `params` (NULL or zero hints: the worst case)                                                      */
int slhip_settle_scratch_bytes(uint32_t n_scenes, const slhip_settle_params* params, uint64_t* bytes_out);

/* Boolean any-overlap query per body against all OTHER bodies of its scene (and the plane if
 * present): d_flags[body] = 1 if it collides.  Replaces Scene::isObjectColliding /
 * checkCollisions (scene.cpp:355-385, :914-925).                                             */
int slhip_overlap_any(const slhip_settle_scene* d_scenes, uint32_t n_scenes,
                      const slhip_body* d_bodies, const slhip_hull* d_hulls,
                      const float* d_hull_verts, uint8_t* d_flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * sl.diff half (replaces python/src/diff.cu + bridge_diff.cpp and fuses diff.py:355-523)
 * ------------------------------------------------------------------------------------------- */

/* generate_sobel_valid_mask (bridge_diff.cpp:13-69, CPU-loop semantics: the 1-px border stays
 * valid).  d_inst i16[H,W]; d_depth f32 with `depth_stride` floats between pixels (1 for a
 * dense [H,W] image, 4 to read channel 3 of the coordinate target in place); d_valid u8[H,W].  */
int slhip_diff_sobel_valid(const int16_t* d_inst, const float* d_depth, int depth_stride, int H, int W,
                           uint8_t* d_valid, void* stream);

/* dilate_object_mask (bridge_diff.cpp:71-157).  d_coords f32 xyz with `coord_stride` floats
 * between pixels (3 or 4); outputs u8[H,W] and f32[H,W,3]; the 1-px border reads 0.            */
int slhip_diff_dilate(const uint8_t* d_mask, const uint8_t* d_valid, const float* d_coords, int coord_stride,
                      int H, int W, uint8_t* d_out_mask, float* d_out_coords, void* stream);

/* compute_image_space_gradients (diff.py:73-127): d_rgb u8[H,W,4] -> grad_x, grad_y f32[3,H,W],
 * zero where !valid.                                                                           */
int slhip_diff_image_gradients(const uint8_t* d_rgb, const uint8_t* d_valid, int H, int W, float* d_grad_x,
                               float* d_grad_y, void* stream);

/* backpropagate_gradient_to_poses (diff.py:355-523), fused.  d_coord f32[H,W,4] (object xyz,
 * depth), d_inst i16[H,W], d_grad_img f32[3,H,W], h_proj HOST float[16] (row-major projection),
 * d_poses f32[n_obj,16], d_obj_inst i32[n_obj]; scratch d_valid u8[H,W], d_acc f64[n_obj*6];
 * d_out f32[n_obj,6].                                                                          */
int slhip_diff_pose_backward(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                             const float* d_grad_img, const float* h_proj, const float* d_poses,
                             const int32_t* d_obj_inst, int n_obj, int H, int W, uint8_t* d_valid,
                             double* d_acc, float* d_out, void* stream);
/* The same for K pose hypotheses of one scene in ONE launch sequence (BASELINE config C5): d_rgb / d_coord / d_inst are the
 * [K,H,W,..] targets of the hypotheses' renders, d_poses f32 [K,n_obj,16], d_grad_img one image for all (grad_stride_floats
 * = 0) or one per hypothesis (= 3 * H * W); scratch d_valid u8 [K,H,W], d_acc f64 [K,n_obj,6]; d_out f32 [K,n_obj,6].
 * The reference loops hypothesis by hypothesis through its renderer and python/stillleben/diff.py:355-523.       */
int slhip_diff_pose_backward_batch(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                                   const float* d_grad_img, uint64_t grad_stride_floats, const float* h_proj,
                                   const float* d_poses, const int32_t* d_obj_inst, int n_obj, int n_hyp, int H, int W,
                                   uint8_t* d_valid, double* d_acc, float* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image-based lighting precompute (replaces LightMap::load's GL passes, light_map.cpp:360-606, and
 * src/shaders/cubemap_shader_{equirectangular,irradiance,prefilter}.frag, brdf_shader.frag)
 * ------------------------------------------------------------------------------------------- */
/* floats needed for the four buffers of a light map with the given sizes: out[0..3] = env,
 * irradiance, prefilter, BRDF LUT                                                              */
int slhip_light_map_floats(uint32_t env_size, uint32_t env_levels, uint32_t irr_size, uint32_t pre_size,
                           uint32_t pre_levels, uint32_t lut_size, uint64_t out[4]);
/* d_equirect: f32 [H][W][3] equirectangular radiance, row 0 = top (+z up, azimuth atan2(y, x) along
 * the row); fills every buffer of *lm (a HOST struct holding DEVICE pointers).                   */
int slhip_light_map_build(const float* d_equirect, int H, int W, const slhip_light_map* lm, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Camera model ('next' row f4 of SURVEY.md 8f): the step after the render in every data-generation
 * loop.  Replaces python/stillleben/camera_model.py:222-263 (process_deterministic): chromatic
 * aberration (:47-74, affine_grid + bilinear grid_sample, reflection padding) -> 5x5 Gaussian blur
 * (:77-118, zero padding) -> re-exposure (:120-130) -> Poissonian-Gaussian noise (:132-163) ->
 * clamp -> hue jitter (:165-220) -> 5x5 post blur (sigma 0.4) -> clamp, fused into two kernels.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float translation[6];   /* (tx, ty) of R, G, B in normalised [-1,1] image coordinates          */
    float scaling[3];       /* scale of R, G, B                                                    */
    float blur_kernel[25];  /* 5x5 weights of the first blur, row-major (camera_model.py:77-104);
                               used when blur_enabled                                             */
    float post_kernel[25];  /* 5x5 weights of the post blur (sigma = 0.4)                          */
    float exposure_gain;    /* exp(deltaS), rounded to f32 (camera_model.py:130)                   */
    float noise_a, noise_b; /* signal-dependent variance factor, signal-independent std            */
    float hue_shift;        /* -0.5 .. 0.5                                                         */
    uint32_t blur_enabled;  /* blur_sigma > 0                                                      */
    uint32_t noise_enabled; /* do_noise                                                            */
    uint32_t seed_lo, seed_hi; /* counter-based RNG key of this image (noise stage)               */
} slhip_camera_params;      /* 268 bytes */

/* d_in / d_out: f32 [n_images, 3, H, W] (the reference's CHW layout, values in [0,1]); d_tmp:
 * scratch of the same size; d_params: DEVICE array of n_images records.  d_out may alias d_in.   */
int slhip_camera_model(const float* d_in, float* d_out, float* d_tmp, uint32_t n_images, int H, int W,
                       const slhip_camera_params* d_params, void* stream);

/* bp_to_vertices_and_colors (diff.py:215-352, row D6), dense form: for every pixel that belongs to one
 * of the n_obj objects, the negated gradient of the objective w.r.t. the three vertices of its triangle
 * (-bary_k * dL/dX, X = object coordinates of the pixel) and w.r.t. their colours (-bary_k * dL/dI).
 * d_bary f32[H,W,4] (the barycentric target); outputs f32[H,W,3,3], zero where no object matches; the
 * caller selects the rows of one object's pixels (row-major, as the reference's boolean indexing does)
 * and pairs them with the vertex-index target.  d_valid u8[H,W] is scratch.                       */
int slhip_diff_vertex_backward(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                               const float* d_bary, const float* d_grad_img, const float* h_proj,
                               const float* d_poses, const int32_t* d_obj_inst, int n_obj, int H, int W,
                               uint8_t* d_valid, float* d_grad_vertices, float* d_grad_colors, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Library
 * ------------------------------------------------------------------------------------------- */
int slhip_abi_version(void);
const char* slhip_last_error(void);
/* hipSetDevice + sanity check that the device is gfx950.  Replaces Context::CreateCUDA's
 * device selection (reference src/context.cpp:411-560).                                      */
int slhip_device_init(int device_index);

/* Streams confined to a range of compute units of the current device (additive: the reference has
 * one GL context per process and nothing to partition).  A data-generation loop runs the settle of
 * batch k+1 and the render of batch k at the same time; the two halves have opposite resource
 * shapes (settle: 256 VGPRs + 20 KiB LDS per single-wave workgroup held for ~100 ms; render: short
 * 256-thread workgroups), and sharing CUs between them fragments both.  Giving each half its own CU
 * range removes the interference.  Consecutive CUs of the mask order are dealt round-robin over
 * the XCDs, so a contiguous range takes the same share of every XCD.  `*stream_out` is a
 * hipStream_t usable as the `stream` argument of every call above.                              */
int slhip_stream_create_cu_range(uint32_t first_cu, uint32_t n_cus, void** stream_out);
int slhip_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scene synthesis on the device: the host work either side of the settle, moved into HBM.
 *
 * In the reference every scene is built by host C++ behind pybind11: the tabletop set-up of
 * Scene::simulateTableTopScene (src/scene.cpp:612-678: plane yaw, the stack of randomly oriented
 * objects), after the settle Scene::chooseRandomCameraPose (scene.cpp:472-610) and
 * chooseRandomLightDirection (scene.cpp:453-470), and per render the shadow matrices
 * (render_pass.cpp:69-211) and the per-drawable uniforms (render_pass.cpp:534-621,
 * render_shader.cpp:233-265).  For a batch that is ~4 ms of host time per scene against 0.15 ms of
 * device time, so a batch is described ONCE by an asset table (one record per sl.Mesh in use) and two
 * kernels write the records the settle and render kernels consume -- nothing crosses PCIe per scene.
 *
 * Randomness: Philox4x32-10 keyed by (seed_lo, seed_hi), counter (scene id, stream, index,
 * 0x51DE5EED); uniform = ((x >> 8) + 0.5) * 2^-24; normals by Box-Muller on deterministic log /
 * sin / cos polynomials (the DISTRIBUTIONS of the reference are the contract, its libstdc++ streams are
 * not reproducible: it seeds from std::random_device, scene.cpp:147-148).
 * ------------------------------------------------------------------------------------------- */

/* One mesh class: what sl.Mesh (+ the defaults of sl.Object) contributes to a scene. */
typedef struct {
    float mesh_to_object[16];   /* Mesh::pretransform, row-major                                     */
    float bbox_min[4], bbox_max[4]; /* Mesh::bbox() incl. quirk q6 (mesh.cpp:1075-1081), object frame  */
    float com[4];               /* centre of mass (object frame) at the default density               */
    float inv_inertia[12];      /* inverse inertia about the COM, object axes, rows padded to 4       */
    float mass;                 /* density 1000 (object.h:287) x hull volumes                         */
    float mu_s, mu_d, restitution;  /* default material 0.3 / 0.2 / 0.1 (context.cpp:250-252)       */
    float bsphere[4];           /* bounding sphere of all hulls                                       */
    uint32_t hull_begin, hull_end;  /* range in the hull table handed to slhip_settle                */
    uint32_t draw_begin, draw_count; /* sub-mesh draw templates of this class in d_templates          */
    uint32_t n_verts;           /* vertices of the mesh (every draw of the class indexes all of them) */
    uint32_t n_chunks;          /* sum over the class's draws of ceil(n_tris / SLHIP_CHUNK_TRIS)      */
    uint32_t _pad[2];
} slhip_asset;                  /* 224 bytes */

#define SLHIP_SYNTH_SAMPLE_DISTINCT 1u  /* draw each scene's n_objects classes without replacement
                                           (examples/ycb.py:60: random.sample(meshes, 20)); needs
                                           n_objects <= n_assets <= SLHIP_SYNTH_MAX_ASSETS; otherwise
                                           d_asset_ids names every object's class                     */
#define SLHIP_SYNTH_RANDOM_PBR      2u  /* metallic, roughness ~ U(0,1) per object (examples/ycb.py:63-64:
                                           obj.metallic / obj.roughness); otherwise the file's values   */
#define SLHIP_SYNTH_SHADOWS         4u  /* fill shadow_mat of the active light (render_pass.cpp:131-211) */
#define SLHIP_SYNTH_MAX_ASSETS   1024u
#define SLHIP_SYNTH_MAX_OBJECTS    64u  /* == SLHIP_MAX_BODIES */

typedef struct {
    uint32_t n_scenes, n_objects, n_assets, flags;
    uint32_t seed_lo, seed_hi;
    uint32_t scene_id_base;     /* global id of scene 0: shards and steps draw disjoint random streams  */
    uint32_t render_chunk;      /* scenes per slhip_render call: `scene`, `draw`, draw_begin/end and
                                   clip_base in the written records are relative to the first scene of
                                   the call's chunk (scene s belongs to chunk s / render_chunk)         */
    uint32_t max_draws_per_scene;      /* record strides: scene s owns draws [s*max_draws, ...), chunks */
    uint32_t max_chunks_per_scene;     /* [s*max_chunks, ...) and clip vertices [s*max_clip, ...)        */
    uint32_t max_clip_verts_per_scene; /* (chunk-relative); unused slots are written with zero counts   */
    float plane_z;              /* top of the table box: BOX_HALF_EXTENTS.z = 0.04 (scene.cpp:638)       */
    float proj[16];             /* Scene::projectionMatrix, row-major (scene.cpp:222-253)                */
    float proj_inv[16];         /* its inverse (render_pass.cpp:73)                                      */
    float plane_size[2];        /* Scene::backgroundPlaneSize; 0,0 = no plane drawn                      */
    float manual_exposure;
    float _pad0;
    float light_color[4];       /* light 0; its direction is drawn per scene (scene.cpp:453-470)         */
    float ambient[4];
} slhip_synth_params;           /* 224 bytes */

typedef struct {
    uint32_t asset;             /* class of the object                                                  */
    uint32_t instance_index;    /* 1 + position in the scene (scene.cpp:285-287)                         */
    float metallic, roughness;  /* per-object override, < 0 = the file's value (render_shader.cpp:355-377) */
} slhip_synth_object;

typedef struct {
    float plane_pose[16];       /* Scene::backgroundPlanePose set by the tabletop set-up (scene.cpp:650-657) */
    float camera_pose[16];      /* out of slhip_synth_place: Scene::cameraPose                            */
} slhip_synth_scene;

/* Tabletop set-up of every scene of the batch (scene.cpp:612-678): d_bodies [n_scenes * n_objects],
 * d_settle_scenes [n_scenes] (has_plane = 1), d_objects [n_scenes * n_objects], d_scenes [n_scenes].
 * d_asset_ids: u16 [n_scenes * n_objects] or NULL with SLHIP_SYNTH_SAMPLE_DISTINCT.                   */
int slhip_synth_stage(const slhip_synth_params* params, const slhip_asset* d_assets, const uint16_t* d_asset_ids,
                      slhip_body* d_bodies, slhip_settle_scene* d_settle_scenes, slhip_synth_object* d_objects,
                      slhip_synth_scene* d_scenes, void* stream);

/* After the settle: camera pose, light direction, shadow matrix, and the slhip_scene / slhip_draw /
 * slhip_chunk records of the batch (strides from `params`), ready for slhip_render chunk by chunk.
 * d_templates: the sub-mesh draws of every class with all static fields filled (materials, textures,
 * vertex / index ranges, class index); transforms, ids and bases are written here.                    */
int slhip_synth_place(const slhip_synth_params* params, const slhip_asset* d_assets, const slhip_draw* d_templates,
                      const slhip_body* d_bodies, const slhip_synth_object* d_objects, slhip_synth_scene* d_scenes,
                      slhip_scene* d_out_scenes, slhip_draw* d_out_draws, slhip_chunk* d_out_chunks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Host side of the per-object API (sl.Scene / sl.RenderPass): record assembly in C++, one call per batch.
 * The reference does this work in C++ as well -- renderer.render(scene) (python/src/py_render_pass.cpp:252-258) runs
 * RenderPass::render (src/render_pass.cpp:303-796), which computes the shadow matrices (:69-211) and uploads the
 * per-drawable uniforms (:534-621; RenderShader::setTransformations / setMaterial, render_shader.cpp:233-265, :326-417).
 * Inputs are flat HOST arrays; nothing here touches the device.
 * ------------------------------------------------------------------------------------------- */
/* An object of a scene: pose + what sl.Object overrides of its mesh's draw templates.        */
typedef struct {
    float pose[16];              /* object to world, row-major (Object::pose)                  */
    float bbox_center[4];        /* mesh bbox centre (object frame); w = bbox diagonal / 2 (shadow fit, render_pass.cpp:87,180) */
    float color[4];              /* Object options "color" ...                                 */
    uint32_t force_color;        /* ... used instead of the material's base colour when set ("force_color") */
    uint32_t tmpl_begin, tmpl_count;   /* the mesh's draw templates (one per sub-mesh) in `templates` */
    uint32_t instance_index;
    float metallic, roughness;   /* per-object override, < 0: the material's (render_shader.cpp:355-377) */
    uint32_t casts_shadows;
    uint32_t _pad;
} slhip_host_object;             /* 128 bytes */

/* A scene: camera, lights, background plane, its objects [obj_begin, obj_end).               */
typedef struct {
    float proj[16], proj_inv[16];    /* projection (Scene::projectionMatrix) and its inverse (float64 inverse rounded) */
    float camera_pose[16];           /* camera to world                                        */
    float light_dir[SLHIP_NUM_LIGHTS][4], light_color[SLHIP_NUM_LIGHTS][4];
    float ambient[4];
    float plane_pose[16];            /* background plane pose (scene.cpp:629-663)              */
    float plane_size[2];
    int32_t plane_template;          /* draw template of the background plane in `templates`, -1: no plane */
    float manual_exposure;
    uint32_t obj_begin, obj_end;
    uint32_t light_map;
    uint32_t bg_tex[3];
} slhip_host_scene;                  /* 408 bytes */

/* Shadow-map matrices of one scene (computeFrustumCorners + computeShadowMapMatrix, render_pass.cpp:69-211):
 * out48 = NUM_LIGHTS row-major 4x4 (identity for inactive lights or a non-finite fit).      */
int slhip_host_shadow_matrices(const slhip_host_scene* scene, const slhip_host_object* objects, float* out48);
/* Matrix4::normalMatrix(): inverse transpose of the upper 3x3 of m16 (float64 cofactors), 3 rows padded to 4. */
int slhip_host_normal_matrix(const float* m16, float* out12);
/* Records a batch needs: draws (plane + one per object sub-mesh) and raster chunks (<= SLHIP_CHUNK_TRIS triangles each). */
int slhip_records_count(const slhip_host_scene* scenes, uint32_t n_scenes, const slhip_host_object* objects,
                        const slhip_draw* templates, uint32_t* n_draws, uint32_t* n_chunks);
/* Fills srec[n_scenes], drec[<= draw_capacity], crec[<= chunk_capacity] in scene / object / sub-mesh order: the draw
 * templates with scene, prim_base, clip_base, object_to_world, normal_to_world, the per-object overrides; the scene
 * records with world_to_cam, cam_position, lights and (with_shadows) the shadow matrices.   */
int slhip_records_build_render(const slhip_host_scene* scenes, uint32_t n_scenes, const slhip_host_object* objects,
                               const slhip_draw* templates, uint32_t with_shadows, slhip_scene* srec, slhip_draw* drec,
                               uint32_t draw_capacity, slhip_chunk* crec, uint32_t chunk_capacity);

/* Host geometry of the collision-shape stage (Mesh::loadPhysics, src/mesh.cpp:335-470, where the reference calls its
 * vendored V-HACD and PhysX's convex cooking; csrc/slhip_hull.cpp).
 * slhip_host_convex_hull: quick-hull of n points (xyz, double): up to tri_capacity index triples into `points` (a hull of n points has
 * at most 2 n - 4 triangles), outward oriented.  Returns 0; 1 when the points span no volume (*n_tris_out = 0); -1 on error.
 * slhip_host_fill_holes: solid fill of a voxel grid [nx][ny][nz] (bytes, C order, in place): every empty cell that cannot be reached
 * from the border through empty face neighbours becomes 1 -- V-HACD's inside / outside classification.                           */
int slhip_host_convex_hull(const double* points, uint32_t n, uint32_t* tris_out, uint32_t tri_capacity, uint32_t* n_tris_out);
int slhip_host_fill_holes(uint8_t* grid, uint32_t nx, uint32_t ny, uint32_t nz);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU exchange: scenes are independent, every rank (one process per GPU, as
 * the reference runs it: python/src/py_context.cpp:34-52) settles and renders its own shard; the one
 * exchange step is the all-gather of rendered batches, RCCL over xGMI.  RCCL is bound at run time
 * (the copy the process already maps -- PyTorch's -- else librccl.so.1 of the ROCm install).
 * ------------------------------------------------------------------------------------------- */
#define SLHIP_COMM_ID_BYTES 128   /* == NCCL_UNIQUE_ID_BYTES */
typedef struct slhip_comm slhip_comm;
/* Rank 0 draws an id and hands it to the other ranks out of band (the host layer uses the
 * torch.distributed store; a file or MPI works as well).                                            */
int slhip_comm_unique_id(uint8_t id_out[SLHIP_COMM_ID_BYTES]);
/* Collective over all ranks; binds the communicator to the calling thread's current HIP device.     */
int slhip_comm_create(const uint8_t id[SLHIP_COMM_ID_BYTES], int n_ranks, int rank, slhip_comm** comm_out);
int slhip_comm_destroy(slhip_comm* comm);
int slhip_comm_info(const slhip_comm* comm, int* n_ranks, int* rank);
/* d_recv (n_ranks * bytes) receives every rank's d_send (bytes) in rank order; asynchronous on `stream`. */
int slhip_allgather(slhip_comm* comm, const void* d_send, void* d_recv, uint64_t bytes, void* stream);
/* The buffers of one rendered chunk (rgb, coord, class, instance, normals) in ONE fused RCCL group:
 * d_recv[i] (n_ranks * bytes[i]) <- all ranks' d_send[i].                                           */
int slhip_allgather_group(slhip_comm* comm, uint32_t n_buffers, const void* const* d_send, void* const* d_recv,
                          const uint64_t* bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLHIP_H */
