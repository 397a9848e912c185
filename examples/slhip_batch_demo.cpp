// slhip_batch_demo.cpp -- the hot path driven from plain C++ through the C-ABI of libslhip.so (include/slhip.h):
// no Python, no torch.  This is what the reference's host C++ (Scene::simulateTableTopScene + RenderPass::render behind
// pybind11, /root/reference/src/scene.cpp:612-759, src/render_pass.cpp:303-796) would call for a batch of scenes:
//
//   slhip_synth_stage -> slhip_settle -> slhip_synth_place -> slhip_render
//
// Input: an asset blob written by tools/export_assets.py (mesh pool, hull table, asset table, draw templates, the batch's
// slhip_synth_params and slhip_settle_params).  Output: the settled body records, the camera poses and the instance masks
// + depth of every scene (raw binary), which tests/test_gpu_cabi_demo.py compares bit for bit with the Python host path.
//
//   slhip_batch_demo <assets.bin> <out.bin> [width height]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "slhip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(3); } } while (0)
#define SL_OK(x) do { int s_ = (x); if (s_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, s_, slhip_last_error()); std::exit(4); } } while (0)

struct Section { std::vector<char> bytes; };

static std::vector<Section> read_blob(const char* path)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    char magic[8];
    uint32_t n = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "SLASSET1", 8) != 0 || std::fread(&n, 4, 1, f) != 1) {
        std::fprintf(stderr, "%s: not an asset blob\n", path);
        std::exit(2);
    }
    std::vector<Section> out(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t sz = 0;
        if (std::fread(&sz, 8, 1, f) != 1) std::exit(2);
        out[i].bytes.resize(sz);
        if (sz && std::fread(out[i].bytes.data(), 1, sz, f) != sz) std::exit(2);
    }
    std::fclose(f);
    return out;
}

template <class T>
static T* upload(const Section& s, size_t min_bytes = 16)
{
    void* d = nullptr;
    const size_t n = s.bytes.size() > min_bytes ? s.bytes.size() : min_bytes;
    HIP_OK(hipMalloc(&d, n));
    HIP_OK(hipMemset(d, 0, n));
    if (!s.bytes.empty()) HIP_OK(hipMemcpy(d, s.bytes.data(), s.bytes.size(), hipMemcpyHostToDevice));
    return static_cast<T*>(d);
}

template <class T>
static T* dalloc(size_t count)
{
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, count * sizeof(T) > 16 ? count * sizeof(T) : 16));
    return static_cast<T*>(d);
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s assets.bin out.bin [width height]\n", argv[0]);
        return 2;
    }
    const uint32_t W = argc > 4 ? (uint32_t)std::atoi(argv[3]) : 320u, H = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 240u;
    // sections, in the order tools/export_assets.py writes them
    enum { POS, NRM, UV, COL, TAN, IDX, TEX, HULLS, HULL_VERTS, ASSETS, TEMPLATES, SYNTH, SETTLE, N_SECTIONS };
    const std::vector<Section> S = read_blob(argv[1]);
    if (S.size() != N_SECTIONS) { std::fprintf(stderr, "asset blob: %zu sections, expected %d\n", S.size(), (int)N_SECTIONS); return 2; }
    if (S[SYNTH].bytes.size() != sizeof(slhip_synth_params) || S[SETTLE].bytes.size() != sizeof(slhip_settle_params)) {
        std::fprintf(stderr, "asset blob: parameter records have the wrong size\n");
        return 2;
    }
    slhip_synth_params sp;
    slhip_settle_params prm;
    std::memcpy(&sp, S[SYNTH].bytes.data(), sizeof(sp));
    std::memcpy(&prm, S[SETTLE].bytes.data(), sizeof(prm));
    const uint32_t n = sp.n_scenes, nb = sp.n_scenes * sp.n_objects;

    SL_OK(slhip_device_init(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    slhip_mesh_pool pool;
    std::memset(&pool, 0, sizeof(pool));
    pool.d_pos = upload<float>(S[POS]); pool.d_nrm = upload<float>(S[NRM]); pool.d_uv = upload<float>(S[UV]);
    pool.d_col = upload<float>(S[COL]); pool.d_tan = upload<float>(S[TAN]); pool.d_idx = upload<uint32_t>(S[IDX]);
    pool.d_tex = upload<uint8_t>(S[TEX]);
    pool.n_vertices = S[POS].bytes.size() / 16; pool.n_indices = S[IDX].bytes.size() / 4; pool.n_tex_bytes = S[TEX].bytes.size();
    slhip_hull* d_hulls = upload<slhip_hull>(S[HULLS]);
    float* d_hull_verts = upload<float>(S[HULL_VERTS]);
    slhip_asset* d_assets = upload<slhip_asset>(S[ASSETS]);
    slhip_draw* d_templates = upload<slhip_draw>(S[TEMPLATES]);

    slhip_body* d_bodies = dalloc<slhip_body>(nb);
    slhip_settle_scene* d_sscenes = dalloc<slhip_settle_scene>(n);
    slhip_synth_object* d_objects = dalloc<slhip_synth_object>(nb);
    slhip_synth_scene* d_scenes = dalloc<slhip_synth_scene>(n);
    slhip_scene* d_srec = dalloc<slhip_scene>(n);
    slhip_draw* d_drec = dalloc<slhip_draw>((size_t)n * sp.max_draws_per_scene);
    slhip_chunk* d_crec = dalloc<slhip_chunk>((size_t)n * sp.max_chunks_per_scene);

    // 1. tabletop set-up, 2. settle, 3. camera / light / draw records -- all on the device, nothing read back in between
    SL_OK(slhip_synth_stage(&sp, d_assets, nullptr, d_bodies, d_sscenes, d_objects, d_scenes, stream));
    uint64_t settle_bytes = 0;
    SL_OK(slhip_settle_scratch_bytes(n, &prm, &settle_bytes));
    void* d_settle_scratch = nullptr;
    HIP_OK(hipMalloc(&d_settle_scratch, settle_bytes));
    SL_OK(slhip_settle(d_sscenes, n, d_bodies, d_hulls, d_hull_verts, &prm, d_settle_scratch, settle_bytes, stream));
    uint32_t refused = 0;
    SL_OK(slhip_settle_status(d_settle_scratch, n, nullptr, &refused, stream));
    SL_OK(slhip_synth_place(&sp, d_assets, d_templates, d_bodies, d_objects, d_scenes, d_srec, d_drec, d_crec, stream));

    // 4. render: instance mask + object coordinates / depth (rgb too, so that the shadow pass, SSAO and the tone map run)
    const uint32_t S_RES = 2048, QCAP = 1u << 20;
    uint64_t sz[7];
    SL_OK(slhip_render_scratch_bytes(n, W, H, S_RES, QCAP, sz));
    slhip_render_scratch scr = {};
    std::memset(&scr, 0, sizeof(scr));
    void* p[7];
    for (int i = 0; i < 7; ++i) HIP_OK(hipMalloc(&p[i], sz[i] > 16 ? sz[i] : 16));
    scr.d_vis = (uint64_t*)p[0]; scr.d_hdr = (float*)p[1]; scr.d_ao = (float*)p[2]; scr.d_shadow = (float*)p[3];
    scr.d_queue = (uint32_t*)p[4]; scr.d_lum = (float*)p[5]; scr.d_shadow_tiles = (uint32_t*)p[6];
    scr.queue_capacity = QCAP; scr.shadow_res = S_RES;
    scr.n_clip_verts = n * sp.max_clip_verts_per_scene;
    HIP_OK(hipMalloc((void**)&scr.d_clip, (size_t)scr.n_clip_verts * 9 * 16));   // 4 clip planes + 80 B per vertex of vertex cache
    scr.d_vattr = scr.d_clip + (size_t)scr.n_clip_verts * 4 * 4;
    const size_t P = (size_t)W * H;
    slhip_render_out out;
    std::memset(&out, 0, sizeof(out));
    out.d_rgb = dalloc<uint8_t>(n * P * 4); out.d_coord = dalloc<float>(n * P * 4);
    out.d_class = dalloc<uint16_t>(n * P); out.d_instance = dalloc<uint16_t>(n * P);
    out.d_normals = dalloc<float>(n * P * 4); out.d_cam_coord = dalloc<float>(n * P * 4);
    const uint32_t flags = SLHIP_OUT_GT6 | SLHIP_OUT_CAM_COORD | SLHIP_RENDER_SSAO | SLHIP_RENDER_SHADOWS | SLHIP_RENDER_SHADOW_RESET;
    // one call renders the whole batch here (render_chunk == n_scenes in the exported parameters)
    SL_OK(slhip_render(&pool, d_srec, d_drec, d_crec, n, n * sp.max_draws_per_scene, n * sp.max_chunks_per_scene, W, H, flags,
                       nullptr, &out, &scr, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<slhip_body> bodies(nb);
    std::vector<slhip_synth_scene> scenes(n);
    std::vector<uint16_t> inst(n * P);
    std::vector<float> coord(n * P * 4);
    std::vector<uint8_t> rgb(n * P * 4);
    HIP_OK(hipMemcpy(bodies.data(), d_bodies, nb * sizeof(slhip_body), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scenes.data(), d_scenes, n * sizeof(slhip_synth_scene), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(inst.data(), out.d_instance, inst.size() * 2, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(coord.data(), out.d_coord, coord.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rgb.data(), out.d_rgb, rgb.size(), hipMemcpyDeviceToHost));
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) { std::perror(argv[2]); return 2; }
    std::fwrite(bodies.data(), sizeof(slhip_body), nb, o);
    std::fwrite(scenes.data(), sizeof(slhip_synth_scene), n, o);
    std::fwrite(inst.data(), 2, inst.size(), o);
    std::fwrite(coord.data(), 4, coord.size(), o);
    std::fwrite(rgb.data(), 1, rgb.size(), o);
    std::fclose(o);
    size_t covered = 0;
    for (uint16_t v : inst) covered += v != 0;
    std::printf("slhip_batch_demo: %u scenes x %u objects settled (%u frames) and rendered at %ux%u; %zu object pixels; "
                "%u scenes refused by the settle\n", n, sp.n_objects, prm.frames, W, H, covered, refused);
    return 0;
}
