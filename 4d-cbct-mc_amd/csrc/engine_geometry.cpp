// engine_geometry.cpp -- C ABI of geometry changes on a resident context (4-D scans, cbctmc/mc/simulation.py:527-710) and of the
// voxel-file writers.
#include "engine_internal.hpp"

using namespace mcgpu;

extern "C" {

// ---- 4-D support: one resident context, many (geometry, projection angle) jobs (cbctmc/mc/simulation.py:527-710)
int mcgpu_set_projection_angles(mcgpu_ctx* ctx, int n, const float* angles_deg) {
  ABI_BEGIN
  require(ctx && angles_deg && n >= 2 && n <= kMaxProjections, -1, "!!ERROR!! mcgpu_set_projection_angles: need 2..1024 angles");
  HostModel& H = ctx->host;
  require(H.cfg.num_projections >= 2, -2,
          "!!ERROR!! mcgpu_set_projection_angles: the input file must define a CT trajectory (more than one projection)");
  H.cfg.enable_specific_angles = 1;
  H.cfg.specific_angles.assign(angles_deg, angles_deg + n);
  H.cfg.num_projections = n;
  H.source.resize(1);    // pose 0 is the input file's (MC-GPU_v1.3.cu:3313); the others follow the angles
  H.detector.resize(1);
  build_ct_trajectory(H);
  if (ctx->has_device) {
    DeviceModel& D = ctx->dev;
    HIP_TRY(hipSetDevice(D.device_id));
    HIP_TRY(hipDeviceSynchronize());
    D.src_all = D.put(H.source);  // the old arrays stay allocated until the context is destroyed (180 B per projection)
    D.det_all = D.put(H.detector);
  }
  return 0;
  ABI_END
}

int mcgpu_set_geometry_arrays(mcgpu_ctx* ctx, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density) {
  ABI_BEGIN
  require(ctx && n && spacing_cm && material && density && n[0] > 0 && n[1] > 0 && n[2] > 0, -1, "!!ERROR!! mcgpu_set_geometry_arrays: bad argument");
  HostModel& H = ctx->host;
  VoxelGrid v;
  for (int k = 0; k < 3; ++k) {
    v.n[k] = n[k];
    v.voxel_size[k] = spacing_cm[k];
    v.size_bbox[k] = v.n[k] * v.voxel_size[k];
    v.inv_voxel_size[k] = 1.0f / v.voxel_size[k];
  }
  const size_t nvox = v.count();
  v.material.assign(material, material + nvox);
  v.density.resize(nvox);
  for (int k = 0; k < kMaxMaterials; ++k) v.density_max[k] = -999.0f;
  // densities as the voxel file would carry them ("%.6f", cbctmc/mc/voxel_data.pyx:25), so that handing arrays over
  // in-process gives the tables -- and therefore the tallies -- of the file-based flow
  std::unordered_map<uint32_t, float> q;
  for (size_t i = 0; i < nvox; ++i) {
    uint32_t b;
    memcpy(&b, &density[i], 4);
    auto it = q.find(b);
    float d;
    if (it != q.end()) d = it->second;
    else {
      char t[64];
      snprintf(t, sizeof t, "%.6f", (double)density[i]);
      d = strtof(t, nullptr);
      q.emplace(b, d);
    }
    const int mat = material[i];
    require(mat >= 1 && mat <= kMaxMaterials, -2, "!!ERROR load_voxels!! Voxel material number out of range!!");
    require(d >= 1.0e-9f, -2, "!!ERROR load_voxels!! Voxel density can not be 0 or negative");
    v.density[i] = d;
    if (d > v.density_max[mat - 1]) v.density_max[mat - 1] = d;
  }
  // Everything that can fail is built beside the live model and swapped in at the end: after an error return the context
  // is what it was before the call.  The Woodcock majorant and the set of loaded materials depend on the volume
  // (MC-GPU_v1.3.cu:2220-2233,2294-2296), so the material tables are rebuilt.
  MaterialTables mat;
  load_material_files(H.cfg.file_materials, v, mat);
  int roi[6];
  for (int k = 0; k < 6; ++k) roi[k] = H.cfg.dose_roi[k];
  if (roi[1] > -1)
    for (int ax = 0; ax < 3; ++ax) roi[2 * ax + 1] = std::min(roi[2 * ax + 1], v.n[ax] - 1);
  std::swap(H.voxels, v);
  std::swap(H.mat, mat);
  int roi_old[6];
  for (int k = 0; k < 6; ++k) { roi_old[k] = H.cfg.dose_roi[k]; H.cfg.dose_roi[k] = roi[k]; }
  if (ctx->has_device) {
    const int dev = ctx->dev.device_id;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipDeviceSynchronize());
    DeviceModel old = std::move(ctx->dev);  // stays allocated until the new model is up
    ctx->dev = DeviceModel();
    try {
      upload_model(*ctx, dev);
    } catch (...) {
      ctx->dev.release();
      ctx->dev = std::move(old);
      std::swap(H.voxels, v);
      std::swap(H.mat, mat);
      for (int k = 0; k < 6; ++k) H.cfg.dose_roi[k] = roi_old[k];
      throw;
    }
    for (int k = 0; k < 5; ++k) ctx->dev.sched[k] = old.sched[k];  // the tuned FAST schedule survives a geometry change
    ctx->dev.sched_set = old.sched_set;
    apply_schedule(ctx->dev);
    old.release();  // NB: the dose tallies belong to a geometry and restart from zero with the new one
  }
  ctx->host_voxels_stale = false;
  ctx->table_cache.clear();
  return 0;
  ABI_END
}

int mcgpu_warp_geometry(mcgpu_ctx* ctx, const float* displacement, int frame, int default_material, float default_density) {
  ABI_BEGIN
  require(ctx && ctx->has_device && displacement && (frame == 0 || frame == 1), -1, "!!ERROR!! mcgpu_warp_geometry: bad argument (the context needs a device)");
  HostModel& H = ctx->host;
  DeviceModel& D = ctx->dev;
  require(D.vol_kind == kVolU8, -5, "!!ERROR!! mcgpu_warp_geometry: needs a palette volume (<= 256 distinct (material, density) pairs); use mcgpu_set_geometry_arrays");
  require(default_material >= 1 && default_material <= kMaxMaterials && D.compact_of[default_material - 1] >= 0, -5,
          "!!ERROR!! mcgpu_warp_geometry: the default material has no data file in this simulation");
  int default_index = -1;
  {
    char t[64];
    snprintf(t, sizeof t, "%.6f", (double)default_density);  // densities as a voxel file would carry them
    const float dq = strtof(t, nullptr);
    for (int e = 0; e < D.palette_size && default_index < 0; ++e) {
      int mc;
      memcpy(&mc, &D.palette_host[2 * e + 1], 4);
      if (mc == D.compact_of[default_material - 1] && D.palette_host[2 * e] == dq) default_index = e;
    }
  }
  require(default_index >= 0, -5, "!!ERROR!! mcgpu_warp_geometry: the default (material, density) is not in the palette; use mcgpu_set_geometry_arrays");
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipDeviceSynchronize());
  const size_t nvox = H.voxels.count();
  const size_t nsub = (size_t)D.sub_n[0] * D.sub_n[1] * D.sub_n[2];
  if (!D.vol_base) {  // first call: what is resident now is the base geometry of every later warp
    D.vol_base = D.put(std::vector<unsigned char>(D.vol_bytes, 0));
    HIP_TRY(hipMemcpy(D.vol_base, D.vol, D.vol_bytes, hipMemcpyDeviceToDevice));
    D.sub_first = D.put(std::vector<unsigned short>(nsub, 0));
    D.brick_first = D.put(std::vector<unsigned short>((size_t)D.brick_count, 0));
    D.code_of_dev = D.put(std::vector<unsigned char>(D.code_of, D.code_of + 256));
    D.rebuild_out = D.put(std::vector<unsigned int>(32, 0u));
    D.dvf = D.put(std::vector<float>(3 * nvox, 0.f));
  }
  HIP_TRY(hipMemcpy(D.dvf, displacement, 3 * nvox * 4, hipMemcpyHostToDevice));
  GeometryRebuild g;
  g.nx = H.voxels.n[0]; g.ny = H.voxels.n[1]; g.nz = H.voxels.n[2];
  g.brick_shift = D.brick_shift;
  for (int k = 0; k < 3; ++k) { g.bn[k] = D.brick_n[k]; g.sn[k] = D.sub_n[k]; }
  g.base_idx = D.vol_base; g.dvf = D.dvf; g.default_index = (unsigned char)default_index;
  g.idx = (unsigned char*)D.vol;
  g.sub_first = D.sub_first; g.brick_first = D.brick_first;
  g.sub = D.sub; g.bricks = D.bricks; g.code_of = D.code_of_dev; g.background = D.background;
  g.rec = D.tile_rec;
  for (int k = 0; k < 3; ++k) g.rn[k] = D.rec_n[k];
  g.out = D.rebuild_out;
  const bool allow_exterior = !D.knobs.no_exterior;
  HIP_TRY(launch_geometry_rebuild(g, frame, allow_exterior, nullptr));
  unsigned int out[17];
  HIP_TRY(hipMemcpy(out, D.rebuild_out, sizeof out, hipMemcpyDeviceToHost));  // waits for the kernels
  // largest density per material among the palette entries that occur -> Woodcock majorant (the only table that depends on it)
  for (int m = 0; m < kMaxMaterials; ++m) H.voxels.density_max[m] = -999.0f;
  for (int e = 0; e < D.palette_size; ++e)
    if (out[e >> 5] & (1u << (e & 31))) {
      int mc;
      memcpy(&mc, &D.palette_host[2 * e + 1], 4);
      for (int m = 0; m < kMaxMaterials; ++m)
        if (D.compact_of[m] == mc) H.voxels.density_max[m] = std::max(H.voxels.density_max[m], D.palette_host[2 * e]);
    }
  rebuild_woodcock(H.mat, H.voxels.density_max);
  {
    std::vector<float> wood(2 * (size_t)H.mat.num_values);
    for (int i = 0; i < H.mat.num_values; ++i) { wood[2 * i] = H.mat.woodcock[i].x; wood[2 * i + 1] = H.mat.woodcock[i].y; }
    HIP_TRY(hipMemcpy(D.woodcock, wood.data(), wood.size() * 4, hipMemcpyHostToDevice));
    const std::vector<float> coarse = coarse_woodcock(H);  // the FAST kernel's LDS copy of the majorant follows the table
    HIP_TRY(hipMemcpy(D.wood_coarse, coarse.data(), coarse.size() * 4, hipMemcpyHostToDevice));
  }
  D.sub_mixed = (int)out[16];
  // The object region (box and, where it pays, elliptic cylinder) and the first-level codes follow from the bricks' classification
  // exactly as at upload (mark_exterior_region): 64 KB of `brick_first` come down, 16 KB of codes go up.
  {
    const int had_exterior = D.has_exterior;
    std::vector<unsigned short> bf((size_t)D.brick_count);
    HIP_TRY(hipMemcpy(bf.data(), D.brick_first, bf.size() * 2, hipMemcpyDeviceToHost));
    std::vector<unsigned char> object((size_t)D.brick_count), exterior;
    for (int b = 0; b < D.brick_count; ++b) object[(size_t)b] = (bf[(size_t)b] == 0x100 || (int)bf[(size_t)b] != D.background) ? 1 : 0;
    mark_exterior_region(H, D, object, true, exterior);
    std::vector<unsigned char> bricks((size_t)D.brick_bytes, 0xFF);
    D.bricks_mixed = 0;
    for (int b = 0; b < D.brick_count; ++b) {
      const int code = exterior[(size_t)b] ? 14 : (bf[(size_t)b] == 0x100 ? 0xF : (int)D.code_of[bf[(size_t)b]]);
      D.bricks_mixed += (code == 0xF);
      const int sh = (b & 1) * 4;
      bricks[(size_t)(b >> 1)] = (unsigned char)((bricks[(size_t)(b >> 1)] & ~(0xF << sh)) | (code << sh));
    }
    HIP_TRY(hipMemcpy(D.bricks, bricks.data(), bricks.size(), hipMemcpyHostToDevice));
    for (int a = 0; a < 3; ++a) { D.cold_host.objbox_lo[a] = D.objbox_lo[a]; D.cold_host.objbox_hi[a] = D.objbox_hi[a]; }
    for (int a = 0; a < 2; ++a) { D.cold_host.ell_c[a] = D.ell_c[a]; D.cold_host.ell_inv[a] = D.ell_inv[a]; }
    if (D.has_exterior) {
      // code 14 means "background outside the object region": its palette slot must name the background even when the BASE
      // geometry had no exterior (its object box spanned the whole brick grid) and the warp made one
      D.brick_palette[14] = D.background;
      D.cold_host.brick_palette[14] = D.background;
    }
    if (D.has_exterior || had_exterior) HIP_TRY(hipMemcpy(D.cold, &D.cold_host, sizeof D.cold_host, hipMemcpyHostToDevice));
  }
  ctx->host_voxels_stale = true;
  ctx->table_cache.clear();
  return 0;
  ABI_END
}

int mcgpu_warp_volume(mcgpu_ctx* ctx, const int n[3], const uint8_t* material, const float* density, const float* displacement,
                      int default_material, float default_density, uint8_t* material_out, float* density_out) {
  ABI_BEGIN
  require(ctx && ctx->has_device && n && material && density && displacement && material_out && density_out, -1,
          "!!ERROR!! mcgpu_warp_volume: bad argument (the context needs a device)");
  HIP_TRY(hipSetDevice(ctx->dev.device_id));
  const size_t nvox = (size_t)n[0] * n[1] * n[2];
  unsigned char *m_in = nullptr, *m_out = nullptr;
  float *d_in = nullptr, *d_out = nullptr, *u = nullptr;
  hipError_t err = hipSuccess;
  auto step = [&](hipError_t e) { if (err == hipSuccess) err = e; };
  step(hipMalloc((void**)&m_in, nvox)); step(hipMalloc((void**)&m_out, nvox));
  step(hipMalloc((void**)&d_in, nvox * 4)); step(hipMalloc((void**)&d_out, nvox * 4)); step(hipMalloc((void**)&u, nvox * 12));
  if (err == hipSuccess) {
    step(hipMemcpy(m_in, material, nvox, hipMemcpyHostToDevice));
    step(hipMemcpy(d_in, density, nvox * 4, hipMemcpyHostToDevice));
    step(hipMemcpy(u, displacement, nvox * 12, hipMemcpyHostToDevice));
    if (err == hipSuccess) step(launch_warp(n[0], n[1], n[2], m_in, d_in, u, (unsigned char)default_material, default_density, m_out, d_out, nullptr));
    step(hipMemcpy(material_out, m_out, nvox, hipMemcpyDeviceToHost));
    step(hipMemcpy(density_out, d_out, nvox * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(m_in); (void)hipFree(m_out); (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(u);
  HIP_TRY(err);
  return 0;
  ABI_END
}

int mcgpu_write_voxel_file(const char* path, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density,
                           int gzip) {
  ABI_BEGIN
  require(path && n && spacing_cm && material && density, -1, "!!ERROR!! mcgpu_write_voxel_file: null argument");
  write_voxel_file(path, n, spacing_cm, material, density, gzip != 0);
  return 0;
  ABI_END
}

int mcgpu_write_voxel_binary(const char* path, const int n[3], const float spacing_cm[3], const uint8_t* material, const float* density) {
  ABI_BEGIN
  require(path && n && spacing_cm && material && density, -1, "!!ERROR!! mcgpu_write_voxel_binary: null argument");
  write_voxel_binary(path, n, spacing_cm, material, density);
  return 0;
  ABI_END
}

}  // extern "C"
