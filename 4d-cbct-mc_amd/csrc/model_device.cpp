// model_device.cpp -- the device-resident model of a context: tables, volume, brick grids, LDS layout, launch arguments.
// Replaces init_CUDA_device (docker/mcgpu/MC-GPU_v1.3.cu:2454-2724).
#include "engine_internal.hpp"

namespace mcgpu {

// Majorant mean free path per coarse energy bin (device_model.hpp: LdsLayout::wood): the table is piecewise linear, mfp(E) =
// x_i + E y_i on table bin i, so its minimum over a coarse bin is taken at the ends of its table bins.  One part in 1e6 below that,
// so that the float product the kernel forms (mfp x density x sigma) stays a probability.
std::vector<float> coarse_woodcock(const HostModel& H) {
  const int nv = H.mat.num_values, nc = (nv + (1 << kWoodShift) - 1) >> kWoodShift;
  const double e0 = (double)H.mat.e0, de = 1.0 / (double)H.mat.ide;
  std::vector<float> out((size_t)nc, 0.f);
  for (int c = 0; c < nc; ++c) {
    double lo = 1.0e30;
    for (int i = c << kWoodShift; i < std::min(nv, (c + 1) << kWoodShift); ++i) {
      const double x = (double)H.mat.woodcock[i].x, y = (double)H.mat.woodcock[i].y;
      lo = std::min(lo, std::min(x + (e0 + i * de) * y, x + (e0 + (i + 1) * de) * y));
    }
    out[(size_t)c] = (float)(lo * (1.0 - 1.0e-6));
  }
  return out;
}

// The object region of a u8 volume and the bricks outside it.  `object[b]` != 0: brick b holds something that is not homogeneous
// background.  Region = the bounding box of those bricks AND -- bodies in a CBCT volume are round, a quarter of their bounding box
// is corner air -- an elliptic cylinder (axis z) with the centre and the aspect of that box, scaled until every corner of every
// object brick is inside; kept only if it puts at least 5 % of the box's bricks outside (else the kernel would pay its quadratic
// for nothing; MCGPU_NO_ELLIPSE: never).  Sets D.objbox_*, D.ell_*, D.has_exterior, D.bricks_exterior; exterior[b] != 0: brick b lies
// wholly outside the region.  Shared by the upload of a geometry and by the device-side geometry change (mcgpu_warp_geometry).
void mark_exterior_region(const HostModel& H, DeviceModel& D, const std::vector<unsigned char>& object, bool have_background,
                          std::vector<unsigned char>& exterior) {
  const int k = D.brick_shift, nvx[3] = {H.voxels.n[0], H.voxels.n[1], H.voxels.n[2]};
  exterior.assign((size_t)D.brick_count, 0);
  D.has_exterior = 0;
  D.bricks_exterior = 0;
  D.ell_inv[0] = D.ell_inv[1] = 0.f;
  int lo[3] = {D.brick_n[0], D.brick_n[1], D.brick_n[2]}, hi[3] = {-1, -1, -1};
  auto coords = [&](int b, int c3[3]) { c3[0] = b % D.brick_n[0]; c3[1] = (b / D.brick_n[0]) % D.brick_n[1]; c3[2] = b / (D.brick_n[0] * D.brick_n[1]); };
  for (int b = 0; b < D.brick_count; ++b) {
    if (!object[(size_t)b]) continue;
    int c3[3];
    coords(b, c3);
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c3[a]); hi[a] = std::max(hi[a], c3[a]); }
  }
  if (!have_background || hi[0] < 0 || D.knobs.no_exterior) return;
  const float bs = (float)(1 << k);
  auto brick_rect = [&](const int c3[3], float r[4]) {
    r[0] = (float)c3[0] * bs * H.voxels.voxel_size[0]; r[1] = std::min((float)(c3[0] + 1) * bs, (float)nvx[0]) * H.voxels.voxel_size[0];
    r[2] = (float)c3[1] * bs * H.voxels.voxel_size[1]; r[3] = std::min((float)(c3[1] + 1) * bs, (float)nvx[1]) * H.voxels.voxel_size[1];
  };
  float box_lo[3], box_hi[3];
  for (int a = 0; a < 3; ++a) {
    box_lo[a] = (float)(lo[a] << k) * H.voxels.voxel_size[a];
    box_hi[a] = (float)std::min((hi[a] + 1) << k, nvx[a]) * H.voxels.voxel_size[a];
  }
  const float ecx = 0.5f * (box_lo[0] + box_hi[0]), ecy = 0.5f * (box_lo[1] + box_hi[1]);
  const float ea = 0.5f * (box_hi[0] - box_lo[0]), eb = 0.5f * (box_hi[1] - box_lo[1]);
  float inv[2] = {0.f, 0.f};
  if (ea > 0.f && eb > 0.f && !knob_set("MCGPU_NO_ELLIPSE")) {
    double s2 = 0.0;
    for (int b = 0; b < D.brick_count; ++b) {
      if (!object[(size_t)b]) continue;
      int c3[3];
      float r[4];
      coords(b, c3);
      brick_rect(c3, r);
      for (int c = 0; c < 4; ++c) {
        const double dx = ((c & 1) ? r[1] : r[0]) - ecx, dy = ((c & 2) ? r[3] : r[2]) - ecy;
        s2 = std::max(s2, dx * dx / ((double)ea * ea) + dy * dy / ((double)eb * eb));
      }
    }
    s2 *= 1.0 + 1.0e-4;
    inv[0] = (float)(1.0 / (s2 * (double)ea * ea));
    inv[1] = (float)(1.0 / (s2 * (double)eb * eb));
  }
  auto outside_cylinder = [&](const int c3[3]) {  // wholly outside: the point of the brick's rectangle nearest to the axis is
    float r[4];
    brick_rect(c3, r);
    const float px = std::min(std::max(ecx, r[0]), r[1]) - ecx, py = std::min(std::max(ecy, r[2]), r[3]) - ecy;
    return px * px * inv[0] + py * py * inv[1] > 1.001f;
  };
  long in_box = 0, cut = 0;
  if (inv[0] > 0.f) {
    for (int b = 0; b < D.brick_count; ++b) {
      int c3[3];
      coords(b, c3);
      if (c3[0] < lo[0] || c3[0] > hi[0] || c3[1] < lo[1] || c3[1] > hi[1] || c3[2] < lo[2] || c3[2] > hi[2]) continue;
      ++in_box;
      cut += outside_cylinder(c3) ? 1 : 0;
    }
    if (cut * 20 < in_box) inv[0] = inv[1] = 0.f;  // the cylinder would hardly trim the box: not worth its arithmetic
  }
  long outside = 0;
  for (int b = 0; b < D.brick_count; ++b) {
    int c3[3];
    coords(b, c3);
    bool out = c3[0] < lo[0] || c3[0] > hi[0] || c3[1] < lo[1] || c3[1] > hi[1] || c3[2] < lo[2] || c3[2] > hi[2];
    if (!out && inv[0] > 0.f) out = outside_cylinder(c3);
    if (out) { exterior[(size_t)b] = 1; ++outside; }
  }
  if (outside == 0) return;
  D.has_exterior = 1;
  D.bricks_exterior = (int)outside;
  for (int a = 0; a < 3; ++a) { D.objbox_lo[a] = box_lo[a]; D.objbox_hi[a] = box_hi[a]; }
  D.ell_c[0] = ecx; D.ell_c[1] = ecy;
  D.ell_inv[0] = inv[0]; D.ell_inv[1] = inv[1];
}

void read_env_knobs(DeviceModel& D) {
  auto env_int = [](const char* name, int dflt) { return knob_int(name, dflt); };
  DeviceModel::Knobs k;
  k.exterior_mode = env_int("MCGPU_EXTERIOR_MODE", 3);
  k.compat_thresh[0] = env_int("MCGPU_COMPAT_THRESH_COMPTON", -1);
  k.compat_thresh[1] = env_int("MCGPU_COMPAT_THRESH_RAYLEIGH", -1);
  k.compat_thresh[2] = env_int("MCGPU_COMPAT_THRESH_NEW", -1);
  k.compat_thresh[3] = env_int("MCGPU_COMPAT_THRESH_TAKE", -1);
  k.compat_stats = env_int("MCGPU_COMPAT_STATS", 0) != 0;
  k.blocks_per_cu = std::max(0, env_int("MCGPU_BLOCKS_PER_CU", 0));
  k.grid_spare_percent = std::max(0, env_int("MCGPU_GRID_SPARE_PERCENT", 0));
  static const char* const kSched[5] = {"MCGPU_THRESH_COMPTON", "MCGPU_THRESH_RAYLEIGH", "MCGPU_THRESH_NEW", "MCGPU_FLYABLE_LOW", "MCGPU_SWAP_BATCH"};
  for (int i = 0; i < 5; ++i) k.sched_override[i] = env_int(kSched[i], -1);
  k.slot_trade = env_int("MCGPU_SLOT_TRADE", 3);
  k.hold_q = env_int("MCGPU_HOLD_Q", 6) & 15;
  k.no_exterior = knob_set("MCGPU_NO_EXTERIOR");
  k.fast_sched = env_int("MCGPU_FAST_SCHED", D.knobs.fast_sched) != 0 ? 1 : 0;
  k.segment_loop = env_int("MCGPU_SEGMENT_LOOP", -1);
  D.knobs = k;
}

// The FAST scheduler's parameters live in TrackCold (device memory read through the scalar cache): effective value =
// environment override, else the schedule set by mcgpu_set_fast_schedule.  Uploads only when something changed, after the
// device has drained (callers are set-up paths, never a launch).
void apply_schedule(DeviceModel& D) {
  if (!D.cold) return;
  TrackCold& ch = D.cold_host;
  int want[5];
  // the workgroup-level pool fills its batches from the whole workgroup: its thresholds are "lanes a batch must fill"
  static const int kWgSched[5] = {56, 56, 56, 40, 40};
  for (int i = 0; i < 5; ++i)
    want[i] = D.knobs.sched_override[i] >= 0 ? D.knobs.sched_override[i] : ((D.knobs.fast_sched == 1 && !D.sched_set) ? kWgSched[i] : D.sched[i]);
  want[3] = std::max(1, want[3]);
  want[4] = std::max(1, want[4]);
  // bit 0: slots traded before flight, bit 1: before the Compton and tally/source services; bits 8-11: hold_q (sixteenths
  // of the flying lanes that end a flight segment at the latest)
  const int trade = D.knobs.slot_trade | (D.knobs.hold_q << 8);
  if (ch.trade_slots == trade && ch.thresh_compton == want[0] && ch.thresh_rayleigh == want[1] && ch.thresh_new == want[2] && ch.flyable_low == want[3] &&
      ch.swap_batch == want[4])
    return;
  ch.thresh_compton = want[0]; ch.thresh_rayleigh = want[1]; ch.thresh_new = want[2]; ch.flyable_low = want[3]; ch.swap_batch = want[4];
  ch.trade_slots = trade;
  HIP_TRY(hipSetDevice(D.device_id));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(D.cold, &ch, sizeof ch, hipMemcpyHostToDevice));
}

// COMPAT kernel: bounds of S0(E) = sum_i f_i n_i(E, theta = pi) (K.cu:1300-1314), the normalisation of the Compton angle test
// "xi S0 > S(theta) T(tau)" (K.cu:1367-1372).  Computing it costs a second pass over the 29-40 electron shells of a tissue per
// event -- 46 % of the kernel's Compton work -- and the test rarely needs its exact value.  S0 does not decrease with E: a
// shell enters at E > U_i with a positive term, and each term grows with E (p_z(theta = pi) grows with E (E - U_i), the
// profile integral n_i with p_z).  So S0 at the lower / upper edge of an energy bin bounds it inside the bin.  The edges are
// evaluated here in double, one bin of slack on either side absorbs the float rounding of the kernel's bin index, and a
// relative margin of 1e-4 covers the float arithmetic of the reference's own S0 (terms accurate to ~2e-6 of f_i, forty
// additions of 6e-8 each).  A test that both bounds decide alike is decided; for the rest (well below 1 % with 1024 bins) the kernel computes S0.
// Layout: float2 {lo, hi} at [row * kS0Bins + bin]; row = compact material index with `compact_of`, else material number - 1.
static std::vector<float> build_s0_bounds(const HostModel& H, const int* compact_of, int rows, float* emin_out, float* inv_w_out) {
  const double emin = H.mat.e0, emax = H.mat.e0 + (double)(H.mat.num_values - 1) / H.mat.ide, w = (emax - emin) / kS0Bins;
  const double mc2 = (double)510998.918f, c1 = (double)0.707106781186545f, c2 = (double)1.4142135623731f;
  std::vector<float> bounds((size_t)2 * kS0Bins * std::max(rows, 1), 0.f);
  for (int m = 0; m < kMaxMaterials; ++m) {
    const int mc = compact_of ? compact_of[m] : (m < rows ? m : -1);
    if (mc < 0) continue;
    const int n = std::min(H.mat.noscco[m], kMaxShells);
    double fsum = 0.0;
    for (int i = 0; i < n; ++i) fsum += (double)H.mat.fco[m + i * kMaxMaterials];
    auto s0_at = [&](double E) {
      double acc = 0.0;
      for (int i = 0; i < n; ++i) {
        const double U = H.mat.uico[m + i * kMaxMaterials], J = H.mat.fj0[m + i * kMaxMaterials], f = H.mat.fco[m + i * kMaxMaterials];
        if (!(U < E)) continue;
        const double aux = E * (E - U) * 2.0;
        const double pz = J * (aux - U * mc2) / (std::sqrt(aux + aux + U * U) * mc2);
        const double a = pz > 0.0 ? c1 + pz * c2 : c1 - pz * c2;
        const double t = 0.5 * std::exp(0.5 - a * a);
        acc += f * (pz > 0.0 ? 1.0 - t : t);
      }
      return acc;
    };
    // the monotonicity argument needs shells with f >= 0, J > 0, U >= 0 (every PENELOPE table has them); a file that breaks it
    // gets bounds that decide nothing: the kernel then computes S0 for every test, like the reference
    bool regular = true;
    for (int i = 0; i < n; ++i)
      regular = regular && H.mat.fco[m + i * kMaxMaterials] >= 0.f && H.mat.fj0[m + i * kMaxMaterials] > 0.f && H.mat.uico[m + i * kMaxMaterials] >= 0.f;
    if (!regular) {
      for (int k = 0; k < kS0Bins; ++k) {
        bounds[2 * ((size_t)mc * kS0Bins + k)] = 0.f;
        bounds[2 * ((size_t)mc * kS0Bins + k) + 1] = 3.0e38f;
      }
      continue;
    }
    std::vector<double> edge(kS0Bins + 1);
    for (int k = 0; k <= kS0Bins; ++k) edge[k] = s0_at(emin + k * w);
    for (int k = 0; k < kS0Bins; ++k) {
      const double lo = k >= 1 ? edge[k - 1] * (1.0 - 1e-4) : 0.0;
      const double hi = (k + 2 <= kS0Bins ? edge[k + 2] : fsum) * (1.0 + 1e-4);
      bounds[2 * ((size_t)mc * kS0Bins + k)] = std::nextafterf((float)lo, -1.0f);
      bounds[2 * ((size_t)mc * kS0Bins + k) + 1] = std::nextafterf((float)hi, 3.0e38f);
    }
  }
  if (emin_out) *emin_out = (float)emin;
  if (inv_w_out) *inv_w_out = (float)(1.0 / w);
  return bounds;
}

// Build the palette-compressed volume and the compact-material tables and upload everything.
void upload_model(mcgpu_ctx& C, int device_id) {
  const HostModel& H = C.host;
  DeviceModel& D = C.dev;
  HIP_TRY(hipSetDevice(device_id));
  D.device_id = device_id;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  D.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  read_env_knobs(D);

  // compact material numbering
  D.nmat = 0;
  for (int m = 0; m < kMaxMaterials; ++m) D.compact_of[m] = H.mat.used[m] ? D.nmat++ : -1;
  const int nmat = D.nmat, nv = H.mat.num_values;

  // ---- volume -> palette indices
  const size_t nvox = H.voxels.count();
  std::unordered_map<uint64_t, int> index_of;
  std::vector<float> palette;  // {density, bits(mat_c)}
  std::vector<uint16_t> idx16(nvox);
  bool overflow = false;
  {
    uint64_t last_key = ~0ull;
    int last_idx = -1;
    for (size_t i = 0; i < nvox; ++i) {
      uint32_t db;
      memcpy(&db, &H.voxels.density[i], 4);
      const uint64_t key = ((uint64_t)H.voxels.material[i] << 32) | db;
      if (key != last_key) {
        auto it = index_of.find(key);
        if (it == index_of.end()) {
          if (index_of.size() >= 65536) { overflow = true; break; }
          const int mc = D.compact_of[H.voxels.material[i] - 1];
          if (mc < 0) throw Error(-2, "!!ERROR!! A voxel uses material " + std::to_string((int)H.voxels.material[i]) + " but no data file was given for it.");
          float mcf;
          memcpy(&mcf, &mc, 4);
          last_idx = (int)index_of.size();
          index_of.emplace(key, last_idx);
          palette.push_back(H.voxels.density[i]);
          palette.push_back(mcf);
        } else {
          last_idx = it->second;
        }
        last_key = key;
      }
      idx16[i] = (uint16_t)last_idx;
    }
  }
  if (overflow) {
    D.vol_kind = kVolRaw;
    std::vector<float> raw(2 * nvox);
    for (size_t i = 0; i < nvox; ++i) {
      const int mc = D.compact_of[H.voxels.material[i] - 1];
      if (mc < 0) throw Error(-2, "!!ERROR!! A voxel uses a material without data file.");
      raw[2 * i] = H.voxels.density[i];
      memcpy(&raw[2 * i + 1], &mc, 4);
    }
    D.vol = D.put(raw);
    D.vol_bytes = raw.size() * 4;
    D.palette_size = 0;
    D.palette = D.put(std::vector<float>(2, 0.f));
  } else if (index_of.size() <= 256) {
    D.vol_kind = kVolU8;
    {
      // the reference's default for voxels warped in from outside the volume (air: material 1 at 0.0013 g/cm^3,
      // cbctmc/mc/geometry.py:403-418) gets a palette entry even when no voxel holds it yet, so that a geometry can be
      // warped on the device without touching the palette (mcgpu_warp_geometry)
      const float air = 0.0013f;
      uint32_t db;
      memcpy(&db, &air, 4);
      const uint64_t key = ((uint64_t)1 << 32) | db;
      if (!index_of.count(key) && index_of.size() < 256 && D.compact_of[0] >= 0) {
        const int mc = D.compact_of[0];
        float mcf;
        memcpy(&mcf, &mc, 4);
        index_of.emplace(key, (int)index_of.size());
        palette.push_back(air);
        palette.push_back(mcf);
      }
    }
    D.palette_host = palette;
    std::vector<uint8_t> idx8(nvox);
    for (size_t i = 0; i < nvox; ++i) idx8[i] = (uint8_t)idx16[i];
    D.palette_size = (int)index_of.size();
    D.palette = D.put(palette);
    // brick grid: smallest power-of-two brick (>= 4 voxels) that keeps the grid within the LDS budget
    const int nx = H.voxels.n[0], ny = H.voxels.n[1], nz = H.voxels.n[2];
    {
      // device layout: tiles of 4x4x4 voxels = one 64-byte sector = one sub-brick of the second level (device_model.hpp:
      // tiled_voxel); the padding voxels of edge tiles repeat the tile's first voxel and are never addressed
      const unsigned int snx = (unsigned int)((nx + 3) >> 2), sny = (unsigned int)((ny + 3) >> 2), snz = (unsigned int)((nz + 3) >> 2);
      const size_t tiles = (size_t)snx * sny * snz;
      if (tiles * 64 >= (1ULL << 31)) throw Error(-2, "!!ERROR!! voxel grid too large for the 32-bit voxel index of the kernel");
      std::vector<uint8_t> tiled(tiles * 64);
      for (size_t t = 0; t < tiles; ++t) {
        const int x0 = (int)(t % snx) << 2, y0 = (int)((t / snx) % sny) << 2, z0 = (int)(t / ((size_t)snx * sny)) << 2;
        const uint8_t pad = idx8[((size_t)z0 * ny + y0) * nx + x0];
        for (int dz = 0; dz < 4; ++dz)
          for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) {
              const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
              tiled[t * 64 + (size_t)(dz * 16 + dy * 4 + dx)] = (x < nx && y < ny && z < nz) ? idx8[((size_t)z * ny + y) * nx + x] : pad;
            }
      }
      D.vol = D.put(tiled);
      D.vol_bytes = tiled.size();
    }
    int k = 2;
    auto nb = [&](int n, int sh) { return (n + (1 << sh) - 1) >> sh; };
    const char* mb = knob_str("MCGPU_MAX_BRICKS");  // tuning knob: a coarser grid frees LDS
    long max_bricks = mb ? std::min<long>(std::max<long>(atol(mb), 1), kMaxBricks) : kMaxBricks;
    {
      // The FAST kernel wants two 1024-thread workgroups per CU, i.e. an LDS image of at most 80 KB.  Everything but the
      // brick grid is fixed by the materials in use (22 tissue materials: 458 Compton shells = 7.3 KB against 1.4 KB for
      // the Catphan set), so the grid gets what is left after the tables, the history slots and a coarse bracket table.
      int shells = 0;
      for (int m = 0; m < kMaxMaterials; ++m)
        if (D.compact_of[m] >= 0) shells += std::min(H.mat.noscco[m], kMaxShells);
      const int ns = std::min(H.spectrum.num_bins + 1, kMaxSpectrumBins) + 1;
      const int nc = (nv + (1 << 9) - 1) >> 9;  // brackets no coarser than 2^9 table bins
      const long fixed = std::max(shells, 1) * 16 + std::max(nmat, 1) * 8 + ns * 10 + (16 + (long)index_of.size()) * 8 + 2 * kMaxMaterials * 8 +
                         (long)kSlotWords * kPoolParked * kPoolBlockThreads * 4 + (D.knobs.fast_sched == 1 ? kPoolQueueBytes : 0) + nc * nmat * 2 + nc * 4 +
                         12 * 16 +
                         (((nv + (1 << kWoodShift) - 1) >> kWoodShift) * 4 + 16);
      const long left = 160 * 1024 / 2 - fixed;
      if (left > 0) max_bricks = std::min(max_bricks, std::max(2 * left, 512L));
    }
    while ((long)nb(nx, k) * nb(ny, k) * nb(nz, k) > max_bricks) ++k;
    D.brick_shift = k;
    D.brick_n[0] = nb(nx, k); D.brick_n[1] = nb(ny, k); D.brick_n[2] = nb(nz, k);
    D.brick_count = D.brick_n[0] * D.brick_n[1] * D.brick_n[2];
    std::vector<int> first(D.brick_count, -1);
    std::vector<unsigned char> mixed(D.brick_count, 0);
    // second level: sub-bricks of 4^3 voxels, dense over the volume (first2: palette entry, 0x100 = mixed)
    D.sub_n[0] = (nx + 3) >> 2; D.sub_n[1] = (ny + 3) >> 2; D.sub_n[2] = (nz + 3) >> 2;
    const size_t nsub = (size_t)D.sub_n[0] * D.sub_n[1] * D.sub_n[2];
    std::vector<short> first2(nsub, -1);
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y) {
        const size_t row = ((size_t)z * ny + y) * nx;
        const size_t brow = ((size_t)(z >> k) * D.brick_n[1] + (y >> k)) * D.brick_n[0];
        const size_t srow = ((size_t)(z >> 2) * D.sub_n[1] + (y >> 2)) * D.sub_n[0];
        for (int x = 0; x < nx; ++x) {
          const int b = (int)(brow + (x >> k)), v = idx8[row + x];
          if (first[b] < 0) first[b] = v;
          else if (first[b] != v) mixed[b] = 1;
          short& f2 = first2[srow + (x >> 2)];
          if (f2 < 0) f2 = (short)v;
          else if (f2 != v) f2 = 0x100;
        }
      }
    // 4-bit codes: the 14 most frequent palette entries among homogeneous bricks get codes 0..13, every other
    // brick (mixed, or a rarer homogeneous one) is 0xF = "read the voxel"; code 14 = EXTERIOR (below)
    std::vector<long> homogeneous(256, 0);
    for (int b = 0; b < D.brick_count; ++b)
      if (!mixed[b] && first[b] >= 0) ++homogeneous[first[b]];
    std::vector<int> order(256);
    for (int i = 0; i < 256; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return homogeneous[a] > homogeneous[b]; });
    int code_of[256];
    for (int i = 0; i < 256; ++i) code_of[i] = 0xF;
    for (int c = 0; c < 14; ++c) {
      D.brick_palette[c] = 0;
      if (homogeneous[order[c]] > 0) { code_of[order[c]] = c; D.brick_palette[c] = order[c]; }
    }
    D.brick_palette[14] = D.brick_palette[15] = 0;
    for (int i = 0; i < 256; ++i) D.code_of[i] = (unsigned char)code_of[i];
    D.background = order[0];
    // Exterior (mark_exterior_region): outside the object region every brick is homogeneous background, and the FAST kernel crosses
    // it with one exact free-path sample instead of delta-tracking through it (track_pool.inc: exterior_hop).
    {
      const int bg = order[0];
      std::vector<unsigned char> object((size_t)D.brick_count, 0);
      for (int b = 0; b < D.brick_count; ++b) object[(size_t)b] = (mixed[b] || first[b] != bg) ? 1 : 0;
      std::vector<unsigned char> exterior;
      mark_exterior_region(H, D, object, homogeneous[bg] > 0, exterior);
      for (int b = 0; b < D.brick_count; ++b)
        if (exterior[(size_t)b]) first[b] = -2;  // marks EXTERIOR for the encoder below
      if (D.has_exterior) D.brick_palette[14] = bg;
    }
    D.brick_bytes = (D.brick_count + 1) / 2;
    std::vector<unsigned char> bricks(D.brick_bytes, 0xFF);
    D.bricks_mixed = 0;
    D.bricks_exterior = 0;
    for (int b = 0; b < D.brick_count; ++b) {
      int code = 0xF;
      if (first[b] == -2) { code = 14; ++D.bricks_exterior; }
      else if (!mixed[b] && first[b] >= 0) code = code_of[first[b]];
      D.bricks_mixed += (code == 0xF);
      const int sh = (b & 1) * 4;
      bricks[b >> 1] = (unsigned char)((bricks[b >> 1] & ~(0xF << sh)) | (code << sh));
    }
    D.bricks = D.put(bricks);
    {
      // Second-level codes (same 4-bit coding, no EXTERIOR): a flight step that lands in a mixed brick asks this table,
      // which stays in L2 (0.5-1 MB), before it asks the volume (64-128 MiB: Infinity Cache / HBM).  On a body-filling
      // volume 78 % of the tissue voxels lie in mixed 16^3 bricks but only 24 % in mixed 4^3 sub-bricks, and the voxel
      // gathers of the flight step were what bound that workload (1.45 KB of fabric traffic per history at 5e9 histories/s).
      std::vector<unsigned char> sub((nsub + 1) / 2, 0xFF);
      D.sub_mixed = 0;
      // Round 2 (x-fastest rows): worth its dependent L2 round trip where most bricks a photon meets are mixed (thorax +24 %).
      // Round 3: the volume is stored in 4x4x4 TILES, one tile = one 64-byte sector = one sub-brick -- asking the volume
      // directly now costs one sector like asking this table, without the second dependent round trip, and the tile it
      // brings in serves the neighbouring voxels of later photons.  Measured on one box (tools/ab_second_level.sh): thorax 14.27
      // -> 13.68 ms, CIRS 6.52 -> 6.25, Catphan 4.17 -> 4.09 with the table OFF.  So it is off unless MCGPU_SUB_BRICKS=1
      // asks for it (kept: it halves the fabric traffic where that is what binds, and the tests hold both routes to the
      // same tallies).
      const char* knob = knob_str("MCGPU_SUB_BRICKS");
      const bool off = knob ? atoi(knob) == 0 : true;
      for (size_t b = 0; b < nsub; ++b) {
        const int code = (!off && first2[b] >= 0 && first2[b] < 0x100) ? code_of[first2[b]] : 0xF;
        D.sub_mixed += (code == 0xF);
        const int sh = (int)(b & 1) * 4;
        sub[b >> 1] = (unsigned char)((sub[b >> 1] & ~(0xF << sh)) | (code << sh));
      }
      D.sub = off ? nullptr : D.put(sub);
    }
    {
      // Tile records (device_model.hpp: TileRecord): the hot set of the voxel gathers is every 64-byte tile of every MIXED brick.
      // Where that set is far beyond the L2 (4 MB per XCD) -- body-filling tissue volumes: thorax 22 MB -- the launch is bound by
      // the line fills of those gathers (profiles/r04p_*: ONE more cold line per mixed step doubles the thorax's kernel time, one
      // more load from the SAME line costs 2 %), and the records shrink the set fourfold.  MCGPU_TILE_RECORDS=0/1 overrides.
      long long hot_tiles = 0;
      for (size_t b = 0; b < nsub; ++b) {
        const int bx = (int)((b % D.sub_n[0]) << 2) >> k, by = (int)(((b / D.sub_n[0]) % D.sub_n[1]) << 2) >> k, bz = (int)((b / ((size_t)D.sub_n[0] * D.sub_n[1])) << 2) >> k;
        hot_tiles += mixed[((size_t)bz * D.brick_n[1] + by) * D.brick_n[0] + bx] ? 1 : 0;
      }
      D.tiles_in_mixed_bricks = hot_tiles;
      const char* knob = knob_str("MCGPU_TILE_RECORDS");
      const bool on = knob ? atoi(knob) != 0 : hot_tiles * 64 > (8LL << 20);
      D.rec_n[0] = (D.sub_n[0] + 1) >> 1; D.rec_n[1] = (D.sub_n[1] + 1) >> 1; D.rec_n[2] = (D.sub_n[2] + 1) >> 1;
      D.tile_rec = nullptr;
      if (on) {
        std::vector<TileRecord> rec((size_t)D.rec_n[0] * D.rec_n[1] * D.rec_n[2] * 8, TileRecord{0u, 0u, 0ULL});
        for (size_t t = 0; t < nsub; ++t) {
          const unsigned int tx = (unsigned int)(t % D.sub_n[0]), ty = (unsigned int)((t / D.sub_n[0]) % D.sub_n[1]), tz = (unsigned int)(t / ((size_t)D.sub_n[0] * D.sub_n[1]));
          short v64[64];
          for (int v = 0; v < 64; ++v) {
            const int x = (int)(tx << 2) + (v & 3), y = (int)(ty << 2) + ((v >> 2) & 3), z = (int)(tz << 2) + (v >> 4);
            v64[v] = (x >= nx || y >= ny || z >= nz) ? (short)-1 : (short)idx8[((size_t)z * ny + y) * nx + x];  // padding of an edge tile: never addressed
          }
          const TileRecord r = encode_tile_record(v64);
          rec[tile_record_index(tx, ty, tz, (unsigned int)D.rec_n[0], (unsigned int)(D.rec_n[0] * D.rec_n[1]))] = r;
        }
        D.tile_rec = D.put(rec);
      }
    }
  } else {
    D.vol_kind = kVolU16;
    D.vol = D.put(idx16);
    D.vol_bytes = nvox * 2;
    D.palette_size = (int)index_of.size();
    D.palette = D.put(palette);
  }

  // ---- cross-section records
  std::vector<float> wood(2 * (size_t)nv), rec(8 * (size_t)nv * nmat, 0.f);
  for (int i = 0; i < nv; ++i) { wood[2 * i] = H.mat.woodcock[i].x; wood[2 * i + 1] = H.mat.woodcock[i].y; }
  for (int i = 0; i < nv; ++i)
    for (int m = 0; m < kMaxMaterials; ++m) {
      const int mc = D.compact_of[m];
      if (mc < 0) continue;
      float* r = &rec[8 * ((size_t)mc * nv + i)];  // material-major rows (track_common.inc: table_row)
      const Float3& a = H.mat.a[(size_t)i * kMaxMaterials + m];
      const Float3& b = H.mat.b[(size_t)i * kMaxMaterials + m];
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = b.x; r[4] = b.y; r[5] = b.z;
      r[6] = H.mat.pmax[(size_t)(i + 1) * kMaxMaterials + m];
      r[7] = 0.f;
    }
  D.woodcock = D.put(wood);
  D.wood_coarse = D.put(coarse_woodcock(H));
  D.mfp = D.put(rec);
  {
    std::vector<float> tot(2 * (size_t)nv * nmat);
    for (size_t k = 0; k < (size_t)nv * nmat; ++k) { tot[2 * k] = rec[8 * k]; tot[2 * k + 1] = rec[8 * k + 3]; }
    D.mfp_tot = D.put(tot);
    // Brackets of the total cross section for the FAST flight step: per (coarse energy bin = 2^shift table bins,
    // material) the centre of [min, max] of a_tot + b_tot * E over the coarse bin as fp16, and per coarse bin one
    // relative half width covering every material.  A step whose random number falls outside
    // [1 - m*hi, 1 - m*lo) is decided from LDS alone; only the narrow band in between fetches the exact value, so the
    // decisions are those of the exact test.  The LDS image takes the finest table that still leaves two workgroups per CU.
    D.sig_tot_host = tot;
  }
  std::vector<float> xco(kRayleighPoints * nmat), pco(xco), aco(xco), bco(xco);
  std::vector<unsigned char> itl(kRayleighPoints * nmat), itu(itl);
  std::vector<float> fco(kMaxShells * nmat, 0.f), uico(fco), fj0(fco);
  std::vector<int> nosc(std::max(nmat, 1), 0);
  std::vector<float> shell_cut(kMaxShells * std::max(nmat, 1), 1.0f);
  std::vector<unsigned char> shell_alias(kMaxShells * std::max(nmat, 1), 0);
  for (int m = 0; m < kMaxMaterials; ++m) {
    const int mc = D.compact_of[m];
    if (mc < 0) continue;
    for (int i = 0; i < kRayleighPoints; ++i) {
      xco[mc * kRayleighPoints + i] = H.mat.xco[m * kRayleighPoints + i];
      pco[mc * kRayleighPoints + i] = H.mat.pco[m * kRayleighPoints + i];
      aco[mc * kRayleighPoints + i] = H.mat.aco[m * kRayleighPoints + i];
      bco[mc * kRayleighPoints + i] = H.mat.bco[m * kRayleighPoints + i];
      itl[mc * kRayleighPoints + i] = H.mat.itlco[m * kRayleighPoints + i];
      itu[mc * kRayleighPoints + i] = H.mat.ituco[m * kRayleighPoints + i];
    }
    for (int s = 0; s < kMaxShells; ++s) {
      fco[s * nmat + mc] = H.mat.fco[m + s * kMaxMaterials];
      uico[s * nmat + mc] = H.mat.uico[m + s * kMaxMaterials];
      fj0[s * nmat + mc] = H.mat.fj0[m + s * kMaxMaterials];
    }
    nosc[mc] = H.mat.noscco[m];
    {
      // FAST Compton sampler (track_common.inc: compton_draw): Walker alias table of the shell weights f_i (Vose's
      // construction, in double): column k keeps shell k below cut[k] and maps the rest of the column to alias[k]
      const int n = std::min(H.mat.noscco[m], kMaxShells);
      double F = 0.0;
      for (int s = 0; s < n; ++s) F += (double)H.mat.fco[m + s * kMaxMaterials];
      std::vector<double> q(n);
      std::vector<int> small, large;
      for (int s = 0; s < n; ++s) {
        q[s] = F > 0.0 ? (double)H.mat.fco[m + s * kMaxMaterials] * n / F : 1.0;
        (q[s] < 1.0 ? small : large).push_back(s);
        shell_cut[s * nmat + mc] = 1.0f;
        shell_alias[s * nmat + mc] = (unsigned char)s;
      }
      while (!small.empty() && !large.empty()) {
        const int a = small.back(), b = large.back();
        small.pop_back();
        shell_cut[a * nmat + mc] = (float)q[a];
        shell_alias[a * nmat + mc] = (unsigned char)b;
        q[b] -= 1.0 - q[a];
        if (q[b] < 1.0) { large.pop_back(); small.push_back(b); }
      }
    }
  }
  D.xco = D.put(xco); D.pco = D.put(pco); D.aco = D.put(aco); D.bco = D.put(bco);
  D.itl = D.put(itl); D.itu = D.put(itu);
  D.fco = D.put(fco); D.uico = D.put(uico); D.fj0 = D.put(fj0);
  D.s0_bounds = D.put(build_s0_bounds(H, D.compact_of, nmat, &D.s0_emin, &D.s0_inv_w));  // COMPAT: bounds of S0 per (material, energy bin)
  D.noscco = D.put(nosc);
  D.shell_cut = D.put(shell_cut);
  D.shell_alias = D.put(shell_alias);
  D.espc = D.put(std::vector<float>(H.spectrum.espc, H.spectrum.espc + kMaxSpectrumBins));
  D.cutoff = D.put(std::vector<float>(H.spectrum.cutoff, H.spectrum.cutoff + kMaxSpectrumBins));
  D.alias = D.put(std::vector<short>(H.spectrum.alias, H.spectrum.alias + kMaxSpectrumBins));
  // ---- LDS image of the kernels (byte offsets; track_common.inc: stage_tables)
  {
    LdsLayout& Y = D.lds;
    int off = 0;
    auto take = [&](int bytes, int align) { off = (off + align - 1) / align * align; const int at = off; off += bytes; return at; };
    const int ns = std::min(H.spectrum.num_bins + 1, kMaxSpectrumBins) + 1;
    int total_shells = 0;
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] >= 0) { D.shell_first[D.compact_of[m]] = total_shells; total_shells += std::min(H.mat.noscco[m], kMaxShells); }
    Y.shells = take(std::max(total_shells, 1) * 16, 16);
    Y.nosc = take(std::max(nmat, 1) * 8, 16);
    Y.espc = take(ns * 4, 16);
    Y.cutoff = take(ns * 4, 16);
    Y.alias = take(ns * 2, 16);
    Y.pal = take(D.vol_kind == kVolU8 ? (16 + D.palette_size) * 8 : 0, 16);
    Y.brick = take(D.vol_kind == kVolU8 ? D.brick_bytes : 0, 16);
    Y.dose_mat = take(2 * kMaxMaterials * 8, 16);
    Y.slots = take(0, 16);  // the COMPAT kernel's image ends here
    take(kSlotWords * kPoolParked * kPoolBlockThreads * 4, 16);
    Y.queues = D.knobs.fast_sched == 1 ? take(kPoolQueueBytes, 16) : 0;
    Y.wood = take(((nv + (1 << kWoodShift) - 1) >> kWoodShift) * 4, 16);
    Y.sig_mid = Y.sig_w = off;
    D.sig_shift = -1;
    if (!knob_set("MCGPU_NO_BRACKETS") && nmat > 0) {
      const int budget = 160 * 1024 / 2;  // two 1024-thread workgroups per CU
      for (int shift = 6; shift <= 12; ++shift) {
        const int nc = (nv + (1 << shift) - 1) >> shift;
        const int need = (off + 15) / 16 * 16 + (nc * nmat * 2 + 15) / 16 * 16 + (nc * 4 + 15) / 16 * 16;
        if (need > budget) continue;
        D.sig_shift = shift;
        D.sig_coarse = nc;
        Y.sig_mid = take(nc * nmat * 2, 16);
        Y.sig_w = take(nc * 4, 16);
        break;
      }
    }
    Y.total = (off + 15) / 16 * 16;
  }
  if (D.sig_shift >= 0) {
    const int nc = D.sig_coarse, shift = D.sig_shift;
    auto to_half = [](float f) -> unsigned short {  // round to nearest even; inputs are positive normal numbers
      uint32_t x; memcpy(&x, &f, 4);
      const int e = (int)((x >> 23) & 0xFF) - 127 + 15;
      uint32_t m = x & 0x7FFFFF;
      if (e <= 0) return 0;
      if (e >= 31) return 0x7BFF;
      uint32_t h = ((uint32_t)e << 10) | (m >> 13);
      const uint32_t rem = m & 0x1FFF;
      if (rem > 0x1000 || (rem == 0x1000 && (h & 1))) ++h;
      return (unsigned short)std::min<uint32_t>(h, 0x7BFF);
    };
    auto from_half = [](unsigned short h) -> double { return std::ldexp((double)((h & 0x3FF) | 0x400), (int)(h >> 10) - 25); };
    std::vector<unsigned short> mid((size_t)nc * nmat, 0);
    std::vector<float> wv(nc, 0.f);
    const double e0 = H.mat.e0, ide = H.mat.ide;
    for (int c = 0; c < nc; ++c) {
      double wmax = 0.0;
      for (int mc = 0; mc < nmat; ++mc) {
        double lo = 1e300, hi = -1e300;
        for (int i = c << shift; i < std::min(nv, (c + 1) << shift); ++i) {
          const double a = D.sig_tot_host[2 * ((size_t)mc * nv + i)], b = D.sig_tot_host[2 * ((size_t)mc * nv + i) + 1];
          // the kernel evaluates a + b * E for E in [E_i, E_{i+1}) (one table bin; a little beyond for float rounding)
          const double ea = e0 + (i - 0.01) / ide, eb = e0 + (i + 1.01) / ide;
          lo = std::min(lo, std::min(a + b * ea, a + b * eb));
          hi = std::max(hi, std::max(a + b * ea, a + b * eb));
        }
        if (!(lo > 0.0)) lo = std::min(1e-30, hi > 0.0 ? hi : 1e-30);
        const unsigned short hbits = to_half((float)(0.5 * (lo + hi)));
        mid[(size_t)c * nmat + mc] = hbits;
        const double m = from_half(hbits);
        if (hbits == 0 || hbits == 0x7BFF || !(m > 0.0)) { wmax = 1e30; continue; }  // not representable: the band is everything
        wmax = std::max(wmax, std::max((hi - m) / m, (m - lo) / m));
      }
      wv[c] = (float)std::min(wmax * 1.001 + 1e-5, 1e30);
    }
    D.sig_mid = D.put(mid);
    D.sig_w = D.put(wv);
  }
  D.num_spectrum_bins = H.spectrum.num_bins;
  {
    TrackCold cold;
    memset(&cold, 0, sizeof cold);
    cold.xco = D.xco; cold.pco = D.pco; cold.aco = D.aco; cold.bco = D.bco; cold.itl = D.itl; cold.itu = D.itu;
    cold.fco = D.fco; cold.uico = D.uico; cold.fj0 = D.fj0; cold.noscco = D.noscco;
    cold.s0_bounds = D.s0_bounds; cold.s0_emin = D.s0_emin; cold.s0_inv_w = D.s0_inv_w;
    cold.shell_cut = D.shell_cut; cold.shell_alias = D.shell_alias;
    cold.espc = D.espc; cold.cutoff = D.cutoff; cold.alias = D.alias;
    cold.bricks = D.bricks;
    cold.sig_mid = D.sig_mid; cold.sig_w = D.sig_w;
    cold.wood_coarse = D.wood_coarse;
    cold.woodcock = D.woodcock;
    cold.lds = D.lds;
    for (int c = 0; c < 16; ++c) cold.brick_palette[c] = D.brick_palette[c];
    // dose tallies (read_input :1868-1893, init_CUDA_device :2636-2657,2694-2720)
    const SimConfig& cfg = H.cfg;
    if (cfg.flag_material_dose == 1) {
      D.dose_materials = D.put(std::vector<unsigned long long>(2 * kMaxMaterials, 0ULL));
      D.dose_flags |= kDoseMaterials;
    }
    if (cfg.dose_roi[1] > -1) {
      D.dose_roi_voxels = (size_t)(cfg.dose_roi[1] - cfg.dose_roi[0] + 1) * (size_t)(cfg.dose_roi[3] - cfg.dose_roi[2] + 1) *
                          (size_t)(cfg.dose_roi[5] - cfg.dose_roi[4] + 1);
      D.dose_voxels = D.put(std::vector<unsigned long long>(2 * D.dose_roi_voxels, 0ULL));
      D.dose_flags |= kDoseVoxels;
    }
    cold.dose_voxels = D.dose_voxels;
    cold.dose_materials = D.dose_materials;
    cold.mfp = D.mfp; cold.e0 = H.mat.e0; cold.ide = H.mat.ide;
    for (int k = 0; k < 3; ++k) cold.bbox[k] = H.voxels.size_bbox[k];
    for (int k = 0; k < 6; ++k) cold.dose_roi[k] = cfg.dose_roi[k];
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] >= 0) cold.material_of_compact[D.compact_of[m]] = m;
    for (int m = 0; m < kMaxMaterials; ++m) cold.shell_first[m] = D.shell_first[m];
    for (int k = 0; k < 3; ++k) { cold.objbox_lo[k] = D.objbox_lo[k]; cold.objbox_hi[k] = D.objbox_hi[k]; }
    for (int k = 0; k < 2; ++k) { cold.ell_c[k] = D.ell_c[k]; cold.ell_inv[k] = D.ell_inv[k]; }
    cold.thresh_compton = cold.thresh_rayleigh = cold.thresh_new = cold.flyable_low = cold.swap_batch = cold.trade_slots = -1;  // apply_schedule
    {
      // azimuthal aperture of the beam (the same for every projection: the pose rotates the beam frame, MC-GPU_v1.3.cu:3280-3434)
      cold.fan_ratio_lo = -3.0e38f;
      cold.fan_ratio_hi = 3.0e38f;
      bool same = !H.source.empty();
      for (const SourcePose& sp : H.source) same = same && sp.phi_low == H.source[0].phi_low && sp.D_phi == H.source[0].D_phi;
      if (same) {
        const double lo = (double)H.source[0].phi_low, hi = lo + (double)H.source[0].D_phi;
        if (lo > 1.0e-3 && hi < 3.14159265358979323846 - 1.0e-3 && hi > lo) {
          const double r_hi = std::cos(lo) / std::sin(lo), r_lo = std::cos(hi) / std::sin(hi);  // cot decreases on (0, pi)
          const double margin = 2.0e-6 * (r_hi - r_lo);
          cold.fan_ratio_lo = (float)(r_lo + margin);
          cold.fan_ratio_hi = (float)(r_hi - margin);
        }
      }
    }
    D.cold_host = cold;
    D.cold = D.put(std::vector<TrackCold>(1, cold));
    D.src_all = D.put(H.source);
    D.det_all = D.put(H.detector);
  }
  D.work_counter = D.put(std::vector<unsigned long long>((size_t)kNumCounters * kCounterStride, 0ULL));
  HIP_TRY(hipEventCreate(&D.ev_start));
  HIP_TRY(hipEventCreate(&D.ev_stop));
  apply_schedule(D);
  HIP_TRY(hipDeviceSynchronize());
}

void require(bool ok, int code, const char* msg) { if (!ok) throw Error(code, msg); }

TrackArgs make_args(const mcgpu_ctx& C, int p) {
  const HostModel& H = C.host;
  const DeviceModel& D = C.dev;
  TrackArgs A;
  memset(&A, 0, sizeof A);
  A.vol = D.vol; A.palette = D.palette; A.vol_kind = D.vol_kind; A.palette_size = D.palette_size;
  A.brick_shift = D.brick_shift; A.brick_nx = D.brick_n[0]; A.brick_nxy = D.brick_n[0] * D.brick_n[1];
  A.brick_bytes = D.vol_kind == kVolU8 ? D.brick_bytes : 0;
  A.sub = D.vol_kind == kVolU8 ? D.sub : nullptr; A.sub_nx = D.sub_n[0]; A.sub_nxy = D.sub_n[0] * D.sub_n[1];
  A.sub_kind = A.sub ? 1 : 0;
  A.rec_nx = D.rec_n[0]; A.rec_nxy = D.rec_n[0] * D.rec_n[1];
  if (D.vol_kind == kVolU8 && D.tile_rec) { A.sub = reinterpret_cast<const unsigned char*>(D.tile_rec); A.sub_kind = 2; }  // the records win over the code table
  A.lds = D.lds;
  A.nx = H.voxels.n[0]; A.ny = H.voxels.n[1]; A.nz = H.voxels.n[2]; A.nxy = A.nx * A.ny;
  for (int k = 0; k < 3; ++k) {
    A.inv_vs[k] = H.voxels.inv_voxel_size[k];
    A.bbox[k] = H.voxels.size_bbox[k];
    // upper clamp of the FAST kernel: bbox - EPS_SOURCE (MC-GPU_v1.3.h:87), lowered until it indexes the last voxel
    float hi = A.bbox[k] - 0.000015f;
    while ((int)(hi * A.inv_vs[k]) > H.voxels.n[k] - 1) hi = std::nextafter(hi, 0.0f);
    A.bbox_hi[k] = hi;
  }
  require((long long)A.nx * A.ny < (1LL << 24) && (long long)H.voxels.count() < (1LL << 31), -2,
          "!!ERROR!! voxel grid too large for the 32-bit voxel index of the kernel");
  A.e0 = H.mat.e0; A.ide = H.mat.ide; A.num_values = H.mat.num_values; A.nmat = D.nmat;
  A.woodcock = D.woodcock; A.mfp = D.mfp; A.mfp_tot = D.mfp_tot; A.sig_shift = D.sig_shift;
  A.cold = D.cold;
  A.nbins = H.spectrum.num_bins;
  A.src = D.src_all + p; A.det = D.det_all + p;
  A.stream_key = (unsigned)p;
  A.dose_flags = D.dose_flags;

  A.sched_kind = D.knobs.fast_sched;
  A.has_exterior = (D.vol_kind == kVolU8 && D.has_exterior) ? D.knobs.exterior_mode : 0;  // bit 0: hop during flight, bit 1: hop at the source
  // batching thresholds of the COMPAT kernel (lanes of a wave64 holding such a history in their registers or their parking slot)
  // The Compton batch of the COMPAT kernel walks every electron shell of the material several times in the reference's own
  // arithmetic: with tissue tables (29-40 shells) it is 60-73 % of the kernel and wants FULL batches -- threshold 40 of 64 lanes
  // instead of 20: thorax +39 %, CIRS +32 % (tools/compat_sweep.py) -- while the 4-12 shells of the Catphan's plastics prefer
  // photons back in flight early (40: -21 %).  Chosen from the mean shell count of the materials in use; tallies do not depend on it.
  // Second sweep: the tally/source batch is cheap and should not hold lanes back (24 -> 12..16 lanes), which in turn lets the
  // Compton batch wait for 48: thorax 1.9e8 -> 2.95e8, CIRS 4.6e8 -> 6.5e8, Catphan 1.58e9 -> 1.66e9 histories/s.
  // Round 3, with resumable Compton trials and two batches per lane (track_kernel.inc), the same thresholds are still the best of
  // the sweep (profiles/r03u_compat_sweep.txt): thorax 4.1e8, CIRS 9.2e8, Catphan 1.8e9; exchanging the two histories of a lane
  // pays from 1-8 takers on (thresh_take), 16+ loses.
  // Later in round 3 the Compton batch lost two thirds of its cost (S0 bounds, in-place Klein-Nishina rejections) and the wave takes
  // four flight steps between two looks at its state: 48 / 8 / 16 on tissue and 32 / 4 / 24 on plastics sit on flat optima
  // (profiles/r03y_compat_sweep.txt).
  int shells = 0, used = 0;
  for (int m = 0; m < kMaxMaterials; ++m)
    if (D.compact_of[m] >= 0) { shells += std::min(H.mat.noscco[m], kMaxShells); ++used; }
  const bool many_shells = used > 0 && shells >= 20 * used;
  A.segment_loop = D.knobs.segment_loop >= 0 ? D.knobs.segment_loop : ((A.sub_kind == 2 || many_shells) ? 1 : 0);  // track_pool.inc: kSegmentLoop
  A.thresh_compton = D.knobs.compat_thresh[0] >= 0 ? D.knobs.compat_thresh[0] : (many_shells ? 48 : 32);
  A.thresh_rayleigh = D.knobs.compat_thresh[1] >= 0 ? D.knobs.compat_thresh[1] : (many_shells ? 8 : 4);
  A.thresh_new = D.knobs.compat_thresh[2] >= 0 ? D.knobs.compat_thresh[2] : (many_shells ? 16 : 24);
  A.thresh_take = D.knobs.compat_thresh[3] >= 0 ? D.knobs.compat_thresh[3] : 2;
  // One source of truth: the FAST kernel reads mfp / e0 / ide / bbox / lds from TrackCold, written once in upload_model
  // (they would otherwise sit in scalar registers for the whole persistent loop); the COMPAT kernel reads the copies above.
  if (D.cold) {
    const TrackCold& ch = D.cold_host;
    require(ch.mfp == A.mfp && ch.e0 == A.e0 && ch.ide == A.ide && memcmp(ch.bbox, A.bbox, sizeof A.bbox) == 0 && memcmp(&ch.lds, &A.lds, sizeof A.lds) == 0, -9,
            "!!ERROR!! internal: TrackCold and the launch arguments disagree on mfp / e0 / ide / bbox / lds (upload_model and make_args have drifted apart)");
  }
  return A;
}


// After mcgpu_warp_geometry the voxels exist on the device only; whoever needs them on the host calls this first.
void sync_host_voxels(mcgpu_ctx& C) {
  if (!C.host_voxels_stale) return;
  HostModel& H = C.host;
  DeviceModel& D = C.dev;
  HIP_TRY(hipSetDevice(D.device_id));
  const size_t nvox = H.voxels.count();
  std::vector<unsigned char> idx(nvox);
  {  // the device volume is tiled (device_model.hpp: tiled_voxel)
    std::vector<unsigned char> tiled(D.vol_bytes);
    HIP_TRY(hipMemcpy(tiled.data(), D.vol, D.vol_bytes, hipMemcpyDeviceToHost));
    const int nx = H.voxels.n[0], ny = H.voxels.n[1], nz = H.voxels.n[2];
    const unsigned int snx = (unsigned int)D.sub_n[0], snxy = (unsigned int)(D.sub_n[0] * D.sub_n[1]);
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) idx[((size_t)z * ny + y) * nx + x] = tiled[tiled_voxel((unsigned)x, (unsigned)y, (unsigned)z, snx, snxy)];
  }
  int mat_of[256];
  float dens_of[256];
  for (int e = 0; e < D.palette_size; ++e) {
    int mc;
    memcpy(&mc, &D.palette_host[2 * e + 1], 4);
    int number = 1;
    for (int m = 0; m < kMaxMaterials; ++m)
      if (D.compact_of[m] == mc) number = m + 1;
    mat_of[e] = number;
    dens_of[e] = D.palette_host[2 * e];
  }
  for (size_t i = 0; i < nvox; ++i) { H.voxels.material[i] = (uint8_t)mat_of[idx[i]]; H.voxels.density[i] = dens_of[idx[i]]; }
  C.host_voxels_stale = false;
}

const void* host_table(mcgpu_ctx& C, const std::string& name, size_t& bytes) {
  HostModel& H = C.host;
  if (name == "voxel_mat_dens") sync_host_voxels(C);
  auto cache = [&](const void* p, size_t n) -> const void* {
    auto& v = C.table_cache[name];
    v.assign((const unsigned char*)p, (const unsigned char*)p + n);
    bytes = n;
    return v.data();
  };
#define DIRECT(vec) do { bytes = (vec).size() * sizeof((vec)[0]); return (const void*)(vec).data(); } while (0)
  if (name == "source_data") DIRECT(H.source);
  if (name == "detector_data") DIRECT(H.detector);
  if (name == "mfp_woodcock") DIRECT(H.mat.woodcock);
  if (name == "woodcock_coarse") {  // what the FAST kernel stages in LDS (LdsLayout::wood), computed from the table above
    const std::vector<float> w = coarse_woodcock(H);
    return cache(w.data(), w.size() * sizeof(float));
  }
  if (name == "mfp_a") DIRECT(H.mat.a);
  if (name == "mfp_b") DIRECT(H.mat.b);
  if (name == "xco") DIRECT(H.mat.xco);
  if (name == "pco") DIRECT(H.mat.pco);
  if (name == "aco") DIRECT(H.mat.aco);
  if (name == "bco") DIRECT(H.mat.bco);
  if (name == "pmax") DIRECT(H.mat.pmax);
  if (name == "itlco") DIRECT(H.mat.itlco);
  if (name == "ituco") DIRECT(H.mat.ituco);
  if (name == "fco") DIRECT(H.mat.fco);
  if (name == "uico") DIRECT(H.mat.uico);
  if (name == "fj0") DIRECT(H.mat.fj0);
#undef DIRECT
  if (name == "s0_bounds") {  // COMPAT kernel: {lo, hi} of S0 per (material number - 1, energy bin), then {emin, 1 / bin width}
    float emin = 0.f, inv_w = 0.f;
    std::vector<float> b = build_s0_bounds(H, nullptr, kMaxMaterials, &emin, &inv_w);
    b.push_back(emin);
    b.push_back(inv_w);
    return cache(b.data(), b.size() * sizeof(float));
  }
  if (name == "noscco") return cache(H.mat.noscco, sizeof H.mat.noscco);
  if (name == "espc") return cache(H.spectrum.espc, sizeof H.spectrum.espc);
  if (name == "espc_cutoff") return cache(H.spectrum.cutoff, sizeof H.spectrum.cutoff);
  if (name == "espc_alias") return cache(H.spectrum.alias, sizeof H.spectrum.alias);
  if (name == "density_max") return cache(H.voxels.density_max, sizeof H.voxels.density_max);
  if (name == "density_nominal") return cache(H.mat.density_nominal, sizeof H.mat.density_nominal);
  if (name == "voxel_size") return cache(H.voxels.voxel_size, sizeof H.voxels.voxel_size);
  if (name == "inv_voxel_size") return cache(H.voxels.inv_voxel_size, sizeof H.voxels.inv_voxel_size);
  if (name == "size_bbox") return cache(H.voxels.size_bbox, sizeof H.voxels.size_bbox);
  if (name == "voxel_mat_dens") {  // reference layout: float2 {material + 0.0001f, density} (MC-GPU_v1.3.cu:2135-2136)
    auto& v = C.table_cache[name];
    const size_t n = H.voxels.count();
    v.resize(n * 8);
    float* f = (float*)v.data();
    for (size_t i = 0; i < n; ++i) { f[2 * i] = (float)(H.voxels.material[i]) + 0.0001f; f[2 * i + 1] = H.voxels.density[i]; }
    bytes = v.size();
    return v.data();
  }
  return nullptr;
}


}  // namespace mcgpu
