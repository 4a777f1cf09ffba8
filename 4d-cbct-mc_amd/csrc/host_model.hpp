// host_model.hpp -- host-side data model of the MI355X-native MC CBCT projection engine.
//
// Everything the photon-history kernel needs, as parsed from the reference's wire formats
// (SURVEY.md Appendix A): the MC-GPU `.in` file, the (gzip) text voxel file, the `.mcgpu`
// material files and the `.spc` spectrum.  Arithmetic follows the reference host code
// (docker/mcgpu/MC-GPU_v1.3.cu: read_input :1240-1895, load_voxels :1996-2145, load_material
// :2177-2443, set_CT_trajectory :3280-3434, init_energy_spectrum/IRND0 :3498-3734) so that
// the resulting tables are bit-identical to the reference's; the code itself is new.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mcgpu {

constexpr int kMaxMaterials = 25;        // MC-GPU_v1.3.h:64
constexpr int kMaxShells = 40;           // MC-GPU_v1.3.h:65
constexpr int kRayleighPoints = 128;     // MC-GPU_v1.3.h:66
constexpr int kMaxRayleighBins = 25005;  // MC-GPU_v1.3.h:67
constexpr int kMaxSpectrumBins = 256;    // MC-GPU_v1.3.h:70
constexpr int kMaxProjections = 1024;    // MC-GPU_v1.3.h:59
constexpr double kPi = 3.14159265358979323846;

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// One x-ray source pose per projection.  Field order == reference `source_struct`
// (MC-GPU_v1.3.h:155-169, 80 bytes) so host dumps can be compared byte for byte.
struct SourcePose {
  float pos[3];
  float dir[3];
  float rot_fan[9];
  float cos_theta_low, phi_low, D_cos_theta, D_phi, max_height_at_y1cm;
};
static_assert(sizeof(SourcePose) == 80, "SourcePose must match the reference wire layout");

// One detector pose per projection.  Field order == reference `detector_struct`
// (MC-GPU_v1.3.h:190-208, 100 bytes).
struct DetectorPose {
  float sdd, lateral_displacement;
  float corner_min[3];
  float center[3];
  float rot_inv[9];
  float width_X, height_Z, inv_pixel_size_X, inv_pixel_size_Z;
  int nx, nz;
  int total_pixels;
  int rotation_flag;
};
static_assert(sizeof(DetectorPose) == 100, "DetectorPose must match the reference wire layout");

struct Spectrum {  // reference `source_energy_struct`, MC-GPU_v1.3.h:173-183
  int num_bins = 0;
  float espc[kMaxSpectrumBins] = {};
  float cutoff[kMaxSpectrumBins] = {};
  short alias[kMaxSpectrumBins] = {};
  float mean_energy = 0.f;
};

struct VoxelGrid {  // reference `voxel_struct` + voxel array (kept split: 5 B/voxel instead of 8)
  int n[3] = {0, 0, 0};
  float voxel_size[3] = {0, 0, 0};
  float inv_voxel_size[3] = {0, 0, 0};
  float size_bbox[3] = {0, 0, 0};
  std::vector<uint8_t> material;  // 1-based material number, x fastest
  std::vector<float> density;     // g/cm^3
  float density_max[kMaxMaterials];
  size_t count() const { return (size_t)n[0] * n[1] * n[2]; }
};

struct Float3 { float x, y, z; };
struct Float2 { float x, y; };

struct MaterialTables {
  int num_values = 0;   // energy grid points (24001 for the bundled 5-125 keV files)
  float e0 = 0.f, ide = 0.f;
  double delta_e = 0.0;
  float density_nominal[kMaxMaterials];
  bool used[kMaxMaterials];
  std::vector<Float2> woodcock;  // [num_values]   mfp_min(E) = x + E*y
  std::vector<double> mfp_total_file;  // [num_values*25] total mean free path as read (what the Woodcock minimum is taken over)
  std::vector<Float3> a, b;      // [num_values*25] {total, Compton, Rayleigh} inverse MFP / rho
  // Rayleigh (reference `rayleigh_struct`, MC-GPU_v1.3.h:251-264)
  std::vector<float> xco, pco, aco, bco;  // [128*25], index i + 128*mat
  std::vector<float> pmax;                // [25005*25], index bin*25 + mat
  std::vector<uint8_t> itlco, ituco;      // [128*25]
  // Compton (reference `compton_struct`, MC-GPU_v1.3.h:238-248)
  std::vector<float> fco, uico, fj0;      // [25*40], index mat + 25*shell
  int noscco[kMaxMaterials];
};

struct SimConfig {
  unsigned long long total_histories = 0;
  int seed = 0, gpu_id = 0, threads_per_block = 0, histories_per_thread = 0;
  int num_projections = 1;
  int enable_specific_angles = 0;
  std::vector<float> specific_angles;
  double D_angle = -1.0, angularROI_0 = 0.0, angularROI_1 = 360.0, initial_angle = 0.0;
  double SRotAxisD = -1.0, vertical_translation = 0.0;
  int flag_material_dose = 0;
  int dose_roi[6] = {32500, -32500, 32500, -32500, 32500, -32500};  // xmin,xmax,ymin,ymax,zmin,zmax (0-based)
  std::string file_voxels, file_output, file_dose_output, file_spectrum;
  std::vector<std::string> file_materials;  // up to 25, may contain empty strings
};

struct HostModel {
  SimConfig cfg;
  std::vector<SourcePose> source;      // [num_projections]
  std::vector<DetectorPose> detector;  // [num_projections]
  Spectrum spectrum;
  VoxelGrid voxels;
  MaterialTables mat;
};

// ---- parsers / builders (host_model.cpp) ----
void parse_input_file(const std::string& path, HostModel& m);              // read_input
void build_ct_trajectory(HostModel& m);                                    // set_CT_trajectory
void load_spectrum(const std::string& path, Spectrum& s);                  // init_energy_spectrum + IRND0
void load_voxel_file(const std::string& path, VoxelGrid& v, int n_threads = 0);  // load_voxels
void load_material_files(const std::vector<std::string>& files, const VoxelGrid& v, MaterialTables& t);  // load_material
void load_model(const std::string& input_path, HostModel& m);              // all of the above, reference order
// Woodcock majorant table from the total mean free paths as read and the largest density of every material in the volume
// (load_material :2294-2296 and :2433-2441): all a geometry change needs when the set of materials stays the same
void rebuild_woodcock(MaterialTables& t, const float density_max[kMaxMaterials]);

// RANECU seed stepping between projections (update_seed_PRNG, MC-GPU_v1.3.cu:3456-3485)
int ranecu_mul_mod(int m, int a, int s);                                   // abMODm, MC-GPU_kernel_v1.3.cu:919
int ranecu_advance_seed(int batch_number, unsigned long long total_histories, int seed);

// Launch sizing (MC-GPU_v1.3.cu:823-841): returns blocks; may raise hpt; total = blocks*threads*hpt.
struct LaunchShape { int blocks; int threads; int hpt; unsigned long long total_histories; };
LaunchShape reference_launch_shape(unsigned long long histories, int threads_per_block, int hpt);

// Projection angle bookkeeping of report_image (MC-GPU_v1.3.cu:2787-2803)
std::string projection_file_name(const HostModel& m, int p);
// ASCII projection writer (report_image, MC-GPU_v1.3.cu:2783-2953); returns bytes written.
size_t write_projection_ascii(const HostModel& m, int p, const uint64_t* image, unsigned long long total_histories,
                              double seconds, const std::string& file_name, int n_threads = 0);

// The same file from data lines formatted elsewhere (ascii_device.hip): NORM of the pixel values, and the writer
double projection_norm(const HostModel& m, unsigned long long total_histories);
size_t write_projection_preformatted(const HostModel& m, int p, const char* text, size_t text_bytes, double energy_integral, double maximum,
                                     long max_pixel, unsigned long long total_histories, double seconds, const std::string& file_name,
                                     int n_threads = 0);

// Dose reports (report_voxels_dose :2976-3199, report_materials_dose :3214-3262, material masses :579-585); the text the
// reference prints to stdout is appended to `log`.
void material_masses(const HostModel& m, double mass[kMaxMaterials]);
void write_voxel_dose_report(const HostModel& m, const uint64_t* voxels_edep, unsigned long long histories_per_projection,
                             double seconds, std::string& log);
void format_materials_dose_report(const HostModel& m, const uint64_t* materials_dose, unsigned long long histories_per_projection,
                                  const double* mass, std::string& log);

// Projection post-processing and MetaImage stacks (postprocess.cpp)
struct MhaStack;
void finalize_projection_host(const HostModel& m, const uint64_t* image, unsigned long long total_histories, int crop_nx, float* planes,
                              int n_threads = 0);
MhaStack* mha_create(const std::string& path, int nx, int ny, int nslices, double sx, double sy);
void mha_append(MhaStack* s, const float* plane);
void mha_write_slice(MhaStack* s, int k, const float* plane);
float mha_finish(MhaStack* s, bool replace_zeros);
void mha_read(const std::string& path, int dims[3], std::vector<float>& data);
void gaussian_filter_2d(float* img, int ny, int nx, double sigma_y, double sigma_x);
void normalize_stack(const std::string& total_path, const std::string& air_path, double sigma_y, double sigma_x, const std::string& out_path,
                     double sx, double sy);

// Fast text/binary voxel writers (cbctmc/mc/voxel_data.pyx + mcgpu_geometry header fields)
size_t write_voxel_file(const std::string& path, const int n[3], const float spacing_cm[3], const uint8_t* material,
                        const float* density, bool gzip);
// Binary sidecar `<name>.voxbin` of a voxel file: the arrays the text parse would yield
// (densities quantised through "%.6f" like the text), palette-compressed when the volume holds <= 65536 distinct
// (material, density) pairs.  load_model() prefers a sidecar that is not older than the text file.
size_t write_voxel_binary(const std::string& path, const int n[3], const float spacing_cm[3], const uint8_t* material,
                          const float* density);
void load_voxel_binary(const std::string& path, VoxelGrid& v);
std::string voxel_sidecar_path(const std::string& voxel_file);  // geometry.vox[.gz] -> geometry.voxbin

}  // namespace mcgpu
