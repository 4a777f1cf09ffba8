// host_model.cpp -- parsers and table builders for the reference's wire formats.
//
// New code; the arithmetic (which operand is float, which is double, in which order) follows
// the reference host code so that every table is bit-identical to what MC-GPU v1.3 builds
// from the same files.  Reference citations are docker/mcgpu/MC-GPU_v1.3.cu:<line>.
#include "host_model.hpp"
#include "knobs.hpp"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace mcgpu {
namespace {

// The reference expands DEG2RAD/RAD2DEG textually without parentheses (MC-GPU_v1.3.h:73-75):
// x*DEG2RAD == (x*PI)/180.0 and x*RAD2DEG == (x*180.0)/PI.  Same rounding here.
inline double deg2rad(double x) { return x * kPi / 180.0; }
inline double rad2deg(double x) { return x * 180.0 / kPi; }

[[noreturn]] void fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

// Line reader over a whole text file with the reference's two flavours of reading:
// raw lines (fgets, <=249 chars) and "trimmed" lines (fgets_trimmed, :1935-1965).
class TextFile {
 public:
  explicit TextFile(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) fail(-1, "!!read_input ERROR!! Input file not found or not readable. Input file name: '%s'", path.c_str());
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data_.append(buf, n);
    fclose(f);
  }
  // fgets(line, 250, f): at most 249 characters, newline kept.
  bool raw(std::string& line) {
    if (pos_ >= data_.size()) return false;
    size_t end = pos_, lim = std::min(data_.size(), pos_ + 249);
    while (end < lim && data_[end] != '\n') ++end;
    if (end < lim) ++end;  // include '\n'
    line.assign(data_, pos_, end - pos_);
    pos_ = end;
    return true;
  }
  // Skip blank/comment-only lines, strip leading blanks, cut at '#' or end of line.
  bool trimmed(std::string& out) {
    std::string line;
    out.clear();
    while (out.empty()) {
      if (!raw(line)) return false;
      size_t a = 0;
      while (a < line.size() && line[a] == ' ') ++a;
      size_t b = a;
      while (b < line.size() && line[b] != '\n' && line[b] != '#') ++b;
      out.assign(line, a, b - a);
    }
    return true;
  }
  // Advance until a raw line contains `marker`.
  void seek(const char* marker, const char* what) {
    std::string line;
    while (raw(line))
      if (line.find(marker) != std::string::npos) return;
    fail(-2, "!!read_input ERROR!! Input file is not readable or does not contain the string '%s'!!", what);
  }

 private:
  std::string data_;
  size_t pos_ = 0;
};

// trim_name (:1907-1926): leading blanks dropped, stop at blank or '#'.
std::string trim_name(const std::string& s) {
  size_t a = 0;
  while (a < s.size() && s[a] == ' ') ++a;
  size_t b = a;
  while (b < s.size() && s[b] != ' ' && s[b] != '#' && s[b] != '\n' && s[b] != '\r' && s[b] != '\t') ++b;
  return s.substr(a, b - a);
}

bool starts_yes(const std::string& s) { return !s.compare(0, 2, "YE") || !s.compare(0, 2, "Ye") || !s.compare(0, 2, "ye"); }
bool starts_no(const std::string& s) { return !s.compare(0, 2, "NO") || !s.compare(0, 2, "No") || !s.compare(0, 2, "no"); }

// Detector-to-+Y rotation (rot_inv) from rotation angles about X and Z (:1766-1781, :3381-3393).
void set_rot_inv(float* r, double rotX, double rotZ) {
  const double cX = cos(rotX), cZ = cos(rotZ), sX = sin(rotX), sZ = sin(rotZ);
  r[0] = (float)cZ;         r[1] = (float)(-sZ);      r[2] = 0.0f;
  r[3] = (float)(cX * sZ);  r[4] = (float)(cX * cZ);  r[5] = (float)(-sX);
  r[6] = (float)(sX * sZ);  r[7] = (float)(sX * cZ);  r[8] = (float)cX;
}
// +Y-to-beam rotation for the fan source (rot_fan) (:1825-1838, :3408-3421).
void set_rot_fan(float* r, double rotX, double rotZ) {
  const double cX = cos(rotX), cZ = cos(rotZ), sX = sin(rotX), sZ = sin(rotZ);
  r[0] = (float)cZ;  r[1] = (float)(-cX * sZ);  r[2] = (float)(sX * sZ);
  r[3] = (float)sZ;  r[4] = (float)(cX * cZ);   r[5] = (float)(-sX * cZ);
  r[6] = 0.0f;       r[7] = (float)sX;          r[8] = (float)cX;
}
// Angle that brings direction (u,v) onto +Y by a rotation about Z (:1755-1764, :3370-3376).
double rot_z_to_plus_y(float u, float v) {
  if ((u * u + v * v) > 1.0e-8) {
    const double c = acos((double)u / sqrt((double)(u * u + v * v)));  // float sum of squares, double sqrt/divide (C semantics of the reference)
    return (v >= 0.0f) ? 0.5 * kPi - c : 0.5 * kPi - (-c);
  }
  return 0.0;
}
void corner_from_center(DetectorPose& d) {
  const float cx = d.center[0], cy = d.center[1], cz = d.center[2];
  d.corner_min[0] = cx * d.rot_inv[0] + cy * d.rot_inv[1] + cz * d.rot_inv[2];
  d.corner_min[1] = cx * d.rot_inv[3] + cy * d.rot_inv[4] + cz * d.rot_inv[5];
  d.corner_min[2] = cx * d.rot_inv[6] + cy * d.rot_inv[7] + cz * d.rot_inv[8];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// `.in` file  (read_input, :1240-1895; format: SURVEY.md Appendix A.1)
// ---------------------------------------------------------------------------------------------
void parse_input_file(const std::string& path, HostModel& m) {
  TextFile f(path);
  SimConfig& c = m.cfg;
  std::string line;
  m.source.assign(1, SourcePose{});
  m.detector.assign(1, DetectorPose{});
  SourcePose& s0 = m.source[0];
  DetectorPose& d0 = m.detector[0];

  // -- SECTION SIMULATION CONFIG (:1280-1312)
  f.seek("SECTION SIMULATION CONFIG v.2009-05-12", "SECTION SIMULATION CONFIG v.2009-05-12");
  f.trimmed(line);
  c.total_histories = (unsigned long long)(strtod(line.c_str(), nullptr) + 0.0001);
  f.trimmed(line); c.seed = (int)strtol(line.c_str(), nullptr, 10);
  f.trimmed(line); c.gpu_id = (int)strtol(line.c_str(), nullptr, 10);
  f.trimmed(line); c.threads_per_block = (int)strtol(line.c_str(), nullptr, 10);
  if (c.threads_per_block <= 0 || (c.threads_per_block % 32) != 0)
    fail(-2, "!!read_input ERROR!! The input number of GPU threads per block must be a multiple of 32. Input value: %d !!",
         c.threads_per_block);
  f.trimmed(line); c.histories_per_thread = (int)strtol(line.c_str(), nullptr, 10);
  if (c.histories_per_thread <= 0) fail(-2, "!!read_input ERROR!! Histories per thread must be positive.");

  // -- SECTION SOURCE (:1315-1395)
  f.seek("SECTION SOURCE v.2011-07-12", "SECTION SOURCE v.2011-07-12");
  f.trimmed(line); c.file_spectrum = trim_name(line);
  f.trimmed(line);
  if (sscanf(line.c_str(), "%f %f %f", &s0.pos[0], &s0.pos[1], &s0.pos[2]) != 3)
    fail(-2, "!!read_input ERROR!! Could not read the source position.");
  f.trimmed(line);
  if (sscanf(line.c_str(), "%f %f %f", &s0.dir[0], &s0.dir[1], &s0.dir[2]) != 3)
    fail(-2, "!!read_input ERROR!! Could not read the source direction.");
  {
    const double inv = 1.0 / sqrt((double)(s0.dir[0] * s0.dir[0] + s0.dir[1] * s0.dir[1] + s0.dir[2] * s0.dir[2]));
    for (int k = 0; k < 3; ++k) s0.dir[k] = (float)(((double)s0.dir[k]) * inv);
  }
  f.trimmed(line);
  double phi1 = 0, phi2 = 0, theta = 0;
  if (sscanf(line.c_str(), "%lf %lf %lf", &phi1, &phi2, &theta) != 3)
    fail(-2, "!!read_input ERROR!! Expecting three fan beam aperture angles (phi1 phi2 theta).");
  double phi = phi1 + phi2;
  if (theta > 180.0) fail(-2, "!!read_input ERROR!! Input polar aperture must be in [0,180] deg.!");
  if (phi > 360.0) fail(-2, "!!read_input ERROR!! Input azimuthal aperture must be in [0,360] deg.!");
  s0.cos_theta_low = (float)cos(deg2rad(90.0 - 0.5 * theta));
  s0.D_cos_theta = (float)(-2.0 * s0.cos_theta_low);
  s0.phi_low = (float)deg2rad(90.0 - phi1);
  s0.D_phi = (float)deg2rad(phi);
  s0.max_height_at_y1cm = (float)tan(deg2rad(0.5 * theta));
  // Pencil beams.  The production (C++/nvcc) reference resolves abs(double) to fabs (:1383,:1390).
  if (fabs(theta) < 1.0e-7) { theta = 1.0e-7; s0.cos_theta_low = 0.f; s0.D_cos_theta = 0.f; s0.max_height_at_y1cm = 0.f; }
  if (fabs(phi) < 1.0e-7) { phi = 1.0e-7; s0.phi_low = (float)deg2rad(90.0); s0.D_phi = 0.f; }

  // -- SECTION IMAGE DETECTOR (:1398-1465)
  f.seek("SECTION IMAGE DETECTOR v.2009-12-02", "SECTION IMAGE DETECTOR v.2009-12-02");
  f.trimmed(line); c.file_output = trim_name(line);
  f.trimmed(line);
  float fx = 0, fz = 0;
  if (sscanf(line.c_str(), "%f %f", &fx, &fz) != 2) fail(-2, "!!read_input ERROR!! Could not read the number of pixels.");
  d0.nx = (int)(fx + 0.001f);
  d0.nz = (int)(fz + 0.001f);
  d0.total_pixels = d0.nx * d0.nz;
  if (d0.total_pixels < 1 || d0.total_pixels > 99999999)
    fail(-2, "!!read_input ERROR!! The input number of pixels is incorrect. Input: X_pix = %d, Y_pix = %d, total_num_pix = %d!!",
         d0.nx, d0.nz, d0.total_pixels);
  f.trimmed(line);
  if (sscanf(line.c_str(), "%f %f", &d0.width_X, &d0.height_Z) != 2) fail(-2, "!!read_input ERROR!! Could not read the image size.");
  d0.inv_pixel_size_X = d0.nx / d0.width_X;
  d0.inv_pixel_size_Z = d0.nz / d0.height_Z;
  f.trimmed(line); d0.sdd = strtof(line.c_str(), nullptr);
  f.trimmed(line); d0.lateral_displacement = strtof(line.c_str(), nullptr);  // parsed, never used in transport
  float dc[3];
  for (int k = 0; k < 3; ++k) dc[k] = s0.pos[k] + s0.dir[k] * d0.sdd;
  if (d0.sdd < 1.0e-6) fail(-2, "!!read_input ERROR!! The source-to-detector distance must be positive. Input: sdd=%f!!", d0.sdd);
  if (phi < -1.0e-7) {  // fit to detector width (:1451-1457; D_phi keeps the negative input, as in the reference)
    phi1 = rad2deg(atan((d0.width_X / 2.0) / d0.sdd));
    phi2 = phi1;
    s0.phi_low = (float)deg2rad(90.0 - phi1);
    s0.D_phi = (float)deg2rad(phi);
  }
  if (theta < -1.0e-7) {  // fit to detector height (:1459-1465)
    theta = rad2deg(2.0 * atan(0.5 * d0.height_Z / d0.sdd));
    s0.cos_theta_low = (float)cos(deg2rad(90.0 - 0.5 * theta));
    s0.D_cos_theta = (float)(-2.0 * s0.cos_theta_low);
    s0.max_height_at_y1cm = (float)tan(deg2rad(0.5 * theta));
  }

  // -- SECTION ANGLES OF PROJ (:1472-1533)
  f.seek("SECTION ANGLES OF PROJ v.2023-09-06", "SECTION ANGLES OF PROJ v.2023-09-06");
  f.trimmed(line);
  if (starts_yes(line)) c.enable_specific_angles = 1;
  else if (starts_no(line)) c.enable_specific_angles = 0;
  else fail(-2, "!!read_input ERROR!! Answer YES or NO in the first line of 'SECTION ANGLES OF PROJ'. Input text: %s", line.c_str());
  c.specific_angles.clear();
  for (;;) {
    if (!f.raw(line)) fail(-2, "!!read_input ERROR!! Input file does not contain the string 'SECTION CT SCAN TRAJECTORY v.2011-10-25'!!");
    if (line.find("SECTION CT SCAN TRAJECTORY v.2011-10-25") != std::string::npos) break;
    float ang = 99999.f;
    if (sscanf(line.c_str(), "%f", &ang) == 1 && ang != 99999.f) {
      if ((int)c.specific_angles.size() >= kMaxProjections) fail(-2, "!!Too many angles are specified (maximum %d)!!", kMaxProjections);
      c.specific_angles.push_back(ang);
    }
  }
  if (c.enable_specific_angles == 1 && c.specific_angles.empty())
    fail(-2, "!!read_input ERROR!! In 'SECTION ANGLES OF PROJ' no angle was specified!!");

  // -- SECTION CT SCAN TRAJECTORY (:1535-1615)
  f.trimmed(line);
  c.num_projections = (int)strtol(line.c_str(), nullptr, 10);
  if (c.num_projections == 0) c.num_projections = 1;
  if (c.enable_specific_angles == 1) c.num_projections = (int)c.specific_angles.size();
  if (c.num_projections < 0) fail(-2, "!!read_input ERROR!! A negative number of projections is not supported.");
  if (c.num_projections > 1 && fabs(s0.dir[2]) > 0.00001f)
    fail(-2, "!!read_input ERROR!! CT scans can only be simulated when the source direction is perpendicular to the Z axis (w=0).");
  if (c.num_projections > kMaxProjections)
    fail(-2, "!!read_input ERROR!! The input number of projections is too large (MAX_NUM_PROJECTIONS=%d).", kMaxProjections);
  if (c.num_projections != 1 || c.enable_specific_angles == 1) {
    f.trimmed(line);
    c.D_angle = deg2rad(strtod(line.c_str(), nullptr));
    double a = acos((double)s0.dir[0]);
    if (s0.dir[1] < 0) a = -a;
    if (a < 0.0) a += 2.0 * kPi;
    a -= kPi;
    if (a < 0.0) a += 2.0 * kPi;
    if (c.enable_specific_angles == 1) {
      a = deg2rad(c.specific_angles[0]);
      if (a >= (2.0 * kPi - 0.0001)) a -= 2.0 * kPi;
    }
    c.initial_angle = a;
    f.trimmed(line);
    double r0 = 0, r1 = 0;
    sscanf(line.c_str(), "%lf %lf", &r0, &r1);
    c.angularROI_0 = deg2rad(r0 - 0.00001);
    c.angularROI_1 = deg2rad(r1 + 0.00001);
    f.trimmed(line);
    c.SRotAxisD = strtod(line.c_str(), nullptr);
    if (c.SRotAxisD < 0.0 || c.SRotAxisD > d0.sdd)
      fail(-2, "!!read_input ERROR!! Invalid source-to-rotation axis distance! Input: %f (sdd=%f).", c.SRotAxisD, d0.sdd);
    f.trimmed(line);
    c.vertical_translation = strtod(line.c_str(), nullptr);
  }

  // -- SECTION DOSE DEPOSITION (:1619-1709)
  f.seek("SECTION DOSE DEPOSITION v.2012-12-12", "SECTION DOSE DEPOSITION v.2012-12-12");
  f.trimmed(line);
  if (starts_yes(line)) c.flag_material_dose = 1;
  else if (starts_no(line)) c.flag_material_dose = 0;
  else fail(-2, "!!read_input ERROR!! Answer YES or NO in the first two lines of 'SECTION DOSE DEPOSITION'. Input text: %s", line.c_str());
  f.trimmed(line);
  if (starts_yes(line)) {
    f.trimmed(line); c.file_dose_output = trim_name(line);
    for (int ax = 0; ax < 3; ++ax) {
      f.trimmed(line);
      int lo = 0, hi = 0;
      sscanf(line.c_str(), "%d %d", &lo, &hi);
      c.dose_roi[2 * ax] = lo - 1;
      c.dose_roi[2 * ax + 1] = hi - 1;
    }
    for (int ax = 0; ax < 3; ++ax)
      if (c.dose_roi[2 * ax] > c.dose_roi[2 * ax + 1] || c.dose_roi[2 * ax] < 0)
        fail(-2, "!!read_input ERROR!! The input region-of-interest in 'SECTION DOSE DEPOSITION' is not valid.");
  } else if (starts_no(line)) {
    for (int ax = 0; ax < 3; ++ax) { c.dose_roi[2 * ax] = 32500; c.dose_roi[2 * ax + 1] = -32500; }
  } else {
    fail(-2, "!!read_input ERROR!! Answer YES or NO in the first two lines of 'SECTION DOSE DEPOSITION'. Input text: %s", line.c_str());
  }

  // -- SECTION VOXELIZED GEOMETRY FILE / MATERIAL FILE LIST (:1713-1745)
  f.seek("SECTION VOXELIZED GEOMETRY FILE v.2009-11-30", "SECTION VOXELIZED GEOMETRY FILE v.2009-11-30");
  f.trimmed(line); c.file_voxels = trim_name(line);
  f.seek("SECTION MATERIAL", "SECTION MATERIAL FILE LIST");
  c.file_materials.clear();
  for (int i = 0; i < kMaxMaterials; ++i) {
    if (!f.trimmed(line)) break;
    c.file_materials.push_back(trim_name(line));
  }

  // -- projection-0 detector rotation (:1750-1814)
  const double rotX0 = acos((double)s0.dir[2]) - 0.5 * kPi;
  const double rotZ0 = rot_z_to_plus_y(s0.dir[0], s0.dir[1]);
  set_rot_inv(d0.rot_inv, rotX0, rotZ0);
  for (int k = 0; k < 3; ++k) d0.center[k] = dc[k];
  if (s0.dir[1] > 0.99999f && c.num_projections == 1) {
    d0.rotation_flag = 0;
    for (int k = 0; k < 3; ++k) d0.corner_min[k] = dc[k];
  } else {
    d0.rotation_flag = 1;
    corner_from_center(d0);
  }
  d0.corner_min[0] = (float)(d0.corner_min[0] - 0.5 * d0.width_X);
  d0.corner_min[2] = (float)(d0.corner_min[2] - 0.5 * d0.height_Z);
  // -- projection-0 fan rotation (:1820-1841)
  if (d0.rotation_flag == 1) {
    const double rX = 0.5 * kPi - acos((double)s0.dir[2]);
    const double rZ = atan2((double)s0.dir[1], (double)s0.dir[0]) - 0.5 * kPi;
    set_rot_fan(s0.rot_fan, rX, rZ);
  }
}

// ---------------------------------------------------------------------------------------------
// CT trajectory (set_CT_trajectory, :3280-3434)
// ---------------------------------------------------------------------------------------------
void build_ct_trajectory(HostModel& m) {
  const SimConfig& c = m.cfg;
  const int np = c.num_projections;
  if (np == 1) return;
  m.source.resize(np);
  m.detector.resize(np);
  const SourcePose s0 = m.source[0];
  const DetectorPose d0 = m.detector[0];
  float cr[3];
  cr[0] = (float)(s0.pos[0] + s0.dir[0] * c.SRotAxisD);
  cr[1] = (float)(s0.pos[1] + s0.dir[1] * c.SRotAxisD);
  cr[2] = s0.pos[2];
  double ang;
  if (c.enable_specific_angles == 0) {
    ang = acos((double)s0.dir[0]);
    if (s0.dir[1] < 0) ang = -ang;
    if (ang < 0.0) ang += 2.0 * kPi;
    ang -= kPi;
    if (ang < 0.0) ang += 2.0 * kPi;
  } else {
    ang = deg2rad(c.specific_angles[0]);
    if (ang >= (2.0 * kPi - 0.0001)) ang -= 2.0 * kPi;
  }
  for (int i = 1; i < np; ++i) {
    SourcePose& s = m.source[i];
    DetectorPose& d = m.detector[i];
    s = SourcePose{};
    d = DetectorPose{};
    s.cos_theta_low = s0.cos_theta_low; s.phi_low = s0.phi_low; s.D_cos_theta = s0.D_cos_theta;
    s.D_phi = s0.D_phi; s.max_height_at_y1cm = s0.max_height_at_y1cm;
    d.sdd = d0.sdd; d.lateral_displacement = d0.lateral_displacement; d.width_X = d0.width_X; d.height_Z = d0.height_Z;
    d.inv_pixel_size_X = d0.inv_pixel_size_X; d.inv_pixel_size_Z = d0.inv_pixel_size_Z;
    d.nx = d0.nx; d.nz = d0.nz; d.total_pixels = d0.total_pixels; d.rotation_flag = d0.rotation_flag;
    if (c.enable_specific_angles) {
      ang = deg2rad(c.specific_angles[i]);
      if (ang >= (2.0 * kPi - 0.0001)) ang -= 2.0 * kPi;
    } else {
      ang += c.D_angle;
      if (ang >= (2.0 * kPi - 0.0001)) ang -= 2.0 * kPi;
    }
    s.pos[0] = (float)(cr[0] + c.SRotAxisD * cos(ang));
    s.pos[1] = (float)(cr[1] + c.SRotAxisD * sin(ang));
    s.pos[2] = (float)(m.source[i - 1].pos[2] + c.vertical_translation);
    s.dir[0] = cr[0] - s.pos[0];
    s.dir[1] = cr[1] - s.pos[1];
    s.dir[2] = 0.0f;
    const double nrm = 1.0 / sqrt((double)s.dir[0] * (double)s.dir[0] + (double)s.dir[1] * (double)s.dir[1]);
    s.dir[0] = (float)(((double)s.dir[0]) * nrm);
    s.dir[1] = (float)(((double)s.dir[1]) * nrm);
    d.center[0] = s.pos[0] + s.dir[0] * d.sdd;
    d.center[1] = s.pos[1] + s.dir[1] * d.sdd;
    d.center[2] = s.pos[2];
    const double rotZ = rot_z_to_plus_y(s.dir[0], s.dir[1]);
    set_rot_inv(d.rot_inv, 0.0, rotZ);
    corner_from_center(d);
    d.corner_min[0] = (float)(d.corner_min[0] - 0.5 * d.width_X);
    d.corner_min[2] = (float)(d.corner_min[2] - 0.5 * d.height_Z);
    set_rot_fan(s.rot_fan, 0.0, -rotZ);
  }
}

// ---------------------------------------------------------------------------------------------
// Spectrum + Walker alias tables (init_energy_spectrum :3498-3587, IRND0 :3675-3734)
// ---------------------------------------------------------------------------------------------
void load_spectrum(const std::string& path, Spectrum& s) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) fail(-1, "!!init_energy_spectrum ERROR!! Error trying to read the energy spectrum input file \"%s\".", path.c_str());
  fclose(fp);
  TextFile f(path);
  s = Spectrum{};
  float prob_bin[kMaxSpectrumBins];
  float e_low = 0.f, prob = 0.f;
  int bin = -1;
  std::string line;
  do {
    ++bin;
    if (bin >= kMaxSpectrumBins)
      fail(-1, "!!init_energy_spectrum ERROR!!: too many energy bins in the input spectrum (MAX_ENERGY_BINS=%d).", kMaxSpectrumBins);
    if (!f.trimmed(line))
      fail(-1, "!!init_energy_spectrum ERROR!! The input file for the x ray spectrum (%s) is not readable or incomplete "
               "(a negative probability marks the end of the spectrum).", path.c_str());
    prob = -123456789.0f;
    sscanf(line.c_str(), "%f %f", &e_low, &prob);
    prob_bin[bin] = prob;
    s.espc[bin] = e_low;
    if (prob == -123456789.0f) fail(-1, "!!init_energy_spectrum ERROR!!: invalid energy bin number %d?", bin);
    if (e_low < s.espc[std::max(bin - 1, 0)])
      fail(-1, "!!init_energy_spectrum ERROR!!: input energy bins with decreasing energy? espc(%d)=%f", bin, e_low);
  } while (prob > -1.0e-11f);
  s.num_bins = bin;
  for (int i = bin; i < kMaxSpectrumBins; ++i) { s.espc[i] = e_low; prob_bin[i] = 0.0f; }
  float all_e = 0.f, all_p = 0.f;
  for (int i = 0; i < s.num_bins; ++i) {
    all_e += 0.5f * (s.espc[i] + s.espc[i + 1]) * prob_bin[i];
    all_p += prob_bin[i];
  }
  s.mean_energy = all_e / all_p;

  // Walker aliasing (IRND0).
  const int N = s.num_bins;
  double ws = 0.0;
  for (int i = 0; i < N; ++i) {
    if (prob_bin[i] < 0.0f) fail(-1, "!!ERROR!! IRND0: Walker sampling initialization. Negative point probability? W(%d)=%f", i, prob_bin[i]);
    ws += prob_bin[i];
  }
  ws = ((double)N) / ws;
  for (int i = 0; i < N; ++i) { s.alias[i] = (short)i; s.cutoff[i] = (float)(prob_bin[i] * ws); }
  // Entries past N are never meant to be sampled; ranecu()==1.0f (p~3e-8) indexes entry N: keep it benign.
  for (int i = N; i < kMaxSpectrumBins; ++i) { s.alias[i] = (short)std::max(N - 1, 0); s.cutoff[i] = 0.f; }
  if (N == 1) return;
  for (int it = 0; it < N - 1; ++it) {
    float hlow = 1.0f, high = 1.0f;
    int ilow = -1, ihigh = -1;
    for (int j = 0; j < N; ++j) {
      if (s.alias[j] == j) {
        if (s.cutoff[j] < hlow) { hlow = s.cutoff[j]; ilow = j; }
        else if (s.cutoff[j] > high) { high = s.cutoff[j]; ihigh = j; }
      }
    }
    if (ilow == -1 || ihigh == -1) return;
    s.alias[ilow] = (short)ihigh;
    s.cutoff[ihigh] = high + hlow - 1.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// Voxel file (load_voxels, :1996-2145; format: SURVEY.md Appendix A.2).  The whole (gunzipped)
// text is parsed by several threads: pass 1 counts data lines per chunk, pass 2 converts.
// ---------------------------------------------------------------------------------------------
namespace {

// Exactly-rounded "%d %f" for the plain decimal notation the reference writer emits; anything
// unusual (exponents, >15 digits, near-tie cases) goes through strtof.
inline bool parse_voxel_line(const char* p, const char* end, int& mat, float& dens) {
  while (p < end && (*p == ' ' || *p == '\t')) ++p;
  if (p >= end) return false;
  char* q;
  long mv = strtol(p, &q, 10);
  if (q == p) return false;
  mat = (int)mv;
  p = q;
  while (p < end && (*p == ' ' || *p == '\t')) ++p;
  const char* s = p;
  uint64_t mant = 0;
  int digits = 0, frac = 0;
  bool dot = false, simple = (p < end && ((*p >= '0' && *p <= '9') || *p == '.'));
  while (simple && p < end) {
    const char ch = *p;
    if (ch >= '0' && ch <= '9') { mant = mant * 10 + (uint64_t)(ch - '0'); ++digits; if (dot) ++frac; ++p; }
    else if (ch == '.' && !dot) { dot = true; ++p; }
    else break;
  }
  if (simple && digits > 0 && digits <= 15 && frac <= 15 && (p >= end || *p == '\n' || *p == '\r' || *p == ' ' || *p == '#')) {
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const double d = (double)mant / p10[frac];  // exact operands -> correctly rounded double
    const float fl = (float)d;
    // double rounding guard: if d sits (almost) on a float rounding boundary, defer to strtof
    const float lo = nextafterf(fl, -INFINITY), hi = nextafterf(fl, INFINITY);
    const double mid_lo = 0.5 * ((double)fl + (double)lo), mid_hi = 0.5 * ((double)fl + (double)hi);
    const double tol = fabs(d) * 4.5e-16;
    if (fabs(d - mid_lo) > tol && fabs(d - mid_hi) > tol) { dens = fl; return true; }
  }
  dens = strtof(s, &q);
  return q != s;
}

inline bool is_skipped_line(const char* p, const char* end) {
  // reference rule (:2109): skip when char 0 or char 1 is '\n' or '#'
  const char c0 = p < end ? p[0] : '\n';
  const char c1 = (p + 1) < end ? p[1] : '\n';
  return c0 == '\n' || c1 == '\n' || c0 == '#' || c1 == '#';
}

std::string gunzip_all(const std::string& path, const char* err_fmt) {
  gzFile g = gzopen(path.c_str(), "rb");
  if (!g) fail(-2, err_fmt, path.c_str());
  gzbuffer(g, 1 << 20);
  std::string out;
  std::vector<char> buf(8 << 20);
  int n;
  while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
  gzclose(g);
  return out;
}

}  // namespace

void load_voxel_file(const std::string& path, VoxelGrid& v, int n_threads) {
  const std::string text = gunzip_all(path, "!! fopen ERROR load_voxels!! File %s does not exist!!");
  const char* base = text.data();
  const char* end = base + text.size();
  const char* p = base;
  auto next_line = [&](const char*& b, const char*& e) -> bool {
    if (p >= end) return false;
    b = p;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    e = nl ? nl + 1 : end;
    p = e;
    return true;
  };
  const char *lb, *le;
  bool found = false;
  while (next_line(lb, le))
    if (std::string(lb, le).find("[SECTION VOXELS") != std::string::npos) { found = true; break; }
  if (!found) fail(-2, "!!Reading ERROR load_voxels!! File is not readable or does not contain the string '[SECTION VOXELS HEADER'!!");
  if (!next_line(lb, le) || sscanf(std::string(lb, le).c_str(), "%d %d %d", &v.n[0], &v.n[1], &v.n[2]) != 3)
    fail(-2, "!!Reading ERROR load_voxels!! Could not read the number of voxels.");
  if (!next_line(lb, le) || sscanf(std::string(lb, le).c_str(), "%f %f %f", &v.voxel_size[0], &v.voxel_size[1], &v.voxel_size[2]) != 3)
    fail(-2, "!!Reading ERROR load_voxels!! Could not read the voxel size.");
  found = false;
  while (next_line(lb, le))
    if (std::string(lb, le).find("[END OF VXH SECTION") != std::string::npos) { found = true; break; }
  if (!found) fail(-2, "!!Reading ERROR load_voxels!! File is not readable or does not contain the string '[END OF VXH SECTION]'!!");
  if (v.n[0] < 1 || v.n[1] < 1 || v.n[2] < 1) fail(-2, "!!ERROR load_voxels!! Invalid number of voxels.");
  for (int k = 0; k < 3; ++k) {
    v.size_bbox[k] = v.n[k] * v.voxel_size[k];
    v.inv_voxel_size[k] = 1.0f / v.voxel_size[k];
  }
  const size_t nvox = v.count();
  v.material.assign(nvox, 0);
  v.density.assign(nvox, 0.f);
  for (int k = 0; k < kMaxMaterials; ++k) v.density_max[k] = -999.0f;

  // chunk the body at line boundaries
  const char* body = p;
  int T = n_threads > 0 ? n_threads : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const size_t body_bytes = (size_t)(end - body);
  if (body_bytes < (1u << 20)) T = 1;
  std::vector<const char*> cuts(T + 1);
  cuts[0] = body;
  cuts[T] = end;
  for (int t = 1; t < T; ++t) {
    const char* c = body + body_bytes * t / T;
    const char* nl = (const char*)memchr(c, '\n', (size_t)(end - c));
    cuts[t] = nl ? nl + 1 : end;
  }
  std::vector<size_t> counts(T, 0);
  auto count_chunk = [&](int t) {
    size_t cnt = 0;
    const char* q = cuts[t];
    const char* e = cuts[t + 1];
    while (q < e) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
      const char* ln_end = nl ? nl + 1 : e;
      if (!is_skipped_line(q, ln_end)) ++cnt;
      q = ln_end;
    }
    counts[t] = cnt;
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(count_chunk, t);
    count_chunk(0);
    for (auto& x : th) x.join();
  }
  std::vector<size_t> first(T + 1, 0);
  for (int t = 0; t < T; ++t) first[t + 1] = first[t] + counts[t];
  if (first[T] < nvox) fail(-2, "!!ERROR load_voxels!! The voxel file ends after %zu voxels; %zu expected.", first[T], nvox);

  std::vector<std::string> errs(T);
  std::vector<std::vector<float>> dmax(T, std::vector<float>(kMaxMaterials, -999.0f));
  auto parse_chunk = [&](int t) {
    size_t idx = first[t];
    const char* q = cuts[t];
    const char* e = cuts[t + 1];
    float* dm = dmax[t].data();
    while (q < e && idx < nvox) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
      const char* ln_end = nl ? nl + 1 : e;
      if (!is_skipped_line(q, ln_end)) {
        int mat = 0;
        float dens = 0.f;
        char msg[256];
        if (!parse_voxel_line(q, ln_end, mat, dens)) {
          snprintf(msg, sizeof msg, "!!ERROR load_voxels!! Expecting to read 2 items (material and density) for voxel number=%zu", idx + 1);
          errs[t] = msg; return;
        }
        if (mat > kMaxMaterials) {
          snprintf(msg, sizeof msg, "!!ERROR load_voxels!! Voxel material number too high!! #mat=%d, MAX_MATERIALS=%d, voxel number=%zu", mat, kMaxMaterials, idx + 1);
          errs[t] = msg; return;
        }
        if (mat < 1) {
          snprintf(msg, sizeof msg, "!!ERROR load_voxels!! Voxel material number can not be zero or negative!! #mat=%d, voxel number=%zu", mat, idx + 1);
          errs[t] = msg; return;
        }
        if (dens < 1.0e-9f) {
          snprintf(msg, sizeof msg, "!!ERROR load_voxels!! Voxel density can not be 0 or negative: #mat=%d, density=%f, voxel number=%zu", mat, dens, idx + 1);
          errs[t] = msg; return;
        }
        if (dens > dm[mat - 1]) dm[mat - 1] = dens;
        v.material[idx] = (uint8_t)mat;
        v.density[idx] = dens;
        ++idx;
      }
      q = ln_end;
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(parse_chunk, t);
    parse_chunk(0);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < T; ++t)
    if (!errs[t].empty()) fail(-2, "%s", errs[t].c_str());
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < kMaxMaterials; ++k) v.density_max[k] = std::max(v.density_max[k], dmax[t][k]);
}

// Binary sidecar written by write_voxel_binary (report.cpp): same VoxelGrid as the text parse, without the parse.
void load_voxel_binary(const std::string& path, VoxelGrid& v) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) fail(-1, "!! fopen ERROR load_voxels!! File %s does not exist!!", path.c_str());
  char magic[8];
  uint32_t version = 0, kind = 0, npal = 0;
  bool ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, "MCGVOX1\n", 8) == 0 && fread(&version, 4, 1, fp) == 1 && version == 1 &&
            fread(&kind, 4, 1, fp) == 1 && kind <= 2 && fread(v.n, 4, 3, fp) == 3 && fread(v.voxel_size, 4, 3, fp) == 3 &&
            fread(&npal, 4, 1, fp) == 1 && npal <= 65536;
  if (!ok || v.n[0] < 1 || v.n[1] < 1 || v.n[2] < 1) { fclose(fp); fail(-2, "!!Reading ERROR load_voxels!! %s is not a voxel sidecar of this engine.", path.c_str()); }
  for (int k = 0; k < 3; ++k) {
    v.size_bbox[k] = v.n[k] * v.voxel_size[k];
    v.inv_voxel_size[k] = 1.0f / v.voxel_size[k];
  }
  const size_t nvox = v.count();
  std::vector<uint32_t> pm(npal);
  std::vector<float> pd(npal);
  for (uint32_t k = 0; k < npal && ok; ++k) ok = fread(&pm[k], 4, 1, fp) == 1 && fread(&pd[k], 4, 1, fp) == 1;
  v.material.assign(nvox, 0);
  v.density.assign(nvox, 0.f);
  if (ok && kind == 0) ok = fread(v.material.data(), 1, nvox, fp) == nvox && fread(v.density.data(), 4, nvox, fp) == nvox;
  else if (ok && kind == 1) {
    ok = fread(v.material.data(), 1, nvox, fp) == nvox;  // indices first, expanded in place below
    if (ok)
      for (size_t i = 0; i < nvox; ++i) {
        const uint32_t k = v.material[i];
        if (k >= npal) { ok = false; break; }
        v.density[i] = pd[k];
        v.material[i] = (uint8_t)pm[k];
      }
  } else if (ok) {
    std::vector<uint16_t> idx(nvox);
    ok = fread(idx.data(), 2, nvox, fp) == nvox;
    if (ok)
      for (size_t i = 0; i < nvox; ++i) {
        const uint32_t k = idx[i];
        if (k >= npal) { ok = false; break; }
        v.density[i] = pd[k];
        v.material[i] = (uint8_t)pm[k];
      }
  }
  fclose(fp);
  if (!ok) fail(-2, "!!ERROR load_voxels!! The voxel sidecar %s is truncated or corrupt.", path.c_str());
  // the checks of the text loader (:2118-2134) and the per-material maximum density (:2136-2137)
  for (int k = 0; k < kMaxMaterials; ++k) v.density_max[k] = -999.0f;
  for (size_t i = 0; i < nvox; ++i) {
    const int mat = v.material[i];
    if (mat < 1 || mat > kMaxMaterials) fail(-2, "!!ERROR load_voxels!! Voxel material number out of range!! #mat=%d, voxel number=%zu", mat, i + 1);
    if (v.density[i] < 1.0e-9f) fail(-2, "!!ERROR load_voxels!! Voxel density can not be 0 or negative: #mat=%d, density=%f, voxel number=%zu", mat, v.density[i], i + 1);
    if (v.density[i] > v.density_max[mat - 1]) v.density_max[mat - 1] = v.density[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Material files (load_material, :2177-2443; format: SURVEY.md Appendix A.3)
// ---------------------------------------------------------------------------------------------
void load_material_files(const std::vector<std::string>& files, const VoxelGrid& v, MaterialTables& t) {
  t = MaterialTables{};
  float density_max[kMaxMaterials];
  for (int k = 0; k < kMaxMaterials; ++k) {
    density_max[k] = v.density_max[k];
    t.density_nominal[k] = -1.0f;
    t.used[k] = false;
    t.noscco[k] = 0;
  }
  t.xco.assign(kRayleighPoints * kMaxMaterials, 0.f);
  t.pco = t.xco; t.aco = t.xco; t.bco = t.xco;
  t.itlco.assign(kRayleighPoints * kMaxMaterials, 0);
  t.ituco = t.itlco;
  t.pmax.assign((size_t)kMaxRayleighBins * kMaxMaterials, 0.f);
  t.fco.assign(kMaxMaterials * kMaxShells, 0.f);
  t.uico = t.fco; t.fj0 = t.fco;
  double delta_e = -99999.0;

  for (int mat = 0; mat < kMaxMaterials && mat < (int)files.size(); ++mat) {
    if (files[mat].empty() || files[mat][0] == '\n') continue;
    const std::string text = gunzip_all(files[mat], "!!fopen ERROR!! Material file '%s' does not exist!!");
    const char* p = text.data();
    const char* end = p + text.size();
    auto next_line = [&](std::string& out) -> bool {
      if (p >= end) return false;
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* e = nl ? nl + 1 : end;
      out.assign(p, std::min<size_t>((size_t)(e - p), 249));
      p = e;
      return true;
    };
    std::string line;
    bool ok = false;
    while (next_line(line))
      if (line.find("[NOMINAL DENSITY") != std::string::npos) { ok = true; break; }
    if (!ok) fail(-2, "!!Reading ERROR!! File '%s' is not readable or does not contain the string '[NOMINAL DENSITY'!!", files[mat].c_str());
    next_line(line);
    if (sscanf(line.c_str(), "# %f", &t.density_nominal[mat]) != 1)
      fail(-2, "!!Reading ERROR!! Could not read the nominal density in '%s'.", files[mat].c_str());
    if (!(density_max[mat] > 0)) {
      if (mat == 0) density_max[mat] = 0.01f * t.density_nominal[mat];  // first file fixes the grid (:2229-2230)
      else continue;
    }
    t.used[mat] = true;
    next_line(line);
    next_line(line);
    int nval = 0;
    sscanf(line.c_str(), "# %d", &nval);
    if (mat == 0) {
      t.num_values = nval;
      if (nval > kMaxRayleighBins) fail(-2, "!!load_material ERROR!! Too many energy bins (Input bins=%d, MAX_ENERGYBINS_RAYLEIGH=%d)!!", nval, kMaxRayleighBins);
      if (nval < 2) fail(-2, "!!load_material ERROR!! Too few energy bins in '%s'.", files[mat].c_str());
      t.woodcock.assign(nval, Float2{99999999.99f, 0.f});
      t.mfp_total_file.assign((size_t)nval * kMaxMaterials, 0.0);
      t.a.assign((size_t)nval * kMaxMaterials, Float3{0, 0, 0});
      t.b.assign((size_t)nval * kMaxMaterials, Float3{0, 0, 0});
    } else if (nval != t.num_values) {
      fail(-2, "!!load_material ERROR!! Incorrect number of energy values given in material '%s': input=%d, expected=%d", files[mat].c_str(), nval, t.num_values);
    }
    next_line(line);
    next_line(line);
    double e_last = -1.0;
    for (int i = 0; i < nval; ++i) {
      if (!next_line(line)) fail(-2, "!!load_material ERROR!! Unexpected end of file in '%s'.", files[mat].c_str());
      double d_e, d_ray, d_co, d_ph, d_tot, d_pmax;
      if (sscanf(line.c_str(), "  %le  %le  %le  %le  %le  %le", &d_e, &d_ray, &d_co, &d_ph, &d_tot, &d_pmax) != 6)
        fail(-2, "!!load_material ERROR!! Could not read mean free path row %d in '%s'.", i, files[mat].c_str());
      t.mfp_total_file[(size_t)i * kMaxMaterials + mat] = d_tot;
      Float3& a = t.a[(size_t)i * kMaxMaterials + mat];
      a.x = (float)(1.0 / (d_tot * t.density_nominal[mat]));
      a.y = (float)(1.0 / (d_co * t.density_nominal[mat]));
      a.z = (float)(1.0 / (d_ray * t.density_nominal[mat]));
      t.pmax[(size_t)i * kMaxMaterials + mat] = (float)d_pmax;
      if (i == 0 && mat == 0) t.e0 = (float)d_e;
      if (i == 0) {
        if (fabs(d_e - t.e0) > 1.0e-9)
          fail(-2, "!!load_material ERROR!! Incorrect first energy value given in material '%s': input=%f, expected=%f", files[mat].c_str(), d_e, t.e0);
      } else if (i == 1) {
        delta_e = d_e - e_last;
      } else if ((fabs((d_e - e_last) - delta_e)) / delta_e > 0.001) {
        fail(-2, "!!ERROR reading material data!! The energy step between mean free path values is not constant!! #value = %d in '%s'", i, files[mat].c_str());
      }
      e_last = d_e;
    }
    t.ide = (float)(1.0f / delta_e);
    t.delta_e = delta_e;
    for (int i = 0; i < nval - 1; ++i) {
      const size_t bin = (size_t)i * kMaxMaterials + mat;
      t.b[bin].x = (float)((t.a[bin + kMaxMaterials].x - t.a[bin].x) / delta_e);
      t.b[bin].y = (float)((t.a[bin + kMaxMaterials].y - t.a[bin].y) / delta_e);
      t.b[bin].z = (float)((t.a[bin + kMaxMaterials].z - t.a[bin].z) / delta_e);
    }
    t.b[(size_t)(nval - 1) * kMaxMaterials + mat] = t.b[(size_t)(nval - 2) * kMaxMaterials + mat];
    for (int i = 0; i < nval; ++i) {
      const double d_e = t.e0 + i * delta_e;
      const size_t bin = (size_t)i * kMaxMaterials + mat;
      t.a[bin].x = (float)(t.a[bin].x - d_e * t.b[bin].x);
      t.a[bin].y = (float)(t.a[bin].y - d_e * t.b[bin].y);
      t.a[bin].z = (float)(t.a[bin].z - d_e * t.b[bin].z);
    }
    // Rayleigh RITA rows (:2360-2394)
    ok = false;
    while (next_line(line))
      if (line.find("[DATA VALUES") != std::string::npos) { ok = true; break; }
    if (!ok) fail(-2, "!!End-of-file ERROR!! Rayleigh data not found: \"#[DATA VALUES...\" in file '%s'.", files[mat].c_str());
    next_line(line);
    int nray = 0;
    sscanf(line.c_str(), "# %d", &nray);
    if (nray != kRayleighPoints)
      fail(-2, "!!ERROR!! The number of values for Rayleigh sampling is different than the allocated space: input=%d, NP_RAYLEIGH=%d. File='%s'", nray, kRayleighPoints, files[mat].c_str());
    next_line(line);
    for (int i = 0; i < nray; ++i) {
      const int bin = kRayleighPoints * mat + i;
      // ITL / ITU are read as numbers and truncated: the reference's aluminium table writes them as "1.0 4.0", on which
      // the reference's unchecked "%d %d" stops after "1" and leaves ITU uninitialised (:2387-2392; DESIGN.md deviation 10)
      double itl = 0.0, itu = 0.0;
      next_line(line);
      if (sscanf(line.c_str(), "  %e  %e  %e  %e  %lf  %lf", &t.xco[bin], &t.pco[bin], &t.aco[bin], &t.bco[bin], &itl, &itu) != 6)
        fail(-2, "!!ERROR!! Could not read Rayleigh sampling row %d in '%s'.", i, files[mat].c_str());
      t.itlco[bin] = (uint8_t)(int)itl;
      t.ituco[bin] = (uint8_t)(int)itu;
    }
    // Compton shells (:2397-2426)
    ok = false;
    while (next_line(line))
      if (line.find("[NUMBER OF SHELLS") != std::string::npos) { ok = true; break; }
    if (!ok) fail(-2, "!!End-of-file ERROR!! Compton data not found: \"[NUMBER OF SHELLS]\" in file '%s'.", files[mat].c_str());
    next_line(line);
    int nsh = 0;
    sscanf(line.c_str(), "# %d", &nsh);
    if (nsh > kMaxShells) fail(-2, "!!ERROR!! Too many shells for Compton interactions in file '%s': input=%d, MAX_SHELLS=%d", files[mat].c_str(), nsh, kMaxShells);
    t.noscco[mat] = nsh;
    next_line(line);
    for (int i = 0; i < nsh; ++i) {
      const int bin = mat + i * kMaxMaterials;
      next_line(line);  // KZCO and KSCO (unused by the reference too, :2424-2426) may be written as "13.0 0.0"
      if (sscanf(line.c_str(), " %e  %e  %e", &t.fco[bin], &t.uico[bin], &t.fj0[bin]) != 3)
        fail(-2, "!!ERROR!! Could not read Compton shell %d in '%s'.", i, files[mat].c_str());
    }
  }
  if (t.num_values < 2) fail(-2, "!!load_material ERROR!! No material data were read (the first material file fixes the energy grid).");
  rebuild_woodcock(t, v.density_max);
}

// Woodcock majorant: minimum over the materials in the volume of mfp_total * rho_nominal / rho_max(material) per energy
// bin (:2294-2296), then slope and re-basing (:2433-2441).  The reference stops the slope loop one entry short and then
// re-bases the last entry with uninitialised memory; here the last bin re-uses the previous slope (only reachable for
// E == table maximum).
void rebuild_woodcock(MaterialTables& t, const float density_max_in[kMaxMaterials]) {
  const int nv = t.num_values;
  t.woodcock.assign(nv, Float2{99999999.99f, 0.f});
  for (int mat = 0; mat < kMaxMaterials; ++mat) {
    if (!t.used[mat]) continue;
    float dmax = density_max_in[mat];
    if (!(dmax > 0)) {
      if (mat == 0) dmax = 0.01f * t.density_nominal[mat];  // the first material is always loaded (:2229-2230)
      else continue;
    }
    for (int i = 0; i < nv; ++i) {
      const float temp_mfp = (float)(t.mfp_total_file[(size_t)i * kMaxMaterials + mat] * t.density_nominal[mat] / dmax);
      if (temp_mfp < t.woodcock[i].x) t.woodcock[i].x = temp_mfp;
    }
  }
  const double delta_e = t.delta_e;
  for (int i = 0; i < nv - 1; ++i) t.woodcock[i].y = (float)((t.woodcock[i + 1].x - t.woodcock[i].x) / delta_e);
  t.woodcock[nv - 1].y = t.woodcock[nv - 2].y;
  for (int i = 0; i < nv; ++i) t.woodcock[i].x = (float)(t.woodcock[i].x - (t.e0 + i * delta_e) * t.woodcock[i].y);
}

// ---------------------------------------------------------------------------------------------
void load_model(const std::string& input_path, HostModel& m) {
  parse_input_file(input_path, m);
  load_spectrum(m.cfg.file_spectrum, m.spectrum);
  build_ct_trajectory(m);
  {
    // a binary sidecar next to the voxel file (geometry.voxbin, not older than the text) replaces the text parse
    const std::string side = voxel_sidecar_path(m.cfg.file_voxels);
    struct stat st_side, st_text;
    const bool have_side = stat(side.c_str(), &st_side) == 0;
    const bool have_text = stat(m.cfg.file_voxels.c_str(), &st_text) == 0;
    const bool is_side = m.cfg.file_voxels.size() > 7 && m.cfg.file_voxels.compare(m.cfg.file_voxels.size() - 7, 7, ".voxbin") == 0;
    if (is_side) load_voxel_binary(m.cfg.file_voxels, m.voxels);
    else if (have_side && !knob_set("MCGPU_IGNORE_VOXBIN") && (!have_text || st_side.st_mtime >= st_text.st_mtime)) load_voxel_binary(side, m.voxels);
    else load_voxel_file(m.cfg.file_voxels, m.voxels);
  }
  // the dose ROI may be given larger than the volume: clip its upper corner (load_voxels, :2058-2064)
  if (m.cfg.dose_roi[1] > -1)
    for (int ax = 0; ax < 3; ++ax) m.cfg.dose_roi[2 * ax + 1] = std::min(m.cfg.dose_roi[2 * ax + 1], m.voxels.n[ax] - 1);
  load_material_files(m.cfg.file_materials, m.voxels, m.mat);
  // consistency check of main() (:565-577)
  const float emax = m.mat.e0 + (m.mat.num_values - 1) / m.mat.ide;
  if (m.spectrum.espc[0] < m.mat.e0 || m.spectrum.espc[m.spectrum.num_bins] > emax)
    fail(-1, "!!ERROR!! The input x-ray source energy spectrum minimum (%.3f eV) and maximum (%.3f eV) energy values are outside "
             "the tabulated energy interval for the material properties tables (from %.3f to %.3f eV)!!",
         m.spectrum.espc[0], m.spectrum.espc[m.spectrum.num_bins], m.mat.e0, emax);
}

// ---------------------------------------------------------------------------------------------
// RANECU host arithmetic (abMODm K.cu:919-950, update_seed_PRNG H.cu:3456-3485)
// ---------------------------------------------------------------------------------------------
int ranecu_mul_mod(int m, int a, int s) {
  int p = -m;
  while (a > 32768) {
    if (a & 1) { p += s; if (p > 0) p -= m; }
    a >>= 1;
    s = (s - m) + s;
    if (s < 0) s += m;
  }
  const int q = m / a;
  const int k = s / q;
  s = a * (s - k * q) - k * (m - q * a);
  while (s < 0) s += m;
  p += s;
  if (p < 0) p += m;
  return p;
}

int ranecu_advance_seed(int batch_number, unsigned long long total_histories, int seed) {
  if (batch_number == 0) return seed;
  const int m1 = 2147483563, a1 = 40014;
  unsigned long long leap = total_histories * (unsigned long long)(batch_number * 256);
  int y = 1, z = a1;
  for (;;) {
    if (leap & 1ULL) {
      leap >>= 1;
      y = ranecu_mul_mod(m1, z, y);
      if (leap == 0) break;
    } else {
      leap >>= 1;
    }
    z = ranecu_mul_mod(m1, z, z);
  }
  return ranecu_mul_mod(m1, seed, y);
}

LaunchShape reference_launch_shape(unsigned long long histories, int threads_per_block, int hpt) {
  LaunchShape L;
  int total_threads = (int)(((double)histories) / ((double)hpt) + 0.9990);
  int blocks = (int)(((double)total_threads) / ((double)threads_per_block) + 0.9990);
  if (blocks > 65535) {
    blocks = 65000;
    hpt = (int)(((double)histories) / ((double)(blocks * threads_per_block)) + 0.9990);
  } else if (blocks < 1) {
    blocks = 1;
  }
  L.blocks = blocks;
  L.threads = threads_per_block;
  L.hpt = hpt;
  L.total_histories = ((unsigned long long)(blocks * threads_per_block)) * (unsigned long long)hpt;
  return L;
}

}  // namespace mcgpu
