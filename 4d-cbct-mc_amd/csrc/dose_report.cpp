// dose_report.cpp -- reports of the two dose tallies (wire formats of the reference).
//
//  * write_voxel_dose_report(): report_voxels_dose (docker/mcgpu/MC-GPU_v1.3.cu:2976-3199): the ASCII file with the
//    z-plane at the height of the focal spot ("%.6lf %.6lf" = dose [eV/g per history], 2 sigma), the two binary
//    float32 volumes `<file>.raw` / `<file>_2sigma.raw`, and the per-material table derived from the voxel tally.
//  * format_materials_dose_report(): report_materials_dose (:3214-3262), the table of the per-material tally.
//  * material_masses(): total mass of each material in the phantom (main(), :579-585).
// Data lines, binary files and table rows are byte-identical to the reference's for equal tallies; the comment
// header names this engine.  The text the reference prints to stdout is appended to `log`.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host_model.hpp"

namespace mcgpu {
namespace {

void appendf(std::string& s, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  s += buf;
}

}  // namespace

void material_masses(const HostModel& m, double mass[kMaxMaterials]) {
  const VoxelGrid& v = m.voxels;
  const double voxel_volume = 1.0 / (((double)v.inv_voxel_size[0]) * ((double)v.inv_voxel_size[1]) * ((double)v.inv_voxel_size[2]));
  for (int k = 0; k < kMaxMaterials; ++k) mass[k] = 0.0;
  const size_t n = v.count();
  for (size_t k = 0; k < n; ++k) mass[v.material[k] - 1] += ((double)v.density[k]) * voxel_volume;  // same summation order
}

void write_voxel_dose_report(const HostModel& m, const uint64_t* edep /* {x, y} pairs, ROI order */,
                             unsigned long long histories_per_projection, double seconds, std::string& log) {
  const SimConfig& c = m.cfg;
  const VoxelGrid& v = m.voxels;
  const int* roi = c.dose_roi;
  const int num_projections = c.num_projections;
  log += "\n\n          *** VOXEL ROI DOSE TALLY REPORT ***\n\n";
  FILE* fp = fopen(c.file_dose_output.c_str(), "w");
  if (!fp) throw Error(-3, "!!fopen ERROR report_voxels_dose!! File " + c.file_dose_output + " can not be opened!!");
  FILE* fmean = fopen((c.file_dose_output + ".raw").c_str(), "w");
  FILE* fsig = fopen((c.file_dose_output + "_2sigma.raw").c_str(), "w");
  if (!fmean || !fsig) {
    fclose(fp);
    if (fmean) fclose(fmean);
    if (fsig) fclose(fsig);
    throw Error(-3, "!!fopen ERROR report_voxels_dose!! Binary file " + c.file_dose_output + ".raw can not be opened!!");
  }
  const int DX = roi[1] - roi[0] + 1, DY = roi[3] - roi[2] + 1, DZ = roi[5] - roi[4] + 1;
  int z_plane = (int)(m.source[0].pos[2] * v.inv_voxel_size[2] + 0.00001f);
  if (z_plane < roi[4] || z_plane > roi[5]) z_plane = (roi[5] + roi[4]) / 2;
  const int z_plane_roi = z_plane - roi[4];
  appendf(log, "              Reporting the 3D voxel dose distribution as binary floats in the .raw file, and the 2D dose for Z plane %d as ASCII text.\n", z_plane);

  fprintf(fp, "# \n");
  fprintf(fp, "#     *****************************************************************************\n");
  fprintf(fp, "#     ***   Monte Carlo CBCT projection engine for AMD MI355X (gfx950)            ***\n");
  fprintf(fp, "#     ***   file contract of MC-GPU v1.3 (voxel dose tally)                       ***\n");
  fprintf(fp, "#     *****************************************************************************\n");
  fprintf(fp, "# \n");
  fprintf(fp, "#  *** SIMULATION IN THE GPU USING HIP ***\n");
  fprintf(fp, "#\n");
  fprintf(fp, "#\n");
  fprintf(fp, "#  3D dose deposition map (and dose uncertainty) created tallying the energy deposited by photons inside each voxel of the input geometry.\n");
  fprintf(fp, "#  Electrons were not transported and therefore we are approximating that the dose is equal to the KERMA (energy released by the photons alone).\n");
  fprintf(fp, "#  This approximation is acceptable when there is electronic equilibrium and when the range of the secondary electrons is shorter than the voxel size.\n");
  fprintf(fp, "#  Usually the doses will be acceptable for photon energies below 1 MeV. The dose estimates may not be accurate at the interface of low density volumes.\n");
  fprintf(fp, "#\n");
  fprintf(fp, "#  The 3D dose deposition is reported in binary form in the .raw files (data given as 32-bit floats). \n");
  fprintf(fp, "#  To reduce the memory use and the reporting time this text output reports only the 2D dose at the Z plane at the level\n");
  fprintf(fp, "#  of the source focal spot: z_coord = %d (z_coord in ROI = %d)\n", z_plane, z_plane_roi);
  fprintf(fp, "#\n");
  fprintf(fp, "#  The total dose deposited in each different material is reported to the standard output.\n");
  fprintf(fp, "#  The dose is calculated adding the energy deposited in the individual voxels within the dose ROI and dividing by the total mass of the material in the ROI.\n");
  fprintf(fp, "#\n");
  fprintf(fp, "#\n");
  fprintf(fp, "#  Voxel size:  %lf x %lf x %lf = %lf cm^3\n", 1.0 / (double)(v.inv_voxel_size[0]), 1.0 / (double)(v.inv_voxel_size[1]),
          1.0 / (double)(v.inv_voxel_size[2]), 1.0 / (double)(v.inv_voxel_size[0] * v.inv_voxel_size[1] * v.inv_voxel_size[2]));
  fprintf(fp, "#  Number of voxels in the reported region of interest (ROI) X, Y and Z:\n");
  fprintf(fp, "#      %d  %d  %d\n", DX, DY, DZ);
  fprintf(fp, "#  Coordinates of the ROI inside the voxel volume = X[%d,%d], Y[%d,%d], Z[%d,%d]\n", roi[0] + 1, roi[1] + 1, roi[2] + 1, roi[3] + 1, roi[4] + 1, roi[5] + 1);
  fprintf(fp, "#\n");
  fprintf(fp, "#  Voxel dose units: eV/g per history\n");
  fprintf(fp, "#  X rows given first, then Y, then Z. One blank line separates the different Y, and two blanks the Z values (GNUPLOT format).\n");
  fprintf(fp, "#  The dose distribution is also reported with binary FLOAT values (.raw file) for easy visualization in ImageJ.\n");
  fprintf(fp, "# \n");
  fprintf(fp, "#    [DOSE]   [2*standard_deviation]\n");
  fprintf(fp, "# =====================================\n");

  double max_dose = -1.0, max_dose_sd = -1.0;
  size_t max_geo = 0;
  int max_x = -1, max_y = -1, max_z = -1;
  unsigned long long total_edep = 0;
  const double inv_scale = 1.0 / 100.0;  // SCALE_eV, MC-GPU_v1.3.h:81
  const double inv_N = 1.0 / (double)(histories_per_projection * ((unsigned long long)num_projections));
  double mat_edep[kMaxMaterials], mat_edep2[kMaxMaterials], mat_mass[kMaxMaterials];
  unsigned int mat_voxels[kMaxMaterials];
  for (int i = 0; i < kMaxMaterials; ++i) { mat_edep[i] = mat_edep2[i] = mat_mass[i] = 0.0; mat_voxels[i] = 0; }
  const double voxel_volume = 1.0 / (((double)v.inv_voxel_size[0]) * ((double)v.inv_voxel_size[1]) * ((double)v.inv_voxel_size[2]));
  std::vector<float> row_mean(DX), row_sig(DX);
  size_t voxel = 0;
  for (int k = 0; k < DZ; ++k) {
    for (int j = 0; j < DY; ++j) {
      for (int i = 0; i < DX; ++i) {
        const size_t geo = (size_t)(i + roi[0]) + (size_t)(j + roi[2]) * v.n[0] + (size_t)(k + roi[4]) * v.n[0] * v.n[1];
        const double inv_voxel_mass = 1.0 / (v.density[geo] * voxel_volume);
        const int mat = (int)v.material[geo] - 1;
        mat_mass[mat] += v.density[geo] * voxel_volume;
        mat_edep[mat] += (double)edep[2 * voxel];
        mat_edep2[mat] += (double)edep[2 * voxel + 1];
        mat_voxels[mat]++;
        const double dose = ((double)edep[2 * voxel]) * inv_N * inv_voxel_mass * inv_scale;
        total_edep += edep[2 * voxel];
        double sd = (((double)edep[2 * voxel + 1]) * inv_N * inv_scale * inv_voxel_mass - dose * dose) * inv_N;
        if (sd > 0.0) sd = sqrt(sd);
        if (dose > max_dose) { max_dose = dose; max_dose_sd = sd; max_x = i + roi[0]; max_y = j + roi[2]; max_z = k + roi[4]; max_geo = geo; }
        if (k == z_plane_roi) fprintf(fp, "%.6lf %.6lf\n", dose, 2.0 * sd);
        row_mean[i] = (float)dose;
        row_sig[i] = 2.0f * (float)(sd);
        voxel++;
      }
      fwrite(row_mean.data(), sizeof(float), DX, fmean);
      fwrite(row_sig.data(), sizeof(float), DX, fsig);
      if (k == z_plane_roi) fprintf(fp, "\n");
    }
    if (k == z_plane_roi) fprintf(fp, "\n");
  }
  const unsigned long long n_all = histories_per_projection * ((unsigned long long)num_projections);
  fprintf(fp, "#   ****** DOSE REPORT: TOTAL SIMULATION PERFORMANCE FOR ALL PROJECTIONS ******\n");
  fprintf(fp, "#       Total number of simulated x rays: %lld\n", (long long)n_all);
  fprintf(fp, "#       Simulated x rays per projection:  %lld\n", (long long)histories_per_projection);
  fprintf(fp, "#       Total simulation time [s]:  %.2f\n", seconds);
  if (seconds > 0.000001) fprintf(fp, "#       Total speed [x-rays/s]:  %.2f\n", (double)n_all / seconds);
  fprintf(fp, "\n#       Total energy absorved inside the dose ROI: %.5lf keV/hist\n\n", 0.001 * ((double)total_edep) * inv_N * inv_scale);
  fclose(fp);
  fclose(fmean);
  fclose(fsig);

  appendf(log, "\n              Total energy absorved inside the dose deposition ROI: %.5lf keV/hist\n", 0.001 * ((double)total_edep) * inv_N * inv_scale);
  const double mass_max = voxel_volume * v.density[max_geo];
  appendf(log, "              Maximum voxel dose (+-2 sigma): %lf +- %lf eV/g per history (E_dep_voxel=%lf eV/hist)\n", max_dose, max_dose_sd, (max_dose * mass_max));
  appendf(log, "              for the voxel: material=%d, density=%.8f g/cm^3, voxel_mass=%.8lf g, voxel coord in geometry=(%d,%d,%d)\n\n",
          (int)v.material[max_geo], v.density[max_geo], mass_max, max_x, max_y, max_z);
  log += "              Dose deposited in the different materials inside the input ROI computed post-processing the 3D voxel dose results:\n\n";
  log += "    [MATERIAL]  [DOSE_ROI, eV/g/hist]  [2*std_dev]  [Rel error 2*std_dev, %]  [E_dep [eV/hist]  [MASS_ROI, g]  [NUM_VOXELS_ROI]\n";
  log += "   =============================================================================================================================\n";
  for (int i = 0; i < kMaxMaterials; ++i) {
    if (mat_voxels[i] == 0) continue;
    const double e = mat_edep[i] * inv_N * inv_scale;
    double sd = (mat_edep2[i] * inv_N - e * e) * inv_N;
    if (sd > 0.0) sd = sqrt(sd);
    const double dose = e / mat_mass[i];
    sd = sd / mat_mass[i];
    double rel = 0.0;
    if (dose > 0.0) rel = sd / dose;
    appendf(log, "\t%d\t%.5lf\t\t%.5lf\t\t%.2lf\t\t%.2lf\t\t%.5lf\t%u\n", (i + 1), dose, 2.0 * sd, (2.0 * 100.0 * rel), e, mat_mass[i], mat_voxels[i]);
  }
  log += "\n";
}

void format_materials_dose_report(const HostModel& m, const uint64_t* md /* {x, y} pairs, 25 materials */,
                                  unsigned long long histories_per_projection, const double* mass, std::string& log) {
  log += "\n\n          *** MATERIALS TOTAL DOSE TALLY REPORT ***\n\n";
  log += "              Dose deposited in each material defined in the input file (tallied directly per material, not per voxel):\n";
  log += "              The results of this tally should be equal to the voxel tally doses for an ROI covering all voxels.\n\n";
  log += "    [MAT]  [DOSE, eV/g/hist]  [2*std_dev]  [Rel_error 2*std_dev, %]  [E_dep [eV/hist]  [MASS_TOTAL, g]\n";
  log += "   ====================================================================================================\n";
  const double inv_N = 1.0 / (double)(histories_per_projection * ((unsigned long long)m.cfg.num_projections));
  bool flag = false;
  int max_mat = 0;
  for (int i = 0; i < kMaxMaterials; ++i) {
    if (m.mat.density_nominal[i] < 0.0f) break;  // materials not defined in the input file
    const double e = ((double)md[2 * i]) / 100.0f * inv_N;
    double sd = sqrt((((double)md[2 * i + 1]) * inv_N - e * e) * inv_N);
    const double rel = (e > 0.0) ? sd / e : 0.0;
    const double dose = e / mass[i];
    sd = sd / mass[i];
    appendf(log, "\t%d\t%.5lf\t\t%.5lf\t\t%.2lf\t\t%.2lf\t\t%.5lf\n", (i + 1), dose, 2.0 * sd, 2.0 * 100.0 * rel, e, mass[i]);
    if (md[2 * i] > 1e16 || dose != fabs(dose) || sd != fabs(sd)) {  // overflow / nan watch (:3247)
      flag = true;
      if (md[2 * i] > md[2 * max_mat]) max_mat = i;
    }
  }
  if (flag) {
    log += "\n     WARNING: it is possible that the unsigned long long int counter used to tally the standard deviation overflowed (>2^64).\n";
    log += "              The standard deviation may be incorrectly measured, but it will surely be very small (<< 1%).\n";
    appendf(log, "              Max counter (mat=%d): E_dep = %llu , E_dep^2 = %llu\n\n", max_mat + 1, (unsigned long long)md[2 * max_mat], (unsigned long long)md[2 * max_mat + 1]);
  }
}

}  // namespace mcgpu
