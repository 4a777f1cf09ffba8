/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C harness around the *real* reference engine (the MC-GPU v1.3 CPU build of
 * IPMI-ICNS-UKE/4d-cbct-mc, docker/mcgpu/MC-GPU_v1.3.cu).  The reference translation
 * unit is #included from where it lies under /root/reference (-I given by
 * oracle/Makefile); nothing of it is copied into this repository.  The resulting
 * shared object goes to oracle/_ref/ (git-ignored, travels to the GPU box as a
 * prebuilt binary).
 *
 * Purpose:
 *   - pin the restated oracle (oracle/mcgpu_oracle.c) against the reference's own code
 *     (integer tallies must be identical);
 *   - generate golden vectors (tests/golden, script oracle/gen_golden.py);
 *   - optional "reference" CPU baseline in bench.py (cpu_baseline.kind == "reference").
 *
 * The reference's main() (MC-GPU_v1.3.cu:377) is renamed so that its internal
 * `inline` functions (single TU) can be driven piecewise.
 */
#define main mcgpu_reference_main
#include "MC-GPU_v1.3.cu"
#undef main

#include <stdint.h>

/* ---- persistent state that main() keeps on its stack (MC-GPU_v1.3.cu:443-487) ---- */
static struct voxel_struct    H_voxel_data;
static struct detector_struct H_detector_data[MAX_NUM_PROJECTIONS];
static struct source_struct   H_source_data[MAX_NUM_PROJECTIONS];
static struct source_energy_struct H_source_energy_data;
static struct linear_interp   H_mfp_table_data;
static struct compton_struct  H_compton_table;
static struct rayleigh_struct H_rayleigh_table;
static float2 *H_voxel_mat_dens = NULL;
static unsigned int H_voxel_mat_dens_bytes = 0;
static float H_density_max[MAX_MATERIALS];
static float H_density_nominal[MAX_MATERIALS];
static unsigned long long int *H_image = NULL;
static int H_image_bytes = -1;
static int H_mfp_table_bytes = -1, H_mfp_Woodcock_table_bytes = -1;
static float2 *H_mfp_Woodcock_table = NULL;
static float3 *H_mfp_table_a = NULL, *H_mfp_table_b = NULL;
static short int H_dose_ROI[6];
static ulonglong2 *H_voxels_Edep = NULL;
static int H_voxels_Edep_bytes = 0;
static ulonglong2 H_materials_dose[MAX_MATERIALS];
static double H_mass_materials[MAX_MATERIALS];
static unsigned long long int H_total_histories;
static int H_histories_per_thread, H_seed_input, H_num_threads_per_block, H_gpu_id, H_num_projections;
static int H_flag_material_dose = -2, H_enable_specific_angles = -2;
static double H_D_angle = -1.0, H_angularROI_0 = 0.0, H_angularROI_1 = 360.0, H_initial_angle = 0.0,
              H_SRotAxisD = -1.0, H_vertical_translation = 0.0;
static char H_file_name_voxels[250], H_file_name_materials[MAX_MATERIALS][250], H_file_name_output[250],
            H_file_dose_output[250], H_file_name_espc[250];
static float H_specific_angles[MAX_NUM_ANGLES];
static float H_mean_energy_spectrum = 0.0f;

/* Runs the reference initialisation sequence of main() (MC-GPU_v1.3.cu:490-562). */
int ref_load(const char *input_path)
{
  char *argv[2];
  int kk;
  argv[0] = (char *)"ref";
  argv[1] = (char *)input_path;
  memset(&H_source_energy_data, 0, sizeof(H_source_energy_data));
  memset(H_source_data, 0, sizeof(H_source_data));
  memset(H_detector_data, 0, sizeof(H_detector_data));
  memset(&H_compton_table, 0, sizeof(H_compton_table));
  memset(&H_rayleigh_table, 0, sizeof(H_rayleigh_table));
  for (kk = 0; kk < MAX_MATERIALS; kk++) {
    H_materials_dose[kk].x = 0; H_materials_dose[kk].y = 0; H_density_nominal[kk] = -1.0f;
  }
  H_D_angle = -1.0; H_angularROI_0 = 0.0; H_angularROI_1 = 360.0; H_initial_angle = 0.0;
  H_SRotAxisD = -1.0; H_vertical_translation = 0.0;
  /* NB: argument order as at the reference call site (MC-GPU_v1.3.cu:490): seed before gpu id. */
  read_input(2, argv, 0, &H_total_histories, &H_seed_input, &H_gpu_id, &H_num_threads_per_block,
             &H_histories_per_thread, H_detector_data, &H_image, &H_image_bytes, H_source_data,
             &H_source_energy_data, H_file_name_voxels, H_file_name_materials, H_file_name_output,
             H_file_name_espc, &H_num_projections, &H_D_angle, &H_angularROI_0, &H_angularROI_1,
             &H_initial_angle, &H_voxels_Edep, &H_voxels_Edep_bytes, H_file_dose_output,
             &H_dose_ROI[0], &H_dose_ROI[1], &H_dose_ROI[2], &H_dose_ROI[3], &H_dose_ROI[4], &H_dose_ROI[5],
             &H_SRotAxisD, &H_vertical_translation, &H_flag_material_dose, &H_enable_specific_angles,
             H_specific_angles);
  init_energy_spectrum(H_file_name_espc, &H_source_energy_data, &H_mean_energy_spectrum);
  if (H_num_projections != 1)
    set_CT_trajectory(0, H_num_projections, H_D_angle, H_angularROI_0, H_angularROI_1, H_SRotAxisD,
                      H_source_data, H_detector_data, H_vertical_translation, &H_enable_specific_angles,
                      H_specific_angles);
  load_voxels(0, H_file_name_voxels, H_density_max, &H_voxel_data, &H_voxel_mat_dens, &H_voxel_mat_dens_bytes,
              &H_dose_ROI[1], &H_dose_ROI[3], &H_dose_ROI[5]);
  load_material(0, H_file_name_materials, H_density_max, H_density_nominal, &H_mfp_table_data,
                &H_mfp_Woodcock_table, &H_mfp_Woodcock_table_bytes, &H_mfp_table_a, &H_mfp_table_b,
                &H_mfp_table_bytes, &H_rayleigh_table, &H_compton_table);
  /* The last Woodcock entry is left uninitialised by the reference (slope loop stops one short,
     MC-GPU_v1.3.cu:2434-2441): make the dump deterministic. */
  /* (value is only reachable for E == table maximum exactly) */
  /* copy to the file-scope "CONST" structs exactly as the CPU path does (MC-GPU_v1.3.cu:942-950) */
  source_energy_data_CONST = H_source_energy_data;
  voxel_data_CONST = H_voxel_data;
  mfp_table_data_CONST = H_mfp_table_data;
  dose_ROI_x_min_CONST = H_dose_ROI[0]; dose_ROI_x_max_CONST = H_dose_ROI[1];
  dose_ROI_y_min_CONST = H_dose_ROI[2]; dose_ROI_y_max_CONST = H_dose_ROI[3];
  dose_ROI_z_min_CONST = H_dose_ROI[4]; dose_ROI_z_max_CONST = H_dose_ROI[5];
  fflush(stdout);
  return 0;
}

/* Pointer + size of a named host array/struct of the loaded reference state. */
const void *ref_get(const char *name, long *nbytes)
{
#define RET(p, n) do { *nbytes = (long)(n); return (const void *)(p); } while (0)
  if (!strcmp(name, "voxel_mat_dens"))   RET(H_voxel_mat_dens, H_voxel_mat_dens_bytes);
  if (!strcmp(name, "voxel_data"))       RET(&H_voxel_data, sizeof(H_voxel_data));
  if (!strcmp(name, "mfp_woodcock"))     RET(H_mfp_Woodcock_table, H_mfp_Woodcock_table_bytes);
  if (!strcmp(name, "mfp_a"))            RET(H_mfp_table_a, H_mfp_table_bytes);
  if (!strcmp(name, "mfp_b"))            RET(H_mfp_table_b, H_mfp_table_bytes);
  if (!strcmp(name, "mfp_table_data"))   RET(&H_mfp_table_data, sizeof(H_mfp_table_data));
  if (!strcmp(name, "rayleigh"))         RET(&H_rayleigh_table, sizeof(H_rayleigh_table));
  if (!strcmp(name, "compton"))          RET(&H_compton_table, sizeof(H_compton_table));
  if (!strcmp(name, "source_data"))      RET(H_source_data, sizeof(struct source_struct) * MAX_NUM_PROJECTIONS);
  if (!strcmp(name, "detector_data"))    RET(H_detector_data, sizeof(struct detector_struct) * MAX_NUM_PROJECTIONS);
  if (!strcmp(name, "source_energy"))    RET(&H_source_energy_data, sizeof(H_source_energy_data));
  if (!strcmp(name, "density_max"))      RET(H_density_max, sizeof(H_density_max));
  if (!strcmp(name, "density_nominal"))  RET(H_density_nominal, sizeof(H_density_nominal));
  if (!strcmp(name, "image"))            RET(H_image, H_image_bytes);
  if (!strcmp(name, "materials_dose"))   RET(H_materials_dose, sizeof(H_materials_dose));
  if (!strcmp(name, "voxels_edep"))      RET(H_voxels_Edep, H_voxels_Edep_bytes);
  if (!strcmp(name, "dose_roi"))         RET(H_dose_ROI, sizeof(H_dose_ROI));
  if (!strcmp(name, "mass_materials"))   RET(H_mass_materials, sizeof(H_mass_materials));
  if (!strcmp(name, "specific_angles"))  RET(H_specific_angles, sizeof(H_specific_angles));
#undef RET
  *nbytes = -1;
  return NULL;
}

/* Scalars: histories, seed, gpu id, threads/block, histories/thread, projections, flags. */
void ref_get_scalars(double *out /* [16] */)
{
  out[0] = (double)H_total_histories; out[1] = H_seed_input; out[2] = H_gpu_id;
  out[3] = H_num_threads_per_block;   out[4] = H_histories_per_thread; out[5] = H_num_projections;
  out[6] = H_D_angle; out[7] = H_angularROI_0; out[8] = H_angularROI_1; out[9] = H_initial_angle;
  out[10] = H_SRotAxisD; out[11] = H_vertical_translation; out[12] = H_flag_material_dose;
  out[13] = H_enable_specific_angles; out[14] = H_mean_energy_spectrum; out[15] = 0.0;
}

void ref_clear_image(void)
{
  int kk;
  (void)kk;
  memset(H_image, 0, H_image_bytes);
}

/* Dose tallies accumulate over projections (MC-GPU_v1.3.cu:1062-1165); cleared on request only. */
void ref_clear_dose(void)
{
  int kk;
  for (kk = 0; kk < MAX_MATERIALS; kk++) { H_materials_dose[kk].x = 0; H_materials_dose[kk].y = 0; }
  if (H_voxels_Edep != NULL && H_dose_ROI[1] > -1) memset(H_voxels_Edep, 0, H_voxels_Edep_bytes);
}

/* The CPU history loop of main() (MC-GPU_v1.3.cu:953-958) for batches [batch0, batch0+nbatches). */
int ref_track(int num_p, int seed_input, int batch0, int nbatches, int histories_per_thread)
{
  int b;
  for (b = batch0; b < batch0 + nbatches; b++)
    track_particles(b, histories_per_thread, num_p, seed_input, H_image, H_voxels_Edep, H_voxel_mat_dens,
                    H_mfp_Woodcock_table, H_mfp_table_a, H_mfp_table_b, &H_rayleigh_table, &H_compton_table,
                    H_detector_data, H_source_data, H_materials_dose);
  return 0;
}

/* report_image (MC-GPU_v1.3.cu:2783) on the current image with an explicit output base name. */
int ref_report(const char *out_base, int num_p, unsigned long long total_histories, double seconds)
{
  char base[250];
  strncpy(base, out_base, 249); base[249] = '\0';
  return report_image(base, H_detector_data, H_source_data, H_mean_energy_spectrum, H_image, seconds,
                      total_histories, num_p, H_num_projections, H_D_angle, H_initial_angle, 0, 1,
                      &H_enable_specific_angles, H_specific_angles);
}

/* report_voxels_dose (MC-GPU_v1.3.cu:2976) when the ROI is enabled, then report_materials_dose (:3214), on the
 * current tallies; the material masses are computed as main() does (:579-585). */
int ref_report_dose(const char *out_file, unsigned long long total_histories, double seconds)
{
  char name[250];
  int kk;
  double voxel_volume = 1.0 / ( ((double)H_voxel_data.inv_voxel_size.x) * ((double)H_voxel_data.inv_voxel_size.y) * ((double)H_voxel_data.inv_voxel_size.z) );
  for (kk = 0; kk < MAX_MATERIALS; kk++) H_mass_materials[kk] = 0.0;
  for (kk = 0; kk < (H_voxel_data.num_voxels.x * H_voxel_data.num_voxels.y * H_voxel_data.num_voxels.z); kk++)
    H_mass_materials[((int)H_voxel_mat_dens[kk].x) - 1] += ((double)H_voxel_mat_dens[kk].y) * voxel_volume;
  strncpy(name, out_file, 249); name[249] = '\0';
  if (H_dose_ROI[1] > -1)
    report_voxels_dose(name, H_num_projections, &H_voxel_data, H_voxel_mat_dens, H_voxels_Edep, seconds, total_histories,
                       H_dose_ROI[0], H_dose_ROI[1], H_dose_ROI[2], H_dose_ROI[3], H_dose_ROI[4], H_dose_ROI[5], H_source_data);
  report_materials_dose(H_num_projections, total_histories, H_density_nominal, H_materials_dose, H_mass_materials);
  fflush(stdout);
  return 0;
}

/* ---- known-answer entry points for the small device helpers ---- */
void ref_init_prng(int batch, int hpt, int seed_input, int *out2)
{ int2 s; init_PRNG(batch, hpt, seed_input, &s); out2[0] = s.x; out2[1] = s.y; }
float ref_ranecu(int *seed2)
{ int2 s; float r; s.x = seed2[0]; s.y = seed2[1]; r = ranecu(&s); seed2[0] = s.x; seed2[1] = s.y; return r; }
double ref_ranecu_double(int *seed2)
{ int2 s; double r; s.x = seed2[0]; s.y = seed2[1]; r = ranecu_double(&s); seed2[0] = s.x; seed2[1] = s.y; return r; }
int ref_abmodm(int m, int a, int s) { return abMODm(m, a, s); }
int ref_update_seed(int batch_number, unsigned long long total_histories, int seed)
{ update_seed_PRNG(batch_number, total_histories, &seed); return seed; }
void ref_rotate_double(float *dir3, double costh, double phi)
{ float3 d; d.x = dir3[0]; d.y = dir3[1]; d.z = dir3[2]; rotate_double(&d, costh, phi); dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z; }
void ref_gcoa(float *energy, double *costh, int mat, int *seed2)
{ int2 s; s.x = seed2[0]; s.y = seed2[1]; GCOa(energy, costh, &mat, &s, &H_compton_table); seed2[0] = s.x; seed2[1] = s.y; }
void ref_graa(float energy, double *costh, int mat, int index, int *seed2)
{ int2 s; float pmax = H_rayleigh_table.pmax[(index + 1) * MAX_MATERIALS + mat];
  s.x = seed2[0]; s.y = seed2[1]; GRAa(&energy, costh, &mat, &pmax, &s, &H_rayleigh_table); seed2[0] = s.x; seed2[1] = s.y; }
void ref_source(int num_p, int *seed2, float *pos3, float *dir3, float *energy, int *absvox)
{ int2 s; float3 p, d; s.x = seed2[0]; s.y = seed2[1]; *absvox = 1;
  source(&p, &d, energy, &s, absvox, &H_source_data[num_p], &H_detector_data[num_p]);
  pos3[0] = p.x; pos3[1] = p.y; pos3[2] = p.z; dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z; seed2[0] = s.x; seed2[1] = s.y; }
int ref_sizeof(const char *name)
{
  if (!strcmp(name, "source_struct")) return (int)sizeof(struct source_struct);
  if (!strcmp(name, "detector_struct")) return (int)sizeof(struct detector_struct);
  if (!strcmp(name, "voxel_struct")) return (int)sizeof(struct voxel_struct);
  if (!strcmp(name, "compton_struct")) return (int)sizeof(struct compton_struct);
  if (!strcmp(name, "rayleigh_struct")) return (int)sizeof(struct rayleigh_struct);
  if (!strcmp(name, "source_energy_struct")) return (int)sizeof(struct source_energy_struct);
  return -1;
}
