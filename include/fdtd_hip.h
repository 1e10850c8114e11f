/* fdtd_hip.h — C ABI of libfdtd_hip.so, the MI355X (gfx950) FDTD time-stepper.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (flexcompute/tidy3d) has NO native interface for this path: its solver is a closed cloud
 * service reached through  tidy3d/web/api/webapi.py:49 run() -> :159 upload() -> :266 start()
 * -> :337 monitor() -> :631 load().  The functions below are what a local replacement of that
 * upload/start/monitor/load sequence binds through ctypes:
 *
 *   upload  (webapi.py:159, hdf5 of the Simulation)      -> fdtd_create + fdtd_set_* / fdtd_add_*
 *   start + monitor (webapi.py:266,:337; progress =
 *        (perc_done, field_decay), task_core.py:537)     -> fdtd_run + FdtdProgressFn
 *   load    (webapi.py:631, hdf5 of the monitor data)    -> fdtd_get_monitor / fdtd_get_field
 *   task status "diverged" (webapi.py:370)               -> FdtdStats.diverged
 *
 * Conventions
 *  - plain C; every function returns 0 on success, <0 on error; the message is available from
 *    fdtd_last_error() (thread-local for create, per-handle otherwise).
 *  - the caller owns every host buffer before and after a call (the library copies in/out
 *    synchronously); the library owns all device memory, streams and RCCL communicators.
 *  - one handle = one solve on one GPU (one z-slab of the domain in a multi-GPU run); calls on
 *    a handle are not re-entrant and must come from one host thread.
 *  - field arrays are [nz][ny][nx] C-ordered, x fastest, fp32 (ref monitor.py:35-36: 4 B real /
 *    8 B complex).  Yee staggering as in ref components/grid/grid.py:465-491.
 *  - component ids: 0..5 = Ex,Ey,Ez,Hx,Hy,Hz.
 */
#ifndef FDTD_HIP_H
#define FDTD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct FdtdSolver FdtdSolver;

enum { FDTD_BC_PEC = 0, FDTD_BC_PMC = 1, FDTD_BC_PERIODIC = 2, FDTD_BC_NEIGHBOR = 3 };
enum { FDTD_MON_TIME = 0, FDTD_MON_DFT = 1 };
/* kernel variants of the two main update kernels (A/B-tested by bench.py --variant) */
/* AUTO = FUSED on one GPU (single-sweep E+H update, 48 B/cell-step), two-pass ZMARCH otherwise */
enum { FDTD_VARIANT_AUTO = 0, FDTD_VARIANT_SIMPLE = 1, FDTD_VARIANT_ZMARCH = 2, FDTD_VARIANT_FUSED = 3 };
enum { FDTD_FLAG_TIME_KERNELS = 1 };   /* bracket every main-kernel launch with hipEvents */

typedef struct FdtdConfig {
  int32_t nx, ny, nz;      /* cells of THIS rank's slab (PML cells included)                    */
  int32_t bc[6];           /* xmin,xmax,ymin,ymax,zmin,zmax: FDTD_BC_*; NEIGHBOR only on z faces */
  int32_t device;          /* HIP device ordinal                                                */
  int32_t variant;         /* FDTD_VARIANT_*                                                    */
  int32_t flags;           /* FDTD_FLAG_*                                                       */
  int32_t z_chunk;         /* planes marched per workgroup (0 = default)                        */
  float   ch;              /* dt / mu0: H-update coefficient                                    */
  int32_t reserved[6];
} FdtdConfig;

typedef struct FdtdStats {
  int64_t steps_done;
  int32_t diverged;          /* NaN/Inf seen in the field-energy reduction                       */
  int32_t stopped_early;     /* shutoff reached                                                  */
  double  field_decay;       /* last W / max W,  W = sum|E|^2 + (mu0/eps0) sum|H|^2              */
  double  run_ms;            /* hipEvent time of the last fdtd_run (whole step loop)             */
  double  h_kernel_ms;       /* with FDTD_FLAG_TIME_KERNELS: summed durations of the main H ...  */
  double  e_kernel_ms;       /* ... and E update kernels in the last fdtd_run                    */
  int64_t h_kernel_launches;
  int64_t e_kernel_launches;
  int64_t device_bytes;      /* device memory held by the handle                                 */
  double  fused_kernel_ms;   /* ... and of the fused E+H sweep                                   */
  int64_t fused_kernel_launches;
  int32_t tile_rows;         /* tile shape of the fused sweep in use (after autotuning)           */
  int32_t tile_zchunk;
  int32_t tile_order;        /* 1 = XCD-aware tile order, 0 = plain (chosen by timing both on the first large sweep) */
  int32_t placement;         /* placement probe of the field arrays: (candidates timed << 8) | index kept (0 = the first allocations) */
  float   placement_ms_first;/* three probe sweeps on the first allocations ...                 */
  float   placement_ms_kept; /* ... and on the set that was kept (0 when the probe did not run) */
  int32_t stream_overlap;    /* the engine's two streams: 1 = measured to run concurrently, 0 = not probed yet (no run used
                                both), -1 = they serialised on every attempt (shared hardware queue): ONE stream is used */
  int32_t stream_retries;    /* fresh second streams created until two overlapped                */
  int32_t comm_ranks;        /* ncclCommCount of the halo communicator (0 = none)                */
  int32_t comm_rank;         /* ncclCommUserRank                                                 */
  int64_t two_step_pairs;    /* step pairs advanced by the slab-interleaved two-step schedule (FDTD_OPT_TBLOCK) in the last fdtd_run */
  int32_t tblock_planes;     /* its slab thickness in planes (0 = schedule off)                  */
  int32_t reserved0;         /* graph capture diagnostics: 0 = none attempted, 1 = captured, < 0 = -(100 stage + hipError) */
  int64_t graph_pairs;       /* step pairs replayed as captured hipGraphs in the last fdtd_run (FDTD_OPT_GRAPH) */
  int64_t fused2_pairs;      /* step pairs advanced by the two-steps-per-sweep kernel in the last fdtd_run (FDTD_OPT_TWOSTEP) */
  int64_t fused2_shape;      /* its tile shape: waves per workgroup | planes per chunk << 6 (0 = no pair was taken) */
  int64_t shell_pairs;       /* of those: pairs of a CPML-walled grid — the two-step sweep over the bulk, the shell (CPML slabs + collar)
                                as two single steps beside it on the second stream (FDTD_OPT_SHELL_PAIRS) */
  double  shell_kernel_ms;   /* with FDTD_FLAG_TIME_KERNELS: summed durations of the shell launches (they overlap the bulk sweep) */
  int64_t shell_kernel_launches;
  int64_t shell2_pairs;      /* of those: pairs whose shell went out as shell2_step_kernel launches — two steps per sweep with the CPML
                                recursions carried through both — instead of two single steps (FDTD_OPT_SHELL2) */
  int32_t fused2_off_reason; /* why the last fdtd_run took NO step pairs: FDTD_F2_OFF_* (0 = it took some, or had no chance to: < 2 steps) */
  int32_t struct_bytes;      /* sizeof(FdtdStats) of the library that filled this in (a binding checks it against its own layout) */
  int64_t disp_pairs;        /* of fused2_pairs: pairs that advanced the dispersive (ADE) cells themselves (FDTD_OPT_DISP; round 6) */
  int32_t single_step_reason;/* a run that took pairs AND single steps: the last FDTD_F2_OFF_* that kept a step from opening a pair because of its
                                sources (a TFSF box / mode plane while it injects ...); 0 = none (single steps then are record / decay-check /
                                odd-count steps) */
  int32_t src_paged_pairs;   /* of fused2_pairs: pairs that carried paged source terms (FDTD_OPT_SRC_PAGED; round 6) */
  double  seam_kernel_ms;    /* with FDTD_FLAG_TIME_KERNELS: summed durations of seam_kernel behind the two-step sweeps (until round 6 they
                                were part of fused_kernel_ms, which now brackets the sweep alone — what rocprofv3 reports for the kernel) */
  int64_t seam_kernel_launches;
} FdtdStats;

/* FdtdStats.fused2_off_reason: what keeps a run on single steps (the first reason found) */
enum { FDTD_F2_OFF_NONE = 0,
       FDTD_F2_OFF_DISABLED = 1,          /* FDTD_OPT_TWOSTEP = 0 (or another schedule was asked for) */
       FDTD_F2_OFF_TOO_SMALL = 2,         /* grids below 2^20 cells are bound by launches, not by bytes */
       FDTD_F2_OFF_COMM = 3,              /* z-slab rank (RCCL communicator): ghost planes are exchanged every step */
       FDTD_F2_OFF_PML = 4,               /* CPML present and shell pairs not possible: switched off, slab-kernel CPML asked for,
                                             layers too thick for the grid, absorber layers on another axis */
       FDTD_F2_OFF_ADE = 5,               /* dispersive media whose cells are not confined to a few plane ranges along z (their planes take single
                                             steps as z holes of the bulk; what is left must be worth a two-step launch), or with absorber layers */
       FDTD_F2_OFF_TFSF = 6,              /* a TFSF box while it injects (a plane wave's injection PLANE is a z hole of the bulk: pairs) */
       FDTD_F2_OFF_BOUNDARY = 7,          /* periodic / Bloch faces, PMC on a plus face, rows not a multiple of 4 cells */
       FDTD_F2_OFF_H_SOURCE_ABSORBER = 8, /* magnetic point sources together with absorber layers */
       FDTD_F2_OFF_SEAM_SOURCE = 9,       /* an H_y / H_z source node in the column left of a seam between 256-cell x tiles */
       FDTD_F2_OFF_SOURCES = 10,          /* while they inject: more than 256 source nodes that are not confined to a few planes along z (a mode plane /
                                             current sheet normal to z is a z hole of the bulk: pairs), or H-side nodes without room for their table */
       FDTD_F2_OFF_VARIANT = 11,          /* the run is not on the fused sweep at all (two-pass kernels) */
       FDTD_F2_OFF_SHELL = 12,            /* CPML shell too large a part of the grid for shell pairs to pay (cost model, fdtd_capi.hip shell_why_not) */
       FDTD_F2_OFF_MEMORY = 13 };         /* no room for the third field set that shell pairs / slab pairs keep the middle step in (+ 50 % field memory) */

/* progress callback: (step, time [s], field_decay) -> non-zero aborts the run (Ctrl-C path).
 * Mirrors the (perc_done, field_decay) pair the cloud reports (ref web/core/task_core.py:537). */
typedef int (*FdtdProgressFn)(int64_t step, double time, double field_decay, void* user);

const char* fdtd_last_error(const FdtdSolver* h);   /* h may be NULL: error of the last create */
int  fdtd_device_count(void);

/* Near -> far projection, the integration step on the device (ref components/field_projection.py:360-368
 * `integrate_2d` and :370 `_far_fields_for_surface`).  For one surface of a projection
 * monitor at one frequency: the equivalent currents J_u, J_v, M_u, M_v on the (u, v) lattice of the surface
 * (`currents`: [4][n_u][n_v] complex as (re, im) pairs of doubles; coordinates u[n_u], v[n_v] relative to the
 * monitor's local origin, w0 = the surface's coordinate along its normal; wu, wv = the trapezoid weights of
 * np.trapz, 1 for a single point) are integrated against exp(-i k r_hat . r') for n_dir directions given by their
 * cosines (r_u, r_v, r_w) along u, v and the normal; k = k_re + i k_im in the projection medium.
 * out[n_dir][4][2] = the four integrals (re, im) per direction.  Returns 0, or -1 with fdtd_last_error(NULL). */
int  fdtd_far_field(int device, int n_u, int n_v, const double* u, const double* v, const double* wu, const double* wv,
                    const double* currents, double w0, double k_re, double k_im, int n_dir, const double* r_u,
                    const double* r_v, const double* r_w, double* out);

/* One solve = one handle.  Stands in for SimulationTask.create + upload_simulation of the cloud
 * path (ref web/api/webapi.py:219,237; web/core/task_core.py:121); the grid size is
 * Simulation.grid.num_cells incl. the PML cells (ref simulation.py:4296, grid_spec.py:114-137),
 * the face codes come from Simulation.boundary_spec (ref boundary.py:732; a PEC wall backs every
 * PML, CHANGELOG.md:1290), ch = Simulation.dt / MU_0 (ref simulation.py:4194, constants.py:21). */
int  fdtd_create(const FdtdConfig* cfg, FdtdSolver** out);
/* ref web/api/webapi.py delete(): releases every device resource of the task */
void fdtd_destroy(FdtdSolver* h);

/* 1/primal and 1/dual step vectors of one axis (length = n cells of that axis of this slab;
 * ref grid.py:393-417).  axis 0,1,2 = x,y,z.  For z, n may also be nz + 2: then entry 0 and
 * n-1 are the values of the planes just below / above the slab (needed by the fused sweep on a
 * z-slab of a non-uniform grid); with n == nz they are derived from the boundary condition. */
int fdtd_set_steps(FdtdSolver* h, int axis, const float* inv_primal, const float* inv_dual, int n);

/* material table (index 0 = PEC: ca = cb = 0) and, optionally, the staircased material index
 * volumes mat[3][nz][ny][nx] (uint8, one per E component, ref simulation.py:1135-1241).
 * Without fdtd_set_material every cell uses entry 1.  (Ca, Cb) follow from Medium.permittivity /
 * conductivity (ref medium.py:1499, :1016-1038) or the pole-residue form (ref medium.py:2739). */
int fdtd_set_media(FdtdSolver* h, const float* ca, const float* cb, int n_media);   /* n_media <= 1024 */
int fdtd_set_material(FdtdSolver* h, const uint8_t* mat, size_t bytes);
/* the same with 16-bit indices mat[3][nz][ny][nx] (count = 3 nz ny nx entries): more than 255 media (the
 * reference allows 65,530 structures, ref components/scene.py:52; the device packs three 10-bit indices
 * into one 32-bit word per cell, so 1023 distinct media per simulation) — what sub-pixel averaging and
 * CustomMedium need to quantise permittivity in 0.2 % steps */
int fdtd_set_material16(FdtdSolver* h, const uint16_t* mat, size_t count);

/* CPML tables of one axis, each of length n (identity outside the slabs); the slab index ranges
 * are derived from n_lo/n_hi = Simulation.num_pml_layers (ref simulation.py:1002) and the profile
 * from PMLParams (ref boundary.py:195-254; sampling positions ref plugins/mode/derivatives.py:
 * 174-197; formulas in tidy3d_amd/coeffs.py).  On a z-slab the host passes slab-local tables. */
int fdtd_set_pml(FdtdSolver* h, int axis, int n_lo, int n_hi,
                 const float* kinv_e, const float* b_e, const float* c_e,
                 const float* kinv_h, const float* b_h, const float* c_h, int n);

/* PMC on the PLUS face of an axis (ref boundary.py:45 PMCBoundary; the min face is FDTD_BC_PMC in FdtdConfig.bc).  The host
 * lays the axis out with two ghost cells beyond the wall (its plus face declared FDTD_BC_PEC: plain truncation) and names the
 * wall's cell-boundary index, wall == n_cells(axis) - 2; the library refreshes the mirror images beyond the wall at the
 * start of every step (tangential E and normal H even, normal E and tangential H odd).  wall < 0 switches it off.
 * One GPU (not on z-slabs, not with fdtd_run_bloch). */
int fdtd_set_mirror_plus(FdtdSolver* h, int axis, int wall);

/* Absorber layers of one axis (ref boundary.py:427-476 Absorber, :166-192 AbsorberParams; layer
 * counts = Simulation.num_pml_layers, ref simulation.py:1002): per-step damping factors of length n
 * (1 outside the layers), fb sampled at the cell boundaries, fc at the cell centres; the layers are
 * [0, n_lo) and [n - n_hi, n).  Every component inside them is multiplied once per step by the
 * product of its three axis factors (tidy3d_amd/coeffs.py damping_tables).  On a z-slab the host
 * passes slab-local tables and counts. */
int fdtd_set_absorber(FdtdSolver* h, int axis, int n_lo, int n_hi, const float* fb, const float* fc, int n);

/* pole-residue ADE group: the cells (linear index k*ny*nx + j*nx + i within the slab) of E
 * component `comp` filled with one dispersive medium; kap/bet are n_poles complex pairs
 * (re,im interleaved), cc the memory-term coefficient (ref medium.py:2900-2913). */
int fdtd_add_ade(FdtdSolver* h, int comp, int64_t n_cells, const uint32_t* cell_index,
                 int n_poles, const float* kap, const float* bet, float cc);

/* fully anisotropic bodies (ref tidy3d medium.py:5058 FullyAnisotropicMedium): the off-diagonal coupling of E component `comp`
 * at n nodes.  The sweep advances every component with the diagonal of eps^-1 (table media); behind it
 *     E_comp[cell[i]] += sum_{s<8} w_new[8i+s] * Eb^{n+1}[nbr[8i+s]] - w_old[8i+s] * Eb^n[nbr[8i+s]],
 * slots 0-3: component (comp+1)%3, slots 4-7: (comp+2)%3; nbr = 0xFFFFFFFF: no node (weight ignored).  The host forms
 * w_new = (dt/eps0) g / Cb(nbr), w_old = w_new * Ca(nbr) (tidy3d_amd/spec.py AnisoSet).  Single steps only; not on z-slabs,
 * not with Bloch boundaries. */
int fdtd_add_aniso(FdtdSolver* h, int comp, int64_t n_nodes, const uint32_t* cell_index, const uint32_t* nbr_index,
                   const float* w_new, const float* w_old);

/* current source: F[comp[p]][index[p]] += w_re[p]*Re(wave[n]) - w_im[p]*Im(wave[n]);
 * E components use wave_e (sampled at t_n + dt/2), H components wave_h (t_n);
 * waves are n_steps complex values (re,im interleaved)  (ref source.py:174-193, :543-632). */
int fdtd_add_point_source(FdtdSolver* h, int64_t n_points, const int32_t* comp,
                          const uint32_t* cell_index, const float* w_re, const float* w_im,
                          int64_t n_steps, const float* wave_e, const float* wave_h);

/* total-field/scattered-field source driven by a 1-D auxiliary grid (ref source.py:1204-1257);
 * semantics documented at tidy3d_amd/spec.py TfsfSpec.  aux indices refer to e1 (h_corr) and
 * h1 (e_corr). */
int fdtd_add_tfsf(FdtdSolver* h, int n_aux, const float* ae, const float* be,
                  const float* ah, const float* bh, int src_cell,
                  int64_t n_steps, const float* wave,
                  int64_t n_e, const int32_t* e_comp, const uint32_t* e_index, const float* e_w,
                  const int32_t* e_aux,
                  int64_t n_h, const int32_t* h_comp, const uint32_t* h_index, const float* h_w,
                  const int32_t* h_aux);

/* monitor over the Yee index box [lo, hi) of this slab.  steps: sorted time-step indices on
 * which it records.  DFT monitors: nf frequencies and phase tables [n_rec][nf] complex
 * (re,im interleaved) for E (t_n) and H (t_n + dt/2) samples (ref monitor.py:363-403,
 * time.py:95-105).  Returns the monitor id (>= 0) or <0. */
int fdtd_add_monitor(FdtdSolver* h, int kind, int n_comps, const int32_t* comps,
                     const int32_t lo[3], const int32_t hi[3],
                     int64_t n_rec, const int64_t* steps,
                     int nf, const float* phase_e, const float* phase_h);
/* time: float [n_rec][n_comps][bz][by][bx];  dft: complex64 [nf][n_comps][bz][by][bx] */
int fdtd_get_monitor(FdtdSolver* h, int monitor_id, void* host, size_t bytes);

/* whole-volume field access [nz][ny][nx] (tests, benchmarks, checkpointing); the reference exposes
 * fields only through monitors (ref monitor.py:363), so this has no cloud counterpart */
int fdtd_set_field(FdtdSolver* h, int comp, const float* host, size_t bytes);
int fdtd_get_field(FdtdSolver* h, int comp, float* host, size_t bytes);

/* field-decay / shutoff: evaluate W = sum|E|^2 + (mu0/eps0) sum|H|^2 over the slab every `every` steps
 * (fixed-order reduction: bitwise repeatable); stop when W falls below shutoff * max W after step
 * `ref_step` (ref simulation.py:2089-2096 shutoff, "field decay" of web/core/task_core.py:537).
 * every = 0 disables.  A non-finite W ends the run with FdtdStats.diverged (ref sim_data.py:909). */
int fdtd_set_shutoff(FdtdSolver* h, int every, double shutoff, int64_t ref_step);

/* z-slab decomposition over RCCL (one process per GPU).  Rank 0 creates the id, the host
 * broadcasts it (torch.distributed), every rank calls fdtd_comm_init.  Faces with
 * FDTD_BC_NEIGHBOR exchange ghost planes with rank-1 / rank+1 (periodic z wraps). */
int fdtd_comm_unique_id(char id[128]);
int fdtd_comm_init(FdtdSolver* h, const char id[128], int rank, int n_ranks);

/* advance n_steps time steps starting at the handle's current step counter: start() + monitor()
 * of the cloud path (ref web/api/webapi.py:266,:337); the step count is Simulation.num_time_steps
 * (ref simulation.py:4226). */
int fdtd_run(FdtdSolver* h, int64_t n_steps, FdtdProgressFn progress, void* user);
/* Bloch boundaries (ref boundary.py:55-79 BlochBoundary: F(r + L_a) = bloch_phase_a F(r), complex
 * fields): two handles created from the same configuration carry the real and the imaginary part
 * (the host gives the second one the source weights times -i) and are advanced together;
 * phase[a] = 2 pi bloch_vec of axis a.  A Bloch axis x or y is laid out by the host with one ghost cell
 * at each end (device index 0 and n_real[a] + 1, PEC faces in FdtdConfig; n_real[a] = 0: no ghost
 * cells on that axis); a Bloch z uses the ghost planes of FDTD_BC_PERIODIC z faces.  On a z-slab
 * (FDTD_BC_NEIGHBOR faces) the first handle carries the communicator (fdtd_comm_init) and both parts
 * exchange their ghost planes through it; planes that wrap around a Bloch z axis are rotated where they
 * arrive.  Monitors of the pair are read per handle; a value is re + i im. */
int fdtd_run_bloch(FdtdSolver* h_re, FdtdSolver* h_im, int64_t n_steps, const double phase[3], const int n_real[3],
                   FdtdProgressFn progress, void* user);
/* ref web/api/webapi.py:370 (task status incl. "diverged"), web/core/task_core.py:537 (run info) */
int fdtd_get_stats(FdtdSolver* h, FdtdStats* out);
/* tuning knobs that may change between runs of one handle (bench A/B without re-upload) */
enum { FDTD_OPT_FLAGS = 0, FDTD_OPT_VARIANT = 1, FDTD_OPT_ZCHUNK = 2, FDTD_OPT_ROWS = 3, FDTD_OPT_XCD_REMAP = 4 /* tile order of the sweep: -1 = default (runs of 8 tiles per XCD), 0 = plain, 1 = a contiguous eighth per XCD, G > 1 = runs of G tiles */,
       FDTD_OPT_FUSED_LB = 5,
       FDTD_OPT_PML_FUSED = 6 /* axes (bit mask) whose CPML recursions run inside the fused sweep: -1 = default (one GPU: all; z-slab ranks: slab kernels), 0 = slab kernels, 6 / 7 = y z / all inside the sweep — on z-slab ranks too (set it on every rank) */,
       FDTD_OPT_BND_PLANES = 7 /* planes per boundary chunk of the fused z-slab schedule, 0 = default (2) */,
       FDTD_OPT_AUTOTUNE = 8 /* 1: time a few tile shapes of the fused sweep on the first run of grids >= 2^20 cells (default 0) */,
       FDTD_OPT_PML_SPLIT = 9 /* CPML-carrying step as three launches over interior / edge tiles: -1 = by grid size (default), 0, 1 */,
       FDTD_OPT_PLACEMENT_TRIES = 12, /* alternative placements of the field arrays the first large one-GPU run samples (0 ... 8, default 6; 0 = keep the first allocations): each costs six sweeps and a further copy of the field memory until the probe ends (candidates that lose are held so that the next one lands elsewhere) */
       FDTD_OPT_MEM_HINTS = 11, /* 1 (default): the measured store placement of the sweep (without CPML: non-temporal field stores, H ahead of the row exchange; with CPML: the H-side psi behind the E update); 0: plain stores, fields at the end of the plane, H-side psi in the H phase */
       FDTD_OPT_TBLOCK = 13, /* one-GPU fused runs without CPML / TFSF / periodic z: advance TWO time steps per pass over
                                the grid, slab by slab (slab s+1 takes step n, then slab s takes step n+1 while the
                                intermediate planes are still in the 256 MiB Infinity Cache): planes per slab; 0 = off,
                                -1 = default */
       FDTD_OPT_EDGE_ZCHUNK = 14, /* planes per workgroup of the EDGE launches of a CPML step (tiles that meet a y / z slab):
                                     -1 = default (z-slab ranks, where they share the interior launch's stream: about one wave of
                                     workgroups, at least 2 planes; one GPU: as the interior launch), 0 = as the interior launch, N */
       FDTD_OPT_GRAPH = 15, /* one-GPU fused runs on one stream: steps without monitor records / decay checks replayed as captured
                               hipGraphs of two steps: -1 = default (off: on ROCm 7.2 replaying costs ~3 us per step MORE than the
                               launches it replaces, profiles/r3i), 0 = never, 1 = whenever possible */
       FDTD_OPT_TWOSTEP = 16, /* two time steps per sweep (in-kernel temporal blocking, bit-identical to single steps; non-dispersive media, PEC
                                 walls (PMC allowed on the min faces), absorber layers, point sources while they inject (any sources once
                                 they are spent), small time monitors, DFT monitors, no decay check on the middle step; CPML and periodic
                                 faces through shell pairs (FDTD_OPT_SHELL_PAIRS), z-slab ranks without CPML with their boundary planes as
                                 the shell — everything else takes single steps, FdtdStats.fused2_off_reason says why): -1 = default (on,
                                 tile shape by grid size), 0 = off, else waves per workgroup (4 ... 16; W - 3 rows of a tile are written)
                                 + 64 * planes per chunk (0 = by grid size) */
       FDTD_OPT_SHELL_PAIRS = 17, /* step pairs on grids with CPML or periodic faces (the two-step sweep over the bulk; the shell — CPML slabs +
                                     a two-cell collar, the two rows / planes next to a periodic y / z face — as two single steps beside it
                                     on the second stream, through a third field set; periodic x wraps inside the sweep; bit-identical to
                                     single steps), and on z-slab ranks: -1 = default (on), 0 = off, 2 = shell behind the bulk on ONE stream
                                     (a measuring aid) */
       FDTD_OPT_STRIP = 18, /* x strips of a shell step: planes per workgroup (1 ... 63) + 64 * workgroups per CU their registers are cut for (3 or 4) */
       FDTD_OPT_SHELL2 = 19, /* the shell of a CPML-walled grid (no periodic z, no absorber layers; a periodic x wraps through the boxes' halo
                                lanes, the rows next to a periodic y wrap and the planes of dispersive cells take single steps beside the
                                boxes; sources that inject three or more cells inside the bulk, or on planes that become z holes; z-slab ranks
                                that carry CPML inside their sweeps too) by shell2_step_kernel — two steps per sweep with psi carried, both
                                psi sides ping-ponged, no third field set; bit-identical to single steps: -1 = default (where the cost
                                model likes it), 0 = off (two single steps beside the bulk, FDTD_OPT_SHELL_PAIRS), 1 = wherever possible, 2 / 3 = wherever possible with
                                one launch per instantiation (x / y / z only, all axes) / per box (measuring aids) */
       FDTD_OPT_SHELL2_SHAPE = 20, /* tile shapes of its launches (one per instantiation: x / y / z only, all axes; each over all of its boxes): lanes
                                      per row of the wide boxes (z / y slabs; 3 ... 64, 0 = by box: the shape that wastes the fewest lane-planes)
                                      + 128 * waves per workgroup of the all-axes launch (1 ... 8, default 8) + 1024 * planes per chunk of the wide
                                      boxes (0 = by box) + 2^17 * waves per workgroup of the one-axis launches (1 ... 8, default 6) + 2^21 * planes
                                      per chunk of the x strips (0 = by box); <= 0: defaults */
       FDTD_OPT_DEBUG_SYNC = 21, /* 1: a device-wide synchronisation in front of and behind every launch group of fdtd_run (no two launches ever
                                    overlap): a debugging aid — a schedule whose result then differs from the normal run's has a missing
                                    cross-stream edge.  Default 0 */
       FDTD_OPT_TILE_SPLIT = 22, /* the two-step sweep of a grid with bodies: workgroups whose tile (halo rows and planes included) holds only
                                    the background medium run the plain sweep inside the materials launch; same bits: -1 = default (where at
                                    least one tile in eight is background-only), 0 = never, 1 = wherever such a tile exists */
       FDTD_OPT_DISP = 23, /* dispersive (pole-residue ADE) cells advanced INSIDE the two-step sweeps: their pole states move into paged storage
                              (one block per 256-cell row segment that holds a dispersive cell, two sets) before the first run that may take
                              step pairs, and single steps update them there too.  -1 / 1 = default (on one GPU, where the packed medium words
                              name the ADE group of every dispersive cell), 0 = off: the planes of dispersive cells are z holes of the bulk
                              (single steps, round 5).  Set it before the first fdtd_run. */
       FDTD_OPT_SRC_PAGED = 25, /* step pairs WHILE a TFSF box, a mode plane, a current sheet or any list of more than 256 nodes injects (round 6):
                                   in front of each pair list kernels leave what the lists add at steps n (E side), n + 1 (H side) and n + 1
                                   (E side) in paged storage — one block per 256-cell row segment that holds a source node — and the
                                   two-step sweep, its seam kernel and the shell's boxes add them.  -1 / 1 = default (on one GPU, where no
                                   two lists meet on a node), 0 = off: single steps (or the lists' planes as z holes) while they inject. */
       FDTD_OPT_WHATIF = 24, /* measuring aid (round 6): 1 ... 8 = a what-if instantiation of the vacuum two-step sweep that skips part of its work
                                (csrc/fdtd_kernels2.hpp lists them) — WRONG results, meaningful times; 0 = off (default) */
       FDTD_OPT_SLAB_BOXES_FIRST = 26, /* z-slab ranks that carry CPML, step pairs: the shell's boxes are launched in front of the bulk sweep (1), behind it (0: round 5), beside it on a third stream (2), or 2 for slabs of 96 planes and more, else 1 (3, default) */
       FDTD_OPT_LDS_PAD = 10 /* measuring aid: extra dynamic LDS per workgroup of the sweep in bytes (lowers its occupancy) */ };
int fdtd_set_option(FdtdSolver* h, int key, int value);
int fdtd_reset(FdtdSolver* h);      /* zero fields, auxiliaries, monitors and the step counter */

#ifdef __cplusplus
}
#endif
#endif /* FDTD_HIP_H */
