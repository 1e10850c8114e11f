// Two time steps per sweep: what fdtd_capi.hip and fdtd_fused2.hip share (the kernels live in fdtd_kernels2.hpp and are
// compiled in their own translation unit, fdtd_fused2.hip, with the SLP vectorizer off: packed-fp32 code for this kernel
// spends more on moves into aligned register pairs than it saves — 0.810 -> 0.701 ms per step inside one engine,
// profiles/r3q_*).
#pragma once
#include "fdtd_kernels.hpp"

namespace fdtd {

constexpr int kMaxInj = 256;
struct InjP {
  int n;                                   // 0: the node table is not walked (no source alive, no monitor sampling)
  const int* start;                        // [nz + 2] rows of plane k: [start[k], start[k + 1])
  const int4* ent;                         // (i, j, code, index), sorted by plane, sources first, list order kept:
                                           //   code 0 - 2: E-side source node of that component, index into val / val2
                                           //   code 3 - 5: H-side source node (H_x, H_y, H_z), index into val2
                                           //   code 8 + c: a time monitor's sample of component c (0 - 5) of the middle step -> cap[index]
  const float* val;                        // source terms of step n
  const float* val2;                       // source terms of step n+1 (nullptr: not available; then there are no H-side nodes)
  int e2_in_sweep;                         // the E-side terms of step n+1 are added to E^{n+2} by the sweep (else: by the caller behind it)
  float* cap;                              // samples of the middle step (pair_record_kernel)
  // The middle step over the boxes of DFT monitors: H^{n+1/2} for a record at step n (its H terms), E^{n+1} for a record at step
  // n+1 (its E terms) — accumulated behind the sweep, dft_record_dump_kernel.  Boxes of plane k = dlist[dstart[k] .. dstart[k + 1])
  const int* dstart;
  const int* dlist;
  const struct DumpBox* dboxes;
  float* dump;
};
struct DumpBox {
  int lo0, lo1, lo2, nx, ny, nz;           // the monitor's box
  int off[6];                              // offset of the E_x .. H_z block ([nz][ny][nx], the monitor's cell order) in `dump`, -1 = not needed
};
constexpr int kMaxDumps = 8;
// absorber layers (damp_kernel's per-axis factor tables: fb at cell boundaries, fc at cell centres; 1 outside the layers);
// fb[0] == nullptr: none
struct DampT {
  const float* fb[3];
  const float* fc[3];
  int e2;                                  // E^{n+2} is damped inside the sweep (else: by the caller, behind the sources it applies)
};
// the box a clipped launch of the two-step sweep writes: [i0, i1) x [j0, j1) x [k0, k1), i0 and i1 multiples of 4 — the bulk of a
// grid whose shell (CPML slabs + collar, boundary planes of a z-slab rank) is advanced by single steps (fdtd_capi.hip, shell pairs)
struct ClipP { int i0, i1, j0, j1, k0, k1; };
// tile classes of a launch of a materials instantiation: cls[tile] (logical tile index, y fastest) = 0 where the tile holds only the
// background medium — its workgroup runs the plain sweep; 2 (launches that carry dispersive cells) where a row segment of the tile
// holds a dispersive cell, 1 elsewhere — the materials sweep without the ADE lines.  cls == nullptr: the launch's own instantiation everywhere
struct TileClassP { const unsigned char* cls; };
// Dispersive cells in step pairs (round 6; K4 twice per pair): paged storage — one block [3 components][256 cells] of floats per row
// segment (256 cells of one row = one wavefront of the sweep) that holds a dispersive cell, found through dseg.  `cs` holds the
// memory term cc S(Q^n) of the current state at the dispersive cells (0 elsewhere), written by whichever ADE kernel formed Q^n;
// the sweep subtracts it from E^{n+1} in S2 and leaves E^{n+1} of the rows it owns in `e1` for ade2_kernel.  dseg == nullptr: none.
struct DispP {
  const int* dseg;                         // [nz][ny][ceil(nx / 256)]: block of the row segment, -1 = no dispersive cell in it
  const float* cs;                         // cc S(Q^n)
  float* e1;                               // E^{n+1} behind its ADE update
};
// Source terms of a step pair in paged storage (round 6): what TFSF boxes, mode planes, current sheets and other lists of more than
// kMaxInj nodes add to the fields while they inject — one block [3 components][256 cells] per row segment that holds a source node
// (sseg), three arrays filled in front of every pair by list kernels with the operations of point_source_kernel / tfsf_corr_kernel:
// e1 = the E-side terms of step n (added to E^{n+1} in S2, behind the walls, in front of damping and ADE), h2 = the H-side terms of
// step n+1 (added to H^{n+1/2} at the top of S3), e2 = the E-side terms of step n+1 (added to E^{n+2} in S4).  The H-side terms of
// step n act on H^{n-1/2} in front of the sweep, as always.  One term per node and side (the host checks that no two lists meet
// One term per node, side and LAYER: lists that meet on a node (the two polarisation components of a TFSF box with a pol_angle) go to
// different layers in their launch order, and (E + term_0) + term_1 is what the list kernels would have formed one after the other
// (a node only the second list touches adds a zero first).  sseg == nullptr: none.
struct SrcT {                              // (in device memory: a sweep reads it only in the few row segments that hold source nodes —
  const float* e1;                         //  as kernel arguments the eight words cost the sweep 16 scalar registers it spills for)
  const float* h2;
  const float* e2;
  int use_h2, use_e2;                      // a list has H-side / E-side nodes
  const float* e1b;                        // layer 1 (nullptr: no two lists meet)
  const float* h2b;
  const float* e2b;
};
struct SrcP {
  const int* sseg = nullptr;               // [nz][ny][ceil(nx / 256)]: block of the row segment, -1 = no source node in it
  const SrcT* t = nullptr;
};
// exchange arrays of a workgroup of `waves` rows ([.][waves][64] float4 of dynamic LDS): H1_x H1_z | H2_x H2_z | E1_x E1_z twice, and — in
// most instantiations (fused2_exj) — E_x of the next plane for the row below (fdtd_kernels2.hpp, EXJ)
#if !defined(FDTD_NO_EXJ)
#define FDTD_NO_EXJ 0       // (1: a build without it — the A/B of the other instantiations, variants/libfdtd_hip_noexj.so)
#endif
// (lb: the instantiation's launch bound, 512 / 768 / 1024 threads for <= 8 / <= 12 / <= 16 waves; opt: its OPT word)
constexpr bool fused2_exj(int lb, int opt) { return !FDTD_NO_EXJ && (opt & 8) == 0 && !(lb == 512 && (opt & 2) != 0); }
constexpr int fused2_xch_arrays(int lb, int opt) { return fused2_exj(lb, opt) ? 9 : 8; }
// dynamic LDS of a launch of fused2_step_kernel<lb, opt> with `waves` rows per workgroup (+ the absorber layers' x factors, + the (Ca, Cb) table)
constexpr size_t fused2_lds_bytes(int lb, int opt, int waves) {
  return ((size_t)fused2_xch_arrays(lb, opt) * waves * 64 + ((opt & 8) ? 2 * 64 : 0)) * sizeof(float4) + ((opt & 2) ? (size_t)kMaxMedia * sizeof(float2) : 0);
}
// (everything a workgroup of the sweep keeps in LDS is in this one dynamic allocation — no static arrays in the tile bodies, which a
//  kernel of several bodies would hold once per body — so the 160 KB of a gfx950 CU bound it here, at compile time)
static_assert(fused2_lds_bytes(1024, 1 | 2 | 4 | 16 | 32 | 64, 16) <= 160 * 1024 && fused2_lds_bytes(1024, 2 | 8, 16) <= 160 * 1024 &&
              2 * fused2_lds_bytes(512, 1, 8) <= 160 * 1024 && 2 * fused2_lds_bytes(512, 2 | 8, 8) <= 160 * 1024,
              "fused2_step_kernel: LDS of a workgroup (two per CU for eight waves)");
constexpr int kMaxCap = 1024;
constexpr int kSeamArrays = 13;  // of step one: H1_y, H1_z, E1_x, E1_y, E1_z [c-1], E1_y, E1_z [c]; of step two: H2_x [c-1], H2_y, H2_z [c-2], H2_x, H2_y, H2_z [c]
                                 // (c = first column of the right tile)

// host-side launchers (fdtd_fused2.hip)
void launch_inject_values(hipStream_t st, float* val, const float* w_re, const float* w_im, const float2* wave,
                          long long step, int n);
// waves = rows per workgroup (W - 3 of them written); opt: bit 0 non-temporal stores, bit 1 materials (m.m4 set), bit 2 monitor
// samples in the table, bit 3 absorber layers (dmp.fb[0] set), bit 4 the launch covers the box `clip` only (not with bit 3),
// bit 5 the memory terms of dispersive cells subtracted from E^{n+1} in the sweep (dp.dseg set; with bit 1; always non-temporal stores)
void launch_fused2_step(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                        const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                        int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl = TileClassP{nullptr},
                        const DispP& dp = DispP{nullptr, nullptr, nullptr}, const SrcP& sr = SrcP{});
// the instantiations that add paged source terms (opt bit 6; always with bits 0 and 2): fdtd_fused2s.hip
void launch_fused2_step_src(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                            const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                            int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl, const DispP& dp, const SrcP& sr);
// what-if instantiations (opt >> 8 = 1 ... 8 on top of opt = 1, 16 waves): fdtd_fused2w.hip
void launch_fused2_step_whatif(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                               const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                               int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip);
// the instantiations that carry dispersive cells live in their own translation unit too (fdtd_fused2d.hip)
void launch_fused2_step_disp(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                             const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                             int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl, const DispP& dp);
// the clipped instantiations live in their own translation unit (fdtd_fused2c.hip): the two compile side by side
void launch_fused2_step_clip(hipStream_t st, int waves, int opt, int grid_blocks, const GridP& g, const FieldP& a,
                             const FieldP& b, const StepP& s, const MatP& m, int zchunk, int nbx, int nby, int nbz,
                             int xcd_remap, const InjP& inj, float* seam, const DampT& dmp, const ClipP& clip, const TileClassP& tcl = TileClassP{nullptr});
void launch_inject_table(hipStream_t st, float* tab, long long stride, long long off, const float* w_re, const float* w_im,
                         const float2* wave, long long n_steps, int n);
constexpr int kPairMons = 4;
struct PairRecP {
  int n_mon;
  int pre_done;                  // E^n and the first H half-sample of records at step n were taken in front of the sweep
  BoxP box[kPairMons];
  int nc[kPairMons];
  int comp[kPairMons][6];
  int cap_off[kPairMons];
  float* out_n[kPairMons];       // the record of step n / n+1, nullptr = the monitor does not record then
  float* out_m[kPairMons];
};
void launch_pair_record(hipStream_t st, const PairRecP& r, long long max_cells, const GridP& g, const FieldP& a, const FieldP& b,
                        const float* cap);
// acc[f][slot][cell] += dump[cell] * phase[f]: terms of a DFT record from the sweep's copy of the middle step (the operations of
// dft_record_multi_kernel); n entries: (slot, offset into dump)
struct DftDumpP { int n; int slot[3]; int off[3]; };
void launch_dft_record_dump(hipStream_t st, const DftDumpP& r, const float* dump, float2* acc, long long cells, long long fstride,
                            const float2* phase, int nf);
// (inj: the seam kernel adds the E-side source terms of step n+1 when the sweep did — inj.e2_in_sweep)
void launch_seams(hipStream_t st, const GridP& g, const FieldP& b, const StepP& s, const MatP& m, const float* seam,
                  int n_seams, const DampT& dmp, const ClipP& clip, const InjP& inj, const SrcP& sr = SrcP{});

}  // namespace fdtd
