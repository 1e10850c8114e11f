// shell2_step_kernel (fdtd_shell2.hpp): what fdtd_capi.hip and fdtd_shell2.hip share.
#pragma once
#include "fdtd_fused2.hpp"

namespace fdtd {

struct Shell2P {
  int q;               // lanes per row (3 .. 64): 64 / q rows per wavefront
  int xorg;            // first column of lane 0 of x tile 0 (a multiple of 4, >= 0); tile t starts (q - 2) lanes further per t
  int ci0, ci1;        // columns written: [ci0, ci1), multiples of 4
  int j0, j1;          // rows written
  int jlo;             // halo slots below a tile's first written row: 2 — or 0 for a box that starts on the y-min wall and fits ONE tile row
                       // (nby == 1): its slot 0 is row 0, which needs nothing from below (round 6: the y-min slab of a 512^3 V2 grid, 14 rows,
                       // took three tile rows of 8 slots, or two of 16, where 16 slots hold rows 0 .. 15)
  int k0, k1;          // planes written
  int zchunk;          // planes per workgroup
  int nbx, nby, nbz;   // tiles
  int paged;           // bit 0: a row segment the box visits (halo rows / planes / lanes included) holds a source node (SrcP), bit 1: a
                       // dispersive cell (DispP) — boxes without look nothing up
};

// the middle step over the boxes of DFT monitors (as InjP's dump fields, fdtd_fused2.hpp): H^{n+1/2} for records at step n, E^{n+1} for
// records at step n + 1, written by the box that owns the cell; dstart == nullptr: none
struct Shell2Dump {
  const int* dstart;
  const int* dlist;
  const DumpBox* dboxes;
  float* dump;
};
constexpr int kShell2MaxQ = 64;
constexpr int kShell2Boxes = 12;
struct Shell2M {
  int n;                             // boxes of this launch
  int first[kShell2Boxes + 1];       // workgroups of box q: [first[q], first[q + 1])
  Shell2P box[kShell2Boxes];
};

// host-side launcher (fdtd_shell2.hip): `waves` wavefronts per workgroup (<= 8); axes: the axes whose recursions the boxes can
// meet (1, 2, 4: that axis only; anything else: all)
void launch_shell2_step(hipStream_t st, int waves, bool mat, int axes, const GridP& g, const FieldP& a, const FieldP& b, const StepP& s,
                        const MatP& m, const PmlP* pm, const Shell2M& boxes, const Shell2Dump& dmp,
                        const DispP& dp = DispP{nullptr, nullptr, nullptr}, const SrcP& sr = SrcP{});

}  // namespace fdtd
