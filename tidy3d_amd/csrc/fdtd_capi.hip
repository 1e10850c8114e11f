// libfdtd_hip.so — host side of the C ABI declared in include/fdtd_hip.h.
// Owns device memory, HIP streams/events and the RCCL communicator of one z-slab solver.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "../../include/fdtd_hip.h"
#include "fdtd_kernels.hpp"
#include "fdtd_fused2.hpp"
#include "fdtd_strip.hpp"
#include "fdtd_shell2_host.hpp"
#include "fdtd_aniso.hpp"

using namespace fdtd;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct PmlAxisDev {
  int n_lo = 0, n_hi = 0, n = 0;   // true layer counts (what the slab kernels visit)
  // membership of the psi arrays (same on the E and the H side, fdtd_kernels.hpp PmlAxisP): index ia is
  // held when ia < lo or ia >= hi0; along x both are multiples of 4 cells
  int lo = 0, hi0 = 0, ns = 0;
  float4 *ce4 = nullptr, *ch4 = nullptr;                       // {1/kappa - 1, b, c, 0} per index
  float *kv_e = nullptr, *b_e = nullptr, *c_e = nullptr, *kv_h = nullptr, *b_h = nullptr, *c_h = nullptr;
  // psi arrays: [comp slot 0/1]
  float* psi_e[2] = {nullptr, nullptr};
  float* psi_h[2] = {nullptr, nullptr};
  float* psi_h2[2] = {nullptr, nullptr};   // write set of the in-sweep CPML (ping-pong), lazily allocated
  float* psi_e2[2] = {nullptr, nullptr};   // write set of the E side for shell2_step_kernel (two steps per sweep carry psi: both sides ping-pong), lazily allocated
  float* psi_ht[2] = {nullptr, nullptr};   // temporary sets: the middle step of the z holes of a shell2 pair (single steps beside boxes that
  float* psi_et[2] = {nullptr, nullptr};   // still read the old values), lazily allocated
  size_t psi_count = 0;                    // entries per psi array
  size_t psi_plane = 0;                    // entries of one z-plane of it (x, y axes: the H-side arrays carry one more, the ghost slot)
};

struct AdeGroup {
  int comp;
  int k0 = 0, k1 = 0;              // planes [k0, k1) that hold its cells: launches over other plane ranges are skipped
  std::vector<long long> plane_off;  // sorted lists: entries of plane k are [plane_off[k], plane_off[k + 1]); empty = unsorted
  bool strict = false;               // sorted and no cell listed twice
  long long n;
  uint32_t* cell;
  float* e_old;
  float2* q;
  AdeP p;
  uint32_t* qoff = nullptr;        // slot of entry t in the paged arrays (Disp)
};

// Dispersive cells in step pairs (round 6; fdtd_fused2.hpp DispP).  What a two-step sweep needs of the ADE update of its first step is
// the memory term cc S(Q^n) at the dispersive cells — known before the sweep starts.  Once a run may take such pairs it is kept in
// paged storage (`cs`: one block per row segment that holds a dispersive cell) by every ADE kernel of this handle; the sweep subtracts
// it and leaves E^{n+1} in `e1`; ade2_kernel behind the sweep advances the pole states two steps.  The lists stay what they were.
struct Disp {
  int state = 0;                   // 0 = not tried, 1 = ready, -1 = this problem cannot
  int n_blocks = 0;
  int* dseg = nullptr;             // [nz][ny][nbx]
  std::vector<int> dseg_host;
  float* cs = nullptr;             // [n_blocks][3][256]
  float* e1 = nullptr;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};     // bounding box of the dispersive cells [lo, hi)
  long long pairs = 0;             // pairs of the last run whose sweep carried the ADE update of its first step
};

struct AnisoGroup {                 // fdtd_aniso.hpp: the off-diagonal coupling of E component `comp` inside fully anisotropic bodies
  int comp;
  long long n;
  uint32_t *cell, *nbr;
  float *w_new, *w_old, *old, *delta;
};

struct PointSrc {
  long long n_e = 0, n_h = 0;      // points split by E / H components
  int32_t *comp_e = nullptr, *comp_h = nullptr;
  uint32_t *cell_e = nullptr, *cell_h = nullptr;
  float *wre_e = nullptr, *wim_e = nullptr, *wre_h = nullptr, *wim_h = nullptr;
  float2 *wave_e = nullptr, *wave_h = nullptr;
  long long n_steps = 0;
  int ke0 = 0, ke1 = 0, kh0 = 0, kh1 = 0;   // planes [k0, k1) that hold its E / H points
  uint32_t *soff_e = nullptr, *soff_h = nullptr;   // slots of the nodes in the paged source terms (SrcPaged)
  int layer_e = 0, layer_h = 0;                    // their layers
  std::vector<int32_t> host_comp_e, host_comp_h;   // host copies of the nodes (the two-step sweep applies them in-kernel)
  std::vector<uint32_t> host_cell_e, host_cell_h;
};

// correction list of one side (E or H) of a TFSF box, grouped by target node (tfsf_corr_kernel)
struct TfsfList {
  int k0 = 0, k1 = 0;              // planes [k0, k1) that hold its target nodes
  long long n_targets = 0;
  int32_t *comp = nullptr, *start = nullptr, *aux = nullptr;
  uint32_t* cell = nullptr;
  float* w = nullptr;
  uint32_t* soff = nullptr;        // slots of the target nodes in the paged source terms (SrcPaged)
  int layer = 0;                   // their layer
};

struct Tfsf {
  int n_aux = 0, src_cell = 0;
  float *ae = nullptr, *be = nullptr, *ah = nullptr, *bh = nullptr, *e1 = nullptr, *h1 = nullptr, *wave = nullptr;
  float *e1c = nullptr, *h1c = nullptr;   // replica advanced on the comm stream (pipelined z-slab schedule)
  long long n_steps = 0;
  TfsfList e, h;
};

struct Monitor {
  int kind = 0;
  std::vector<int> comps;
  BoxP box{};
  std::vector<long long> steps;
  size_t next = 0;                 // next entry of `steps` to record
  int nf = 0;
  float2 *phase_e = nullptr, *phase_h = nullptr;   // [n_rec][nf]
  void* data = nullptr;
  size_t data_bytes = 0;
  long long cells = 0;
};

// node table of one kind of step pair of the two-step sweep: the E-side source nodes and the nodes small time monitors
// sample of the middle step (InjP), for one set of recording monitors
struct F2Table {
  bool with_sources = true;        // the source nodes are listed (false: the table of a pair whose source lists are all spent)
  std::vector<int> mons;           // time monitors whose middle-step samples the sweep copies out (ascending)
  std::vector<int> cap_off;        // their offsets into the sample buffer
  std::vector<int> dfts;           // DFT monitors that record at the first (flag 1) / middle (flag 2) step: the sweep copies
  std::vector<int> dft_when;       // H^{n+1/2} / E^{n+1} over their boxes out
  std::vector<std::array<int, 6>> dft_off;     // offsets of their E_x .. H_z blocks in the dump buffer (-1: not needed)
  int* start = nullptr;
  int4* ent = nullptr;
  int *dstart = nullptr, *dlist = nullptr;
  DumpBox* dboxes = nullptr;
  long long dump_floats = 0;
};
struct F2Plan {                    // one step pair: the monitors that record at its first or middle step
  std::vector<int> mons;           // small time monitors
  std::vector<int> dfts;           // DFT monitors recording at the first and / or the middle step
  std::vector<int> dft_when;       // bit 0: at step n, bit 1: at step n + 1
};

}  // namespace

// Paged source terms (round 6; fdtd_fused2.hpp SrcP): step pairs while a TFSF box, a mode plane, a current sheet — any list the
// two-step sweep's node table cannot hold — injects.  In front of every such pair list kernels leave what the lists would add at steps
// n (E side), n + 1 (H side) and n + 1 (E side) in three paged arrays; the sweep, the seam kernel and the shell's boxes add them.
struct SrcPaged {
  int state = 0;                   // 0 = not tried, 1 = ready, -1 = this problem cannot (two lists meet on a node, ...)
  int n_blocks = 0;
  int* sseg = nullptr;             // [nz][ny][nbx]
  float *e1 = nullptr, *h2 = nullptr, *e2 = nullptr;     // [n_blocks][3][256]: layer 0
  float *e1b = nullptr, *h2b = nullptr, *e2b = nullptr;  // layer 1: lists that meet an earlier list on a node (nullptr: none does)
  bool any_h = false;              // a list has H-side nodes
  long long pairs = 0;
  std::vector<int> sseg_host;
  SrcT* tab = nullptr;             // the arrays' addresses, in device memory (SrcP::t)
};

struct FdtdSolver {
  FdtdConfig cfg{};
  GridP g{};
  std::string err;
  std::vector<DevBuf> bufs;          // everything hipMalloc'ed (freed in destroy)
  float* fbase[6] = {};              // allocation base (ghost plane -1)
  FieldP f{};                        // interior plane 0 of the CURRENT field set
  float* fbase2[6] = {};             // second set for the fused (ping-pong) sweep, lazily allocated
  FieldP f2{};
  float* fbase3[6] = {};             // third set: the middle step of a shell pair over the shell (lazily allocated)
  FieldP f3{};
  float* step_base[2] = {};          // 1/primal_z, 1/dual_z with one ghost entry on each side
  size_t field_bytes = 0;
  float *ip[3] = {}, *idl[3] = {};
  uint32_t* mat4 = nullptr;          // packed material words (interior plane 0), nullptr = uniform
  uint32_t* mat4b = nullptr;         // wide layout (more than 1023 media): the second word per cell; mat4 then holds E_x | E_y << 16
  uint32_t* roww = nullptr;          // row-segment words [nz][ny][ceil(nx / 256)]
  std::vector<uint32_t> roww_host;   // host copy of them: the tile classes of the two-step sweep are derived from it (tile_classes)
  struct TileClasses { int W, zc; ClipP box; int nbx, nby, nbz; unsigned char* dev; long long n_bg, n_all; bool disp, src; };
  std::vector<TileClasses> tile_cls; // one entry per launch shape met so far
  int tile_split = -1;               // FDTD_OPT_TILE_SPLIT: background-only tiles on the plain instantiation: -1 = default (where >= 25 % of the tiles are), 0 = never, 1 = always
  float2* lut = nullptr;
  int n_media = 0;
  float ca1 = 1.f, cb1 = 0.f;
  std::vector<float> cb_host;
  PmlAxisDev pml[3];
  // absorber layers: per-axis damping tables (identity tables of length N for axes without layers)
  float *damp_fb[3] = {}, *damp_fc[3] = {};
  int damp_lo[3] = {0, 0, 0}, damp_hi[3] = {0, 0, 0};   // layers: index < lo or index >= hi
  bool has_damp = false;
  std::vector<AdeGroup> ade;
  Disp disp;
  SrcPaged spg;
  std::map<std::array<int, 6>, int> box_paged;       // Shell2P::paged of the shell's boxes met so far (cleared when either paging is set up)
  int spg_on = -1;                   // FDTD_OPT_SRC_PAGED: -1 / 1 = default (on), 0 = off (single steps / z holes while such lists inject, round 5)
  int whatif = 0;                    // FDTD_OPT_WHATIF: a what-if instantiation of the vacuum two-step sweep (fdtd_kernels2.hpp; wrong results, meaningful times)
  int disp_on = -1;                  // FDTD_OPT_DISP: dispersive cells inside the two-step sweeps: -1 = default (on), 0 = off (their planes as z holes, round 5)
  std::vector<AnisoGroup> aniso;
  std::vector<PointSrc> psrc;
  std::vector<Tfsf> tfsf;
  std::vector<Monitor> mons;
  hipStream_t stream = nullptr, comm_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> kev;       // per-launch timing events (FDTD_FLAG_TIME_KERNELS)
  std::vector<int> kev_kind;
  double* energy_dev = nullptr;      // [0] = total, [1 .. 1 + kEnergyBlocks) = per-workgroup partial sums
  int decay_every = 0;
  double shutoff = 0.0;
  long long decay_ref = 0;
  double energy_max = 0.0;
  long long step = 0;
  FdtdStats stats{};
  int zchunk = 2;
  int zchunk_f = 16;                 // planes marched per workgroup by the fused sweep (8 without in-sweep CPML: launch_fused_range)
  int last_zc = 0;
  int rows_f = 3;                    // rows per workgroup of the fused sweep (+1 halo wave = 256 threads:
                                     // ~150 VGPRs without spills, 3 workgroups per CU; measured best, profiles/r01g)
  // tile order of the sweep: -1 = default (runs of kTileRun tiles per XCD), 0 = plain, 1 = one contiguous eighth of
  // the tiles per XCD (round 1), G > 1 = runs of G tiles per XCD; measurements at launch_fused_range
  int xcd_remap = -1;
  int fused_lb = 0;                  // 0 = by workgroup size, else forced __launch_bounds__ variant
  // axis mask of the CPML recursions folded into the fused sweep (single GPU): 0 = slab kernels,
  // 6 = y and z, 7 = all.  Measured on 512^3 + 12-layer PML (profiles/r01h_pml_placement.txt): the
  // extra live registers drop the sweep from 3 to 2 (mask 6) or 1 (mask 7) waves per SIMD, which
  // costs more (+0.41 ms / +3.0 ms) than the slab kernels it removes (0.22 ms / 0.45 ms) -> default 0.
  int pml_fused = -1;                // -1 = default
  // placement of the field arrays: how many alternative sets of allocations the first large run tries (probe_placement)
  int placement_tries = 6;
  int slab_boxes_first = 3;          // FDTD_OPT_SLAB_BOXES_FIRST: the shell's boxes of a CPML slab-rank pair in front of the bulk sweep (1), behind it (0),
                                     // beside it on a stream of their own (2: box_stream), or 2 for slabs of 96 planes and more, else 1 (3, default)
  hipStream_t box_stream = nullptr;  // (made at the first such pair: make_box_stream)
  bool box_stream_tried = false;
  int box_stream_attempts = 0;
  hipEvent_t ev_box = nullptr, ev_box_in = nullptr;      // the boxes done / what st had issued in front of them
  bool placement_done = false;
  float placement_ms[9] = {};      // time of the probe sweeps per candidate (the first is the original)
  int placement_tried = 0, placement_kept = 0;        // candidates timed beyond the original / index of the one kept (0 = original)
  int mem_hints = 1;                 // FDTD_OPT_MEM_HINTS: 1 = non-temporal field stores in the sweep's instantiations without CPML
  int lds_pad = 0;                   // extra dynamic LDS per workgroup of the sweep (bytes): lowers its occupancy — a measuring aid
  int pml_split = -1;                // three launches (interior / y-edge / z-edge tiles): -1 = by grid size, 0 = one launch, 1 = always
  // opt-in (FDTD_OPT_AUTOTUNE): time a few (rows, z-chunk) tile shapes on the first run and keep the
  // best.  Measured (profiles/r01h_autotune.txt): +6 % on a 64-plane slab (3 x 32 instead of 3 x 16),
  // nothing at 512^3 (shapes within noise of each other, so the pick is noise too) and the wrong
  // objective for the pipelined z-slab schedule (it times the whole range) -> off by default.
  int autotune = 0;
  bool tuned = false, user_geometry = false;
  int bnd_planes = 0;                // fused z-slab schedule: planes per boundary chunk (0 = heuristic)
  int rows = 4;
  PmlP* pml_blk[8][2][2] = {};       // device parameter blocks of the in-sweep CPML: [axis mask][psi_h parity][psi_e parity] (E side in place)
  PmlP* pml_blk2[2][2] = {};         // the same for shell2_step_kernel: all axes, both sides read one set and write the other
  PmlP* pml_blk_hole[2][2][2] = {};  // z holes of a shell2 pair: [step of the pair][psi_h parity][psi_e parity] — step one current -> temporary sets, step two temporary -> other sets
  bool pml_blk_hole_ok = false;
  bool pml_blk_ok[8] = {};
  bool pml_blk2_ok = false;
  int pml_parity = 0;
  int pml_e_parity = 0;              // flips when a step pair wrote the E-side psi into the other set (shell2 pairs)
  // RCCL
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  hipEvent_t ev_h_int = nullptr, ev_h_bnd = nullptr, ev_e_int = nullptr, ev_e_bnd = nullptr;
  // the two streams must really overlap (probe_stream_overlap): 0 = not probed, 1 = verified, -1 = one stream in use
  int stream_overlap = 0, stream_retries = 0;
  bool streams_shared = false;       // comm_stream is an alias of stream (fallback)
  // slab-interleaved two-step schedule (fdtd_run): planes per slab; 0 = off, -1 = default
  int tblock = -1;
  int edge_zchunk = -1;
  // Captured step pairs (small grids): a run of steps without monitor records or field-decay checks is replayed as
  // hipGraphs of TWO steps each (set a -> b -> a, psi parity back where it was), one per (field set, psi parity) state
  int use_graph = -1;                // -1 = default (= never: no gain measured, fdtd_run), 0 = never, 1 = whenever possible
  long long* step_dev = nullptr;     // device-side step counter the captured source kernels read
  long long step_dev_value = -1;     // what it holds (host mirror)
  bool step_dev_mode = false;        // launches are being captured: source kernels take step_dev + step_dev_off
  long long step_dev_off = 0;
  long long graph_pairs = 0;
  int mirror_wall[3] = {-1, -1, -1}; // PMC on plus faces: wall index per axis (fdtd_set_mirror_plus), -1 = none
  int graph_status = 0;              // 0 = no capture attempted, 1 = captured, < 0 = -(100 * stage + hipError) of the failed capture              // z-chunk of the edge launches of a CPML step: -1 = about one wave of workgroups, 0 = as the interior, N = planes
  long long two_step_pairs = 0;
  int tblock_used = 0;
  // two time steps per sweep (fused2_step_kernel): waves per workgroup (0 = off) and planes per chunk
  int twostep_w = -1, twostep_zc = 0;      // -1 / 0: chosen by fused2_shape
  int twostep_w_used = 0, twostep_zc_used = 0;
  long long fused2_pairs = 0;
  // shell pairs: the two-step sweep over the bulk of a CPML-walled grid, its shell (slabs + collar) by two single steps beside
  // it (fdtd_run).  shell_on: -1 = default (on), 0 = off
  int shell_on = -1;
  int strip_zc = 4;                   // planes per workgroup of the x strips (4: 1.199 ms per V2 step, 8: 1.205, 16: 1.228 inside one engine, profiles/r4/r4n)
  int strip_occ = 3;                  // their register budget: workgroups per CU (3 or 4)
  long long shell_pairs = 0;
  // shell2 pairs (round 5): the shell of a CPML-walled grid advanced by shell2_step_kernel — two steps per sweep with psi carried —
  // instead of two single steps through the third field set.  shell2_on: -1 = default (where the cost model likes it), 0 = off, 1 = wherever possible
  int shell2_on = -1;
  int shell2_qw = 0, shell2_ww = 8, shell2_zcw = 0;      // lanes per row of the wide boxes (z / y slabs; 0 = by box), waves per workgroup of the launch, planes per chunk (0 = by box)
  int shell2_ws = 6, shell2_zcs = 0;                     // waves per workgroup of the one-axis launches; x strips: planes per chunk (0 = by box)
  long long shell2_pairs = 0;
  int f2_off_reason = 0;              // why the last fdtd_run took no step pairs (FDTD_F2_OFF_*), 0 = it did / could
  int debug_sync = 0;                 // FDTD_OPT_DEBUG_SYNC: a device-wide synchronisation behind every launch group (dbg_sync) — no two launches ever overlap
  hipEvent_t ev_shell_a = nullptr, ev_shell_b = nullptr;
  hipEvent_t ev_rec = nullptr;        // z-slab ranks: a monitor record on the main stream done (the comm stream's next boundary work waits for it)
  float* seam_buf = nullptr;          // intermediate values on the seams between x tiles
  float* inj_val = nullptr;           // source terms applied between the two steps
  float* cap_val = nullptr;           // samples of the middle step (small time monitors)
  float* dump_buf = nullptr;          // H^{n+1/2} over the boxes of DFT monitors recording at the first step of a pair
  long long dump_cap = 0;
  std::deque<F2Table> f2_tables;      // node tables, one per set of recording monitors met so far (a deque: pointers to its entries stay valid)
  size_t inj_sources = 0;             // point-source lists the tables were built from
  float* src_tab = nullptr;           // [step][node] source terms of every step, formed once (nullptr: per pair)
  long long src_tab_steps = 0, src_tab_nodes = 0;
  bool src_on_seam = false;           // an E-side source node lies next to a seam between x tiles: step n+1's terms go behind the launch
  bool src_h_on_seam = false;         // an H_y / H_z source node in the column left of a seam: no pairs while that list is alive
  long long src_nodes = 0;            // source nodes of all lists
  int f2_dyn_reason = 0;              // the last reason a step of this run could not open a pair because of its sources (FDTD_F2_OFF_*)
  long long src_h_nodes = 0;          // H-side source nodes (need the table of all steps)
  bool src_e_in_damp = false;         // an E-side source node lies inside an absorber layer (its damping factor is not exactly 1)
};

namespace {

int fail(FdtdSolver* h, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return -1;
}

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(h, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define NCCLCHK(h, call)                                                                  \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess)                                                                \
      return fail(h, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
int dev_alloc(FdtdSolver* h, T** out, size_t count, bool zero_fill = true) {
  void* p = nullptr;
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  HIPCHK(h, hipMalloc(&p, bytes));
  if (zero_fill) {
    // hipMemset on device memory may return before the fill has run, and it runs on the NULL stream — which the engine's
    // non-blocking streams do not wait for.  Every allocation is handed out filled.
    HIPCHK(h, hipMemset(p, 0, bytes));
    HIPCHK(h, hipStreamSynchronize(nullptr));
  }
  h->bufs.push_back({p, bytes});
  h->stats.device_bytes += (int64_t)bytes;
  *out = reinterpret_cast<T*>(p);
  return 0;
}

template <typename T>
int dev_upload(FdtdSolver* h, T** out, const T* host, size_t count) {
  if (dev_alloc(h, out, count, false)) return -1;
  if (count) HIPCHK(h, hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// The twelve field arrays: one allocation per array.
//
// WHERE they land in device memory decides up to 15 % of the sweep's speed (profiles/r02q_probe_series.jsonl,
// r02t_probe_memory_depth.jsonl: the same kernel on the same box runs 1.174, 1.25 or 1.35 ms per 512^3 step depending
// on the engine's allocations, to +-0.2 % within an engine and with all clocks unchanged; a fresh process's first
// engine normally gets the fast state).  Virtual addresses do not show it, distances between the arrays inside one
// allocation do not change it (r02r), nor does aligning the virtual addresses to the array size — through
// $HSA_MAX_VA_ALIGN or through hipMemAddressReserve + hipMemMap (r02u, r02v: 1.25 ms mapped against 1.17-1.19 ms
// hipMalloc'ed on the same box) — so it is a property of the physical pages, nothing this library controls.
// -DFDTD_PLACEMENT_PROBE compiles the placements those measurements used ($FDTD_FIELD_LAYOUT, scripts/probe_layout.py).
#ifdef FDTD_PLACEMENT_PROBE
// $FDTD_FIELD_LAYOUT, read at every fdtd_create:
//   0  one allocation per array (default)          1  one allocation per field set (six arrays back to back)
//   3  ONE allocation: array c of set q starts at  q * (6 * stride + $FDTD_FIELD_S2) + c * stride  bytes,
//      stride = array bytes + $FDTD_FIELD_S1  (both multiples of 16)
inline long env_long(const char* name, long dflt) {
  const char* e = std::getenv(name);
  return e ? std::atol(e) : dflt;
}
int alloc_field_set(FdtdSolver* h, float** base, size_t fcount, int set) {
  const int mode = (int)env_long("FDTD_FIELD_LAYOUT", 0);
  if (mode == 3 || mode == 4) {
    if (set == 1) return 0;                                   // both sets were placed by the first call
    const size_t s0 = (size_t)env_long("FDTD_FIELD_S0", 0) / 16 * 4;
    const size_t s1 = (size_t)env_long("FDTD_FIELD_S1", 0) / 16 * 4, s2 = (size_t)env_long("FDTD_FIELD_S2", 0) / 16 * 4;
    const size_t stride = fcount + s1;
    float* blk = nullptr;
    if (mode == 4) {                                          // 4: as 3, inside a pool that lives as long as the process
      static float* pool = nullptr;                           //    (the SAME memory for every engine; $FDTD_FIELD_S0 = offset into it)
      const size_t pool_floats = (size_t)env_long("FDTD_FIELD_POOL", 16L << 30) / 4;
      if (!pool) HIPCHK(h, hipMalloc((void**)&pool, pool_floats * 4));
      if (s0 + 12 * stride + s2 > pool_floats) return fail(h, "field pool too small");
      blk = pool + s0;
      HIPCHK(h, hipMemset(blk, 0, (12 * stride + s2) * 4));
    } else if (dev_alloc(h, &blk, 12 * stride + s2)) return -1;
    for (int c = 0; c < 6; ++c) {
      h->fbase[c] = blk + (size_t)c * stride;
      h->fbase2[c] = blk + 6 * stride + s2 + (size_t)c * stride;
    }
    if (env_long("FDTD_DEBUG_ADDR", 0)) std::fprintf(stderr, "field block at %p, stride %zu bytes\n", (void*)blk, stride * 4);
    return 0;
  }
  if (mode == 1) {
    float* blk = nullptr;
    if (dev_alloc(h, &blk, 6 * fcount)) return -1;
    for (int c = 0; c < 6; ++c) base[c] = blk + (size_t)c * fcount;
    return 0;
  }
  // 5: one allocation per array, its size rounded up to a multiple of $FDTD_FIELD_ROUND bytes
  const size_t round = mode == 5 ? (size_t)env_long("FDTD_FIELD_ROUND", 1L << 30) / 4 : 1;
  for (int c = 0; c < 6; ++c) {
    if (dev_alloc(h, &base[c], (fcount + round - 1) / round * round)) return -1;
    if (env_long("FDTD_DEBUG_ADDR", 0)) std::fprintf(stderr, "field set %d array %d at %p\n", set, c, (void*)base[c]);
  }
  return 0;
}

#else
int alloc_field_set(FdtdSolver* h, float** base, size_t fcount, int) {
  for (int c = 0; c < 6; ++c)
    if (dev_alloc(h, &base[c], fcount)) return -1;
  return 0;
}
#endif

constexpr int kBndPlanes = 2;        // boundary chunk of the z-slab schedule, planes per neighbour face (fdtd_run)
constexpr int kPlainZChunk = 8;      // z-chunk of sweeps without in-sweep CPML (launch_fused_range)
constexpr int kTileRun = 8;          // default tile order of the sweep: runs of 8 tiles per XCD (launch_fused_range)
// cost model of shell2 pairs (shell2_why_not): ps per cell and PAIR, measured on MI355X at 512^3 V2 with each launch alone on
// the machine (profiles/r5/r5d, r5e): the clipped bulk sweep 12.9 (with materials; 11.3 without), a cell of a wide box 27, a
// strip cell 41 (the boxes as one launch: 0.69 ms for 21.3 M cells, half of it the strips); the two streams overlap to 0.93 of
// the sum; a single step costs 10.1 ps per cell.  V2: predicted 0.74 of two single steps, measured 0.76; BASELINE config 3 laid
// out with x = 224: predicted 0.83, measured 0.86.
constexpr double kShell2BulkPs = 11.3, kShell2BulkMatPs = 12.9, kShell2WidePs = 27.0, kShell2StripPs = 41.0, kShell2Overlap = 0.93;

inline unsigned nblk(long long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

// planes [k0, k1) touched by a list of linear cell indices (empty list: k0 == k1 == 0)
inline void plane_range(const uint32_t* cell, long long n, long long sxy, int* k0, int* k1) {
  *k0 = *k1 = 0;
  if (n <= 0) return;
  uint32_t lo = cell[0], hi = cell[0];
  for (long long i = 1; i < n; ++i) { lo = std::min(lo, cell[i]); hi = std::max(hi, cell[i]); }
  *k0 = (int)(lo / (uint64_t)sxy);
  *k1 = (int)(hi / (uint64_t)sxy) + 1;
}
inline bool planes_meet(int k0, int k1, int kbeg, int kend) { return k0 < kend && kbeg < k1; }

long long plane_cells(const FdtdSolver* h) { return (long long)h->cfg.nx * h->cfg.ny; }
long long n_cells(const FdtdSolver* h) { return plane_cells(h) * h->cfg.nz; }

float* field_ptr(FdtdSolver* h, int comp) {
  switch (comp) {
    case 0: return h->f.ex; case 1: return h->f.ey; case 2: return h->f.ez;
    case 3: return h->f.hx; case 4: return h->f.hy; default: return h->f.hz;
  }
}

StepP step_params(const FdtdSolver* h) {
  StepP s;
  s.ipx = h->ip[0]; s.ipy = h->ip[1]; s.ipz = h->ip[2];
  s.idx = h->idl[0]; s.idy = h->idl[1]; s.idz = h->idl[2];
  return s;
}

MatP mat_params(const FdtdSolver* h) {
  MatP m;
  m.m4 = h->mat4;
  m.m4b = h->mat4b;
  m.roww = h->roww;
  m.lut = h->lut; m.n_media = h->n_media; m.ca1 = h->ca1; m.cb1 = h->cb1;
  return m;
}

// ---- timing of the main kernels -----------------------------------------------------------
// FDTD_OPT_DEBUG_SYNC: everything issued so far — on either stream — has finished before the next launch group goes out.  Called at
// the head of every launch helper and around every sweep launch: with it no two launches of a run ever overlap, so a result that
// differs from the normal run's points at a missing cross-stream edge (tests/test_gpu_parity.py).
inline void dbg_sync(const FdtdSolver* h) { if (h->debug_sync) (void)hipDeviceSynchronize(); }

void time_begin(FdtdSolver* h, int kind, hipStream_t st) {
  dbg_sync(h);
  if (!(h->cfg.flags & FDTD_FLAG_TIME_KERNELS)) return;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, st);
  h->kev.push_back(a);
  h->kev.push_back(b);
  h->kev_kind.push_back(kind);
}
void time_end(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  if (!(h->cfg.flags & FDTD_FLAG_TIME_KERNELS)) return;
  hipEventRecord(h->kev.back(), st);
}

// ---- launches -------------------------------------------------------------------------------
void launch_h_main(FdtdSolver* h, int kbeg, int kend, hipStream_t st) {
  if (kend <= kbeg) return;
  const GridP& g = h->g;
  const bool vec = (g.nx % 4 == 0) && h->cfg.variant != FDTD_VARIANT_SIMPLE;
  const int V = vec ? 4 : 1;
  const int zc = (h->cfg.variant == FDTD_VARIANT_SIMPLE) ? 1 : h->zchunk;
  const int rows = h->rows;                             // <= 8: __launch_bounds__(512)
  dim3 block(64, rows, 1);
  dim3 grid((g.nx + 64 * V - 1) / (64 * V), (g.ny + rows - 1) / rows, (kend - kbeg + zc - 1) / zc);
  time_begin(h, 0, st);
  if (vec)
    hipLaunchKernelGGL((h_update_kernel<4>), grid, block, 0, st, g, h->f, step_params(h), kbeg, kend, zc);
  else
    hipLaunchKernelGGL((h_update_kernel<1>), grid, block, 0, st, g, h->f, step_params(h), kbeg, kend, zc);
  time_end(h, st);
}

void launch_e_main(FdtdSolver* h, int kbeg, int kend, hipStream_t st) {
  if (kend <= kbeg) return;
  const GridP& g = h->g;
  const bool vec = (g.nx % 4 == 0) && h->cfg.variant != FDTD_VARIANT_SIMPLE;
  const int V = vec ? 4 : 1;
  const int zc = (h->cfg.variant == FDTD_VARIANT_SIMPLE) ? 1 : h->zchunk;
  const bool has_mat = h->mat4 != nullptr;
  const int rows = h->rows;                             // <= 8: __launch_bounds__(512)
  dim3 block(64, rows, 1);
  dim3 grid((g.nx + 64 * V - 1) / (64 * V), (g.ny + rows - 1) / rows, (kend - kbeg + zc - 1) / zc);
  MatP m = mat_params(h);
  StepP s = step_params(h);
  time_begin(h, 1, st);
  if (vec && has_mat)
    hipLaunchKernelGGL((e_update_kernel<4, true>), grid, block, 0, st, g, h->f, s, m, kbeg, kend, zc);
  else if (vec)
    hipLaunchKernelGGL((e_update_kernel<4, false>), grid, block, 0, st, g, h->f, s, m, kbeg, kend, zc);
  else if (has_mat)
    hipLaunchKernelGGL((e_update_kernel<1, true>), grid, block, 0, st, g, h->f, s, m, kbeg, kend, zc);
  else
    hipLaunchKernelGGL((e_update_kernel<1, false>), grid, block, 0, st, g, h->f, s, m, kbeg, kend, zc);
  time_end(h, st);
}

// fused E+H sweep over the planes [kbeg, kend): reads the current set, writes the other one
int ensure_second_set(FdtdSolver* h) {
  const GridP& g = h->g;
  if (h->f2.ex) return 0;
  const size_t fcount = (size_t)g.sxy * (g.nz + 2);
  if (!h->fbase2[0] && alloc_field_set(h, h->fbase2, fcount, 1)) return -1;
  h->f2.ex = h->fbase2[0] + g.sxy; h->f2.ey = h->fbase2[1] + g.sxy; h->f2.ez = h->fbase2[2] + g.sxy;
  h->f2.hx = h->fbase2[3] + g.sxy; h->f2.hy = h->fbase2[4] + g.sxy; h->f2.hz = h->fbase2[5] + g.sxy;
  return 0;
}

bool any_pml(const FdtdSolver* h) {
  for (int a = 0; a < 3; ++a) if (h->pml[a].n_lo + h->pml[a].n_hi > 0) return true;
  return false;
}

// Device parameter blocks of the in-sweep CPML: block [hp][ep] reads psi_h (hp = 0) or psi_h2 (hp = 1) and
// writes the other set; the sweep alternates between them.  The E side is updated in place, in the set that is current
// ([ep]: a shell2 pair leaves it in the other one).  Built on first use.
void fill_pml_axis(const FdtdSolver* h, int a, bool in, bool flip_h, bool flip_e, bool e_in_place, PmlAxisP& A) {
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  const PmlAxisDev& P = h->pml[a];
  A.ce4 = P.ce4; A.ch4 = P.ch4;
  A.kv_e = P.kv_e; A.b_e = P.b_e; A.c_e = P.c_e;
  A.kv_h = P.kv_h; A.b_h = P.b_h; A.c_h = P.c_h;
  float* const e_cur[2] = {flip_e && P.psi_e2[0] ? P.psi_e2[0] : P.psi_e[0], flip_e && P.psi_e2[1] ? P.psi_e2[1] : P.psi_e[1]};
  float* const e_oth[2] = {flip_e || !P.psi_e2[0] ? P.psi_e[0] : P.psi_e2[0], flip_e || !P.psi_e2[1] ? P.psi_e[1] : P.psi_e2[1]};
  A.pe0 = e_cur[0]; A.pe1 = e_cur[1];
  A.pe0n = e_in_place ? e_cur[0] : e_oth[0]; A.pe1n = e_in_place ? e_cur[1] : e_oth[1];
  A.ph0 = flip_h ? P.psi_h2[0] : P.psi_h[0]; A.ph1 = flip_h ? P.psi_h2[1] : P.psi_h[1];
  A.ph0n = flip_h ? P.psi_h[0] : P.psi_h2[0]; A.ph1n = flip_h ? P.psi_h[1] : P.psi_h2[1];
  // an axis without CPML, or one whose recursions stay in the slab kernels, has no members
  A.lo = in ? P.lo : 0; A.hi0 = in ? P.hi0 : N[a]; A.ns = P.ns; A.n = N[a];
}
int ensure_pml_blocks(FdtdSolver* h, int mask) {
  if (h->pml_blk_ok[mask]) return 0;
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.ns == 0) continue;
    for (int q = 0; q < 2; ++q)
      if (!P.psi_h2[q] && dev_alloc(h, &P.psi_h2[q], P.psi_count + P.psi_plane)) return -1;
  }
  // The two sets of an axis keep their identity: `psi_h` / `psi_h2` (`psi_e` / `psi_e2`) swap names on the host after every
  // sweep (shell2 pair), so block [hp][ep] must read what the host calls psi_h (psi_e) when pml_parity == hp (pml_e_parity == ep).
  for (int par = 0; par < 2; ++par) {
    for (int ep = 0; ep < 2; ++ep) {
      PmlP pm{};
      for (int a = 0; a < 3; ++a)
        fill_pml_axis(h, a, h->pml[a].ns > 0 && ((mask >> a) & 1), par != h->pml_parity, ep != h->pml_e_parity, true, pm.ax[a]);
      if (!h->pml_blk[mask][par][ep] && dev_alloc(h, &h->pml_blk[mask][par][ep], 1, false)) return -1;
      if (hipMemcpy(h->pml_blk[mask][par][ep], &pm, sizeof(PmlP), hipMemcpyHostToDevice) != hipSuccess)
        return fail(h, "upload of the CPML parameter block failed");
    }
  }
  h->pml_blk_ok[mask] = true;
  return 0;
}
// the blocks of shell2_step_kernel: all axes with layers, both sides read the current sets and write the other ones; the second
// E-side set is allocated here (and the in-place blocks, which name it for the other parity, are rebuilt on their next use)
int ensure_pml_blocks2(FdtdSolver* h) {
  if (h->pml_blk2_ok) return 0;
  bool fresh = false;
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.ns == 0) continue;
    for (int q = 0; q < 2; ++q) {
      if (!P.psi_h2[q] && dev_alloc(h, &P.psi_h2[q], P.psi_count + P.psi_plane)) return -1;
      if (!P.psi_e2[q]) { if (dev_alloc(h, &P.psi_e2[q], P.psi_count)) return -1; fresh = true; }
    }
  }
  if (fresh) for (bool& ok : h->pml_blk_ok) ok = false;
  for (int par = 0; par < 2; ++par) {
    for (int ep = 0; ep < 2; ++ep) {
      PmlP pm{};
      for (int a = 0; a < 3; ++a)
        fill_pml_axis(h, a, h->pml[a].ns > 0, par != h->pml_parity, ep != h->pml_e_parity, false, pm.ax[a]);
      if (!h->pml_blk2[par][ep] && dev_alloc(h, &h->pml_blk2[par][ep], 1, false)) return -1;
      if (hipMemcpy(h->pml_blk2[par][ep], &pm, sizeof(PmlP), hipMemcpyHostToDevice) != hipSuccess)
        return fail(h, "upload of the CPML parameter block failed");
    }
  }
  h->pml_blk2_ok = true;
  return 0;
}

// The blocks of the z holes of a shell2 pair (planes that take two single steps beside the boxes: a mode plane, the injection plane
// of a plane wave).  The boxes around a hole re-read the OLD psi of its planes (chunk prologues, the plane above a chunk) while the
// hole's first step runs, so that step writes TEMPORARY sets on both sides and the second one reads them and writes the sets the
// boxes write — after the pair every cell's psi sits in the same (new) sets.
int ensure_pml_blocks_hole(FdtdSolver* h) {
  if (h->pml_blk_hole_ok) return 0;
  if (ensure_pml_blocks2(h)) return -1;
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.ns == 0) continue;
    for (int q = 0; q < 2; ++q) {
      if (!P.psi_ht[q] && dev_alloc(h, &P.psi_ht[q], P.psi_count + P.psi_plane)) return -1;
      if (!P.psi_et[q] && dev_alloc(h, &P.psi_et[q], P.psi_count)) return -1;
    }
  }
  for (int step = 0; step < 2; ++step)
    for (int par = 0; par < 2; ++par)
      for (int ep = 0; ep < 2; ++ep) {
        PmlP pm{};
        for (int a = 0; a < 3; ++a) {
          const PmlAxisDev& P = h->pml[a];
          PmlAxisP& A = pm.ax[a];
          fill_pml_axis(h, a, P.ns > 0, par != h->pml_parity, ep != h->pml_e_parity, false, A);     // (coefficients, membership; pointers below)
          float* const h_cur[2] = {const_cast<float*>(A.ph0), const_cast<float*>(A.ph1)};
          float* const h_oth[2] = {A.ph0n, A.ph1n};
          float* const e_cur[2] = {A.pe0, A.pe1};
          float* const e_oth[2] = {A.pe0n, A.pe1n};
          if (step == 0) {
            A.ph0 = h_cur[0]; A.ph1 = h_cur[1]; A.ph0n = P.psi_ht[0]; A.ph1n = P.psi_ht[1];
            A.pe0 = e_cur[0]; A.pe1 = e_cur[1]; A.pe0n = P.psi_et[0]; A.pe1n = P.psi_et[1];
          } else {
            A.ph0 = P.psi_ht[0]; A.ph1 = P.psi_ht[1]; A.ph0n = h_oth[0]; A.ph1n = h_oth[1];
            A.pe0 = P.psi_et[0]; A.pe1 = P.psi_et[1]; A.pe0n = e_oth[0]; A.pe1n = e_oth[1];
          }
          (void)N;
        }
        if (!h->pml_blk_hole[step][par][ep] && dev_alloc(h, &h->pml_blk_hole[step][par][ep], 1, false)) return -1;
        if (hipMemcpy(h->pml_blk_hole[step][par][ep], &pm, sizeof(PmlP), hipMemcpyHostToDevice) != hipSuccess)
          return fail(h, "upload of the CPML parameter block failed");
      }
  h->pml_blk_hole_ok = true;
  return 0;
}

// axes whose CPML recursions can run inside the sweep
int pml_in_sweep_mask(const FdtdSolver* h) {
  int mask = 0;
  for (int a = 0; a < 3; ++a) if (h->pml[a].ns > 0) mask |= 1 << a;
  return mask;
}

// after a sweep that carried CPML recursions: the write set becomes the read set
void swap_psi_h(FdtdSolver* h, int pml_inside) {
  if (!pml_inside) return;
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.ns == 0 || !((pml_inside >> a) & 1)) continue;
    std::swap(P.psi_h[0], P.psi_h2[0]);
    std::swap(P.psi_h[1], P.psi_h2[1]);
  }
  h->pml_parity ^= 1;
}

// after a shell2 pair: the E side too
void swap_psi_e(FdtdSolver* h) {
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.ns == 0 || !P.psi_e2[0]) continue;
    std::swap(P.psi_e[0], P.psi_e2[0]);
    std::swap(P.psi_e[1], P.psi_e2[1]);
  }
  h->pml_e_parity ^= 1;
}

// Tile rows: all of them (ty_n < 0) or  [0, ty_a) + [ty_a + ty_gap, ty_a + ty_gap + (ty_n - ty_a)).
// pml_inside: axes whose CPML recursions this launch carries (its tiles must not touch members of other
// in-sweep axes): 0 -> plain instantiation, 1 -> the x-only one, anything else -> the all-axes one.
// A shell step (ShellSets) names its own read / write sets and psi parity, and the rows [ex_j0, ex_j1) its launch leaves alone.
struct ShellSets { FieldP src, dst; int parity; int ex_j0, ex_j1; const PmlP* pm = nullptr; };     // pm: a parameter block of the caller's (z holes of a shell2 pair)
int launch_fused_range(FdtdSolver* h, int kbeg, int kend, hipStream_t st, int pml_inside = 0, int k2beg = 0,
                       int k2end = 0, int ty_n = -1, int ty_a = 0, int ty_gap = 0, bool edge = false,
                       const ShellSets* sh = nullptr) {
  if (kend <= kbeg) {                 // first plane range empty: the second one takes its place
    if (k2end <= k2beg) return 0;
    kbeg = k2beg; kend = k2end; k2beg = k2end = 0;
  }
  if (ty_n == 0) return 0;
  const GridP& g = h->g;
  if (ensure_second_set(h)) return -1;
  if (pml_inside && ensure_pml_blocks(h, pml_inside)) return -1;
  const int R = h->rows_f;
  // z-chunk: 16 planes per workgroup when CPML runs inside the sweep, 8 when not — measured inside engines: 8 is
  // 0 ... 1.5 % faster on the 512^3 plain / materials sweeps and 2 % slower on the CPML-carrying step (profiles/
  // r03f), 2.5 ... 4.5 % faster on the 64- and 128-plane slabs a rank of an 8- / 4-GPU run holds (r03n).  A shape set
  // through fdtd_set_option or found by the tile-shape probe is kept.
  int zc = (h->user_geometry || h->tuned || pml_inside) ? h->zchunk_f : std::min(h->zchunk_f, kPlainZChunk);
  if (ty_gap == 0 && !edge) h->last_zc = zc;            // (the interior launch of a step, not its edge launches)
  // Edge launches of a CPML step (the few tiles that meet a y / z slab, all-axes instantiation): on a thin z-slab they are a
  // handful of workgroups marching the whole slab one plane at a time — 80 workgroups x 16 planes took 83 us alone on the
  // machine, in front of the 179 us interior launch of the same stream (profiles/r3e: 64-plane slab, CPML on x and y).
  // Shorter chunks turn them into about one wave of workgroups (the prologue plane per chunk is cheap on 6 % of the tiles).
  // On ONE GPU the edge launches run on the second stream beside the interior launch and are off the critical path: there
  // the long chunks stay (512^3 V2 inside one engine, profiles/r3f: 16 planes 1.367 ms, 8 planes 1.377, 4 planes 1.394).
  if (edge && sh && h->edge_zchunk < 0) zc = std::min(zc, 8);       // shell steps (beside the bulk sweep of a shell pair): profiles/r4d
  else if (edge && (h->edge_zchunk > 0 || (h->edge_zchunk < 0 && h->comm != nullptr))) {
    const int nby_e = ty_n < 0 ? (g.ny + R - 1) / R : ty_n;
    const long long wg_planes = (long long)((g.nx + 255) / 256) * nby_e * ((kend - kbeg) + std::max(0, k2end - k2beg));
    const int want = h->edge_zchunk > 0 ? h->edge_zchunk : (int)std::max(2LL, std::min((long long)zc, wg_planes / 1024));
    zc = std::min(zc, want);
  }
  dim3 block(64, R + 1, 1);
  const int nby_all = (g.ny + R - 1) / R;
  if (ty_n < 0) { ty_n = nby_all; ty_a = nby_all; ty_gap = 0; }
  const int nbx = (g.nx + 255) / 256, nby = ty_n, nbz1 = (kend - kbeg + zc - 1) / zc;
  const int nbz = nbz1 + (k2end > k2beg ? (k2end - k2beg + zc - 1) / zc : 0);
  const int total = nbx * nby * nbz;
  // Tile order (kernel: launch argument xcd_remap).  Default: runs of 8 consecutive tiles (y-neighbours) per XCD, the
  // runs round-robin over the XCDs.  Measured INSIDE engines held side by side (the only comparison free of the
  // placement effect; profiles/r03d_probe_grouped_tile_order.jsonl, r03e_*): faster than the plain order in every
  // engine (plain sweep -1.4 ... -5 %, materials -1.2 ... -3.8 %, CPML-carrying three-launch step -1.3 ... -2.3 %)
  // and than the contiguous one-eighth-per-XCD split of round 1 (-1.8 ... -9 %), which moves fewer bytes (7.1 vs
  // 8.6 GB per 512^3 sweep in the plain order) but has the eight XCDs stream eight distant regions of every array.
  // Runs of 6 ... 32 tiles are within 0.5 % of each other.
  const int remap = h->xcd_remap < 0 ? kTileRun : h->xcd_remap;
  dim3 grid(remap ? ((total + 7) / 8) * 8 : total, 1, 1);
  const size_t shmem = ((size_t)2 * 2 * (R + 1) * 64 + (pml_inside ? 6 * 64 : 0)) * sizeof(float4) + (size_t)h->lds_pad;   // both CPML instantiations stage the x coefficients
  const int pmc = h->cfg.bc[4] == FDTD_BC_PMC;
  MatP m = mat_params(h);
  StepP s = step_params(h);
  // register budget follows the workgroup size: __launch_bounds__ of 256 / 512 / 1024 threads
  const int threads = 64 * (R + 1);
  int lb = h->fused_lb ? h->fused_lb : (threads <= 256 ? 256 : (threads <= 512 ? 512 : 1024));
  if (lb < threads) lb = threads <= 512 ? 512 : 1024;
  time_begin(h, sh ? 3 : 2, st);
  const PmlP* pm = pml_inside ? ((sh && sh->pm) ? sh->pm : h->pml_blk[pml_inside][sh ? sh->parity : h->pml_parity][h->pml_e_parity]) : nullptr;
  const FieldP fa = sh ? sh->src : h->f, fb = sh ? sh->dst : h->f2;
  const int ex_j0 = sh ? sh->ex_j0 : 0, ex_j1 = sh ? sh->ex_j1 : 0;
#define FDTD_LAUNCH_FUSED_H(MATV, LBV, PMLV, HV)                                                       \
  hipLaunchKernelGGL((fused_step_kernel<MATV, LBV, PMLV, HV>), grid, block, shmem, st, g, fa, fb, s, m, kbeg, \
                     kend, zc, pmc, nbx, nby, nbz, remap, pm, nbz1, k2beg, k2end, ty_a, ty_gap, ex_j0, ex_j1)
  // Non-temporal field stores (FDTD_OPT_MEM_HINTS) in the instantiations WITHOUT CPML only: measured inside one engine
  // (profiles/r02z_probe_same_engine_cache_hints.jsonl) they take 2.1 % off the plain sweep and 3.1 % off the one with
  // materials, and ADD 13 % to the CPML-carrying ones (2-3 waves per SIMD: the slower store completion is not hidden).
  // With them goes the order of the stores: H ahead of the row exchange instead of behind the E update (-0.5 % there,
  // +12 % in the CPML-carrying instantiations; raising the wave priority while a plane's loads go out: +0.6 %;
  // profiles/r03k_probe_early_h_stores_setprio.jsonl).
  // The CPML-carrying instantiations store their H-side psi behind the E update instead of in the H phase in front of
  // the row exchange: -2.7 ... -3.3 % of the whole CPML step (profiles/r04j_probe_late_psi_h_stores_all.jsonl).
  // Non-temporal field stores on top of that: +13 % again (r04k).
  // Without CPML (256-thread instantiations) the E values of a plane are stored one H phase later, behind the loads of
  // the next plane (12 more live registers): -0.3 ... -0.9 % (r04l); the same in the CPML instantiations: +10 % (r04m).
#define FDTD_LAUNCH_FUSED(MATV, LBV, PMLV)                                                            \
  do {                                                                                                \
    if (PMLV == 0 && LBV == 256 && h->mem_hints) FDTD_LAUNCH_FUSED_H(MATV, 256, 0, 265);              \
    else if (PMLV == 0 && h->mem_hints) FDTD_LAUNCH_FUSED_H(MATV, LBV, 0, 9);                         \
    else if (PMLV != 0 && h->mem_hints) FDTD_LAUNCH_FUSED_H(MATV, LBV, PMLV, 128);                    \
    else FDTD_LAUNCH_FUSED_H(MATV, LBV, PMLV, 0);                                                     \
  } while (0)
  if (pml_inside == 1) {        // x recursions only: every tile of a grid with x layers
    if (h->mat4) { if (lb == 256) FDTD_LAUNCH_FUSED(true, 256, 1); else FDTD_LAUNCH_FUSED(true, 512, 1); }
    else { if (lb == 256) FDTD_LAUNCH_FUSED(false, 256, 1); else FDTD_LAUNCH_FUSED(false, 512, 1); }
  } else if (pml_inside) {      // all axes (axes outside `pml_inside` have no members in the block)
    if (h->mat4) { if (lb == 256) FDTD_LAUNCH_FUSED(true, 256, 7); else FDTD_LAUNCH_FUSED(true, 512, 7); }
    else { if (lb == 256) FDTD_LAUNCH_FUSED(false, 256, 7); else FDTD_LAUNCH_FUSED(false, 512, 7); }
  } else if (h->mat4) {
    if (lb == 256) FDTD_LAUNCH_FUSED(true, 256, 0); else if (lb == 512) FDTD_LAUNCH_FUSED(true, 512, 0); else FDTD_LAUNCH_FUSED(true, 1024, 0);
  } else {
    if (lb == 256) FDTD_LAUNCH_FUSED(false, 256, 0); else if (lb == 512) FDTD_LAUNCH_FUSED(false, 512, 0); else FDTD_LAUNCH_FUSED(false, 1024, 0);
  }
#undef FDTD_LAUNCH_FUSED
  time_end(h, st);
  return 0;
}

void swap_sets(FdtdSolver* h) {
  std::swap(h->f, h->f2);
  for (int c = 0; c < 6; ++c) std::swap(h->fbase[c], h->fbase2[c]);
}

int launch_fused(FdtdSolver* h, hipStream_t st, int pml_inside) {
  if (launch_fused_range(h, 0, h->g.nz, st, pml_inside)) return -1;
  swap_sets(h);
  swap_psi_h(h, pml_inside);
  return 0;
}

// ---- two time steps per sweep (fdtd_kernels2.hpp) --------------------------------------------------------------------
// What fused2_step_kernel covers: the curl stencil with non-dispersive media (uniform, or packed medium words + (Ca, Cb)
// table: dielectrics, conductors, PEC bodies) inside six PEC walls on one GPU, driven by point sources (electric and magnetic), recorded by
// small time monitors; the min faces may be PMC walls; absorber layers (applied in registers).  Anything else (CPML, ADE,
// TFSF, periodic / Bloch faces, PMC on max faces, z-slabs) takes single steps.
// Tile shape of the two-step sweep: waves per workgroup W (W - 3 rows of a tile are written) and planes per chunk zc (a
// chunk runs zc + 2 plane iterations).  Asked for through FDTD_OPT_TWOSTEP, or (default) the cheapest of W = 8 / 16 x
// zc = 8 ... 64 under a two-parameter model fitted to profiles/r3r_two_step_small.jsonl, r3r_two_step_shapes512.jsonl and
// r3s_two_step_auto_shapes.jsonl:
//   time ~ rounds of workgroups x (zc + 2) x t_W,   16 waves: one workgroup per CU, t = 8.2 us per plane iteration;
//                                                    8 waves: two per CU, t = 6.0 us
// (rounds counted whole while there are fewer than four: the tail of a short launch is real).  It picks the measured best
// or a shape within 3 % of it at 128^3 ... 512^3: 8 waves on grids up to 256^3 (416 workgroups of 5 rows fill the 512 slots
// at once: 195 Gcells/s against 140 with 16 waves), 16 waves from 320^3 on.  Grids below 2^20 cells keep single steps.
// `box` (a clipped launch: the bulk of a shell pair or of a z-slab rank's pair): rows and planes of the box, and chunks of
// EQUAL length — on the 60 bulk planes of a 64-plane slab rank the fixed lengths leave one round of 412 8-wave workgroups
// marching 32 iterations (237 us) where 240 16-wave ones march 22 (profiles/r4e).
bool fused2_shape(const FdtdSolver* h, int* W, int* zc, const ClipP* box = nullptr) {
  const GridP& g = h->g;
  const int nbx = (g.nx + 255) / 256;
  if (h->twostep_w > 0 && h->twostep_zc > 0) { *W = h->twostep_w; *zc = std::max(2, std::min(h->twostep_zc, g.nz)); return true; }
  if (h->twostep_w <= 0 && (long long)g.nx * g.ny * g.nz < (1LL << 20)) return false;
  double best = 0.0;
  bool found = false;
  if (box) {
    const int nyb = box->j1 - box->j0, nzb = box->k1 - box->k0;
    for (int w : {16, 8}) {
      if (h->twostep_w > 0 && w != h->twostep_w) {
        if (w == 16) w = h->twostep_w; else continue;
      }
      const int R = w - 3, nby = (nyb + R - 1) / R;
      // Round 6 (profiles/r6/r6w_slab_shapes.jsonl: 512 x 512 slabs of 64 / 128 / 256 planes, forced shapes against the model of
      // rounds 3 - 5, which took 8 waves everywhere): per plane iteration a 16-wave workgroup takes 6.6 us + 2.1 us x (how full the
      // launch's rounds of 256 are) — 8.6 when two rounds are nearly full — and 1.4 x that when the launch is ONE round (every
      // plane's latency exposed: 16 x 20 on the 60 planes of a 64-plane slab 278 us per pair); an 8-wave one 7.2 us (two per CU), not
      // the 6.0 fitted to 256^3 grids; a chunk costs its two extra iterations and ~1.5 more (prologue, ramp).  64 planes: 8 x 30
      // 253 us per pair, 16 x 10 234; 128 planes: 8 x 18 461, 16 x 21 400; 256 planes: 8 x 23 815, 16 x 42 752 — the shapes this picks.
      // (a rank that carries CPML runs the shell's boxes and the cut planes' single steps beside the bulk: a single round is not left
      //  alone on the machine — 64 planes with layers on x / y: 16 x 20 0.180 ms per step, 16 x 10 0.197, 8 x 30 0.256; 128 planes: 16 x 21
      //  0.338, 16 x 42 0.343, r6x_slab_shapes_pml2)
      const double one_round = any_pml(h) ? 1.1 : 1.4;
      const double slots = w <= 8 ? 512.0 : 256.0;
      for (int nch = 1; nch <= std::max(1, nzb / 4); ++nch) {
        const int c = (nzb + nch - 1) / nch;
        if (c > 64) continue;
        const double wg = (double)nbx * nby * ((nzb + c - 1) / c);
        double rounds = wg / slots;
        if (rounds < 4.0) rounds = std::ceil(rounds);
        const double fill = wg / (std::ceil(wg / slots) * slots);
        const double t8 = 7.2 + (h->mat4 ? 1.6 : 0.0), t16 = (6.6 + 2.1 * fill) * (rounds <= 1.0 ? one_round : 1.0) + (h->mat4 ? 1.2 : 0.0);
        const double t = w <= 8 ? t8 : (w >= 16 ? t16 : t8 + (t16 - t8) * (w - 8) / 8.0);
        const double cost = rounds * (c + 3.5) * t;
        if (!found || cost < best * 0.999) { best = cost; *W = w; *zc = c; found = true; }
      }
    }
    *zc = std::max(2, std::min(*zc, nzb));
    return found;
  }
  for (int w : {16, 8}) {
    if (h->twostep_w > 0 && w != h->twostep_w) {
      if (w == 16) w = h->twostep_w; else continue;        // a requested W: only the chunk length is chosen
    }
    const int R = w - 3, nby = (g.ny + R - 1) / R;
    // (with materials the instantiation sits at 128 VGPRs with 4 spills and looks coefficients up: 512^3 V1 0.800 ms per step
    //  with 16 waves, 0.882 with 8 against 1.136 for single sweeps, profiles/r3zx)
    // (absorber layers damped in registers, 512^3 with 40 layers: 0.959 / 1.014 ms per step with 16 / 8 waves, 1.104 / 1.185 with
    //  materials too, profiles/r3zs)
    // (round 6: 6.0 us per plane iteration of an 8-wave workgroup holds while the launch is ONE round — 256^3: 416 workgroups on 512
    //  slots — launches of several rounds measure 7.2 - 7.7 (slab ranks, r6w) and 320^3 / 384^3 prefer 16 waves: 8 x 16 127 against
    //  16 x 64 136 Gcells/s, 8 x 24 154 against 16 x 48 157, profiles/r6/r6sh_v0_shapes.jsonl)
    const double t16 = 8.2 + (h->mat4 ? 1.2 : 0.0) + (h->has_damp ? 3.0 : 0.0) + (h->mat4 && h->has_damp ? 0.5 : 0.0);
    const double slots = w <= 8 ? 512.0 : 256.0;
    for (int c : {64, 48, 32, 24, 16, 12, 8}) {
      if (c > std::max(8, g.nz)) continue;
      const double wg = (double)nbx * nby * ((g.nz + c - 1) / c);
      double rounds = wg / slots;
      if (rounds < 4.0) rounds = std::ceil(rounds);
      const double t8 = (rounds <= 1.0 ? 6.0 : 6.8) + (h->mat4 ? 1.6 : 0.0) + (h->has_damp ? 2.7 : 0.0);
      const double t = w <= 8 ? t8 : (w >= 16 ? t16 : t8 + (t16 - t8) * (w - 8) / 8.0);
      const double cost = rounds * (c + 2) * t;
      // (16 waves are tried first; 8 waves must be 8 % cheaper under the model to replace them: inside its error the measured
      //  times are equal or favour 16 waves — 512^3 0.703 / 0.709 ms, 1024^3 16 x 64 best, profiles/r3w, r3zw)
      // (on grids below 2^26 cells the model's margin is real: 256^3 8 x 32 183-195 Gcells/s against 168-175 with 16 waves)
      const bool big = (long long)g.nx * g.ny * g.nz >= (1LL << 26);
      if (!found || cost < best * (big && w <= 8 && *W > 8 ? 0.92 : 0.999)) { best = cost; *W = w; *zc = c; found = true; }
    }
  }
  *zc = std::max(2, std::min(*zc, g.nz));
  return found;
}

// Why a run takes no step pairs (FDTD_F2_OFF_*, include/fdtd_hip.h); 0 = nothing in the problem keeps the two-step sweep
// from it.  CPML is not a reason here: the caller then asks shell_eligible.
int fused2_why_not(const FdtdSolver* h, bool slab_rank = false, bool shell = false) {
  if (h->twostep_w == 0) return FDTD_F2_OFF_DISABLED;
  { int W, zc; if (!fused2_shape(h, &W, &zc)) return FDTD_F2_OFF_TOO_SMALL; }
  if (h->comm && !slab_rank) return FDTD_F2_OFF_COMM;
  // (a shell pair: the planes that hold dispersive cells are a z hole of the bulk; round 6: the pair advances them itself once
  //  their memory terms are paged, disp_setup)
  if (!h->ade.empty() && !shell && h->disp.state != 1) return FDTD_F2_OFF_ADE;
  if (!h->aniso.empty()) return FDTD_F2_OFF_ADE;               // fully anisotropic bodies: their coupling follows every single step
  // (sources are judged step by step, fused2_sources_why_not: a TFSF box or a mode plane keeps single steps only while it injects)
  // PEC walls; the min faces may be PMC (the symmetry planes of a half / quarter / eighth domain)
  // (a z-slab rank: a neighbour face is no wall — the sweep stays two planes clear of it, fdtd_run)
  // (a shell pair: a periodic y / z face has the two rows / planes next to it in the shell, periodic x wraps inside the sweep)
  for (int f = 0; f < 6; ++f)
    if (h->cfg.bc[f] != FDTD_BC_PEC && !((f & 1) == 0 && h->cfg.bc[f] == FDTD_BC_PMC) &&
        !(slab_rank && f >= 4 && h->cfg.bc[f] == FDTD_BC_NEIGHBOR) &&
        !(shell && h->cfg.bc[f] == FDTD_BC_PERIODIC && h->cfg.bc[f ^ 1] == FDTD_BC_PERIODIC)) return FDTD_F2_OFF_BOUNDARY;
  if ((h->g.pec_z0 != 0) != (h->cfg.bc[4] == FDTD_BC_PEC) || h->g.nx % 4 || h->g.nz < 2) return FDTD_F2_OFF_BOUNDARY;
  for (int a = 0; a < 3; ++a) if (h->mirror_wall[a] >= 0) return FDTD_F2_OFF_BOUNDARY;
  return 0;
}
bool fused2_eligible(const FdtdSolver* h) { return !any_pml(h) && fused2_why_not(h) == 0; }
// Can the pair (n, n + 1) be taken as far as the sources go?  0 = yes, else the reason.  The sweep applies the node table of ALL
// point-source lists or none: they must be all alive (then at most kMaxInj nodes — a dipole, a few; not a mode plane or a current
// sheet —, no H_y / H_z node left of a tile seam, no magnetic node together with absorber layers, and the table of all steps
// for magnetic nodes) or all spent (then nothing is injected and their number does not matter: a mode source or a TFSF box
// keeps single steps for the length of its pulse, the rest of the run goes out in pairs).  `*alive`: the lists inject at step n.
int fused2_sources_why_not(const FdtdSolver* h, long long n, bool* alive = nullptr) {
  bool any_alive = false, any_spent = false;
  for (const PointSrc& s : h->psrc) {
    if (!s.n_e && !s.n_h) continue;
    if (n < s.n_steps) any_alive = true; else any_spent = true;
  }
  if (alive) *alive = any_alive;
  for (const Tfsf& t : h->tfsf) if (n < t.n_steps) return FDTD_F2_OFF_TFSF;
  if (any_alive && any_spent) return FDTD_F2_OFF_SOURCES;
  if (!any_alive) return 0;
  if (h->src_nodes > kMaxInj) return FDTD_F2_OFF_SOURCES;
  if (h->src_h_nodes > 0 && !h->src_tab) return FDTD_F2_OFF_SOURCES;      // H-side nodes take their terms of step n+1 from the table only
  if (h->has_damp && h->src_h_nodes > 0) return FDTD_F2_OFF_H_SOURCE_ABSORBER;  // absorber layers are applied inside the sweep, H-side sources of step n in front of it
  if (h->src_h_on_seam) return FDTD_F2_OFF_SEAM_SOURCE;                    // the seam kernel rebuilds that value without the source term
  return 0;
}

// A time monitor whose record of a middle step the sweep can copy out on the way: every component of every cell of its box,
// a few hundred values at most (each costs every wave of its plane a scalar table entry).
bool fused2_capturable(const FdtdSolver* h, const Monitor& m) {
  if (m.kind != FDTD_MON_TIME || m.cells <= 0 || (long long)m.comps.size() * m.cells > kMaxCap) return false;
  const BoxP& b = m.box;
  return b.lo0 >= 0 && b.lo1 >= 0 && b.lo2 >= 0 && b.lo0 + b.nx <= h->g.nx && b.lo1 + b.ny <= h->g.ny && b.lo2 + b.nz <= h->g.nz;
}

// The monitors that record at step n or at step n + 1: the pair (n, n + 1) can go out as one sweep if they are all small
// time monitors (at most kPairMons).  The sweep copies their samples of the middle step (E^{n+1}, H^{n+1/2}) out; ONE launch
// behind it (pair_record_kernel) writes everything they record of the pair — E^n and H^{n-1/2} are still in the set the
// sweep read, H^{n+3/2} is in the set it wrote — so such a pair costs no record launch in front of the sweep.
// `box` (shell pairs): the bulk the sweep covers — a monitor that records at either step must lie inside it.
bool fused2_plan(const FdtdSolver* h, long long n, F2Plan* plan, const int* box_lo = nullptr, const int* box_hi = nullptr, bool dft_anywhere = false) {
  plan->mons.clear(); plan->dfts.clear(); plan->dft_when.clear();
  long long total = 0, dump = 0;
  for (size_t q = 0; q < h->mons.size(); ++q) {
    const Monitor& m = h->mons[q];
    bool at_n = false, at_m = false;
    for (size_t r = m.next; r < m.steps.size() && m.steps[r] <= n + 1; ++r) { at_n = at_n || m.steps[r] == n; at_m = at_m || m.steps[r] == n + 1; }
    if (!at_n && !at_m) continue;
    const BoxP& b = m.box;
    const bool inside = b.lo0 >= 0 && b.lo1 >= 0 && b.lo2 >= 0 && b.lo0 + b.nx <= h->g.nx && b.lo1 + b.ny <= h->g.ny && b.lo2 + b.nz <= h->g.nz;
    // (shell2 pairs: the shell's boxes copy the middle step out too — a DFT monitor may reach into the layers, as flux planes and mode
    //  monitors normally do; time monitors' samples come from the bulk sweep's node table only)
    if (box_lo && !(dft_anywhere && m.kind == FDTD_MON_DFT) &&
        (b.lo0 < box_lo[0] || b.lo1 < box_lo[1] || b.lo2 < box_lo[2] || b.lo0 + b.nx > box_hi[0] ||
         b.lo1 + b.ny > box_hi[1] || b.lo2 + b.nz > box_hi[2])) return false;
    if (m.kind == FDTD_MON_DFT) {
      // A DFT record at the FIRST step: E^n is taken in front of the sweep as always; its H terms need H^{n+1/2}.  A record at
      // the MIDDLE step: its E terms need E^{n+1}; its H terms, H^{n+3/2}, are in the write set afterwards.  The sweep copies
      // what is needed of the middle step out over the box (any size).
      if (!inside || m.cells <= 0) return false;
      for (int c : m.comps) if ((c >= 3 && at_n) || (c < 3 && at_m)) dump += m.cells;
      plan->dfts.push_back((int)q);
      plan->dft_when.push_back((at_n ? 1 : 0) | (at_m ? 2 : 0));
      continue;
    }
    if (!fused2_capturable(h, m) || m.comps.size() > 6) return false;
    total += (long long)m.comps.size() * m.cells;
    plan->mons.push_back((int)q);
  }
  return total <= kMaxCap && (int)plan->mons.size() <= kPairMons && (int)plan->dfts.size() <= kMaxDumps && dump <= (1LL << 26);
}

// the source terms of every step, formed once with the operations of point_source_kernel: [step][node], nodes in the order of
// the node tables (E nodes then H nodes of each list).  Rebuilt when lists were added.  0 = fine (src_tab may stay nullptr when
// the table would be too large: then the terms of step n are formed per pair and those of step n+1 go behind the launch)
int fused2_sources(FdtdSolver* h) {
  const GridP& g = h->g;
  if (h->inj_sources != h->psrc.size()) {
    h->f2_tables.clear();
    h->inj_sources = h->psrc.size();
    // the source terms of every step, formed once with the operations of point_source_kernel
    h->src_tab = nullptr;
    h->src_on_seam = false;
    h->src_h_on_seam = false;
    h->src_e_in_damp = false;
    long long nodes = 0, steps = 0;
    h->src_h_nodes = 0;
    for (const PointSrc& s : h->psrc) {
      nodes += s.n_e + s.n_h;
      h->src_h_nodes += s.n_h;
      if (s.n_e || s.n_h) steps = std::max(steps, s.n_steps);
      for (long long t = 0; t < s.n_e; ++t) {
        const int i = (int)(s.host_cell_e[(size_t)t] % g.nx);
        h->src_on_seam = h->src_on_seam || (i % 256 == 255 && i + 1 < g.nx) || (i % 256 == 0 && i > 0);
        if (h->has_damp) {
          const long long cell = s.host_cell_e[(size_t)t];
          const int c3[3] = {i, (int)((cell % g.sxy) / g.nx), (int)(cell / g.sxy)};
          for (int a = 0; a < 3; ++a) h->src_e_in_damp = h->src_e_in_damp || c3[a] < h->damp_lo[a] || c3[a] >= h->damp_hi[a];
        }
      }
      for (long long t = 0; t < s.n_h; ++t) {
        const int i = (int)(s.host_cell_h[(size_t)t] % g.nx), c = s.host_comp_h[(size_t)t];
        h->src_h_on_seam = h->src_h_on_seam || (c != 3 && i % 256 == 255 && i + 1 < g.nx);
      }
    }
    h->src_nodes = nodes;
    if (nodes > 0 && nodes * steps <= (1LL << 24)) {
      if (dev_alloc(h, &h->src_tab, (size_t)(nodes * steps))) return -1;
      long long off = 0;
      for (const PointSrc& s : h->psrc) {          // node order: E nodes then H nodes of each list (as the node table below)
        if (s.n_e) launch_inject_table(h->stream, h->src_tab, nodes, off, s.wre_e, s.wim_e, s.wave_e, s.n_steps, (int)s.n_e);
        off += s.n_e;
        if (s.n_h) launch_inject_table(h->stream, h->src_tab, nodes, off, s.wre_h, s.wim_h, s.wave_h, s.n_steps, (int)s.n_h);
        off += s.n_h;
      }
      h->src_tab_steps = steps; h->src_tab_nodes = nodes;
    }
  }
  return 0;
}

const F2Table* fused2_table(FdtdSolver* h, const F2Plan& plan, bool with_sources) {
  const GridP& g = h->g;
  if (fused2_sources(h)) return nullptr;
  for (const F2Table& t : h->f2_tables)
    if (t.with_sources == with_sources && t.mons == plan.mons && t.dfts == plan.dfts && t.dft_when == plan.dft_when) return &t;
  // the nodes of all E-side lists, sorted by plane; nodes of one plane keep their list order (a node two lists share
  // receives their terms in the order the source kernels would add them); the monitor samples of a plane follow them
  std::vector<std::array<int, 5>> ent;        // k, i, j, code, index
  int off = 0;
  for (const PointSrc& s : h->psrc) {
    if (!with_sources) break;          // (spent lists: nothing is injected, and a mode plane's 10^5 nodes would be walked per plane for nothing)
    for (long long t = 0; t < s.n_e; ++t, ++off) {
      const long long cell = s.host_cell_e[(size_t)t];
      ent.push_back({(int)(cell / g.sxy), (int)(cell % g.nx), (int)((cell % g.sxy) / g.nx), s.host_comp_e[(size_t)t] % 3, off});
    }
    for (long long t = 0; t < s.n_h; ++t, ++off) {      // H-side nodes: codes 3 - 5, terms of step n+1 only (those of step n go out
      const long long cell = s.host_cell_h[(size_t)t];  // in front of the sweep, launch_sources)
      ent.push_back({(int)(cell / g.sxy), (int)(cell % g.nx), (int)((cell % g.sxy) / g.nx), 3 + s.host_comp_h[(size_t)t] % 3, off});
    }
  }
  F2Table tb;
  tb.with_sources = with_sources;
  tb.mons = plan.mons;
  int coff = 0;
  for (size_t q = 0; q < plan.mons.size(); ++q) {
    Monitor& m = h->mons[(size_t)plan.mons[q]];
    tb.cap_off.push_back(coff);
    const BoxP& b = m.box;
    long long idx = 0;
    for (size_t ic = 0; ic < m.comps.size(); ++ic)
      for (long long t = 0; t < m.cells; ++t, ++idx) {
        const int lx = (int)(t % b.nx), ly = (int)((t / b.nx) % b.ny), lz = (int)(t / ((long long)b.nx * b.ny));
        ent.push_back({b.lo2 + lz, b.lo0 + lx, b.lo1 + ly, 8 + m.comps[ic], coff + (int)idx});
      }
    coff += (int)idx;
  }
  std::stable_sort(ent.begin(), ent.end(), [](const std::array<int, 5>& x, const std::array<int, 5>& y) {
    return x[0] != y[0] ? x[0] < y[0] : (x[3] >= 8) < (y[3] >= 8);
  });
  std::vector<int> start((size_t)g.nz + 2, 0);
  std::vector<int4> e4(std::max<size_t>(1, ent.size()));
  for (size_t q = 0; q < ent.size(); ++q) {
    start[(size_t)ent[q][0] + 1]++;
    e4[q].x = ent[q][1]; e4[q].y = ent[q][2]; e4[q].z = ent[q][3]; e4[q].w = ent[q][4];
  }
  for (int k = 0; k <= g.nz; ++k) start[(size_t)k + 1] += start[(size_t)k];
  if (dev_upload(h, &tb.start, (const int*)start.data(), start.size()) ||
      dev_upload(h, &tb.ent, (const int4*)e4.data(), e4.size())) return nullptr;
  // boxes of the DFT monitors that record at the first step, indexed by plane
  tb.dfts = plan.dfts; tb.dft_when = plan.dft_when;
  if (!plan.dfts.empty()) {
    std::vector<DumpBox> boxes;
    std::vector<std::vector<int>> per_plane((size_t)g.nz + 1);
    long long off = 0;
    for (size_t q = 0; q < plan.dfts.size(); ++q) {
      const Monitor& m = h->mons[(size_t)plan.dfts[q]];
      DumpBox bx{m.box.lo0, m.box.lo1, m.box.lo2, m.box.nx, m.box.ny, m.box.nz, {-1, -1, -1, -1, -1, -1}};
      const int when = plan.dft_when[q];
      for (int c : m.comps) if ((c >= 3 && (when & 1)) || (c < 3 && (when & 2))) { bx.off[c] = (int)off; off += m.cells; }
      tb.dft_off.push_back({bx.off[0], bx.off[1], bx.off[2], bx.off[3], bx.off[4], bx.off[5]});
      bool any = false;
      for (int c = 0; c < 6; ++c) any = any || bx.off[c] >= 0;
      if (!any) continue;                                                     // nothing of the middle step is needed
      for (int k = m.box.lo2; k < m.box.lo2 + m.box.nz; ++k) per_plane[(size_t)k].push_back((int)boxes.size());
      boxes.push_back(bx);
    }
    tb.dump_floats = off;
    if (!boxes.empty()) {
      std::vector<int> ds((size_t)g.nz + 2, 0), dl;
      for (int k = 0; k <= g.nz; ++k) { ds[(size_t)k] = (int)dl.size(); if (k < g.nz) for (int b : per_plane[(size_t)k]) dl.push_back(b); }
      ds[(size_t)g.nz + 1] = (int)dl.size();
      if (dl.empty()) dl.push_back(0);
      if (dev_upload(h, &tb.dstart, (const int*)ds.data(), ds.size()) || dev_upload(h, &tb.dlist, (const int*)dl.data(), dl.size()) ||
          dev_upload(h, &tb.dboxes, (const DumpBox*)boxes.data(), boxes.size())) return nullptr;
    }
  }
  h->f2_tables.push_back(tb);
  return &h->f2_tables.back();
}

// Tile classes of a launch shape of the two-step sweep (round 5): cls[tile] = 1 where any row segment the workgroup of that tile
// computes E for (its rows j0-2 .. j0+R, planes k0-1 .. k1) differs from the background word, 0 where the tile is background only —
// there the materials launch runs the plain sweep (fused2_step_kernel; the uniform coefficients ARE the table's entry 1: the same
// bits).
// `disp` (a launch that advances dispersive cells): class 2 where a row segment the workgroup visits holds one, 1 for other tiles with bodies.
// `src` (a launch that adds paged source terms): + 4 where a row segment the workgroup visits holds a source node — the other tiles run
// the instantiation without those lines (a launch over a uniform medium has this bit only).
const FdtdSolver::TileClasses* tile_classes(FdtdSolver* h, int W, int zc, const ClipP& box, int nbx, int nby, int nbz, bool disp = false, bool src = false) {
  const bool mats = h->mat4 && !h->roww_host.empty();
  if ((!mats && !src) || h->tile_split == 0) return nullptr;
  for (const auto& t : h->tile_cls)
    if (t.W == W && t.zc == zc && t.nbx == nbx && t.nby == nby && t.nbz == nbz && t.box.j0 == box.j0 && t.box.j1 == box.j1 &&
        t.box.k0 == box.k0 && t.box.k1 == box.k1 && t.disp == disp && t.src == src) return &t;
  const GridP& g = h->g;
  const int R = W - 3, nbx_all = (g.nx + 255) / 256;
  std::vector<unsigned char> cls((size_t)nbx * nby * nbz, 0);
  long long n_bg = 0;
  for (int tz = 0; tz < nbz; ++tz)
    for (int tx = 0; tx < nbx; ++tx)
      for (int ty = 0; ty < nby; ++ty) {
        const int k0 = box.k0 + tz * zc, k1 = std::min(k0 + zc, box.k1);
        const int j0 = box.j0 + ty * R;
        unsigned char c = 0;
        for (int k = std::max(k0 - 1, 0); mats && k <= std::min(k1, g.nz - 1) && !c; ++k)
          for (int j = std::max(j0 - 2, 0); j <= std::min(j0 + R, g.ny - 1); ++j)
            if (h->roww_host[((size_t)k * g.ny + j) * nbx_all + tx] != kBgWord) { c = 1; break; }
        if (disp && c)
          for (int k = std::max(k0 - 1, 0); k <= std::min(k1, g.nz - 1) && c < 2; ++k)
            for (int j = std::max(j0 - 2, 0); j <= std::min(j0 + R, g.ny - 1); ++j)
              if (h->disp.dseg_host[((size_t)k * g.ny + j) * nbx_all + tx] >= 0) { c = 2; break; }
        n_bg += !c;
        if (src) {
          bool found = false;
          for (int k = std::max(k0 - 1, 0); k <= std::min(k1, g.nz - 1) && !found; ++k)
            for (int j = std::max(j0 - 2, 0); j <= std::min(j0 + R, g.ny - 1); ++j)
              if (h->spg.sseg_host[((size_t)k * g.ny + j) * nbx_all + tx] >= 0) { found = true; break; }
          if (found) c |= 4;
        }
        cls[((size_t)tz * nbx + tx) * nby + ty] = c;
      }
  FdtdSolver::TileClasses e{W, zc, box, nbx, nby, nbz, nullptr, n_bg, (long long)cls.size(), disp, src};
  if (dev_upload(h, &e.dev, (const unsigned char*)cls.data(), cls.size())) return nullptr;
  h->tile_cls.push_back(e);
  return &h->tile_cls.back();
}

// steps n and n + 1 in one sweep: set a (E^n, H^{n-1/2}) -> set b (E^{n+2}, H^{n+3/2}).  The E-side sources of step n are
// applied inside the kernel; so are those of step n + 1 (*sources2_done) when their terms come from the table of all steps
// and no source node lies next to a seam — else the caller applies them afterwards.  The middle step is copied out for the
// monitors of `tb` (fused2_plan); pair_record (called by the caller behind the launch) writes their records.
// With `clip` the launch covers that box only (the bulk of a shell pair): nothing outside it is written, the E-side sources
// of step n + 1 are left to the caller, and the sets are NOT swapped (the shell launches beside it still name them).
// `use_disp`: the sweep subtracts the memory terms of the dispersive cells from E^{n+1} and leaves it for launch_ade2, which the
// caller issues behind the launch (and behind the sources / damping of step n + 1 it applies itself).  `e2_clip`: a clipped launch
// may apply the E-side sources of step n + 1 itself (the caller's ONE launch covers every source node).
// `sr`: paged source terms of this pair (the caller has filled them, spg_fill; the node table then lists no sources).
int launch_fused2(FdtdSolver* h, long long n, hipStream_t st, const F2Table* tb, bool* sources2_done, bool* damp2_done = nullptr,
                  const ClipP* clip = nullptr, bool use_disp = false, bool e2_clip = false, const SrcP& sr = SrcP{}) {
  const bool inject = tb->with_sources;      // (false: the lists are spent, or — shell pairs with z holes — their planes take single steps)
  const GridP& g = h->g;
  if (ensure_second_set(h)) return -1;
  int W = 16, zc = 32;
  const ClipP box = clip ? *clip : ClipP{0, g.nx, 0, g.ny, 0, g.nz};
  fused2_shape(h, &W, &zc, (clip && h->comm) ? clip : nullptr);
  // (beside the shell launches of a shell pair shorter chunks do better — workgroups retire, and hand their CU to a shell
  //  workgroup, twice as often: 512^3 V2 inside one engine 16 x 16 1.206 / 1.219 ms per step, 16 x 24 1.225, 16 x 32 1.249 / 1.231,
  //  profiles/r4d)
  // (... up to 512^3; larger grids have tile rounds to spare and prefer the long chunks of the plain sweep: V2 at 768^3 117.6 /
  //  119.6 / 123.1 Gcells/s with 16 / 32 / 64 planes, at 1024^3 127.2 / 131.5 / 134.7, profiles/r4/r4u)
  if (clip && !h->comm && h->twostep_zc <= 0 && (long long)g.nx * g.ny * g.nz < (1LL << 28)) zc = std::min(zc, 16);
  zc = std::max(2, std::min(zc, box.k1 - box.k0));
  h->twostep_w_used = W; h->twostep_zc_used = zc;
  const int R = W - 3;
  const int nbx = (g.nx + 255) / 256, nby = (box.j1 - box.j0 + R - 1) / R;
  const int nbz = (box.k1 - box.k0 + zc - 1) / zc;
  const int n_seams = nbx - 1 + ((clip && h->cfg.bc[0] == FDTD_BC_PERIODIC) ? 1 : 0);        // (periodic x: the wrap is a seam too)
  if (!h->seam_buf && (nbx > 1 || h->cfg.bc[0] == FDTD_BC_PERIODIC) &&
      dev_alloc(h, &h->seam_buf, (size_t)nbx * kSeamArrays * (size_t)(g.nz + 2) * (size_t)g.ny)) return -1;
  if (!h->inj_val && dev_alloc(h, &h->inj_val, (size_t)kMaxInj)) return -1;
  if (!h->cap_val && dev_alloc(h, &h->cap_val, (size_t)kMaxCap)) return -1;
  InjP inj{};
  *sources2_done = false;
  {
    bool alive = false, alive2 = true;
    for (const PointSrc& s : h->psrc)
      if (inject && (s.n_e || s.n_h)) { alive = alive || n < s.n_steps; alive2 = alive2 && n + 1 < s.n_steps; }
    // (all lists alive or all spent: fused2_sources_uniform.  Spent lists add nothing — not + 0 — so their table is only
    //  walked when there are none or when they are alive: pairs with monitor samples AND spent sources are not taken, fdtd_run)
    if (alive && h->src_tab && n + 1 < h->src_tab_steps) {
      inj.val = h->src_tab + n * h->src_tab_nodes;
      if (alive2) {
        inj.val2 = h->src_tab + (n + 1) * h->src_tab_nodes;
        // (round 6: a node next to a seam no longer sends them behind the launch — seam_kernel adds the terms of the values it repairs)
        if (!clip || e2_clip) { inj.e2_in_sweep = 1; *sources2_done = true; }
      }
    } else if (alive) {
      int off = 0;
      for (const PointSrc& s : h->psrc) {
        if (s.n_e) launch_inject_values(st, h->inj_val + off, s.wre_e, s.wim_e, s.wave_e, n, (int)s.n_e);
        off += (int)(s.n_e + s.n_h);
      }
      inj.val = h->inj_val;
    }
    inj.n = (alive || !tb->mons.empty()) ? 1 : 0;
    inj.start = tb->start; inj.ent = tb->ent; inj.cap = h->cap_val;
    if (tb->dstart) {
      if (tb->dump_floats > h->dump_cap) {
        if (dev_alloc(h, &h->dump_buf, (size_t)tb->dump_floats, false)) return -1;
        h->dump_cap = tb->dump_floats;
      }
      inj.dstart = tb->dstart; inj.dlist = tb->dlist; inj.dboxes = tb->dboxes; inj.dump = h->dump_buf;
      inj.n = 1;
    }
  }
  // absorber layers: H^{n-1/2}, E^{n+1}, H^{n+1/2} are damped in registers; E^{n+2} too unless E-side sources of step n+1 still
  // have to be applied behind the launch (the damping follows the sources: then the caller damps, launch_damp)
  DampT dmp{};
  if (h->has_damp) {
    for (int a = 0; a < 3; ++a) { dmp.fb[a] = h->damp_fb[a]; dmp.fc[a] = h->damp_fc[a]; }
    bool post_sources = false;
    for (const PointSrc& s : h->psrc) post_sources = post_sources || (s.n_e && n + 1 < s.n_steps);
    // (... unless every E-side source node lies outside the layers: there the factor is exactly 1 — damp (E + s) == damp (E) + s bit
    //  for bit — and the sweep damps E^{n+2} itself.  The bench's dipole sits on the seam column 256 of a 512-cell row, which sent
    //  six damping launches over 40 % of the grid behind every pair: 20 % of the `va` step)
    dmp.e2 = (!post_sources || inj.e2_in_sweep || sr.sseg || !h->src_e_in_damp) ? 1 : 0;
    if (damp2_done) *damp2_done = dmp.e2 != 0;
  }
  const int total = nbx * nby * nbz;
  // tile order: runs of 32 tiles per XCD (the single sweep: 8) — inside engines at 512^3: runs of 8 0.7165, 16 0.7143, 32 0.7090,
  // 40 0.730, 64 0.726, plain order 0.7225 ms per step (profiles/r3zm)
  const int remap = h->xcd_remap < 0 ? 32 : h->xcd_remap;
  StepP sp = step_params(h);
  time_begin(h, 2, st);
  const MatP mp = mat_params(h);
  DispP dp{nullptr, nullptr, nullptr};
  if (use_disp) dp = DispP{h->disp.dseg, h->disp.cs, h->disp.e1};
  int opt = (h->mem_hints ? 1 : 0) | (h->mat4 ? 2 : 0) | ((tb->mons.empty() && (!tb->with_sources || h->src_h_nodes == 0) && !tb->dstart) ? 0 : 4) |
            (clip ? 16 : (h->has_damp ? 8 : 0)) | (use_disp ? 32 | 1 : 0);
  if (h->whatif > 0 && h->whatif <= 15 && h->whatif != 9 && opt == 1 && (W == 16 || h->whatif == 13 || h->whatif == 14)) opt |= h->whatif << 8;        // (measuring aid: the vacuum sweep with part of its work skipped)
  SrcP sr_used = sr;
  if (h->whatif == 9 && !sr.sseg && !(opt & 8) && (!(opt & 32) || (opt & 16))) {
    // measuring aid (scripts/probe_bodies.py): the instantiation that adds paged source terms over a map WITHOUT any source segment —
    // every tile dispatches to the bodies the plain launch runs: what the larger kernel costs by itself (results unchanged)
    SrcPaged& S = h->spg;
    const size_t nseg = (size_t)g.nz * g.ny * ((g.nx + 255) / 256);
    if (!S.sseg && S.state != 1) {
      S.sseg_host.assign(nseg, -1);
      SrcT tab{};
      if (dev_upload(h, &S.sseg, (const int*)S.sseg_host.data(), nseg) || dev_upload(h, &S.tab, &tab, 1)) return -1;
    }
    if (S.sseg && S.state != 1) { sr_used.sseg = S.sseg; sr_used.t = S.tab; opt |= 64 | 4 | 1; }
  }
  if (sr.sseg) { opt |= 64 | 4 | 1; *sources2_done = true; h->spg.pairs++; }
  const int blocks = remap ? ((total + 7) / 8) * 8 : total;
  // background-only tiles take the plain sweep inside the materials launch (fdtd_kernels2.hpp, tile classes)
  const FdtdSolver::TileClasses* tc = tile_classes(h, W, zc, box, nbx, nby, nbz, use_disp, sr_used.sseg != nullptr);
  const bool split = tc && (use_disp || sr_used.sseg || (tc->n_bg > 0 && (h->tile_split == 1 || 8 * tc->n_bg >= tc->n_all)));
  launch_fused2_step(st, W, opt, blocks, g, h->f, h->f2, sp, mp, zc, nbx, nby, nbz, remap, inj, h->seam_buf, dmp, box,
                     TileClassP{split ? tc->dev : nullptr}, dp, sr_used);
  time_end(h, st);
  if (n_seams > 0) {
    time_begin(h, 4, st);
    launch_seams(st, g, h->f2, sp, mp, h->seam_buf, n_seams, dmp, box, inj, sr);
    time_end(h, st);
  }
  if (use_disp) h->disp.pairs++;
  if (!clip) swap_sets(h);
  return 0;
}

// ---- shell pairs: two steps per sweep on grids walled by CPML ------------------------------------------------------------
// An error made by ignoring a per-step feature travels one cell per half step, so after a step pair it is confined to the
// feature's cells plus a collar.  The CPML recursions live in slabs on the faces: the two-step sweep (plain formulas) advances
// the BULK  O = [lo + 2, hi0 - 1) per CPML axis (x rounded inwards to multiples of 4) — every intermediate value it forms for O
// lies in cells where the plain update is the exact one — and the SHELL (the rest: slabs + collar, 16 % of a 512^3 grid with 12
// layers) takes two single steps of the production kernels beside it, on the second stream:
//     step one   set A -> set T (third set)   over the shell grown by one cell into the bulk (what step two differentiates)
//                E-side sources of step n, H-side ones of step n + 1 on T
//     step two   set T -> set B               over the shell
// The shell is cut into z slabs (all rows: all-axes instantiation), y slabs between them (tile rows that meet a slab, the
// rows of the bulk excluded: x + y instantiation) and x strips between those (strip_step_kernel).  Bulk and shell read A and
// write disjoint cells of B: no order between them inside a pair.  psi is touched by the shell launches only (H-side
// ping-pong: parity p in step one, p ^ 1 in step two, back to p).  Same kernels / formulas as single steps: the same bits.
struct ShellGeom { int o0[3], o1[3]; };
bool shell_geometry(const FdtdSolver* h, ShellGeom* G) {
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  for (int a = 0; a < 3; ++a) {
    const PmlAxisDev& P = h->pml[a];
    const int lo = P.ns > 0 ? P.lo : 0, hi0 = P.ns > 0 ? P.hi0 : N[a];
    int o0 = lo > 0 ? lo + 2 : 0, o1 = hi0 < N[a] ? hi0 - 1 : N[a];
    // a periodic y / z axis: the bulk stays two cells clear of the wrap (what it reads then lies inside the grid; the shell's
    // single steps wrap as they always did); periodic x wraps inside the sweep
    if (a > 0 && h->cfg.bc[2 * a] == FDTD_BC_PERIODIC) { o0 = 2; o1 = N[a] - 2; }
    if (a == 0) { o0 = (o0 + 3) / 4 * 4; o1 = o1 / 4 * 4; }
    if (o1 - o0 < (a == 0 ? 16 : 8)) return false;        // (also: the rounded x ranges met, lo == hi0 == n)
    G->o0[a] = o0; G->o1[a] = o1;
  }
  return true;
}

void release_buf(FdtdSolver* h, void* p);
int ensure_third_set(FdtdSolver* h) {
  const GridP& g = h->g;
  if (h->f3.ex) return 0;
  const size_t fcount = (size_t)g.sxy * (g.nz + 2);
  if (!h->fbase3[0] && alloc_field_set(h, h->fbase3, fcount, 2)) {
    for (int c = 0; c < 6; ++c) { if (h->fbase3[c]) release_buf(h, h->fbase3[c]); h->fbase3[c] = nullptr; }
    return -1;
  }
  h->f3.ex = h->fbase3[0] + g.sxy; h->f3.ey = h->fbase3[1] + g.sxy; h->f3.ez = h->fbase3[2] + g.sxy;
  h->f3.hx = h->fbase3[3] + g.sxy; h->f3.hy = h->fbase3[4] + g.sxy; h->f3.hz = h->fbase3[5] + g.sxy;
  return 0;
}

// one x strip of a shell step: columns [ci0, ci1), rows [j0, j1), planes [k0, k1)
void launch_strip(FdtdSolver* h, const FieldP& src, const FieldP& dst, const PmlP* pm, int ci0, int ci1, int j0, int j1,
                  int k0, int k1, hipStream_t st) {
  if (ci1 <= ci0 || j1 <= j0 || k1 <= k0) return;
  StripP sp;
  sp.q = std::min(kStripMaxQ, (ci1 - ci0) / 4);             // lanes per row: the strip's width (16 columns: 4; 20: 5), 64 columns per x tile at most
  sp.xorg = ci0; sp.ci0 = ci0; sp.ci1 = ci1; sp.j0 = j0; sp.j1 = j1; sp.kbeg = k0; sp.kend = k1;
  sp.zchunk = h->strip_zc;
  const int slots = (64 / sp.q) * kStripWaves;
  sp.nbx = (ci1 - ci0 + 4 * sp.q - 1) / (4 * sp.q);
  sp.nby = (j1 - j0 + slots - 2) / (slots - 1);
  sp.nbz = (k1 - k0 + sp.zchunk - 1) / sp.zchunk;
  const int pmc = h->cfg.bc[4] == FDTD_BC_PMC;
  const dim3 grid((unsigned)(sp.nbx * sp.nby * sp.nbz)), block(64, kStripWaves);
  time_begin(h, 3, st);
  if (h->mat4 && h->strip_occ == 4) hipLaunchKernelGGL((strip_step_kernel<true, 4>), grid, block, 0, st, h->g, src, dst, step_params(h), mat_params(h), pm, sp, pmc);
  else if (h->mat4) hipLaunchKernelGGL((strip_step_kernel<true, 3>), grid, block, 0, st, h->g, src, dst, step_params(h), mat_params(h), pm, sp, pmc);
  else if (h->strip_occ == 4) hipLaunchKernelGGL((strip_step_kernel<false, 4>), grid, block, 0, st, h->g, src, dst, step_params(h), mat_params(h), pm, sp, pmc);
  else hipLaunchKernelGGL((strip_step_kernel<false, 3>), grid, block, 0, st, h->g, src, dst, step_params(h), mat_params(h), pm, sp, pmc);
  time_end(h, st);
}

// The bulk's planes: the box's z range minus the z HOLES — plane ranges that take single steps with the shell although they lie
// inside the box: the planes that hold dispersive cells (their ADE state advances every step), and, while the lists inject, the
// planes of sources the sweep cannot apply itself (a mode plane, a current sheet, the injection plane of a plane wave: more than
// kMaxInj nodes, or TFSF corrections) — each grown by two planes (the reach of what the sweep gets wrong by ignoring them).
struct ZPlan {
  int n = 0;                 // bulk intervals [a[i], b[i]), ascending
  int a[4] = {}, b[4] = {};
  bool ok = false;           // a plan exists and the cost model likes it
};
bool zplan_build(const FdtdSolver* h, const ShellGeom& G, bool source_holes, ZPlan* P) {
  std::vector<std::pair<int, int>> holes;
  for (const AdeGroup& a : h->ade) if (a.n > 0) holes.push_back({a.k0 - 2, a.k1 + 2});
  if (source_holes) {
    for (const PointSrc& s : h->psrc) {
      if (s.n_e) holes.push_back({s.ke0 - 2, s.ke1 + 2});
      if (s.n_h) holes.push_back({s.kh0 - 2, s.kh1 + 2});
    }
    for (const Tfsf& t : h->tfsf) {
      if (t.e.n_targets) holes.push_back({t.e.k0 - 2, t.e.k1 + 2});
      if (t.h.n_targets) holes.push_back({t.h.k0 - 2, t.h.k1 + 2});
    }
  }
  std::sort(holes.begin(), holes.end());
  P->n = 0;
  int lo = G.o0[2];
  auto close = [&](int hi) {           // the bulk interval [lo, hi): kept when it is worth a launch
    if (hi - lo >= 8) {
      if (P->n == 4) return false;
      P->a[P->n] = lo; P->b[P->n] = hi; P->n++;
    }
    return true;
  };
  for (const auto& hz : holes) {
    if (hz.second <= lo) continue;
    if (hz.first >= G.o1[2]) break;
    if (hz.first > lo && !close(std::min(hz.first, G.o1[2]))) return false;
    lo = std::max(lo, hz.second);
  }
  if (lo < G.o1[2] && !close(G.o1[2])) return false;
  return P->n > 0;
}
// the part of the bulk interval i that belongs to someone else in a shell step that reaches `grow` planes into the bulk
inline void zplan_in(const ZPlan& P, int i, int grow, int nz, int* k0, int* k1) {
  *k0 = P.a[i] + (P.a[i] > 0 ? grow : 0);
  *k1 = P.b[i] - (P.b[i] < nz ? grow : 0);
}

// One step of the shell: set `src` -> set `dst`, H-side psi parity `parity`.  in[a] = [in0[a], in1[a]) (a = x, y): the part of
// axis a that belongs to someone else in this step (the bulk O in step two, O shrunk by one cell on its CPML sides in step one);
// along z the bulk's intervals of `P`, shrunk by `grow` planes each.
int launch_shell_step(FdtdSolver* h, const FieldP& src, const FieldP& dst, int parity, const int in0[3], const int in1[3],
                      int pml_in, hipStream_t st, const ZPlan& P, int grow) {
  const GridP& g = h->g;
  const int R = h->rows_f, nby_all = (g.ny + R - 1) / R;
  ShellSets sh{src, dst, parity, 0, 0};
  // z slabs and z holes: the planes outside the bulk's intervals, all rows (two plane ranges per launch)
  {
    int r0[6], r1[6], nr = 0, lo = 0;
    for (int i = 0; i < P.n; ++i) {
      int k0, k1;
      zplan_in(P, i, grow, g.nz, &k0, &k1);
      if (k0 > lo) { r0[nr] = lo; r1[nr] = k0; ++nr; }
      lo = k1;
    }
    if (lo < g.nz) { r0[nr] = lo; r1[nr] = g.nz; ++nr; }
    for (int q = 0; q < nr; q += 2)
      if (launch_fused_range(h, r0[q], r1[q], st, pml_in, q + 1 < nr ? r0[q + 1] : 0, q + 1 < nr ? r1[q + 1] : 0, -1, 0, 0, true, &sh)) return -1;
  }
  for (int i = 0; i < P.n; ++i) {
    int k0, k1;
    zplan_in(P, i, grow, g.nz, &k0, &k1);
    if (k1 <= k0) continue;
    // y slabs: the planes of the interval, the tile rows that hold a row outside in[1] (the rows inside it excluded)
    if (in0[1] > 0 || in1[1] < g.ny) {
      const int ty_a = in0[1] > 0 ? std::min(nby_all, (in0[1] + R - 1) / R) : 0;
      const int ty_c = in1[1] < g.ny ? std::max(ty_a, in1[1] / R) : nby_all;
      sh.ex_j0 = in0[1]; sh.ex_j1 = in1[1];
      if (launch_fused_range(h, k0, k1, st, pml_in & 3, 0, 0, ty_a + (nby_all - ty_c), ty_a, ty_c - ty_a, true, &sh)) return -1;
      sh.ex_j0 = sh.ex_j1 = 0;
    }
    // x strips: planes of the interval, rows of in[1], the columns outside in[0]
    if (in0[0] > 0 || in1[0] < g.nx) {
      const PmlP* pm = h->pml_blk[pml_in][parity][h->pml_e_parity];
      launch_strip(h, src, dst, pm, 0, in0[0], in0[1], in1[1], k0, k1, st);
      launch_strip(h, src, dst, pm, in1[0], g.nx, in0[1], in1[1], k0, k1, st);
    }
  }
  return 0;
}

bool any_periodic(const FdtdSolver* h) {
  for (int f = 0; f < 6; ++f) if (h->cfg.bc[f] == FDTD_BC_PERIODIC) return true;
  return false;
}
int shell_why_not(const FdtdSolver* h, ShellGeom* G, ZPlan* base, ZPlan* with_src) {
  const int why = fused2_why_not(h, false, true);
  if (why) return why;
  if (!any_pml(h) && !any_periodic(h) && h->ade.empty()) return FDTD_F2_OFF_PML;              // (nothing to do here: the plain pairs cover it)
  if (h->shell_on == 0) return FDTD_F2_OFF_PML;
  if (h->has_damp) return h->ade.empty() ? FDTD_F2_OFF_PML : FDTD_F2_OFF_ADE;   // absorber layers with CPML or dispersive media: single steps
  // the shell runs the CPML recursions inside its sweeps (all axes), as a one-GPU step does by default
  if (64 * (h->rows_f + 1) > 512 || (h->pml_fused >= 0 && (h->pml_fused & pml_in_sweep_mask(h)) != pml_in_sweep_mask(h))) return FDTD_F2_OFF_PML;
  if (!shell_geometry(h, G)) return FDTD_F2_OFF_PML;
  // Is it worth it?  A shell cell costs a single step's bytes and more (fields + psi, through the third set, in launches that
  // cannot fill the machine), twice per pair; a bulk cell costs half a single step's.  Measured on MI355X, 512^3 V2, each kernel
  // alone on the machine (profiles/r4/r4c_kernel_trace_shell_pair_one_stream.txt; scaled to a single step of 10.1 ps per cell):
  // the bulk's two steps 10.3 ps per cell (12 with materials), a slab cell 25 ps per step, a strip cell 42 ps per step (64-byte
  // pieces at a 2 KB stride), and the two streams overlap to about 0.9 of the sum.  V2 (shell: 16 % of the cells): predicted 0.89
  // of two single steps, measured 0.88-0.89 (r4n).  BASELINE config 3 laid out with x = 224 (strips: 13 % of the cells): predicted
  // 1.11, measured 1.14-1.16 — pairs LOSE there (r4o), as on the V2 problem at 320^3 (1.08).  shell_on = 1 forces pairs (tests).
  // Two plans: the bulk's planes with the z holes of the dispersive cells only (`base`), and with those of every source list
  // too (`with_src`: what a pair uses while lists inject that the sweep cannot apply itself).
  auto judge = [&](bool source_holes, ZPlan* P) {
    P->ok = false;
    if (!zplan_build(h, *G, source_holes, P)) return;
    if (h->shell_on != 1) {
      const double N[3] = {(double)h->g.nx, (double)h->g.ny, (double)h->g.nz};
      const double ox = G->o1[0] - G->o0[0], oy = G->o1[1] - G->o0[1];
      double oz = 0.0;
      for (int i = 0; i < P->n; ++i) oz += P->b[i] - P->a[i];
      const double all = N[0] * N[1] * N[2], bulk = ox * oy * oz;
      const double strips = (N[0] - ox) * oy * oz, slabs = all - bulk - strips;
      const double pair_ps = 0.9 * (bulk * (h->mat4 ? 12.0 : 10.3) + 2.0 * (slabs * 25.0 + strips * 42.0));
      const double single_ps = 2.0 * all * 10.1;
      if (pair_ps > 0.97 * single_ps) return;
    }
    P->ok = true;
  };
  judge(false, base);
  judge(true, with_src);
  if (!base->ok) return h->ade.empty() ? FDTD_F2_OFF_SHELL : FDTD_F2_OFF_ADE;
  return 0;
}

// ---- shell2 pairs (round 5): the shell advanced by shell2_step_kernel, two steps per sweep with psi carried --------------------
// The grid is cut into the bulk O (shell_geometry: clipped two-step sweep, plain formulas) and up to six boxes that
// shell2_step_kernel advances beside it — all of them read set A and the current psi sets and write disjoint cells of set B and of
// the other psi sets, so there is no order between any two launches of a pair, no third field set and no middle-step launches:
//   x strips   the columns outside O_x, ALL rows and planes (the corners with the y / z slabs included: all-axes recursions)
//   z slabs    the columns of O_x, all rows, the planes outside O_z
//   y slabs    the columns of O_x, the rows outside O_y, the planes of O_z
// Taken when CPML makes the shell (no absorber layers, no periodic z) and, while source lists inject, when every source node lies
// three or more cells inside O (the boxes apply no sources; the bulk sweep applies its own) — or on planes that become z holes.
// What the boxes cannot advance takes two single steps through set T beside them, psi routed through temporary sets
// (Run::shell2_pair): z holes (the planes of dispersive cells and of big source planes), the two rows on either side of a periodic
// y wrap.  A periodic x needs neither: the boxes' halo lanes hold the wrapped columns (fdtd_shell2.hpp) and there are no x strips.
struct Shell2Box { int i0, i1, j0, j1, k0, k1; bool strip; int axes; };
constexpr int kShell2MaxBoxes = 24;
// Every box is cut so that most of its cells meet the recursions of ONE axis (the middle of an x strip: x; a y slab: y; the
// middle rows of a z slab: z) — their instantiation carries 32 psi registers instead of 96 — and the edges and corners, where two
// or three slabs cross, go out as small boxes of the all-axes instantiation.
int shell2_boxes_by_axes(const FdtdSolver* h, const ShellGeom& G, Shell2Box out[kShell2MaxBoxes]);
// (z_lo, z_hi: the planes of the x strips and — where they reach beyond G's plane range — of the z slabs: the whole grid, or the
//  segment of one bulk interval when z holes cut the run of planes)
int shell2_boxes(const FdtdSolver* h, const ShellGeom& G, Shell2Box out[kShell2MaxBoxes], int z_lo = 0, int z_hi = -1) {
  if (z_hi < 0) z_hi = h->g.nz;
  if (h->shell2_on != 2 && h->shell2_on != 3) {      // the default: six boxes, all of the all-axes instantiation (one launch)
    const int nx = h->g.nx, ny = h->g.ny, nz = h->g.nz;
    int n = 0;
    (void)nz;
    // (a periodic y: the two rows next to the wrap on either side belong to nobody here — they take single steps beside the
    //  boxes, Run::shell2_pair — so the x strips and z slabs end there and there are no y slabs)
    const bool per_y = h->cfg.bc[2] == FDTD_BC_PERIODIC;
    const int r0 = per_y ? G.o0[1] : 0, r1 = per_y ? G.o1[1] : ny;
    if (G.o0[0] > 0) out[n++] = {0, G.o0[0], r0, r1, z_lo, z_hi, true, 7};
    if (G.o1[0] < nx) out[n++] = {G.o1[0], nx, r0, r1, z_lo, z_hi, true, 7};
    if (G.o0[2] > z_lo) out[n++] = {G.o0[0], G.o1[0], r0, r1, z_lo, G.o0[2], false, 7};
    if (G.o1[2] < z_hi) out[n++] = {G.o0[0], G.o1[0], r0, r1, G.o1[2], z_hi, false, 7};
    if (!per_y && G.o0[1] > 0) out[n++] = {G.o0[0], G.o1[0], 0, G.o0[1], G.o0[2], G.o1[2], false, 7};
    if (!per_y && G.o1[1] < ny) out[n++] = {G.o0[0], G.o1[0], G.o1[1], ny, G.o0[2], G.o1[2], false, 7};
    return n;
  }
  return shell2_boxes_by_axes(h, G, out);
}
int shell2_boxes_by_axes(const FdtdSolver* h, const ShellGeom& G, Shell2Box out[kShell2MaxBoxes]) {
  const int nx = h->g.nx, ny = h->g.ny, nz = h->g.nz;
  const int ox0 = G.o0[0], ox1 = G.o1[0], oy0 = G.o0[1], oy1 = G.o1[1], oz0 = G.o0[2], oz1 = G.o1[2];
  int n = 0;
  auto add = [&](int i0, int i1, int j0, int j1, int k0, int k1, bool strip, int axes) {
    if (i1 > i0 && j1 > j0 && k1 > k0) out[n++] = {i0, i1, j0, j1, k0, k1, strip, axes};
  };
  // x strips (all rows and planes): the z-slab planes and the y-slab rows of the planes between them as all-axes boxes, the rest x only
  for (int side = 0; side < 2; ++side) {
    const int i0 = side == 0 ? 0 : ox1, i1 = side == 0 ? ox0 : nx;
    if (i1 <= i0) continue;
    add(i0, i1, 0, ny, 0, oz0, true, 7);
    add(i0, i1, 0, ny, oz1, nz, true, 7);
    add(i0, i1, 0, oy0, oz0, oz1, true, 7);
    add(i0, i1, oy1, ny, oz0, oz1, true, 7);
    add(i0, i1, oy0, oy1, oz0, oz1, true, 1);
  }
  // z slabs (the bulk's columns, all rows): the y-slab rows as all-axes boxes, the rows between them z only
  for (int side = 0; side < 2; ++side) {
    const int k0 = side == 0 ? 0 : oz1, k1 = side == 0 ? oz0 : nz;
    if (k1 <= k0) continue;
    add(ox0, ox1, 0, oy0, k0, k1, false, 7);
    add(ox0, ox1, oy1, ny, k0, k1, false, 7);
    add(ox0, ox1, oy0, oy1, k0, k1, false, 4);
  }
  // y slabs (the bulk's columns and planes): y only
  add(ox0, ox1, 0, oy0, oz0, oz1, false, 2);
  add(ox0, ox1, oy1, ny, oz0, oz1, false, 2);
  return n;
}
// tile shape of one box: lanes per row q for `W` waves per workgroup — the shape that wastes the fewest lane-planes (row slots of
// the last tile row and the three halo slots, halo lanes and the last x tile's idle lanes, the two extra iterations per chunk)
void shell2_shape(const FdtdSolver* h, const Shell2Box& bx, int W, int zc_cap, Shell2P* out) {
  const GridP& g = h->g;
  const bool per_x = h->cfg.bc[0] == FDTD_BC_PERIODIC;          // (the row wraps: a halo lane on both sides, holding the wrapped columns)
  const int halo_l = (bx.i0 > 0 || per_x) ? 1 : 0, halo_r = (bx.i1 < g.nx || per_x) ? 1 : 0;
  const int lanes_w = (bx.i1 - bx.i0) / 4;
  const int L = lanes_w + halo_l + halo_r;                   // lanes a row needs: the written ones and a halo lane on every side that is no wall
  const int rows = bx.j1 - bx.j0, nzb = bx.k1 - bx.k0;
  Shell2P best{};
  double best_cost = 0.0;
  int forced_q = bx.strip ? 0 : h->shell2_qw;
  if (forced_q > 0 && (64 / std::max(3, std::min(forced_q, L))) * W < 4) forced_q = 0;      // (a shape without a row to write: by box)
  for (int q = 3; q <= kShell2MaxQ; ++q) {
    if (forced_q > 0 && q != std::max(3, std::min(forced_q, L))) continue;
    if (q > std::max(3, L)) break;
    const int S = (64 / q) * W;
    // (a box that starts on the y-min wall — a wall, not a periodic wrap — and fits one tile row without the two halo slots below: jlo = 0)
    const bool wall_lo = bx.j0 == 0 && h->cfg.bc[2] != FDTD_BC_PERIODIC && rows <= S - 1 && !getenv("FDTD_NO_JLO");      // ($FDTD_NO_JLO: a debugging aid)
    const int jlo = wall_lo ? 0 : 2;
    const int R = S - 1 - jlo;
    if (R < 1) continue;
    // tiles overlap by two lanes: tile t holds lanes t (q - 2) ... t (q - 2) + q - 1 of the row; lv = the last lane that must come out right
    const int lv = halo_r ? L - 2 : L - 1, qv = halo_r ? q - 2 : q - 1;
    const int nbx = 1 + std::max(0, (lv - qv + q - 3) / (q - 2));
    const int nby = (rows + R - 1) / R;
    int zc = bx.strip ? h->shell2_zcs : h->shell2_zcw;
    if (zc <= 0) {
      // planes per workgroup: chunks of equal length, zc_cap planes at most (each chunk pays two extra iterations and a prologue:
      // 512^3 V2 inside one engine 8 planes 1.054, 16: 1.031 / 1.054, 32: 1.005 / 1.028, 64: 0.998 / 1.026 ms per step, profiles/r5);
      // the caller lowers the cap until the launch as a whole gives the machine two rounds of workgroups
      const int nch = std::max(1, (nzb + zc_cap - 1) / zc_cap);
      zc = (nzb + nch - 1) / nch;
    }
    zc = std::max(1, std::min(zc, nzb));
    const int nbz = (nzb + zc - 1) / zc;
    const double cost = (double)nbx * nby * nbz * (zc + 2.5) * W;          // wave-iterations of the launch
    if (best.q == 0 || cost < best_cost * 0.999) {
      best_cost = cost;
      best.q = q; best.xorg = bx.i0 - 4 * halo_l;
      best.ci0 = bx.i0; best.ci1 = bx.i1; best.j0 = bx.j0; best.j1 = bx.j1; best.k0 = bx.k0; best.k1 = bx.k1;
      best.jlo = jlo;
      best.zchunk = zc; best.nbx = nbx; best.nby = nby; best.nbz = nbz;
    }
  }
  *out = best;
}
// The boxes of the shell as ONE launch of the all-axes instantiation (the default): six launches one behind the other left the
// machine half empty between them (0.92 against 0.69 ms per 512^3 V2 pair alone on the machine, profiles/r5/r5e), and one launch
// per instantiation — x / y / z only at three waves per SIMD for 94 % of the cells, all axes for the edges and corners — lost
// more to the four half-empty launches than the third wave bought (0.84 ms, profiles/r5/r5f).  shell2_on = 2 / 3: one launch per
// instantiation / per box (measuring aids).
int shell2_box_paged(FdtdSolver* h, const Shell2Box& bx);
void launch_shell2_boxes(FdtdSolver* h, const Shell2Box* bx, int n, const PmlP* pm, hipStream_t st, const F2Table* tb, bool use_disp = false,
                         const SrcP& sr = SrcP{}) {
  const DispP dp = use_disp ? DispP{h->disp.dseg, h->disp.cs, h->disp.e1} : DispP{nullptr, nullptr, nullptr};
  // (the DFT monitors of the pair's plan that reach into the shell: the boxes copy the middle step out over them; the dump buffer was sized by launch_fused2)
  Shell2Dump dmp{};
  if (tb && tb->dstart && h->dump_buf) { dmp.dstart = tb->dstart; dmp.dlist = tb->dlist; dmp.dboxes = tb->dboxes; dmp.dump = h->dump_buf; }
  const bool by_axes = h->shell2_on == 2 || h->shell2_on == 3;
  for (int axes : {1, 4, 2, 7}) {
    if (!by_axes && axes != 7) continue;
    Shell2M mb{};
    const int W = std::max(1, std::min(8, (by_axes && axes != 7) ? h->shell2_ws : h->shell2_ww));
    auto flush = [&]() {
      if (mb.n == 0) return;
      time_begin(h, 3, st);
      launch_shell2_step(st, W, h->mat4 != nullptr, axes, h->g, h->f, h->f2, step_params(h), mat_params(h), pm, mb, dmp, dp, sr);
      time_end(h, st);
      mb = Shell2M{};
    };
    // planes per chunk: 32 at most, fewer while the boxes of this launch together make less than two rounds of workgroups
    int zc_cap = 32;
    for (; zc_cap > 8; zc_cap -= (zc_cap > 16 ? 8 : 4)) {
      long long wgs = 0;
      for (int q = 0; q < n; ++q) {
        if (by_axes && bx[q].axes != axes) continue;
        Shell2P sp{};
        shell2_shape(h, bx[q], W, zc_cap, &sp);
        wgs += (long long)sp.nbx * sp.nby * sp.nbz;
      }
      if (wgs >= 2 * 256 * (W <= 4 ? 2 : 1)) break;
    }
    for (int q = 0; q < n; ++q) {
      if (by_axes && bx[q].axes != axes) continue;
      shell2_shape(h, bx[q], W, zc_cap, &mb.box[mb.n]);
      mb.box[mb.n].paged = (sr.sseg || dp.dseg) ? shell2_box_paged(h, bx[q]) : 0;
      mb.first[mb.n + 1] = mb.first[mb.n] + mb.box[mb.n].nbx * mb.box[mb.n].nby * mb.box[mb.n].nbz;
      mb.n++;
      if (mb.n == kShell2Boxes || h->shell2_on == 3) flush();
    }
    flush();
  }
}
// every node of every point-source list three or more cells inside the bulk on the axes / sides that carry a shell
bool shell2_sources_deep(const FdtdSolver* h, const ShellGeom& G) {
  const GridP& g = h->g;
  const int N[3] = {g.nx, g.ny, g.nz};
  auto deep = [&](long long cell) {
    const int c[3] = {(int)(cell % g.nx), (int)((cell % g.sxy) / g.nx), (int)(cell / g.sxy)};
    for (int a = 0; a < 3; ++a) {
      if (G.o0[a] > 0 && c[a] < G.o0[a] + 3) return false;
      if (G.o1[a] < N[a] && c[a] >= G.o1[a] - 3) return false;
    }
    return true;
  };
  for (const PointSrc& s : h->psrc) {
    for (long long t = 0; t < s.n_e; ++t) if (!deep(s.host_cell_e[(size_t)t])) return false;
    for (long long t = 0; t < s.n_h; ++t) if (!deep(s.host_cell_h[(size_t)t])) return false;
  }
  return true;
}
int shell2_why_not(const FdtdSolver* h, ShellGeom* G) {
  const int why = fused2_why_not(h, false, true);
  if (why) return why;
  if (!any_pml(h) || h->shell2_on == 0 || h->shell_on == 0) return FDTD_F2_OFF_PML;
  // periodic x (wraps through halo lanes) and y (its rows next to the wrap take single steps beside the boxes) — one launch only;
  // a periodic z would put holes on the grid's ends: the round-4 form
  if (h->cfg.bc[4] == FDTD_BC_PERIODIC) return FDTD_F2_OFF_BOUNDARY;
  if (any_periodic(h) && (h->shell2_on == 2 || h->shell2_on == 3)) return FDTD_F2_OFF_BOUNDARY;
  // (dispersive cells: their planes are z holes of the bulk, as in the round-4 form — Run::setup_pairs checks that they lie inside it)
  if (!h->ade.empty() && (h->shell2_on == 2 || h->shell2_on == 3)) return FDTD_F2_OFF_ADE;
  if (h->has_damp) return FDTD_F2_OFF_PML;
  if ((long long)h->g.sxy * 4 >= (1LL << 32)) return FDTD_F2_OFF_PML;           // (32-bit lane offsets inside a plane)
  if (!shell_geometry(h, G)) return FDTD_F2_OFF_PML;
  if (h->shell2_on < 1) {
    // Is it worth it?  (constants and their source: kShell2*Ps above)
    const double N[3] = {(double)h->g.nx, (double)h->g.ny, (double)h->g.nz};
    const double ox = G->o1[0] - G->o0[0], oy = G->o1[1] - G->o0[1], oz = G->o1[2] - G->o0[2];
    const double all = N[0] * N[1] * N[2], bulk = ox * oy * oz, strips = (N[0] - ox) * N[1] * N[2], wide = all - bulk - strips;
    const double pair_ps = kShell2Overlap * (bulk * (h->mat4 ? kShell2BulkMatPs : kShell2BulkPs) + wide * kShell2WidePs + strips * kShell2StripPs);
    if (pair_ps > 0.95 * 2.0 * all * 10.1) return FDTD_F2_OFF_SHELL;
  }
  return 0;
}

// H-side source terms of step n act on H^{n-1/2} in place in FRONT of a pair's sweep: then the small time monitors of the pair take
// E^n and their first H half-sample in front of those (record_monitors at the top of the step) and pair_record adds the rest.
// Round 6: the H-side corrections of a TFSF box count too — since such boxes inject inside pairs (paged source terms) a probe on
// one of their H nodes recorded half the term of step n too much (1e-12 of the field on the pulse's rising edge: one random device
// case in 200, scripts/fuzz_round6.py seed 11 case 47).
bool h_terms_in_front(const FdtdSolver* h) {
  if (h->src_h_nodes > 0) return true;
  for (const Tfsf& t : h->tfsf) if (t.h.n_targets) return true;
  return false;
}
// behind the sweep of the pair (n, n + 1) (the sets are swapped: h->f2 = what it read, h->f = what it wrote): everything the
// monitors of `tb` record of steps n and n + 1, in one launch
void pair_record(FdtdSolver* h, const F2Table* tb, long long n, hipStream_t st) {
  dbg_sync(h);
  // DFT monitors: a record at step n took E^n in front of the sweep — its H terms from the sweep's copy of H^{n+1/2}; a record at
  // step n+1 takes its E terms from the copy of E^{n+1} here and its H terms (H^{n+3/2}) from the write set, record_monitors
  for (size_t q = 0; q < tb->dfts.size(); ++q) {
    Monitor& m = h->mons[(size_t)tb->dfts[q]];
    for (int pass = 0; pass < 2; ++pass) {
      if (m.next >= m.steps.size() || m.steps[m.next] != n + pass) continue;
      DftDumpP r{};
      for (size_t ic = 0; ic < m.comps.size(); ++ic)
        if ((m.comps[ic] >= 3) == (pass == 0)) { r.slot[r.n] = (int)ic; r.off[r.n] = tb->dft_off[q][(size_t)m.comps[ic]]; r.n++; }
      if (r.n > 0)
        launch_dft_record_dump(st, r, h->dump_buf, reinterpret_cast<float2*>(m.data), m.cells, (long long)m.comps.size() * m.cells,
                               (const float2*)((pass == 0 ? m.phase_h : m.phase_e) + (long long)m.next * m.nf), m.nf);
      if (pass == 0) m.next++;
    }
  }
  if (tb->mons.empty()) return;
  PairRecP r{};
  r.pre_done = h_terms_in_front(h);            // (then fdtd_run has not skipped them at the top of the step)
  long long max_cells = 0;
  for (size_t q = 0; q < tb->mons.size(); ++q) {
    Monitor& m = h->mons[(size_t)tb->mons[q]];
    const long long rs = (long long)m.comps.size() * m.cells;
    r.box[q] = m.box;
    r.nc[q] = (int)m.comps.size();
    for (size_t ic = 0; ic < m.comps.size(); ++ic) r.comp[q][ic] = m.comps[ic];
    r.cap_off[q] = tb->cap_off[q];
    if (m.next < m.steps.size() && m.steps[m.next] == n) { r.out_n[q] = reinterpret_cast<float*>(m.data) + (long long)m.next * rs; m.next++; }
    if (m.next < m.steps.size() && m.steps[m.next] == n + 1) { r.out_m[q] = reinterpret_cast<float*>(m.data) + (long long)m.next * rs; m.next++; }
    max_cells = std::max(max_cells, m.cells);
  }
  r.n_mon = (int)tb->mons.size();
  launch_pair_record(st, r, max_cells, h->g, h->f2, h->f, h->cap_val);
}

// Placement of the field arrays.  Where the twelve arrays land in device memory moves the sweep by up to 15 % (DESIGN.md
// section 7; 1.08 ... 1.24 ms per 512^3 step over the 40 engines of profiles/r03d-r03t) and nothing in HIP steers it — but
// it can be sampled: before its first large run an engine allocates up to `placement_tries` further sets of the
// twelve arrays, times two PAIRS of plain sweeps (a -> b, b -> a: what a run does) on a copy of the live fields in each, and
// keeps the fastest set, copying the fields over.  Costs ten-odd sweeps and, for their duration, twice the field memory
// (skipped when that is not free).  For fused runs of at least 2^24 cells (2^22 per rank on z-slabs).  Measured (profiles/
// r03u_probe_placement_probe.jsonl, 10 engines, 3 candidates each): the set kept is 0.8 ... 8.1 % faster than the first
// allocations (mean 4 %), and the first allocations never won.
// What a run does: PAIRS of sweeps, set a -> b then b -> a.  The two directions read and write different arrays and differ
// in speed with the placement (one-directional timing said 1.094 ms per sweep where the run then took 1.146 ms per step,
// profiles/r3j), so both are timed: one warm-up pair, two timed pairs.  Advances the fields of the set it runs on.
float time_sweep_pairs(FdtdSolver* h, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  // a run the two-step sweep covers is timed on THAT kernel (its access pattern — 16 rows per workgroup, 32-plane chunks — is
  // not the single sweep's): launches without sources or monitor samples, set a -> b -> a as in the run
  const F2Table* tb = nullptr;
  if (fused2_eligible(h)) { F2Plan none; tb = fused2_table(h, none, false); }
  for (int k = 0; k < 6; ++k) {
    if (k == 2) hipEventRecord(e0, st);
    if (tb) {
      bool unused = false;
      if (launch_fused2(h, (1LL << 60), st, tb, &unused)) return -1.f;        // (a step no source list reaches: nothing is added)
    } else {
      if (launch_fused_range(h, 0, h->g.nz, st)) return -1.f;
      swap_sets(h);
    }
  }
  hipEventRecord(e1, st);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
void set_field_views(FdtdSolver* h) {
  const long long o = h->g.sxy;
  h->f.ex = h->fbase[0] + o; h->f.ey = h->fbase[1] + o; h->f.ez = h->fbase[2] + o;
  h->f.hx = h->fbase[3] + o; h->f.hy = h->fbase[4] + o; h->f.hz = h->fbase[5] + o;
  h->f2.ex = h->fbase2[0] + o; h->f2.ey = h->fbase2[1] + o; h->f2.ez = h->fbase2[2] + o;
  h->f2.hx = h->fbase2[3] + o; h->f2.hy = h->fbase2[4] + o; h->f2.hz = h->fbase2[5] + o;
}
void release_buf(FdtdSolver* h, void* p) {
  for (size_t i = 0; i < h->bufs.size(); ++i)
    if (h->bufs[i].p == p) {
      hipFree(p);
      h->stats.device_bytes -= (int64_t)h->bufs[i].bytes;
      h->bufs.erase(h->bufs.begin() + (long)i);
      return;
    }
}
int probe_placement(FdtdSolver* h, hipStream_t st) {
  h->placement_done = true;
#ifdef FDTD_PLACEMENT_PROBE
  return 0;                                      // (measurement build: the arrays may live in one pooled allocation)
#else
  if (ensure_second_set(h)) return -1;
  const size_t fcount = (size_t)h->g.sxy * (h->g.nz + 2);
  const size_t need = 12 * fcount * sizeof(float);
  const int flags = h->cfg.flags;
  h->cfg.flags &= ~FDTD_FLAG_TIME_KERNELS;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {      // no timing, no probe: keep what we have
    if (e0) hipEventDestroy(e0);
    (void)hipGetLastError();
    h->cfg.flags = flags;
    return 0;
  }
  // Every set is timed on the SAME data — a copy of the live fields — in the mode a run uses (time_sweep_pairs); the
  // live set is timed in place, once, and restored from the first candidate's copy.
  float best = -1.f;
  float* cur[12];                                 // the set that holds the fields: at most this one and one candidate exist
  for (int c = 0; c < 6; ++c) { cur[c] = h->fbase[c]; cur[6 + c] = h->fbase2[c]; }
  auto copy6 = [&](float* const* dst, float* const* src) -> int {
    for (int c = 0; c < 6; ++c)
      if (hipMemcpyAsync(dst[c], src[c], fcount * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(h, "probe_placement: copy failed");
    return 0;
  };
  int rc = 0;
  // (round 6: candidates that lose are HELD until the probe ends — freed at once, their blocks came straight back as the next
  //  candidate and the probe timed the same placement again; up to 8 tries)
  std::vector<float*> held;
  for (int t = 0; t < h->placement_tries && t < 8 && !rc; ++t) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + need / 4) break;
    float* cand[12] = {};
    bool ok = true;
    for (int c = 0; c < 12 && ok; ++c) ok = dev_alloc(h, &cand[c], fcount) == 0;
    if (!ok) {                                    // out of memory after all: give back what was taken
      for (float* q : cand) if (q) release_buf(h, q);
      (void)hipGetLastError();
      h->err.clear();
      break;
    }
    auto drop_cand = [&]() { for (int c = 0; c < 12; ++c) held.push_back(cand[c]); };
    // dev_alloc zero-fills on the NULL stream, which this engine's non-blocking stream does not wait for: without this the
    // memset of a candidate could land AFTER the copy below and the "restored" fields came back partly zero (caught by
    // tests/test_gpu_production_path.py on the first visit with this probe, profiles/r3l)
    if (hipDeviceSynchronize() != hipSuccess) { drop_cand(); rc = fail(h, "probe_placement: %s", hipGetErrorString(hipGetLastError())); break; }
    rc = copy6(cand, cur);                        // the live fields (E, H of the current set), ghost planes included
    if (!rc && best < 0.f) {                      // the incumbent, in place; its fields come back from the copy
      best = time_sweep_pairs(h, st, e0, e1);
      h->placement_ms[0] = best;
      if (best < 0.f) rc = -1;
      if (!rc) rc = copy6(cur, cand);
    }
    if (rc) { drop_cand(); break; }
    for (int c = 0; c < 6; ++c) { h->fbase[c] = cand[c]; h->fbase2[c] = cand[6 + c]; }
    set_field_views(h);
    const float ms = time_sweep_pairs(h, st, e0, e1);
    for (int c = 0; c < 6; ++c) { h->fbase[c] = cur[c]; h->fbase2[c] = cur[6 + c]; }
    set_field_views(h);
    h->placement_ms[t + 1] = ms;
    h->placement_tried = t + 1;
    if (ms < 0.f) { drop_cand(); rc = -1; break; }
    // The candidate wins only by a clear margin.  Measured on 8 engines of one process (profiles/r3k): moves for a 0.6 ... 1.8 %
    // advantage in the probe ended 0.6 ... 2 % SLOWER in the run that followed, moves for 3.4 % and 5.8 % ended 2.6 % and
    // 4.5 % faster (first allocations 1.113 / 1.145 ms per step -> 1.084 / 1.093): the probe is there to escape a bad
    // placement, not to polish a good one.
    if (ms < 0.98f * best) {
      rc = copy6(cand, cur);
      if (!rc) rc = copy6(cand + 6, cur + 6);
      if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(h, "probe_placement: %s", hipGetErrorString(hipGetLastError()));
      if (rc) { drop_cand(); break; }
      for (int c = 0; c < 12; ++c) { held.push_back(cur[c]); cur[c] = cand[c]; }
      best = ms;
      h->placement_kept = t + 1;
    } else {
      if (hipStreamSynchronize(st) != hipSuccess) rc = fail(h, "probe_placement: %s", hipGetErrorString(hipGetLastError()));
      drop_cand();
    }
  }
  for (int c = 0; c < 6; ++c) { h->fbase[c] = cur[c]; h->fbase2[c] = cur[6 + c]; }
  set_field_views(h);
  if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(h, "probe_placement: %s", hipGetErrorString(hipGetLastError()));
  for (float* q : held) release_buf(h, q);
  if (getenv("FDTD_PLACEMENT_LOG")) {
    fprintf(stderr, "[placement] kept %d of 1 + %d:", h->placement_kept, h->placement_tried);
    for (int t = 0; t <= h->placement_tried; ++t) fprintf(stderr, " %.3f", h->placement_ms[t]);
    fprintf(stderr, " ms\n");
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->cfg.flags = flags;
  return rc;
#endif
}

// Tile-shape autotuning of the fused sweep.  Which (rows, z-chunk) shape is fastest depends on how
// the tile count quantises onto 256 CUs x 3 workgroups and on the slab height (profiles/
// r01g_probe_geometry.jsonl: 107 .. 111 Gcells/s across shapes at 512^3; a 64-plane slab prefers
// taller chunks).  The sweep reads set A and writes set B only, so timing it has no side effect
// and — the arithmetic being independent of the launch geometry — no effect on the results.
int autotune_fused(FdtdSolver* h, hipStream_t st) {
  h->tuned = true;
  const int nz = h->g.nz;
  if (ensure_second_set(h)) return -1;
  const int flags = h->cfg.flags;
  h->cfg.flags &= ~FDTD_FLAG_TIME_KERNELS;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int rows_c[2] = {3, 2};
  const int zc_c[4] = {16, 24, 32, 8};
  int best_r = h->rows_f, best_z = h->zchunk_f;
  float best = 1e30f;
  for (int r : rows_c)
    for (int z : zc_c) {
      if (z > nz && z != 16) continue;
      h->rows_f = r; h->zchunk_f = z;
      if (launch_fused_range(h, 0, nz, st)) return -1;             // warm-up (instruction cache, TLB)
      hipEventRecord(e0, st);
      for (int k = 0; k < 2; ++k) if (launch_fused_range(h, 0, nz, st)) return -1;
      hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) { best = ms; best_r = r; best_z = z; }
    }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->rows_f = best_r; h->zchunk_f = best_z;
  h->cfg.flags = flags;
  return 0;
}

// two spins of 200 us, one on each stream: 1 = they ran side by side, 0 = one after the other (one hardware queue), -1 = error
int streams_run_side_by_side(FdtdSolver* h, hipStream_t a, hipStream_t b) {
#if defined(__HIPCC__)
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device) != hipSuccess || khz <= 0) khz = 100000;
  const long long ticks = (long long)khz / 5;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) return -1;
  int res = -1;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 0LL);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 0LL);
  if (hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess) {
    hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    hipEventRecord(e1, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    hipEventRecord(e2, b);
    if (hipEventSynchronize(e1) == hipSuccess && hipEventSynchronize(e2) == hipSuccess) {
      float t1 = 0.f, t2 = 0.f;
      hipEventElapsedTime(&t1, e0, e1);
      hipEventElapsedTime(&t2, e0, e2);
      res = std::max(t1, t2) < 0.32f ? 1 : 0;
    }
  }
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
  return res;
#else
  (void)h; (void)a; (void)b;
  return 1;
#endif
}
// the third stream of CPML slab-rank pairs (the shell's boxes beside the bulk and the cut planes' hole): one that shares a hardware queue
// with neither of the engine's two streams — fresh streams are tried, the rejected ones held meanwhile (the runtime hands a new stream the
// least-loaded queue); none found: box_stream stays null and the boxes keep their place in front of the bulk.  Called between runs' launches
// only at the first such pair (it synchronises the two streams).
int make_box_stream(FdtdSolver* h) {
  h->box_stream_tried = true;
  std::vector<hipStream_t> rejected;
  hipStream_t found = nullptr;
  for (int attempt = 0; attempt < 6 && !found; ++attempt) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
    const int a = streams_run_side_by_side(h, s, h->stream), b = a == 1 ? streams_run_side_by_side(h, s, h->comm_stream) : 0;
    if (a == 1 && b == 1) found = s; else rejected.push_back(s);
    if (a < 0 || b < 0) break;
  }
  for (hipStream_t r : rejected) hipStreamDestroy(r);
  (void)hipGetLastError();
  h->box_stream = found;
  h->box_stream_attempts = (int)rejected.size() + (found ? 1 : 0);
  if (found) {
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_box, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_box_in, hipEventDisableTiming));
  }
  return 0;
}
// Do the engine's two streams really run concurrently?  The HIP runtime multiplexes a process's streams onto
// $GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that land on ONE queue execute their launches one after
// the other.  For this engine that is a silent 3x cliff: the boundary chunks and the interior sweep of a z-slab step
// (0.83 instead of 0.28 ms, profiles/r04u), the edge and interior launches of a CPML step.  So it is MEASURED, once,
// before the first run that uses both streams: a 200 us spin kernel on each; if the pair takes two spins instead of one,
// a fresh second stream is tried (the runtime gives a new stream the least-loaded queue of its priority class), a few
// times over; if none overlaps, the engine uses ONE stream for both roles and says so in FdtdStats.stream_overlap.
int probe_stream_overlap(FdtdSolver* h) {
  if (h->stream_overlap != 0) return 0;
#if defined(__HIPCC__)
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device) != hipSuccess || khz <= 0) khz = 100000;
  const long long ticks = (long long)khz / 5;                 // 200 us of the constant-rate clock
  hipEvent_t e0, e1, e2;
  HIPCHK(h, hipEventCreate(&e0));
  HIPCHK(h, hipEventCreate(&e1));
  HIPCHK(h, hipEventCreate(&e2));
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
  std::vector<hipStream_t> rejected;
  int rc = 0;
  bool ok = false;
  for (int attempt = 0; attempt < 6 && !rc; ++attempt) {
    hipStream_t a = h->stream, b = h->comm_stream;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 0LL);       // (code object resident, queues awake)
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 0LL);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { rc = fail(h, "probe_stream_overlap: %s", hipGetErrorString(hipGetLastError())); break; }
    hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    hipEventRecord(e1, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    hipEventRecord(e2, b);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventSynchronize(e2) != hipSuccess) { rc = fail(h, "probe_stream_overlap: %s", hipGetErrorString(hipGetLastError())); break; }
    float t1 = 0.f, t2 = 0.f;
    hipEventElapsedTime(&t1, e0, e1);
    hipEventElapsedTime(&t2, e0, e2);
    if (std::max(t1, t2) < 0.32f) { ok = true; break; }      // one spin (0.2 ms) + launch latencies; two spins would be >= 0.4
    // serialised: another second stream
    hipStream_t fresh = nullptr;
    if (hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, attempt % 2 ? least : greatest) != hipSuccess) break;
    rejected.push_back(h->comm_stream);
    h->comm_stream = fresh;
    h->stream_retries++;
  }
  for (hipStream_t r : rejected) hipStreamDestroy(r);
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
  if (rc) return rc;
  if (ok) { h->stream_overlap = 1; return 0; }
  // fallback: one stream carries both roles (correct — every cross-stream edge becomes stream order — just not overlapped)
  hipStreamDestroy(h->comm_stream);
  h->comm_stream = h->stream;
  h->streams_shared = true;
  h->stream_overlap = -1;
  std::fprintf(stderr, "libfdtd_hip: the engine's two HIP streams do not run concurrently (shared hardware queue; "
                       "GPU_MAX_HW_QUEUES too low for this process?) - using one stream\n");
  return 0;
#else
  return 0;        // (the emulator runs every launch to completion: nothing to measure)
#endif
}

// periodic z, fused sweep: the prologue recomputes H^{n+1/2}[-1] from ghost copies of E (all three
// components) and H_x, H_y of plane nz-1; the top plane needs E_x, E_y of plane 0
// PMC on plus faces: refresh the mirror images beyond the walls (start of a step: E^n, H^{n-1/2} are complete)
// (k0, k1): the planes of this call — x / y walls are refreshed plane by plane; a z wall (the last rank's: images in its last two
//  planes, of the two below them) with the call that holds all four of its planes (fdtd_run checks that one does)
void fill_mirror(FdtdSolver* h, hipStream_t st, int k0 = 0, int k1 = -1) {
  dbg_sync(h);
  const GridP& g = h->g;
  if (k1 < 0) k1 = g.nz;
  if (k1 <= k0) return;
  for (int a = 0; a < 3; ++a) {
    const int N = h->mirror_wall[a];
    if (N < 0) continue;
    if (a == 2 && !(k0 <= N - 2 && N + 2 <= k1 && k0 >= 0 && k1 <= g.nz)) continue;
    const long long lines = (a == 0) ? (long long)g.ny * (k1 - k0) : (a == 1 ? (long long)g.nx * (k1 - k0) : g.sxy);
    hipLaunchKernelGGL(mirror_fill_kernel, dim3(nblk(lines)), dim3(256), 0, st, g, h->f, a, N, k0, k1 - k0);
  }
}

// one xy-plane device to device: a copy (the runtime's blit), or a kernel node while a graph is being captured
void copy_plane(FdtdSolver* h, float* dst, const float* src, hipStream_t st) {
  const long long pc = (long long)h->cfg.nx * h->cfg.ny;
  if (h->step_dev_mode) hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((pc + 255) / 256)), dim3(256), 0, st, dst, src, pc);
  else hipMemcpyAsync(dst, src, pc * 4, hipMemcpyDeviceToDevice, st);
}
void fill_ghost_fused(FdtdSolver* h, hipStream_t st, const FieldP* fs = nullptr) {
  dbg_sync(h);
  if (h->cfg.bc[4] != FDTD_BC_PERIODIC) return;
  const FieldP F = fs ? *fs : h->f;
  const long long pc = plane_cells(h);
  const long long top = (long long)(h->g.nz - 1) * pc;
  float* lo[5] = {F.ex, F.ey, F.ez, F.hx, F.hy};
  for (float* p : lo) copy_plane(h, p - pc, p + top, st);
  copy_plane(h, F.ex + (long long)h->g.nz * pc, F.ex, st);
  copy_plane(h, F.ey + (long long)h->g.nz * pc, F.ey, st);
}

// slabs of one axis: E-side ranges [0,n_lo) and [N-n_hi+1,N); H-side [0,n_lo) and [N-n_hi,N)
// Axis order: H side x, y, z; E side y, z, x — so that a sweep that folds only the y and z
// recursions into the fused kernel (x stays a slab kernel before / after it) sums in the same order.
void launch_pml(FdtdSolver* h, bool e_side, int kbeg, int kend, hipStream_t st, int axes = 7) {
  dbg_sync(h);
  const GridP& g = h->g;
  const int N[3] = {g.nx, g.ny, g.nz};
  for (int ai = 0; ai < 3; ++ai) {
    const int a = e_side ? (ai + 1) % 3 : ai;
    PmlAxisDev& P = h->pml[a];
    if (P.n_lo + P.n_hi == 0 || !((axes >> a) & 1)) continue;
    const int c1 = (a + 1) % 3, c2 = (a + 2) % 3;
    // the two faces of an axis touch disjoint cells: ONE launch (blockIdx.y = face) — small and
    // mid-size grids are bound by the number of dependent launches, not by the slab work
    SlabP sl[2];
    long long cells[2] = {0, 0};
    int n_sl = 0;
    for (int side = 0; side < 2; ++side) {
      int s_lo, s_n, base;
      if (side == 0) { s_lo = 0; s_n = P.n_lo; base = 0; }
      else if (e_side) { s_lo = N[a] - P.n_hi + 1; s_n = P.n_hi - 1; base = P.lo + (s_lo - P.hi0); }
      else { s_lo = N[a] - P.n_hi; s_n = P.n_hi; base = P.lo + (s_lo - P.hi0); }
      if (s_n <= 0) continue;
      SlabP q;
      q.a = a; q.s_lo = s_lo; q.s_n = s_n; q.psi_base = base;
      q.psi_ns = P.ns;
      q.kbeg = kbeg; q.kend = kend; q.kpsi0 = 0;
      if (a == 2) {          // intersect the slab with the launch z-range
        int lo = s_lo > kbeg ? s_lo : kbeg;
        int hi = (s_lo + s_n) < kend ? (s_lo + s_n) : kend;
        if (hi <= lo) continue;
        q.kbeg = lo; q.kend = hi;
      }
      const long long bx = (a == 0) ? s_n : g.nx, by = (a == 1) ? s_n : g.ny;
      const long long total = bx * by * (q.kend - q.kbeg);
      if (total <= 0) continue;
      sl[n_sl] = q; cells[n_sl] = total; ++n_sl;
    }
    if (n_sl == 0) continue;
    if (n_sl == 1) sl[1] = sl[0];
    const bool vec4 = (a != 0) && (g.nx % 4 == 0);      // y / z slabs: float4 along x
    const long long most = std::max(cells[0], cells[1]);
    const dim3 grid(nblk(vec4 ? most / 4 : most), n_sl);
    if (vec4 && e_side)
      hipLaunchKernelGGL(pml_e4_kernel, grid, dim3(256), 0, st, g, sl[0], sl[1], field_ptr(h, c1),
                         field_ptr(h, c2), (const float*)field_ptr(h, 3 + c1), (const float*)field_ptr(h, 3 + c2),
                         P.psi_e[0], P.psi_e[1], (const float4*)P.ce4, (const float*)h->idl[a], (const uint32_t*)h->mat4,
                         (const float2*)h->lut, h->cb1, (const uint32_t*)h->mat4b);
    else if (vec4)
      hipLaunchKernelGGL(pml_h4_kernel, grid, dim3(256), 0, st, g, sl[0], sl[1], field_ptr(h, 3 + c1),
                         field_ptr(h, 3 + c2), (const float*)field_ptr(h, c1), (const float*)field_ptr(h, c2),
                         P.psi_h[0], P.psi_h[1], (const float4*)P.ch4, (const float*)h->ip[a]);
    else if (e_side)
      hipLaunchKernelGGL(pml_e_kernel, grid, dim3(256), 0, st, g, sl[0], sl[1], field_ptr(h, c1),
                         field_ptr(h, c2), (const float*)field_ptr(h, 3 + c1), (const float*)field_ptr(h, 3 + c2),
                         P.psi_e[0], P.psi_e[1], (const float4*)P.ce4, (const float*)h->idl[a], (const uint32_t*)h->mat4,
                         (const float2*)h->lut, h->cb1, (const uint32_t*)h->mat4b);
    else
      hipLaunchKernelGGL(pml_h_kernel, grid, dim3(256), 0, st, g, sl[0], sl[1], field_ptr(h, 3 + c1),
                         field_ptr(h, 3 + c2), (const float*)field_ptr(h, c1), (const float*)field_ptr(h, c2),
                         P.psi_h[0], P.psi_h[1], (const float4*)P.ch4, (const float*)h->ip[a]);
  }
}

// absorber layers over the planes [kbeg, kend): E components (end of the E phase) or H components
// (start of the H phase, before the H-side corrections).  One launch per axis and face.
void launch_damp(FdtdSolver* h, bool e_side, int kbeg, int kend, hipStream_t st) {
  dbg_sync(h);
  if (!h->has_damp || kend <= kbeg) return;
  const GridP& g = h->g;
  const int N[3] = {g.nx, g.ny, g.nz};
  DampP d;
  for (int a = 0; a < 3; ++a) {
    d.fb[a] = h->damp_fb[a]; d.fc[a] = h->damp_fc[a];
    d.lo[a] = h->damp_lo[a]; d.hi[a] = h->damp_hi[a];
  }
  const int off = e_side ? 0 : 3;
  float *f0 = field_ptr(h, off), *f1 = field_ptr(h, off + 1), *f2 = field_ptr(h, off + 2);
  for (int a = 0; a < 3; ++a)
    for (int side = 0; side < 2; ++side) {
      int s_lo = side == 0 ? 0 : h->damp_hi[a];
      int s_hi = side == 0 ? h->damp_lo[a] : N[a];
      if (a == 2) { s_lo = std::max(s_lo, kbeg); s_hi = std::min(s_hi, kend); }
      if (s_hi <= s_lo) continue;
      const long long bx = (a == 0) ? (s_hi - s_lo) : g.nx, by = (a == 1) ? (s_hi - s_lo) : g.ny;
      const long long bz = (a == 2) ? (s_hi - s_lo) : (kend - kbeg);
      if (a != 0 && g.nx % 4 == 0)
        hipLaunchKernelGGL(damp4_kernel, dim3(nblk(bx / 4 * by * bz)), dim3(256), 0, st, g, d, f0, f1, f2,
                           e_side ? 0 : 1, a, s_lo, s_hi - s_lo, kbeg, kend);
      else
        hipLaunchKernelGGL(damp_kernel, dim3(nblk(bx * by * bz)), dim3(256), 0, st, g, d, f0, f1, f2,
                           e_side ? 0 : 1, a, s_lo, s_hi - s_lo, kbeg, kend);
    }
}

void launch_sources(FdtdSolver* h, bool e_side, long long n, int kbeg, int kend, hipStream_t st, bool replica = false,
                    const FieldP* fs = nullptr) {
  dbg_sync(h);
  if (kend <= kbeg) return;
  const long long zlo = (long long)kbeg * h->g.sxy, zhi = (long long)kend * h->g.sxy;
  const int off = e_side ? 0 : 3;
  float *f0 = field_ptr(h, off), *f1 = field_ptr(h, off + 1), *f2 = field_ptr(h, off + 2);
  if (fs) { f0 = e_side ? fs->ex : fs->hx; f1 = e_side ? fs->ey : fs->hy; f2 = e_side ? fs->ez : fs->hz; }
  for (Tfsf& t : h->tfsf) {
    if (n >= t.n_steps) continue;
    const TfsfList& L = e_side ? t.e : t.h;
    // E-side corrections read the incident H (h1), H-side ones the incident E (e1)
    const float* aux = e_side ? (replica ? t.h1c : t.h1) : (replica ? t.e1c : t.e1);
    if (L.n_targets && planes_meet(L.k0, L.k1, kbeg, kend))
      hipLaunchKernelGGL(tfsf_corr_kernel, dim3(nblk(L.n_targets)), dim3(256), 0, st, f0, f1, f2,
                         (const int32_t*)L.comp, (const uint32_t*)L.cell, (const int32_t*)L.start, (const float*)L.w,
                         (const int32_t*)L.aux, aux, L.n_targets, zlo, zhi);
  }
  for (PointSrc& s : h->psrc) {
    if (n >= s.n_steps) continue;
    if (e_side && s.n_e && planes_meet(s.ke0, s.ke1, kbeg, kend))
      hipLaunchKernelGGL(point_source_kernel, dim3(nblk(s.n_e)), dim3(256), 0, st, f0, f1, f2,
                         (const int32_t*)s.comp_e, (const uint32_t*)s.cell_e, (const float*)s.wre_e,
                         (const float*)s.wim_e, (const float2*)s.wave_e, h->step_dev_mode ? h->step_dev_off : n, s.n_e, zlo, zhi,
                         (const long long*)(h->step_dev_mode ? h->step_dev : nullptr));
    if (!e_side && s.n_h && planes_meet(s.kh0, s.kh1, kbeg, kend))
      hipLaunchKernelGGL(point_source_kernel, dim3(nblk(s.n_h)), dim3(256), 0, st, f0, f1, f2,
                         (const int32_t*)s.comp_h, (const uint32_t*)s.cell_h, (const float*)s.wre_h,
                         (const float*)s.wim_h, (const float2*)s.wave_h, h->step_dev_mode ? h->step_dev_off : n, s.n_h, zlo, zhi,
                         (const long long*)(h->step_dev_mode ? h->step_dev : nullptr));
  }
}

// the 1-D incident grids advance once per phase on the main stream (not z-range dependent)
void advance_tfsf_aux(FdtdSolver* h, bool e_side, long long n, hipStream_t st, bool replica = false) {
  dbg_sync(h);
  for (Tfsf& t : h->tfsf) {
    if (n >= t.n_steps) continue;
    float* e1 = replica ? t.e1c : t.e1;
    float* h1 = replica ? t.h1c : t.h1;
    if (e_side)
      hipLaunchKernelGGL(tfsf_aux_e_kernel, dim3(1), dim3(1024), 0, st, e1, (const float*)h1,
                         (const float*)t.ae, (const float*)t.be, t.n_aux, t.src_cell, (const float*)t.wave,
                         h->step_dev_mode ? h->step_dev_off : n, (const long long*)(h->step_dev_mode ? h->step_dev : nullptr));
    else
      hipLaunchKernelGGL(tfsf_aux_h_kernel, dim3(nblk(t.n_aux)), dim3(256), 0, st, h1, (const float*)e1,
                         (const float*)t.ah, (const float*)t.bh, t.n_aux);
  }
}

// fully anisotropic bodies: E^n of the neighbour nodes, saved in front of the E update ...
void aniso_save(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  for (AnisoGroup& a : h->aniso) {
    const float* b1 = field_ptr(h, (a.comp + 1) % 3);
    const float* b2 = field_ptr(h, (a.comp + 2) % 3);
    hipLaunchKernelGGL(aniso_save_kernel, dim3(nblk(8 * a.n)), dim3(256), 0, st, b1, b2, (const uint32_t*)a.nbr, a.old, 8 * a.n);
  }
}
// ... and the coupling, behind the E update and its sources: every component's correction from the unpatched values, then applied
void aniso_apply(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  for (AnisoGroup& a : h->aniso) {
    const float* b1 = field_ptr(h, (a.comp + 1) % 3);
    const float* b2 = field_ptr(h, (a.comp + 2) % 3);
    hipLaunchKernelGGL(aniso_delta_kernel, dim3(nblk(a.n)), dim3(256), 0, st, b1, b2, (const uint32_t*)a.nbr, (const float*)a.w_new,
                       (const float*)a.w_old, (const float*)a.old, a.delta, a.n);
  }
  for (AnisoGroup& a : h->aniso)
    hipLaunchKernelGGL(aniso_apply_kernel, dim3(nblk(a.n)), dim3(256), 0, st, field_ptr(h, a.comp), (const uint32_t*)a.cell,
                       (const float*)a.delta, a.n);
}

void launch_ade(FdtdSolver* h, int kbeg, int kend, hipStream_t st, const FieldP* fs = nullptr) {
  dbg_sync(h);
  if (kend <= kbeg) return;
  const long long zlo = (long long)kbeg * h->g.sxy, zhi = (long long)kend * h->g.sxy;
  for (AdeGroup& a : h->ade) {
    if (!planes_meet(a.k0, a.k1, kbeg, kend)) continue;
    // a sorted list: the launch covers the entries of the planes [kbeg, kend) only — the boundary chunks of a z-slab step
    // (2 planes each, three launches per component and step) no longer scan the whole list three times
    long long t0 = 0, t1 = a.n;
    if (!a.plane_off.empty()) { t0 = a.plane_off[(size_t)std::max(kbeg, 0)]; t1 = a.plane_off[(size_t)std::min(kend, h->g.nz)]; }
    if (t1 <= t0) continue;
    float* ef = fs ? (a.comp == 0 ? fs->ex : (a.comp == 1 ? fs->ey : fs->ez)) : field_ptr(h, a.comp);
    const bool paged = h->disp.state == 1;      // (the memory term of the next step goes to the paged array the two-step sweeps read)
    hipLaunchKernelGGL(ade_kernel, dim3(nblk(t1 - t0)), dim3(256), 0, st, ef,
                       (const uint32_t*)a.cell + t0, a.e_old + t0, a.q + t0, t1 - t0, a.n, zlo, zhi, a.p,
                       paged ? h->disp.cs : nullptr, paged ? (const uint32_t*)a.qoff + t0 : nullptr);
  }
}

// behind a two-step sweep that carried the ADE update of its first step (launch_fused2 with use_disp; set B = `fs` or the current
// set holds E^{n+2} with the sources and damping of step n + 1 applied): both steps of the pole states, E^{n+2} corrected
void launch_ade2(FdtdSolver* h, hipStream_t st, const FieldP* fs = nullptr) {
  dbg_sync(h);
  for (AdeGroup& a : h->ade) {
    float* ef = fs ? (a.comp == 0 ? fs->ex : (a.comp == 1 ? fs->ey : fs->ez)) : field_ptr(h, a.comp);
    hipLaunchKernelGGL(ade2_kernel, dim3(nblk(a.n)), dim3(256), 0, st, ef, (const uint32_t*)a.cell, a.e_old, a.q, a.n, a.p,
                       (const float*)h->disp.e1, h->disp.cs, (const uint32_t*)a.qoff);
  }
}

// One-off, before the first run that may take step pairs: the memory terms cc S(Q) of all dispersive cells go to PAGED storage
// (fdtd_fused2.hpp DispP) that the two-step sweeps address by position, and every ADE launch of the handle keeps them current from
// then on.  Needs sorted lists without repeats (one slot per entry), one GPU, packed medium words (the instantiations that carry
// the ADE lines are materials ones).  state = 1 (ready) or -1 (this problem keeps single steps / z holes as in round 5).
// -1 = a HIP error.
int disp_setup(FdtdSolver* h) {
  Disp& D = h->disp;
  if (D.state != 0) return 0;
  D.state = -1;
  const GridP& g = h->g;
  if (h->ade.empty() || h->disp_on == 0 || !h->mat4 || h->mat4b || h->comm || !h->aniso.empty() || g.nx % 4 != 0) return 0;
  for (const AdeGroup& a : h->ade) if (!a.strict) return 0;
  const int nbx = (g.nx + 255) / 256;
  const size_t nseg = (size_t)g.nz * g.ny * nbx;
  hipStream_t st = h->stream;
  auto give_up = [&]() { (void)hipGetLastError(); h->err.clear(); for (AdeGroup& a : h->ade) a.qoff = nullptr; return 0; };
  if (dev_alloc(h, &D.dseg, nseg)) return give_up();
  for (const AdeGroup& a : h->ade)
    hipLaunchKernelGGL(disp_mark_kernel, dim3(nblk(a.n)), dim3(256), 0, st, (const uint32_t*)a.cell, a.n, g.nx, nbx, D.dseg);
  D.dseg_host.assign(nseg, 0);
  HIPCHK(h, hipMemcpyAsync(D.dseg_host.data(), D.dseg, nseg * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  long long nb = 0;
  for (size_t q = 0; q < nseg; ++q) D.dseg_host[q] = D.dseg_host[q] ? (int)nb++ : -1;
  const long long floats = nb * 3 * 256;
  if (nb == 0 || floats >= (1LL << 32)) return give_up();             // (qoff is 32 bits wide)
  HIPCHK(h, hipMemcpyAsync(D.dseg, D.dseg_host.data(), nseg * sizeof(int), hipMemcpyHostToDevice, st));
  const int big = 1 << 30;
  const int box0[6] = {big, -1, big, -1, big, -1};
  int* box = nullptr;
  if (dev_alloc(h, &D.cs, (size_t)floats) || dev_alloc(h, &D.e1, (size_t)floats) || dev_upload(h, &box, box0, 6)) return give_up();
  for (AdeGroup& a : h->ade) {
    if (dev_alloc(h, &a.qoff, (size_t)a.n, false)) return give_up();
    hipLaunchKernelGGL(disp_qoff_kernel, dim3(nblk(a.n)), dim3(256), 0, st, (const uint32_t*)a.cell, a.n, g.nx, g.ny, nbx,
                       (const int*)D.dseg, a.comp, (const float2*)a.q, a.p, a.qoff, D.cs, box);
  }
  int boxh[6];
  HIPCHK(h, hipMemcpyAsync(boxh, box, sizeof(boxh), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  D.n_blocks = (int)nb;
  for (int a = 0; a < 3; ++a) { D.lo[a] = boxh[2 * a]; D.hi[a] = boxh[2 * a + 1] + 1; }
  h->tile_cls.clear();
  h->box_paged.clear();
  D.state = 1;
  return 0;
}
// every dispersive cell `margin` cells (x: 4) or more inside the box [lo, hi) — a bulk the shell's boxes never recompute
bool disp_inside(const FdtdSolver* h, const int lo[3], const int hi[3], int margin) {
  const Disp& D = h->disp;
  if (D.state != 1) return false;
  for (int a = 0; a < 3; ++a) {
    const int m = a == 0 ? std::max(margin ? 8 : 4, margin) : margin;       // (x: whole float4 lanes; the boxes' halo lanes reach 8 columns)
    if (D.lo[a] < lo[a] + m || D.hi[a] > hi[a] - m) return false;
  }
  return true;
}

// One-off: slots for the nodes of every source list in paged storage (fdtd_fused2.hpp SrcP).  Needs one GPU and at most two LAYERS:
// a list goes to layer 0, or — where it meets an earlier list of its side (E / H) on a node: the second polarisation component of
// a TFSF box — to layer 1, which the kernels add behind layer 0 as the list kernels would one after the other; a list that meets a
// layer-1 list, or lists a node twice, leaves the problem to single steps.  state = 1 (ready) or -1.  -1 = a HIP error.
int spg_setup(FdtdSolver* h) {
  SrcPaged& S = h->spg;
  if (S.state != 0) return 0;
  S.state = -1;
  const GridP& g = h->g;
  if (h->spg_on == 0 || h->comm || g.nx % 4 != 0) return 0;
  long long total = 0;
  for (const PointSrc& s : h->psrc) total += s.n_e + s.n_h;
  for (const Tfsf& t : h->tfsf) total += t.e.n_targets + t.h.n_targets;
  if (total == 0) return 0;
  const int nbx = (g.nx + 255) / 256;
  const size_t nseg = (size_t)g.nz * g.ny * nbx;
  hipStream_t st = h->stream;
  auto give_up = [&]() { (void)hipGetLastError(); h->err.clear(); return 0; };
  if (dev_alloc(h, &S.sseg, nseg)) return give_up();
  // every (cells, comps, n, soff, layer, H side?) of every list, in the order launch_sources adds them: TFSF boxes, then point lists
  auto each = [&](auto fn) {
    for (Tfsf& t : h->tfsf) {
      if (t.e.n_targets) fn((const uint32_t*)t.e.cell, (const int32_t*)t.e.comp, t.e.n_targets, &t.e.soff, &t.e.layer, false);
      if (t.h.n_targets) fn((const uint32_t*)t.h.cell, (const int32_t*)t.h.comp, t.h.n_targets, &t.h.soff, &t.h.layer, true);
    }
    for (PointSrc& s : h->psrc) {
      if (s.n_e) fn((const uint32_t*)s.cell_e, (const int32_t*)s.comp_e, s.n_e, &s.soff_e, &s.layer_e, false);
      if (s.n_h) fn((const uint32_t*)s.cell_h, (const int32_t*)s.comp_h, s.n_h, &s.soff_h, &s.layer_h, true);
    }
  };
  each([&](const uint32_t* cell, const int32_t*, long long n, uint32_t**, int*, bool) {
    hipLaunchKernelGGL(disp_mark_kernel, dim3(nblk(n)), dim3(256), 0, st, cell, n, g.nx, nbx, S.sseg);
  });
  std::vector<int>& seg = S.sseg_host;
  seg.assign(nseg, 0);
  HIPCHK(h, hipMemcpyAsync(seg.data(), S.sseg, nseg * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  long long nb = 0;
  for (size_t q = 0; q < nseg; ++q) seg[q] = seg[q] ? (int)nb++ : -1;
  const long long floats = nb * 3 * 256;
  if (nb == 0 || floats >= (1LL << 32)) return give_up();
  HIPCHK(h, hipMemcpyAsync(S.sseg, seg.data(), nseg * sizeof(int), hipMemcpyHostToDevice, st));
  int *occ[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *box = nullptr, *hit = nullptr;       // [side][layer]
  const int big = 1 << 30;
  const int box0[6] = {big, -1, big, -1, big, -1};
  if (dev_alloc(h, &S.e1, (size_t)floats) || dev_alloc(h, &S.h2, (size_t)floats) || dev_alloc(h, &S.e2, (size_t)floats) ||
      dev_upload(h, &box, box0, 6) || dev_alloc(h, &hit, 2)) return give_up();
  for (int sd = 0; sd < 2; ++sd) for (int ly = 0; ly < 2; ++ly) if (dev_alloc(h, &occ[sd][ly], (size_t)floats)) return give_up();
  bool failed = false, hip_failed = false;
  int layers = 1;
  S.any_h = false;
  each([&](const uint32_t* cell, const int32_t* comp, long long n, uint32_t** soff, int* layer, bool h_side) {
    if (failed || hip_failed) return;
    if (dev_alloc(h, soff, (size_t)n, false)) { failed = true; return; }
    S.any_h = S.any_h || h_side;
    hipLaunchKernelGGL(src_soff_kernel, dim3(nblk(n)), dim3(256), 0, st, cell, comp, n, g.nx, g.ny, nbx, (const int*)S.sseg, *soff, box);
    // the layer: above every earlier list of this side it meets
    int hh[2] = {0, 0};
    auto probe = [&](int ly, int mark) {
      if (hipMemsetAsync(hit, 0, 2 * sizeof(int), st) != hipSuccess) { hip_failed = true; return; }
      hipLaunchKernelGGL(src_layer_kernel, dim3(nblk(n)), dim3(256), 0, st, (const uint32_t*)*soff, n, occ[h_side ? 1 : 0][ly], hit, mark);
      if (hipMemcpyAsync(hh, hit, sizeof(hh), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) hip_failed = true;
    };
    probe(1, 0);
    if (hip_failed) return;
    if (hh[0] > 0) { failed = true; return; }            // it would need a third layer
    probe(0, 0);
    if (hip_failed) return;
    *layer = hh[0] > 0 ? 1 : 0;
    layers = std::max(layers, *layer + 1);
    probe(*layer, 1);
    if (!hip_failed && hh[1] > 0) failed = true;         // a node listed twice
  });
  for (int sd = 0; sd < 2; ++sd) for (int ly = 0; ly < 2; ++ly) release_buf(h, occ[sd][ly]);
  if (hip_failed) return fail(h, "spg_setup: a HIP call failed");
  if (failed) return give_up();
  if (layers > 1 && (dev_alloc(h, &S.e1b, (size_t)floats) || dev_alloc(h, &S.h2b, (size_t)floats) || dev_alloc(h, &S.e2b, (size_t)floats)))
    return give_up();
  HIPCHK(h, hipStreamSynchronize(st));
  const SrcT tab{S.e1, S.h2, S.e2, S.any_h ? 1 : 0, 1, S.e1b, S.h2b, S.e2b};
  if (dev_upload(h, &S.tab, &tab, 1)) return give_up();
  S.n_blocks = (int)nb;
  S.state = 1;
  h->box_paged.clear();
  return 0;
}
// Shell2P::paged of a box: does a row segment it visits — rows j0 - 2 .. j1, planes k0 - 1 .. k1, columns i0 - 8 .. i1 + 7 — hold a
// source node (bit 0) / a dispersive cell (bit 1)?  Found once per box on the host maps of the segments.
int shell2_box_paged(FdtdSolver* h, const Shell2Box& bx) {
  const std::array<int, 6> key{bx.i0, bx.i1, bx.j0, bx.j1, bx.k0, bx.k1};
  auto it = h->box_paged.find(key);
  if (it != h->box_paged.end()) return it->second;
  const GridP& g = h->g;
  const int nbx = (g.nx + 255) / 256;
  const bool per_x = h->cfg.bc[0] == FDTD_BC_PERIODIC;
  const int t0 = per_x ? 0 : std::max(0, bx.i0 - 8) / 256, t1 = per_x ? nbx - 1 : std::min(g.nx - 1, bx.i1 + 7) / 256;
  int flags = 0;
  const std::vector<int>* maps[2] = {h->spg.state == 1 ? &h->spg.sseg_host : nullptr, h->disp.state == 1 ? &h->disp.dseg_host : nullptr};
  for (int m = 0; m < 2; ++m) {
    if (!maps[m] || maps[m]->empty()) continue;
    bool found = false;
    for (int k = std::max(0, bx.k0 - 1); k <= std::min(g.nz - 1, bx.k1) && !found; ++k)
      for (int j = std::max(0, bx.j0 - 2); j <= std::min(g.ny - 1, bx.j1) && !found; ++j)
        for (int t = t0; t <= t1; ++t)
          if ((*maps[m])[((size_t)k * g.ny + j) * nbx + t] >= 0) { found = true; break; }
    if (found) flags |= 1 << m;
  }
  h->box_paged[key] = flags;
  return flags;
}
// the three arrays (per layer) of the pair (n, n + 1), and the incident grids of the TFSF boxes advanced through both steps — in the
// order of two single steps (the H-side terms of step n were added to H^{n-1/2} in place just before, launch_sources):
//   incident H(n) | E-side terms of n | incident E(n) | H-side terms of n+1 | incident H(n+1) | E-side terms of n+1 | incident E(n+1)
void spg_fill(FdtdSolver* h, long long n, hipStream_t st) {
  SrcPaged& S = h->spg;
  auto fill = [&](float* val0, float* val1, bool e_side, long long step) {
    for (Tfsf& t : h->tfsf) {
      const TfsfList& L = e_side ? t.e : t.h;
      if (!L.n_targets) continue;
      hipLaunchKernelGGL(src_fill_tfsf_kernel, dim3(nblk(L.n_targets)), dim3(256), 0, st, L.layer ? val1 : val0, (const uint32_t*)L.soff,
                         (const int32_t*)L.start, (const float*)L.w, (const int32_t*)L.aux, (const float*)(e_side ? t.h1 : t.e1), L.n_targets,
                         step >= t.n_steps ? 1 : 0);
    }
    for (PointSrc& s : h->psrc) {
      const long long nn = e_side ? s.n_e : s.n_h;
      if (!nn) continue;
      hipLaunchKernelGGL(src_fill_points_kernel, dim3(nblk(nn)), dim3(256), 0, st, (e_side ? s.layer_e : s.layer_h) ? val1 : val0,
                         (const uint32_t*)(e_side ? s.soff_e : s.soff_h), (const float*)(e_side ? s.wre_e : s.wre_h),
                         (const float*)(e_side ? s.wim_e : s.wim_h), (const float2*)(e_side ? s.wave_e : s.wave_h), step, nn,
                         step >= s.n_steps ? 1 : 0);
    }
  };
  advance_tfsf_aux(h, false, n, st, false);
  fill(S.e1, S.e1b, true, n);
  advance_tfsf_aux(h, true, n, st, false);
  if (S.any_h) fill(S.h2, S.h2b, false, n + 1);
  advance_tfsf_aux(h, false, n + 1, st, false);
  fill(S.e2, S.e2b, true, n + 1);
  advance_tfsf_aux(h, true, n + 1, st, false);
}
SrcP spg_params(const FdtdSolver* h) {
  const SrcPaged& S = h->spg;
  SrcP p;
  p.sseg = S.sseg; p.t = S.tab;
  return p;
}

// z boundary conditions of a single slab (no neighbour): fill ghost planes
void fill_ghost_h(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  const int bc0 = h->cfg.bc[4];
  if (bc0 == FDTD_BC_PERIODIC) {
    copy_plane(h, h->f.hx - pc, h->f.hx + (long long)(h->g.nz - 1) * pc, st);
    copy_plane(h, h->f.hy - pc, h->f.hy + (long long)(h->g.nz - 1) * pc, st);
  } else if (bc0 == FDTD_BC_PMC) {
    hipLaunchKernelGGL(negate_copy_kernel, dim3(nblk(pc)), dim3(256), 0, st, h->f.hx - pc, (const float*)h->f.hx, pc);
    hipLaunchKernelGGL(negate_copy_kernel, dim3(nblk(pc)), dim3(256), 0, st, h->f.hy - pc, (const float*)h->f.hy, pc);
  }
}
void fill_ghost_e(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  if (h->cfg.bc[5] == FDTD_BC_PERIODIC) {
    hipMemcpyAsync(h->f.ex + (long long)h->g.nz * pc, h->f.ex, pc * 4, hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(h->f.ey + (long long)h->g.nz * pc, h->f.ey, pc * 4, hipMemcpyDeviceToDevice, st);
  }
}

// ---- halo exchange over RCCL ----------------------------------------------------------------
// H phase: my top Hx,Hy plane -> upper neighbour's ghost(-1); E phase: my bottom Ex,Ey plane
// -> lower neighbour's ghost(nz).  Periodic z wraps rank n-1 <-> 0.
int exchange(FdtdSolver* h, bool e_side, hipStream_t st) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  const int nz = h->g.nz;
  const bool has_lo = h->cfg.bc[4] == FDTD_BC_NEIGHBOR, has_hi = h->cfg.bc[5] == FDTD_BC_NEIGHBOR;
  const int lo = (h->rank - 1 + h->n_ranks) % h->n_ranks, hi = (h->rank + 1) % h->n_ranks;
  NCCLCHK(h, ncclGroupStart());
  if (!e_side) {
    if (has_hi) {
      NCCLCHK(h, ncclSend(h->f.hx + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
      NCCLCHK(h, ncclSend(h->f.hy + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
    }
    if (has_lo) {
      NCCLCHK(h, ncclRecv(h->f.hx - pc, pc, ncclFloat, lo, h->comm, st));
      NCCLCHK(h, ncclRecv(h->f.hy - pc, pc, ncclFloat, lo, h->comm, st));
    }
  } else {
    if (has_lo) {
      NCCLCHK(h, ncclSend(h->f.ex, pc, ncclFloat, lo, h->comm, st));
      NCCLCHK(h, ncclSend(h->f.ey, pc, ncclFloat, lo, h->comm, st));
    }
    if (has_hi) {
      NCCLCHK(h, ncclRecv(h->f.ex + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
      NCCLCHK(h, ncclRecv(h->f.ey + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
    }
  }
  NCCLCHK(h, ncclGroupEnd());
  return 0;
}

// Fused sweep on z-slabs: the chunk prologue recomputes H^{n+1/2}[-1] from E^n[-1] (all three
// components) and the corrected H^{n-1/2}_{x,y}[-1]; the top plane needs E^n_{x,y}[nz].
//   exchange_fused_h : corrected H_x,H_y of my top plane  -> upper neighbour's ghost(-1)
//   exchange_fused_e : E_x,E_y,E_z of my top plane -> upper ghost(-1); E_x,E_y of my bottom plane
//                      -> lower neighbour's ghost(nz)
int exchange_fused_h(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  const int nz = h->g.nz;
  const bool has_lo = h->cfg.bc[4] == FDTD_BC_NEIGHBOR, has_hi = h->cfg.bc[5] == FDTD_BC_NEIGHBOR;
  const int lo = (h->rank - 1 + h->n_ranks) % h->n_ranks, hi = (h->rank + 1) % h->n_ranks;
  NCCLCHK(h, ncclGroupStart());
  if (has_hi) {
    NCCLCHK(h, ncclSend(h->f.hx + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
    NCCLCHK(h, ncclSend(h->f.hy + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
  }
  if (has_lo) {
    NCCLCHK(h, ncclRecv(h->f.hx - pc, pc, ncclFloat, lo, h->comm, st));
    NCCLCHK(h, ncclRecv(h->f.hy - pc, pc, ncclFloat, lo, h->comm, st));
  }
  NCCLCHK(h, ncclGroupEnd());
  return 0;
}

int exchange_fused_e(FdtdSolver* h, hipStream_t st) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  const int nz = h->g.nz;
  const bool has_lo = h->cfg.bc[4] == FDTD_BC_NEIGHBOR, has_hi = h->cfg.bc[5] == FDTD_BC_NEIGHBOR;
  const int lo = (h->rank - 1 + h->n_ranks) % h->n_ranks, hi = (h->rank + 1) % h->n_ranks;
  float* e3[3] = {h->f.ex, h->f.ey, h->f.ez};
  // RCCL pairs the k-th send to a peer with the k-th receive from it: when lo == hi (one or two
  // ranks, periodic z) the posting order below must mirror the peer's: [to-hi sends][to-lo sends]
  // on the sending side <-> [from-lo receives][from-hi receives] on the receiving side.
  NCCLCHK(h, ncclGroupStart());
  if (has_hi)
    for (float* p : e3) NCCLCHK(h, ncclSend(p + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
  if (has_lo) {
    NCCLCHK(h, ncclSend(h->f.ex, pc, ncclFloat, lo, h->comm, st));
    NCCLCHK(h, ncclSend(h->f.ey, pc, ncclFloat, lo, h->comm, st));
    for (float* p : e3) NCCLCHK(h, ncclRecv(p - pc, pc, ncclFloat, lo, h->comm, st));
  }
  if (has_hi) {
    NCCLCHK(h, ncclRecv(h->f.ex + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
    NCCLCHK(h, ncclRecv(h->f.ey + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
  }
  NCCLCHK(h, ncclGroupEnd());
  return 0;
}

// Pipelined z-slab schedule: ONE exchange per step carries everything the neighbours' boundary
// chunks need for the next sweep — up: E_x,E_y,E_z and the pre-corrected H_x,H_y of my top plane
// (-> upper ghost(-1)); down: E_x,E_y of my bottom plane (-> lower ghost(nz)).
// `fs`: the set whose planes travel (default: the current one) — the middle step of a slab pair ships the third set's.
// psi_set: which H-side psi arrays travel (the sender's newest top plane -> slot nz of the set the receiver's next sweep reads):
// 0 = the current set, 1 = the temporary set of a z hole (after the first step of a CPML slab pair), 2 = the other set of the
// ping-pong (after its second step, before the swap)
int exchange_fused_all(FdtdSolver* h, hipStream_t st, bool pml_with_sweep = false, const FieldP* fs = nullptr, int psi_set = 0) {
  dbg_sync(h);
  const long long pc = plane_cells(h);
  const int nz = h->g.nz;
  const bool has_lo = h->cfg.bc[4] == FDTD_BC_NEIGHBOR, has_hi = h->cfg.bc[5] == FDTD_BC_NEIGHBOR;
  if (!has_lo && !has_hi) return 0;
  const int lo = (h->rank - 1 + h->n_ranks) % h->n_ranks, hi = (h->rank + 1) % h->n_ranks;
  const FieldP F = fs ? *fs : h->f;
  float* up5[5] = {F.ex, F.ey, F.ez, F.hx, F.hy};
  // With the x / y CPML inside the sweep the chunk prologue of the upper rank corrects H[-1] itself and needs the
  // H-side psi of that plane — my top plane, CURRENT set (what the next sweep reads): into slot nz of its arrays.
  // Decided by the configuration alone (both sides must post the same messages), not by what a rank's sweep ends up doing.
  const bool psi_too = pml_with_sweep && h->pml_fused > 0;
  auto psi_of = [&](int a, int q) { const PmlAxisDev& P = h->pml[a]; return psi_set == 1 ? P.psi_ht[q] : (psi_set == 2 ? P.psi_h2[q] : P.psi_h[q]); };
  // posting order mirrors the peer's (see exchange_fused_e): [to-hi][to-lo] <-> [from-lo][from-hi]
  NCCLCHK(h, ncclGroupStart());
  if (has_hi) {
    for (float* p : up5) NCCLCHK(h, ncclSend(p + (long long)(nz - 1) * pc, pc, ncclFloat, hi, h->comm, st));
    if (psi_too)
      for (int a = 0; a < 2; ++a)
        for (int q = 0; q < 2 && h->pml[a].ns > 0; ++q)
          NCCLCHK(h, ncclSend(psi_of(a, q) + (size_t)(nz - 1) * h->pml[a].psi_plane, h->pml[a].psi_plane, ncclFloat, hi, h->comm, st));
  }
  if (has_lo) {
    NCCLCHK(h, ncclSend(F.ex, pc, ncclFloat, lo, h->comm, st));
    NCCLCHK(h, ncclSend(F.ey, pc, ncclFloat, lo, h->comm, st));
    for (float* p : up5) NCCLCHK(h, ncclRecv(p - pc, pc, ncclFloat, lo, h->comm, st));
    if (psi_too)
      for (int a = 0; a < 2; ++a)
        for (int q = 0; q < 2 && h->pml[a].ns > 0; ++q)
          NCCLCHK(h, ncclRecv(psi_of(a, q) + (size_t)nz * h->pml[a].psi_plane, h->pml[a].psi_plane, ncclFloat, lo, h->comm, st));
  }
  if (has_hi) {
    NCCLCHK(h, ncclRecv(F.ex + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
    NCCLCHK(h, ncclRecv(F.ey + (long long)nz * pc, pc, ncclFloat, hi, h->comm, st));
  }
  NCCLCHK(h, ncclGroupEnd());
  return 0;
}

// ---- field energy (K7) ------------------------------------------------------------------------
// W = sum |E|^2 + eta0^2 sum |H|^2 over the slab: fixed launch geometry, two passes, no atomics
int eval_energy(FdtdSolver* h, hipStream_t st, double* out) {
  const long long nc = n_cells(h);
  unsigned blocks = nblk(nc);
  if (blocks > (unsigned)kEnergyBlocks) blocks = kEnergyBlocks;
  hipLaunchKernelGGL(energy_partial_kernel, dim3(blocks), dim3(256), 0, st, (const float*)h->f.ex, (const float*)h->f.ey,
                     (const float*)h->f.ez, (const float*)h->f.hx, (const float*)h->f.hy, (const float*)h->f.hz, nc,
                     h->energy_dev + 1);
  hipLaunchKernelGGL(energy_final_kernel, dim3(1), dim3(256), 0, st, (const double*)(h->energy_dev + 1), (int)blocks,
                     h->energy_dev);
  HIPCHK(h, hipMemcpyAsync(out, h->energy_dev, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return 0;
}

// ---- monitors ---------------------------------------------------------------------------------
void record_monitors(FdtdSolver* h, long long n, bool post, hipStream_t st, const F2Plan* in_sweep = nullptr) {
  dbg_sync(h);
  for (size_t mi = 0; mi < h->mons.size(); ++mi) {
    Monitor& m = h->mons[mi];
    if (m.next >= m.steps.size() || m.steps[m.next] != n) continue;
    if (in_sweep) {                    // monitors recorded behind the two-step sweep (pair_record)
      bool skip = false;
      for (int q : in_sweep->mons) skip = skip || q == (int)mi;
      if (skip) continue;
    }
    const long long rec = (long long)m.next;
    const int nc = (int)m.comps.size();
    // one launch per monitor and phase.  pre (before the H update): E^n, and half of H^{n-1/2} for
    // time monitors; post (after it): the other half of H / the H terms of the running DFT.
    RecP r{};
    for (int ic = 0; ic < nc; ++ic) {
      const int c = m.comps[ic];
      const bool is_h = c >= 3;
      bool take;
      if (m.kind == FDTD_MON_TIME) take = is_h || !post;
      else take = is_h == post;
      if (!take) continue;
      r.f[r.n] = field_ptr(h, c);
      r.slot[r.n] = ic;
      r.scale[r.n] = is_h ? 0.5f : 1.0f;
      r.acc[r.n] = is_h ? 1 : 0;
      r.n++;
    }
    if (r.n > 0 && m.cells > 0) {
      const dim3 grid(nblk(m.cells), r.n);
      if (m.kind == FDTD_MON_TIME) {
        float* out = reinterpret_cast<float*>(m.data) + rec * nc * m.cells;
        hipLaunchKernelGGL(time_record_multi_kernel, grid, dim3(256), 0, st, r, h->g, m.box, out, (long long)m.cells);
      } else {
        const long long fstride = (long long)nc * m.cells;
        hipLaunchKernelGGL(dft_record_multi_kernel, grid, dim3(256), 0, st, r, h->g, m.box,
                           reinterpret_cast<float2*>(m.data), (long long)m.cells, fstride,
                           (const float2*)((post ? m.phase_h : m.phase_e) + rec * m.nf), m.nf);
      }
    }
    if (post) m.next++;
  }
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* fdtd_last_error(const FdtdSolver* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fdtd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fdtd_far_field(int device, int n_u, int n_v, const double* u, const double* v, const double* wu, const double* wv,
                   const double* currents, double w0, double k_re, double k_im, int n_dir, const double* r_u,
                   const double* r_v, const double* r_w, double* out) {
  if (n_u < 1 || n_v < 1 || n_dir < 0 || !u || !v || !wu || !wv || !currents || (n_dir && (!r_u || !r_v || !r_w || !out)))
    return fail(nullptr, "fdtd_far_field: bad argument");
  if (n_dir == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, "fdtd_far_field: hipSetDevice(%d) failed", device);
  const size_t n = (size_t)n_u * n_v;
  const size_t sizes[9] = {(size_t)n_u, (size_t)n_v, (size_t)n_u, (size_t)n_v, 8 * n, (size_t)n_dir, (size_t)n_dir,
                           (size_t)n_dir, 8 * (size_t)n_dir};
  const double* host[8] = {u, v, wu, wv, currents, r_u, r_v, r_w};
  double* dev[9] = {};
  int rc = 0;
  for (int i = 0; i < 9 && !rc; ++i) {
    if (hipMalloc((void**)&dev[i], sizes[i] * sizeof(double)) != hipSuccess) rc = fail(nullptr, "fdtd_far_field: out of device memory");
    else if (i < 8 && hipMemcpy(dev[i], host[i], sizes[i] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(nullptr, "fdtd_far_field: upload failed");
  }
  if (!rc) {
    FarP p;
    p.u = dev[0]; p.v = dev[1]; p.wu = dev[2]; p.wv = dev[3];
    p.cur = reinterpret_cast<const double2*>(dev[4]);
    p.n_u = n_u; p.n_v = n_v; p.w0 = w0; p.k_re = k_re; p.k_im = k_im;
    p.r_u = dev[5]; p.r_v = dev[6]; p.r_w = dev[7]; p.out = dev[8];
    hipLaunchKernelGGL(far_field_kernel, dim3((unsigned)n_dir), dim3(256), 0, 0, p);
    if (hipMemcpy(out, dev[8], sizes[8] * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(nullptr, "fdtd_far_field: %s", hipGetErrorString(hipGetLastError()));
  }
  for (double* q : dev) if (q) hipFree(q);
  return rc;
}

int fdtd_create(const FdtdConfig* cfg, FdtdSolver** out) {
  if (!cfg || !out) return fail(nullptr, "fdtd_create: null argument");
  if (cfg->nx < 1 || cfg->ny < 1 || cfg->nz < 1) return fail(nullptr, "fdtd_create: bad grid %d x %d x %d", cfg->nx, cfg->ny, cfg->nz);
  if ((long long)cfg->nx * cfg->ny * (cfg->nz + 2) >= (1LL << 32))
    return fail(nullptr, "fdtd_create: slab of %d x %d x %d cells exceeds the 2^32 cell index range", cfg->nx, cfg->ny, cfg->nz);
  for (int a = 0; a < 3; ++a) {
    const bool p0 = cfg->bc[2 * a] == FDTD_BC_PERIODIC, p1 = cfg->bc[2 * a + 1] == FDTD_BC_PERIODIC;
    if (p0 != p1 && a < 2) return fail(nullptr, "fdtd_create: periodic boundary must be set on both faces of axis %d", a);
    if (cfg->bc[2 * a + 1] == FDTD_BC_PMC) return fail(nullptr, "fdtd_create: PMC is supported on min faces only");
    if (a < 2 && (cfg->bc[2 * a] == FDTD_BC_NEIGHBOR || cfg->bc[2 * a + 1] == FDTD_BC_NEIGHBOR))
      return fail(nullptr, "fdtd_create: neighbour faces exist only along z");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, "fdtd_create: no HIP device available");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "fdtd_create: device %d out of range (%d devices)", cfg->device, ndev);
  FdtdSolver* h = new FdtdSolver();
  h->cfg = *cfg;
  if (hipSetDevice(cfg->device) != hipSuccess) { delete h; return fail(nullptr, "hipSetDevice(%d) failed", cfg->device); }
  GridP& g = h->g;
  g.nx = cfg->nx; g.ny = cfg->ny; g.nz = cfg->nz;
  g.sxy = (long long)cfg->nx * cfg->ny;
  g.bcx0 = cfg->bc[0]; g.bcx1 = cfg->bc[1]; g.bcy0 = cfg->bc[2]; g.bcy1 = cfg->bc[3];
  g.pec_z0 = cfg->bc[4] == FDTD_BC_PEC;
  g.psi_ghost = cfg->bc[4] == FDTD_BC_NEIGHBOR ? 1 : 0;
  g.ch = cfg->ch;
  // measured on MI355X, 512^3 (profiles/r01a_probe_geometry_512.jsonl): short z-marches win —
  // the 256 MiB Infinity Cache already serves the k+1 plane re-read, and more, smaller
  // workgroups balance the 256 CUs better than long marches.
  h->zchunk = cfg->z_chunk > 0 ? cfg->z_chunk : 2;
  h->zchunk_f = cfg->z_chunk > 0 ? cfg->z_chunk : 16;
  h->user_geometry = cfg->z_chunk > 0;
  h->rows = 4;
  int rc = 0;
  const size_t fcount = (size_t)g.sxy * (g.nz + 2);
  h->field_bytes = fcount * sizeof(float);
  rc = alloc_field_set(h, h->fbase, fcount, 0);
  if (!rc) {
    h->f.ex = h->fbase[0] + g.sxy; h->f.ey = h->fbase[1] + g.sxy; h->f.ez = h->fbase[2] + g.sxy;
    h->f.hx = h->fbase[3] + g.sxy; h->f.hy = h->fbase[4] + g.sxy; h->f.hz = h->fbase[5] + g.sxy;
    rc = dev_alloc(h, &h->energy_dev, 1 + kEnergyBlocks);
  }
  if (!rc) {
    // default: unit steps, vacuum
    const int N[3] = {g.nx, g.ny, g.nz};
    for (int a = 0; a < 3 && !rc; ++a) {
      // the z vectors carry one ghost entry below and above (indices -1 and nz)
      std::vector<float> ones(N[a] + (a == 2 ? 2 : 0), 1.0f);
      rc = dev_upload(h, &h->ip[a], (const float*)ones.data(), ones.size());
      if (!rc) rc = dev_upload(h, &h->idl[a], (const float*)ones.data(), ones.size());
      if (!rc && a == 2) {
        h->step_base[0] = h->ip[2]; h->step_base[1] = h->idl[2];
        h->ip[2] += 1; h->idl[2] += 1;
      }
    }
  }
  if (!rc && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(nullptr, "hipStreamCreate failed");
  // The comm stream carries the ghost exchanges and the boundary chunks.  High priority: its few
  // workgroups are dispatched ahead of the interior sweep's, otherwise the RCCL Send/Recv kernel
  // queues behind a full machine (measured: 138 us instead of 30 us, profiles/r01h_slab_timeline.txt).
  if (!rc) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    if (hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, greatest) != hipSuccess)
      rc = fail(nullptr, "hipStreamCreate failed");
  }
  if (!rc) {
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1);
    hipEventCreateWithFlags(&h->ev_h_int, hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_h_bnd, hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_e_int, hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_e_bnd, hipEventDisableTiming);
  }
  if (rc) {
    g_create_error = h->err.empty() ? g_create_error : h->err;
    fdtd_destroy(h);
    return -1;
  }
  *out = h;
  return 0;
}

void fdtd_destroy(FdtdSolver* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  hipDeviceSynchronize();
  if (h->comm) ncclCommDestroy(h->comm);
  for (DevBuf& b : h->bufs) hipFree(b.p);
  for (hipEvent_t e : h->kev) hipEventDestroy(e);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->ev_h_int) hipEventDestroy(h->ev_h_int);
  if (h->ev_h_bnd) hipEventDestroy(h->ev_h_bnd);
  if (h->ev_e_int) hipEventDestroy(h->ev_e_int);
  if (h->ev_e_bnd) hipEventDestroy(h->ev_e_bnd);
  if (h->ev_shell_a) hipEventDestroy(h->ev_shell_a);
  if (h->ev_rec) hipEventDestroy(h->ev_rec);
  if (h->ev_shell_b) hipEventDestroy(h->ev_shell_b);
  if (h->stream) hipStreamDestroy(h->stream);
  if (h->comm_stream && !h->streams_shared) hipStreamDestroy(h->comm_stream);
  if (h->box_stream) hipStreamDestroy(h->box_stream);
  if (h->ev_box) hipEventDestroy(h->ev_box);
  if (h->ev_box_in) hipEventDestroy(h->ev_box_in);
  delete h;
}

int fdtd_set_steps(FdtdSolver* h, int axis, const float* inv_primal, const float* inv_dual, int n) {
  if (!h) return -1;
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  if (axis < 0 || axis > 2) return fail(h, "fdtd_set_steps: bad axis %d", axis);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (axis == 2 && n == N[2] + 2) {        // ghost entries supplied (z-slab of a non-uniform grid)
    HIPCHK(h, hipMemcpy(h->step_base[0], inv_primal, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->step_base[1], inv_dual, (size_t)n * 4, hipMemcpyHostToDevice));
    return 0;
  }
  if (n != N[axis]) return fail(h, "fdtd_set_steps: axis %d expects %d entries, got %d", axis, N[axis], n);
  HIPCHK(h, hipMemcpy(h->ip[axis], inv_primal, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->idl[axis], inv_dual, (size_t)n * 4, hipMemcpyHostToDevice));
  if (axis == 2) {                         // ghost entries: wrap for periodic z, replicate otherwise
    const bool per = h->cfg.bc[4] == FDTD_BC_PERIODIC;
    const float lo[2] = {per ? inv_primal[n - 1] : inv_primal[0], per ? inv_dual[n - 1] : inv_dual[0]};
    const float hi[2] = {per ? inv_primal[0] : inv_primal[n - 1], per ? inv_dual[0] : inv_dual[n - 1]};
    for (int q = 0; q < 2; ++q) {
      HIPCHK(h, hipMemcpy(h->step_base[q], &lo[q], 4, hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(h->step_base[q] + n + 1, &hi[q], 4, hipMemcpyHostToDevice));
    }
  }
  return 0;
}

int fdtd_set_media(FdtdSolver* h, const float* ca, const float* cb, int n_media) {
  if (!h) return -1;
  if (n_media < 2 || n_media > kMaxMediaWide) return fail(h, "fdtd_set_media: n_media must be in [2, %d], got %d", kMaxMediaWide, n_media);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  std::vector<float2> lut(n_media);
  for (int i = 0; i < n_media; ++i) { lut[i].x = ca[i]; lut[i].y = cb[i]; }
  if (dev_upload(h, &h->lut, (const float2*)lut.data(), lut.size())) return -1;
  h->n_media = n_media;
  h->ca1 = ca[1];
  h->cb1 = cb[1];
  h->cb_host.assign(cb, cb + n_media);
  return 0;
}

extern "C++" {
namespace {
// pack the three index volumes into one word per cell (10 bits per component), derive the
// row-segment words and upload both
template <typename T>
int upload_material(FdtdSolver* h, const T* mat, size_t count) {
  const size_t nc = (size_t)n_cells(h);
  if (count != 3 * nc) return fail(h, "fdtd_set_material: expected %zu entries, got %zu", 3 * nc, count);
  if (h->disp.state == 1) h->tile_cls.clear();
  if (h->n_media == 0) return fail(h, "fdtd_set_material: call fdtd_set_media first");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const GridP& g = h->g;
  const size_t fcount = (size_t)g.sxy * (g.nz + 2);
  std::vector<uint32_t> packed(fcount, kBgWord);                 // ghost planes: background medium
  uint32_t* dst = packed.data() + g.sxy;
  const uint32_t nm = (uint32_t)h->n_media;
  const bool wide = h->n_media > kMaxMedia - 1;           // more media than 10-bit indices name: two words per cell (fdtd_kernels.hpp MatP)
  std::vector<uint32_t> packed_b(wide ? fcount : 0, 1u);
  uint32_t* dst_b = wide ? packed_b.data() + g.sxy : nullptr;
  if (wide) std::fill(packed.begin(), packed.end(), 1u | (1u << 16));
  for (size_t i = 0; i < nc; ++i) {
    const uint32_t m0 = mat[i], m1 = mat[nc + i], m2 = mat[2 * nc + i];
    if (m0 >= nm || m1 >= nm || m2 >= nm) return fail(h, "fdtd_set_material: medium index out of range at cell %zu", i);
    if (wide) { dst[i] = m0 | (m1 << 16); dst_b[i] = m2; }
    else dst[i] = m0 | (m1 << 10) | (m2 << 20);
  }
  h->mat4b = nullptr;
  if (wide) {
    uint32_t* base_b = nullptr;
    if (dev_upload(h, &base_b, (const uint32_t*)packed_b.data(), fcount)) return -1;
    h->mat4b = base_b + g.sxy;
  }
  const int nbx = (g.nx + 255) / 256;
  std::vector<uint32_t> roww((size_t)g.nz * g.ny * nbx);
  for (size_t r = 0; r < (size_t)g.nz * g.ny; ++r)
    for (int bx = 0; bx < nbx; ++bx) {
      const uint32_t* row = dst + r * g.nx;
      const int i1 = std::min(g.nx, (bx + 1) * 256);
      uint32_t w = row[bx * 256];
      for (int i = bx * 256 + 1; i < i1; ++i) if (row[i] != w) { w = kMixedWord; break; }
      roww[r * nbx + bx] = w;
    }
  uint32_t* base = nullptr;
  if (dev_upload(h, &base, (const uint32_t*)packed.data(), fcount)) return -1;
  if (dev_upload(h, &h->roww, (const uint32_t*)roww.data(), roww.size())) return -1;
  h->roww_host = wide ? std::vector<uint32_t>() : roww;
  h->tile_cls.clear();
  h->mat4 = base + g.sxy;
  return 0;
}
}  // namespace
}  // extern "C++"

int fdtd_set_material(FdtdSolver* h, const uint8_t* mat, size_t bytes) {
  if (!h) return -1;
  return upload_material(h, mat, bytes);
}

int fdtd_set_material16(FdtdSolver* h, const uint16_t* mat, size_t count) {
  if (!h) return -1;
  return upload_material(h, mat, count);
}

int fdtd_set_pml(FdtdSolver* h, int axis, int n_lo, int n_hi, const float* kinv_e, const float* b_e,
                 const float* c_e, const float* kinv_h, const float* b_h, const float* c_h, int n) {
  if (!h) return -1;
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  if (axis < 0 || axis > 2 || n != N[axis]) return fail(h, "fdtd_set_pml: bad axis/length");
  if (n_lo < 0 || n_hi < 0 || n_lo + n_hi > n) return fail(h, "fdtd_set_pml: %d + %d layers exceed %d cells", n_lo, n_hi, n);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  PmlAxisDev& P = h->pml[axis];
  P.n_lo = n_lo; P.n_hi = n_hi; P.n = n;
  // membership of the psi arrays: [0, lo) and [hi0, n); along x rounded outwards to multiples of 4 cells
  // (the tables are identity on the cells this adds), the whole axis when the two ranges would meet
  P.lo = n_lo; P.hi0 = n - n_hi;
  if (axis == 0) { P.lo = (n_lo + 3) / 4 * 4; P.hi0 = (n - n_hi) / 4 * 4; }
  if (P.lo >= P.hi0) { P.lo = n; P.hi0 = n; }
  P.ns = P.lo + (n - P.hi0);
  std::vector<float4> ce(n), chh(n);
  std::vector<float> kve(n), kvh(n);
  for (int i = 0; i < n; ++i) {
    kve[i] = kinv_e[i] - 1.f; kvh[i] = kinv_h[i] - 1.f;
    ce[i].x = kve[i]; ce[i].y = b_e[i]; ce[i].z = c_e[i]; ce[i].w = 0.f;
    chh[i].x = kvh[i]; chh[i].y = b_h[i]; chh[i].z = c_h[i]; chh[i].w = 0.f;
  }
  if (dev_upload(h, &P.ce4, (const float4*)ce.data(), (size_t)n) || dev_upload(h, &P.ch4, (const float4*)chh.data(), (size_t)n) ||
      dev_upload(h, &P.kv_e, (const float*)kve.data(), (size_t)n) || dev_upload(h, &P.b_e, b_e, (size_t)n) ||
      dev_upload(h, &P.c_e, c_e, (size_t)n) || dev_upload(h, &P.kv_h, (const float*)kvh.data(), (size_t)n) ||
      dev_upload(h, &P.b_h, b_h, (size_t)n) || dev_upload(h, &P.c_h, c_h, (size_t)n))
    return -1;
  const size_t other = (size_t)n_cells(h) / (size_t)n;
  P.psi_count = other * P.ns;
  // (x, y axes: one more plane behind the H-side arrays — the ghost slot a z-slab rank receives its lower neighbour's
  //  top-plane psi into, fdtd_kernels.hpp GridP::psi_ghost)
  P.psi_plane = axis < 2 ? P.psi_count / (size_t)h->cfg.nz : 0;
  for (int q = 0; q < 2; ++q) {
    if (P.ns > 0 && dev_alloc(h, &P.psi_e[q], P.psi_count)) return -1;
    if (P.ns > 0 && dev_alloc(h, &P.psi_h[q], P.psi_count + P.psi_plane)) return -1;
    P.psi_h2[q] = nullptr;
    P.psi_e2[q] = nullptr; P.psi_ht[q] = nullptr; P.psi_et[q] = nullptr;
  }
  for (bool& ok : h->pml_blk_ok) ok = false;          // parameter blocks are rebuilt on next use
  h->pml_blk2_ok = false; h->pml_blk_hole_ok = false;
  return 0;
}

int fdtd_set_mirror_plus(FdtdSolver* h, int axis, int wall) {
  if (!h) return -1;
  if (axis < 0 || axis > 2) return fail(h, "fdtd_set_mirror_plus: bad axis %d", axis);
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  if (wall >= 0 && (wall < 2 || wall + 2 > N[axis]))        // (rows padded to a multiple of 4 put more PEC cells behind the ghosts)
    return fail(h, "fdtd_set_mirror_plus: two ghost cells are needed beyond the wall (wall %d of %d cells)", wall, N[axis]);
  if (wall >= 0 && h->cfg.bc[2 * axis + 1] != FDTD_BC_PEC) return fail(h, "fdtd_set_mirror_plus: the plus face of the axis must be declared PEC (plain truncation behind the ghost cells)");
  h->mirror_wall[axis] = wall < 0 ? -1 : wall;
  return 0;
}

int fdtd_set_absorber(FdtdSolver* h, int axis, int n_lo, int n_hi, const float* fb, const float* fc, int n) {
  if (!h) return -1;
  if (axis < 0 || axis > 2) return fail(h, "fdtd_set_absorber: bad axis %d", axis);
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  if (n != N[axis]) return fail(h, "fdtd_set_absorber: axis %d has %d cells, got %d", axis, N[axis], n);
  if (n_lo < 0 || n_hi < 0 || n_lo + n_hi > n) return fail(h, "fdtd_set_absorber: bad layer counts %d + %d", n_lo, n_hi);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  // identity tables on the other axes, so that the kernel always forms the full product
  for (int a = 0; a < 3; ++a) {
    if (h->damp_fb[a]) continue;
    std::vector<float> one((size_t)N[a], 1.0f);
    if (dev_upload(h, &h->damp_fb[a], one.data(), one.size())) return -1;
    if (dev_upload(h, &h->damp_fc[a], one.data(), one.size())) return -1;
    h->damp_lo[a] = 0; h->damp_hi[a] = N[a];
  }
  HIPCHK(h, hipMemcpy(h->damp_fb[axis], fb, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->damp_fc[axis], fc, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  h->damp_lo[axis] = n_lo; h->damp_hi[axis] = n - n_hi;
  h->has_damp = false;
  for (int a = 0; a < 3; ++a) if (h->damp_lo[a] > 0 || h->damp_hi[a] < N[a]) h->has_damp = true;
  return 0;
}

int fdtd_add_ade(FdtdSolver* h, int comp, int64_t n, const uint32_t* cell_index, int n_poles, const float* kap,
                 const float* bet, float cc) {
  if (!h) return -1;
  if (comp < 0 || comp > 2) return fail(h, "fdtd_add_ade: comp must be 0..2");
  if (n_poles < 1 || n_poles > kMaxPoles) return fail(h, "fdtd_add_ade: n_poles must be in [1, %d]", kMaxPoles);
  if (n <= 0) return 0;
  if (h->disp.state == 1) return fail(h, "fdtd_add_ade: the memory terms of this handle's dispersive cells are paged already (a run took step pairs); add every group before the first fdtd_run");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  AdeGroup a{};
  a.comp = comp; a.n = n;
  plane_range(cell_index, n, h->g.sxy, &a.k0, &a.k1);
  {
    bool sorted = true;
    a.strict = true;
    for (int64_t i = 1; i < n && sorted; ++i) { sorted = cell_index[i - 1] <= cell_index[i]; a.strict = a.strict && cell_index[i - 1] != cell_index[i]; }
    a.strict = a.strict && sorted;
    if (sorted) {
      a.plane_off.assign((size_t)h->g.nz + 1, n);
      int64_t i = 0;
      for (int k = 0; k <= h->g.nz; ++k) {
        while (i < n && (long long)cell_index[i] < (long long)k * h->g.sxy) ++i;
        a.plane_off[(size_t)k] = i;
      }
    }
  }
  if (dev_upload(h, &a.cell, cell_index, (size_t)n) || dev_alloc(h, &a.e_old, (size_t)n) ||
      dev_alloc(h, &a.q, (size_t)n * n_poles))
    return -1;
  a.p.n_poles = n_poles; a.p.cc = cc;
  for (int k = 0; k < n_poles; ++k) {
    a.p.kap[k].x = kap[2 * k]; a.p.kap[k].y = kap[2 * k + 1];
    a.p.bet[k].x = bet[2 * k]; a.p.bet[k].y = bet[2 * k + 1];
  }
  h->ade.push_back(a);
  return 0;
}

int fdtd_add_aniso(FdtdSolver* h, int comp, int64_t n, const uint32_t* cell_index, const uint32_t* nbr_index, const float* w_new,
                   const float* w_old) {
  if (!h) return -1;
  if (comp < 0 || comp > 2) return fail(h, "fdtd_add_aniso: comp must be 0..2");
  if (n <= 0) return 0;
  const uint64_t ncell = (uint64_t)n_cells(h);
  for (int64_t i = 0; i < n; ++i) if (cell_index[i] >= ncell) return fail(h, "fdtd_add_aniso: cell index out of range");
  for (int64_t q = 0; q < 8 * n; ++q)
    if (nbr_index[q] != kNoNode && nbr_index[q] >= ncell) return fail(h, "fdtd_add_aniso: neighbour index out of range");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  AnisoGroup a{};
  a.comp = comp; a.n = n;
  if (dev_upload(h, &a.cell, cell_index, (size_t)n) || dev_upload(h, &a.nbr, nbr_index, (size_t)8 * n) ||
      dev_upload(h, &a.w_new, w_new, (size_t)8 * n) || dev_upload(h, &a.w_old, w_old, (size_t)8 * n) ||
      dev_alloc(h, &a.old, (size_t)8 * n) || dev_alloc(h, &a.delta, (size_t)n))
    return -1;
  h->aniso.push_back(a);
  return 0;
}

int fdtd_add_point_source(FdtdSolver* h, int64_t n, const int32_t* comp, const uint32_t* cell, const float* w_re,
                          const float* w_im, int64_t n_steps, const float* wave_e, const float* wave_h) {
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const uint64_t ncell = (uint64_t)n_cells(h);
  // Entries that address the same node are merged (their weights multiply the same waveform), so the
  // kernel's plain read-modify-write touches every node once: no race, no atomics, repeatable bits.
  std::map<std::pair<int32_t, uint32_t>, std::pair<double, double>> merged;
  for (int64_t i = 0; i < n; ++i) {
    if (comp[i] < 0 || comp[i] > 5) return fail(h, "fdtd_add_point_source: bad component %d", comp[i]);
    if (cell[i] >= ncell) return fail(h, "fdtd_add_point_source: cell index out of range");
    auto& acc = merged[{comp[i], cell[i]}];
    acc.first += (double)w_re[i];
    acc.second += (double)w_im[i];
  }
  std::vector<int32_t> ce, chh;
  std::vector<uint32_t> le, lh;
  std::vector<float> re_e, im_e, re_h, im_h;
  for (const auto& kv : merged) {
    const int32_t c = kv.first.first;
    if (c < 3) { ce.push_back(c); le.push_back(kv.first.second); re_e.push_back((float)kv.second.first); im_e.push_back((float)kv.second.second); }
    else { chh.push_back(c); lh.push_back(kv.first.second); re_h.push_back((float)kv.second.first); im_h.push_back((float)kv.second.second); }
  }
  PointSrc s{};
  s.n_e = (long long)ce.size(); s.n_h = (long long)chh.size(); s.n_steps = n_steps;
  plane_range(le.data(), s.n_e, h->g.sxy, &s.ke0, &s.ke1);
  plane_range(lh.data(), s.n_h, h->g.sxy, &s.kh0, &s.kh1);
  if (s.n_e) {
    if (dev_upload(h, &s.comp_e, (const int32_t*)ce.data(), ce.size()) || dev_upload(h, &s.cell_e, (const uint32_t*)le.data(), le.size()) ||
        dev_upload(h, &s.wre_e, (const float*)re_e.data(), re_e.size()) || dev_upload(h, &s.wim_e, (const float*)im_e.data(), im_e.size()) ||
        dev_upload(h, &s.wave_e, reinterpret_cast<const float2*>(wave_e), (size_t)n_steps))
      return -1;
  }
  if (s.n_h) {
    if (dev_upload(h, &s.comp_h, (const int32_t*)chh.data(), chh.size()) || dev_upload(h, &s.cell_h, (const uint32_t*)lh.data(), lh.size()) ||
        dev_upload(h, &s.wre_h, (const float*)re_h.data(), re_h.size()) || dev_upload(h, &s.wim_h, (const float*)im_h.data(), im_h.size()) ||
        dev_upload(h, &s.wave_h, reinterpret_cast<const float2*>(wave_h), (size_t)n_steps))
      return -1;
  }
  s.host_comp_e = ce; s.host_comp_h = chh;
  s.host_cell_e = le; s.host_cell_h = lh;
  h->psrc.push_back(s);
  return 0;
}

namespace {
// group a TFSF correction list by target node (component, cell), entries of a node in their given order
int build_tfsf_list(FdtdSolver* h, TfsfList& L, int64_t n, const int32_t* comp, const uint32_t* cell, const float* w,
                    const int32_t* aux, int n_aux_max) {
  if (n <= 0) return 0;
  const uint64_t ncell = (uint64_t)n_cells(h);
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (comp[i] < 0 || comp[i] > 5) return fail(h, "fdtd_add_tfsf: bad component %d", comp[i]);
    if (cell[i] >= ncell) return fail(h, "fdtd_add_tfsf: cell index out of range");
    if (aux[i] < 0 || aux[i] > n_aux_max) return fail(h, "fdtd_add_tfsf: auxiliary index out of range");
    order[(size_t)i] = i;
  }
  std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) {
    return comp[x] != comp[y] ? comp[x] < comp[y] : cell[x] < cell[y];
  });
  std::vector<int32_t> tc, st, ax((size_t)n);
  std::vector<uint32_t> tl;
  std::vector<float> ww((size_t)n);
  for (int64_t e = 0; e < n; ++e) {
    const int64_t i = order[(size_t)e];
    if (e == 0 || comp[i] != tc.back() || cell[i] != tl.back()) { tc.push_back(comp[i]); tl.push_back(cell[i]); st.push_back((int32_t)e); }
    ww[(size_t)e] = w[i];
    ax[(size_t)e] = aux[i];
  }
  st.push_back((int32_t)n);
  L.n_targets = (long long)tc.size();
  plane_range(tl.data(), L.n_targets, h->g.sxy, &L.k0, &L.k1);
  if (dev_upload(h, &L.comp, (const int32_t*)tc.data(), tc.size()) || dev_upload(h, &L.cell, (const uint32_t*)tl.data(), tl.size()) ||
      dev_upload(h, &L.start, (const int32_t*)st.data(), st.size()) || dev_upload(h, &L.w, (const float*)ww.data(), ww.size()) ||
      dev_upload(h, &L.aux, (const int32_t*)ax.data(), ax.size()))
    return -1;
  return 0;
}
}  // namespace

int fdtd_add_tfsf(FdtdSolver* h, int n_aux, const float* ae, const float* be, const float* ah, const float* bh,
                  int src_cell, int64_t n_steps, const float* wave, int64_t n_e, const int32_t* e_comp,
                  const uint32_t* e_index, const float* e_w, const int32_t* e_aux, int64_t n_h,
                  const int32_t* h_comp, const uint32_t* h_index, const float* h_w, const int32_t* h_aux) {
  if (!h) return -1;
  if (n_aux < 4 || src_cell < 1 || src_cell >= n_aux) return fail(h, "fdtd_add_tfsf: bad auxiliary grid");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  Tfsf t{};
  t.n_aux = n_aux; t.src_cell = src_cell;
  t.n_steps = n_steps;
  if (dev_upload(h, &t.ae, ae, (size_t)n_aux + 1) || dev_upload(h, &t.be, be, (size_t)n_aux + 1) ||
      dev_upload(h, &t.ah, ah, (size_t)n_aux) || dev_upload(h, &t.bh, bh, (size_t)n_aux) ||
      dev_alloc(h, &t.e1, (size_t)n_aux + 1) || dev_alloc(h, &t.h1, (size_t)n_aux) ||
      dev_alloc(h, &t.e1c, (size_t)n_aux + 1) || dev_alloc(h, &t.h1c, (size_t)n_aux) ||
      dev_upload(h, &t.wave, wave, (size_t)n_steps))
    return -1;
  // E-side corrections read h1 (n_aux entries), H-side ones e1 (n_aux + 1 entries)
  if (build_tfsf_list(h, t.e, n_e, e_comp, e_index, e_w, e_aux, n_aux - 1) ||
      build_tfsf_list(h, t.h, n_h, h_comp, h_index, h_w, h_aux, n_aux))
    return -1;
  h->tfsf.push_back(t);
  return 0;
}

int fdtd_add_monitor(FdtdSolver* h, int kind, int n_comps, const int32_t* comps, const int32_t lo[3],
                     const int32_t hi[3], int64_t n_rec, const int64_t* steps, int nf, const float* phase_e,
                     const float* phase_h) {
  if (!h) return -1;
  if (kind != FDTD_MON_TIME && kind != FDTD_MON_DFT) return fail(h, "fdtd_add_monitor: bad kind %d", kind);
  const int N[3] = {h->g.nx, h->g.ny, h->g.nz};
  for (int a = 0; a < 3; ++a)
    if (lo[a] < 0 || hi[a] > N[a] || hi[a] <= lo[a]) return fail(h, "fdtd_add_monitor: box [%d,%d) outside axis %d of %d cells", lo[a], hi[a], a, N[a]);
  if (n_comps < 1 || n_comps > 6) return fail(h, "fdtd_add_monitor: n_comps must be 1..6");
  if (kind == FDTD_MON_DFT && nf < 1) return fail(h, "fdtd_add_monitor: a DFT monitor needs frequencies");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  Monitor m;
  m.kind = kind;
  m.comps.assign(comps, comps + n_comps);
  for (int c : m.comps) if (c < 0 || c > 5) return fail(h, "fdtd_add_monitor: bad component %d", c);
  m.box.lo0 = lo[0]; m.box.lo1 = lo[1]; m.box.lo2 = lo[2];
  m.box.nx = hi[0] - lo[0]; m.box.ny = hi[1] - lo[1]; m.box.nz = hi[2] - lo[2];
  m.cells = (long long)m.box.nx * m.box.ny * m.box.nz;
  m.steps.assign(steps, steps + n_rec);
  for (size_t i = 1; i < m.steps.size(); ++i)
    if (m.steps[i] <= m.steps[i - 1]) return fail(h, "fdtd_add_monitor: steps must be strictly increasing");
  m.nf = nf;
  if (kind == FDTD_MON_TIME) {
    m.data_bytes = (size_t)n_rec * n_comps * m.cells * sizeof(float);
    float* d = nullptr;
    if (dev_alloc(h, &d, (size_t)n_rec * n_comps * m.cells)) return -1;
    m.data = d;
  } else {
    m.data_bytes = (size_t)nf * n_comps * m.cells * sizeof(float2);
    float2* d = nullptr;
    if (dev_alloc(h, &d, (size_t)nf * n_comps * m.cells)) return -1;
    m.data = d;
    if (dev_upload(h, &m.phase_e, reinterpret_cast<const float2*>(phase_e), (size_t)n_rec * nf) ||
        dev_upload(h, &m.phase_h, reinterpret_cast<const float2*>(phase_h), (size_t)n_rec * nf))
      return -1;
  }
  h->mons.push_back(m);
  return (int)h->mons.size() - 1;
}

int fdtd_get_monitor(FdtdSolver* h, int id, void* host, size_t bytes) {
  if (!h) return -1;
  if (id < 0 || id >= (int)h->mons.size()) return fail(h, "fdtd_get_monitor: bad id %d", id);
  Monitor& m = h->mons[id];
  if (bytes != m.data_bytes) return fail(h, "fdtd_get_monitor: expected %zu bytes, got %zu", m.data_bytes, bytes);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(host, m.data, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int fdtd_set_field(FdtdSolver* h, int comp, const float* host, size_t bytes) {
  if (!h) return -1;
  if (comp < 0 || comp > 5) return fail(h, "fdtd_set_field: bad component");
  if (bytes != (size_t)n_cells(h) * 4) return fail(h, "fdtd_set_field: expected %zu bytes", (size_t)n_cells(h) * 4);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipMemcpy(field_ptr(h, comp), host, bytes, hipMemcpyHostToDevice));
  // the ADE recursion forms Q^{n+1} from E^{n+1} + E^n: E^n of its cells is the value just uploaded
  for (AdeGroup& a : h->ade)
    if (a.comp == comp) {
      hipLaunchKernelGGL(ade_gather_kernel, dim3(nblk(a.n)), dim3(256), 0, h->stream, (const float*)field_ptr(h, comp),
                         (const uint32_t*)a.cell, a.e_old, a.n);
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  // keep single-slab ghost planes consistent with the new interior
  if (h->comm == nullptr) { fill_ghost_h(h, h->stream); fill_ghost_e(h, h->stream); fill_ghost_fused(h, h->stream); HIPCHK(h, hipStreamSynchronize(h->stream)); }
  else if (comp == 0 || comp == 1 || comp == 3 || comp == 4) {
    // with a communicator the ghost planes come from the neighbour: do one exchange now
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (exchange(h, comp < 3, h->comm_stream)) return -1;
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
  }
  return 0;
}

int fdtd_get_field(FdtdSolver* h, int comp, float* host, size_t bytes) {
  if (!h) return -1;
  if (comp < 0 || comp > 5) return fail(h, "fdtd_get_field: bad component");
  if (bytes != (size_t)n_cells(h) * 4) return fail(h, "fdtd_get_field: expected %zu bytes", (size_t)n_cells(h) * 4);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(host, field_ptr(h, comp), bytes, hipMemcpyDeviceToHost));
  return 0;
}

int fdtd_set_shutoff(FdtdSolver* h, int every, double shutoff, int64_t ref_step) {
  if (!h) return -1;
  h->decay_every = every; h->shutoff = shutoff; h->decay_ref = ref_step;
  return 0;
}

int fdtd_comm_unique_id(char id[128]) {
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return fail(nullptr, "ncclGetUniqueId failed");
  std::memcpy(id, &u, 128);
  return 0;
}

int fdtd_comm_init(FdtdSolver* h, const char id[128], int rank, int n_ranks) {
  if (!h) return -1;
  if (rank < 0 || rank >= n_ranks) return fail(h, "fdtd_comm_init: bad rank %d of %d", rank, n_ranks);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  NCCLCHK(h, ncclCommInitRank(&h->comm, n_ranks, u, rank));
  h->rank = rank; h->n_ranks = n_ranks;
  // what the communicator itself reports goes into FdtdStats (bench.py --gpus N prints it: proof that RCCL saw N ranks)
  int cnt = 0, ur = -1;
  NCCLCHK(h, ncclCommCount(h->comm, &cnt));
  NCCLCHK(h, ncclCommUserRank(h->comm, &ur));
  h->stats.comm_ranks = cnt; h->stats.comm_rank = ur;
  return 0;
}

int fdtd_reset(FdtdSolver* h) {
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int c = 0; c < 6; ++c) HIPCHK(h, hipMemset(h->fbase[c], 0, h->field_bytes));
  for (int c = 0; c < 6; ++c) if (h->fbase2[c]) HIPCHK(h, hipMemset(h->fbase2[c], 0, h->field_bytes));
  for (int c = 0; c < 6; ++c) if (h->fbase3[c]) HIPCHK(h, hipMemset(h->fbase3[c], 0, h->field_bytes));
  for (int a = 0; a < 3; ++a) {
    PmlAxisDev& P = h->pml[a];
    if (P.n == 0) continue;
    for (int s = 0; s < 2; ++s) {
      if (P.psi_e[s]) HIPCHK(h, hipMemset(P.psi_e[s], 0, P.psi_count * 4));
      if (P.psi_h[s]) HIPCHK(h, hipMemset(P.psi_h[s], 0, (P.psi_count + P.psi_plane) * 4));
      if (P.psi_h2[s]) HIPCHK(h, hipMemset(P.psi_h2[s], 0, (P.psi_count + P.psi_plane) * 4));
      if (P.psi_e2[s]) HIPCHK(h, hipMemset(P.psi_e2[s], 0, P.psi_count * 4));
      if (P.psi_et[s]) HIPCHK(h, hipMemset(P.psi_et[s], 0, P.psi_count * 4));
      if (P.psi_ht[s]) HIPCHK(h, hipMemset(P.psi_ht[s], 0, (P.psi_count + P.psi_plane) * 4));
    }
  }
  for (AdeGroup& a : h->ade) {
    HIPCHK(h, hipMemset(a.e_old, 0, (size_t)a.n * 4));
    HIPCHK(h, hipMemset(a.q, 0, (size_t)a.n * a.p.n_poles * 8));
  }
  if (h->disp.state == 1) HIPCHK(h, hipMemset(h->disp.cs, 0, (size_t)h->disp.n_blocks * 3 * 256 * sizeof(float)));
  for (Tfsf& t : h->tfsf) {
    HIPCHK(h, hipMemset(t.e1, 0, ((size_t)t.n_aux + 1) * 4));
    HIPCHK(h, hipMemset(t.h1, 0, (size_t)t.n_aux * 4));
  }
  for (Monitor& m : h->mons) { HIPCHK(h, hipMemset(m.data, 0, m.data_bytes)); m.next = 0; }
  h->step = 0; h->energy_max = 0.0;
  h->stats.steps_done = 0; h->stats.diverged = 0; h->stats.stopped_early = 0; h->stats.field_decay = 1.0;
  return 0;
}

// ---- fdtd_run ---------------------------------------------------------------------------------------------------------------
// One `Run` object per call: its members are the state of the run (what the locals of the former 800-line function held), its
// methods the schedules — each with the streams it issues on and the field sets it reads / writes in its header.  Conventions:
//   st = main stream, cs = comm stream (the second stream of the engine; an alias of st when the two were found not to overlap);
//   set A = h->f (current: E^n, H^{n-1/2}), set B = h->f2 (the other set of the ping-pong), set T = h->f3 (third set: the middle
//   step of shell / slab pairs, round-4 form); "swap" = swap_sets: B becomes current.
//   Cross-stream edges are hipEvents recorded on the producer and waited for on the consumer; nothing waits on the host inside
//   the loop except field-decay checks.  FDTD_OPT_DEBUG_SYNC = 1 puts a device-wide synchronisation behind every launch
//   (time_end) and every step: a schedule whose result then differs from the normal run has a missing edge
//   (tests/test_gpu_parity.py::test_schedules_do_not_depend_on_stream_timing).
}  // extern "C"
namespace {
struct GraphRec { const float* set; int parity; hipGraphExec_t exec; };
struct Run {
  FdtdSolver* h;
  int64_t n_steps;
  FdtdProgressFn progress;
  void* user;
  // the run's configuration (setup)
  bool multi = false, nb_lo = false, nb_hi = false;
  int nz = 0;
  hipStream_t st = nullptr, cs = nullptr;
  bool fused_ok = false, fused = false, fused_multi = false;
  int b_lo = 0, b_hi = 0;              // fused z-slab schedule: boundary planes next to the lower / upper neighbour
  bool primed = false;                 // fused z-slab schedule: monitors pre-recorded, H-side pre-corrections applied, ghost planes in flight
  int pml_in_m = 0;                    // z-slab ranks: axes whose CPML recursions run inside the sweeps
  bool psi_ghosts = false;
  int tb_req = 0;
  bool tb_two_streams = false, tb_ok = false;
  std::vector<hipEvent_t> tb_ev;       // [2 s] = A(s) done, [2 s + 1] = B(s) done (two-stream mode)
  bool split_now = false, graph_ok = false;
  std::vector<GraphRec> graphs;
  bool f2_ok = false, f2s_ok = false, s2_ok = false, s2_deep = false, f2m_ok = false;
  bool s2_disp = false;                // shell2 pairs: every dispersive cell deep inside the bulk — its sweep advances them (no z holes)
  bool pair_disp = false;              // the pair about to be issued does so
  bool spg_ok = false;                 // lists the node table cannot hold go out as paged source terms while they inject (spg_setup)
  bool pair_spg = false;               // the pair about to be issued carries them
  bool f2mc_ok = false, f2mc_deep = false;   // z-slab ranks with CPML: shell2 pairs with the planes next to a cut as z holes (sgm: their geometry)
  ShellGeom sg{}, sgm{};
  ZPlan zp_base, zp_src;               // the bulk's planes: without / with the z holes of the source lists
  ZPlan zp_s2, zp_s2h;                 // shell2 pairs: one interval [o0z, o1z) / the intervals between the z holes of the source lists (ok: usable)
  F2Plan f2_plan;
  int64_t done = 0;
  // the step being issued (begin_step)
  long long n = 0;
  bool rec = false, src_alive = false, pair = false, use_s2 = false;
  int src_why = 0;
  const ZPlan* zp = nullptr;

  // a debugging aid (FDTD_OPT_DEBUG_SYNC): everything issued so far has finished before anything else is issued
  void sync_point() { if (h->debug_sync) (void)hipDeviceSynchronize(); }
  // checks, streams, the variant (fused / two-pass, one GPU / z-slab rank), tile-shape and placement probes (both on st)
  int setup() {
    multi = h->comm != nullptr;     // also true for a 1-rank communicator (self exchange)
    nb_lo = h->cfg.bc[4] == FDTD_BC_NEIGHBOR, nb_hi = h->cfg.bc[5] == FDTD_BC_NEIGHBOR;
    if ((nb_lo || nb_hi) && !multi) return fail(h, "fdtd_run: neighbour faces need fdtd_comm_init");
    if (multi && !h->aniso.empty()) return fail(h, "fdtd_run: fully anisotropic media are not available on z-slabs");
    // (PMC on a plus face of a z-slab rank: x / y walls are local to every plane; a z wall belongs to the rank without an upper
    //  neighbour, whose interior launch must hold the wall's two image planes and the two they mirror)
    // (six planes: the boundary chunk next to the lower neighbour is one plane thick below eight planes, two from there on)
    if (multi && h->mirror_wall[2] >= 0 && (nb_hi || h->mirror_wall[2] != h->g.nz - 2 || h->g.nz < 6))
      return fail(h, "fdtd_run: a PMC plus face along z needs the last z-slab to hold the wall and at least 6 planes");
    nz = h->g.nz;
    // runs that use BOTH streams first make sure the two really overlap (once per engine; falls back to one stream)
    if ((multi || any_pml(h) || h->tblock > 4096) && probe_stream_overlap(h)) return -1;
    st = h->stream, cs = h->comm_stream;
    for (hipEvent_t e : h->kev) hipEventDestroy(e);
    h->kev.clear(); h->kev_kind.clear();
    h->stats.stopped_early = 0;
    HIPCHK(h, hipEventRecord(h->ev0, st));
    if (multi && (nb_lo || nb_hi) && nz < 2) return fail(h, "fdtd_run: a z-slab needs at least 2 planes");
    // the fused sweep is the default single-GPU path whenever rows are float4-aligned
    // (a wide material table — more than 1023 media — has no LDS copy and no packed 10-bit words: the two-pass kernels take it)
  if (h->mat4b && h->cfg.variant == FDTD_VARIANT_FUSED)
    return fail(h, "fdtd_run: more than %d media need the two-pass kernels (FDTD_VARIANT_ZMARCH / AUTO), not FDTD_VARIANT_FUSED", kMaxMedia - 1);
  fused_ok = !h->mat4b && (h->g.nx % 4 == 0) && h->rows_f <= 15 &&
                          (h->cfg.variant == FDTD_VARIANT_FUSED || h->cfg.variant == FDTD_VARIANT_AUTO);
    fused = !multi && fused_ok;
    // with a communicator every rank must take the same path: the fused z-slab schedule runs only on
    // explicit request (the host decides for all ranks, tidy3d_amd/engine.py), AUTO = two-pass
    if (multi && h->cfg.variant == FDTD_VARIANT_FUSED && !(fused_ok && nz >= 4))
      return fail(h, "fdtd_run: the fused z-slab schedule needs nx %% 4 == 0 and >= 4 planes per slab");
    fused_multi = multi && h->cfg.variant == FDTD_VARIANT_FUSED;
    // ---- pipelined fused z-slab schedule (fused_multi) -------------------------------------------
    // Per step, with  b_lo / b_hi  boundary planes next to a neighbour face:
    //   cs: sweep [0,b_lo) + [nz-b_hi,nz)  -> E-side corrections and next step's H-side pre-corrections
    //       of those planes -> ONE exchange (exchange_fused_all), which overlaps the interior sweep
    //   st: sweep [b_lo, nz-b_hi)          -> the same corrections of the interior planes
    // The next boundary sweep needs this exchange and the interior planes next to it (ev_e_int); the
    // next interior sweep needs only the boundary planes next to it (ev_e_bnd, recorded BEFORE the
    // exchange).  Invariant at the top of a step ("primed"): monitors pre-recorded, H-side
    // pre-corrections applied on all planes, ghost planes in flight on cs.  Steps that record
    // monitors, check the field decay or end the run use a joined tail on st instead and re-prime.
    b_lo = 0, b_hi = 0;
    if (fused_multi) {
      // boundary chunk: TWO planes per neighbour face (all the exchange needs, and it starts that much earlier).
      // Measured inside engines on the per-rank proxy, RCCL looped back (profiles/r03y_probe_boundary_chunk_thickness
      // .jsonl): 512 x 512 x 64 plain 0.186 ms per step at 16 planes, 0.175 at 8, 0.167 at 4, 0.163 at 2 and at 1; with
      // materials + CPML 0.319 -> 0.287; 128 planes 0.301 -> 0.297.  (Round 1's kernels preferred 16: r01h.)
      int zb = h->bnd_planes > 0 ? h->bnd_planes : std::min(kBndPlanes, nz / 4);
      zb = std::max(1, std::min(zb, nz / 2));
      b_lo = nb_lo ? zb : 0;
      b_hi = nb_hi ? zb : 0;
      // the z-CPML differentiates along z: its slabs must stay clear of the boundary chunks (whose
      // corrections run on the other stream and before the ghost planes of the new step arrive)
      const PmlAxisDev& pz = h->pml[2];
      if (nb_hi && pz.n_lo > 0) b_hi = std::min(b_hi, nz - pz.n_lo - 1);
      if (nb_lo && pz.n_hi > 0) b_lo = std::min(b_lo, nz - pz.n_hi - 1);
      if ((nb_hi && b_hi < 1) || (nb_lo && b_lo < 1))
        return fail(h, "fdtd_run: the fused z-slab schedule needs at least 2 planes between a slab cut and the z-PML");
    }
    // Tile-shape probing: on request (FDTD_OPT_AUTOTUNE), and by default on one GPU when the default shape
    // launches less than one wave of workgroups (256 CUs x 3): there the z-chunk decides how much of the chip a
    // sweep fills (128^3: 344 workgroups at 16 planes per chunk, 0.045 ms per step; 688 at 8 planes, 0.034 ms —
    // profiles/r01m_narrow_grid_axis_shift.log) and the probe costs a dozen sweeps once.  Results do not depend
    // on the shape.  (autotune == 2 lifts the size threshold: test aid for the emulated library)
    bool under_one_wave = false;
    if (fused && !h->tuned && !h->user_geometry) {
      const long long wgs = (long long)((h->g.nx + 255) / 256) * ((h->g.ny + h->rows_f - 1) / h->rows_f) *
                            ((nz + h->zchunk_f - 1) / h->zchunk_f);
      under_one_wave = wgs < 2048;       // (two waves of workgroups at 4 waves per SIMD)
    }
    if ((fused || fused_multi) && (h->autotune || under_one_wave) && !h->tuned && !h->user_geometry &&
        (n_cells(h) >= (1LL << 20) || h->autotune == 2)) {
      if (autotune_fused(h, st)) return -1;
      if (fused_multi) {           // the boundary-chunk thickness follows the chosen z-chunk
        int zb = h->bnd_planes > 0 ? h->bnd_planes : std::min(kBndPlanes, nz / 4);
        zb = std::max(1, std::min(zb, nz / 2));
        const PmlAxisDev& pz = h->pml[2];
        b_lo = nb_lo ? zb : 0;
        b_hi = nb_hi ? zb : 0;
        if (nb_hi && pz.n_lo > 0) b_hi = std::min(b_hi, nz - pz.n_lo - 1);
        if (nb_lo && pz.n_hi > 0) b_lo = std::min(b_lo, nz - pz.n_hi - 1);
      }
    }
    // (a rank of a z-slab run samples its own slab; nothing is exchanged while it does.  >= 100: any size — test aid)
    if ((fused || fused_multi) && !h->placement_done && (h->placement_tries % 100) > 0 &&
        (n_cells(h) >= (fused ? (1LL << 24) : (1LL << 22)) || h->placement_tries >= 100)) {
      const int tries = h->placement_tries;
      h->placement_tries = tries % 100;
      const int prc = probe_placement(h, st);
      h->placement_tries = tries;
      if (prc) return -1;
    }
    primed = false;
    // z-slab ranks carry the CPML recursions inside their sweeps as one GPU does (same arithmetic and summation order):
    // the x / y recursions are local in z, the z recursion stays two planes clear of the cuts, and the one thing a rank
    // lacks — the H-side psi of its ghost plane -1, for the chunk prologue at plane 0 — comes with the ghost planes
    // (exchange_fused_all).  pml_in_m: axes inside the sweep; bits 0 / 1 agree on all ranks, bit 2 only end ranks have.
    // ON REQUEST only (FDTD_OPT_PML_FUSED > 0 on every rank): measured inside engines on the per-rank proxy (profiles/
    // r04p, r04q: 512 x 512 slabs with CPML on x and y, exchange included) the slab kernels win on thin slabs — 64 planes
    // 0.304 vs 0.352 ms, 128 planes 0.565 vs 0.594 — and tie at 256 (1.069 vs 1.067): the interior goes out as three
    // partial launches on one stream there, and the all-axes instantiation runs its few tiles at 2 waves per SIMD.
    pml_in_m = 0;
    if (fused_multi && any_pml(h) && h->pml_fused > 0 && 64 * (h->rows_f + 1) <= 512)
      pml_in_m = h->pml_fused & pml_in_sweep_mask(h);
    psi_ghosts = fused_multi && (pml_in_m & 3) != 0;
    return 0;
  }
  void e_post(long long n, int k0, int k1, hipStream_t s, bool replica) {
    launch_pml(h, true, k0, k1, s, 7 & ~pml_in_m);
    launch_sources(h, true, n, k0, k1, s, replica);
    launch_damp(h, true, k0, k1, s);       // before the ADE pass: its stored E^{n+1} is the damped one
    launch_ade(h, k0, k1, s);
  }
  void h_pre(long long n, int k0, int k1, hipStream_t s, bool replica) {
    fill_mirror(h, s, k0, k1);             // (E^n and H^{n-1/2} of these planes are complete: the images beyond PMC plus walls first)
    launch_damp(h, false, k0, k1, s);
    launch_sources(h, false, n, k0, k1, s, replica);
    launch_pml(h, false, k0, k1, s, 7 & ~pml_in_m);
  }
  bool rec_at(long long n) {
    for (Monitor& m : h->mons) if (m.next < m.steps.size() && m.steps[m.next] == n) return true;
    return false;
  }
  // all planes on st: monitors of step n, H-side pre-corrections, then the exchange on cs
  int prime(long long n) {
    if (rec_at(n)) record_monitors(h, n, false, st);
    h_pre(n, 0, nz, st, false);
    advance_tfsf_aux(h, false, n, st, false);
    advance_tfsf_aux(h, false, n, st, true);
    HIPCHK(h, hipEventRecord(h->ev_e_int, st));
    HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
    HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
    if (exchange_fused_all(h, cs, psi_ghosts)) return -1;
    primed = true;
    return 0;
  }
  // the schedules a run may use besides single steps: the fused z-slab pipeline, the slab-interleaved two-step schedule, captured step pairs
  int setup_schedules() {
    if (fused_multi) {
      // the comm-stream replica of the 1-D incident grids starts from the main one
      for (Tfsf& t : h->tfsf) {
        HIPCHK(h, hipMemcpyAsync(t.e1c, t.e1, ((size_t)t.n_aux + 1) * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(t.h1c, t.h1, (size_t)t.n_aux * 4, hipMemcpyDeviceToDevice, st));
      }
      if (ensure_second_set(h)) return -1;
    }
    // Two-stream schedule of one step (st = main stream, cs = comm stream):
    //   cs: [H top plane] -> send/recv H -> [E bottom plane] -> send/recv E      (boundary planes first)
    //   st: [H interior ] ----------------> [E interior    ]
    // Cross-stream edges (RAW and WAR), one event each:
    //   ev_e_int : E interior of step n-1 done     -> cs may update/ship H top plane (reads E[nz-1])
    //   ev_e_bnd : E plane 0 of step n-1 done (cs) -> st may run H interior (reads E[0]) and monitors
    //   ev_h_int : H interior done                 -> cs may update E plane 0 (reads H[0])
    //   ev_h_bnd : H top plane done (cs)           -> st may run E interior (reads H[nz-1])
    // Ghost planes are only touched on cs, in stream order.  No host synchronisation in the loop.
    // ---- slab-interleaved two-step schedule (one GPU, fused sweep; FDTD_OPT_TBLOCK) ---------------------------------
    // Two time steps per pass over the grid, slab by slab of T planes:  A(s) = step n on slab s (set a -> set b),
    // B(s) = step n+1 on slab s (b -> a, IN PLACE of what A read), issued  A(0) A(1) B(0) A(2) B(1) ... : B(s) follows
    // A(s+1) because its top plane differentiates E^{n+1} of slab s+1's first plane, and it must not overwrite a's slab s
    // before A(s+1)'s chunk prologue has read its top plane.  What B(s) reads was written two launches earlier — 2 T planes
    // x 6 arrays, within the 256 MiB Infinity Cache for T <= 16 at 512^2 cells per plane — so per step pair the arrays
    // cross the HBM interface about three times (read a, write b, write a) instead of four.  Every correction launch takes
    // a plane range already (the z-slab schedule uses them the same way): H-side pre-corrections of a slab go out in front
    // of its sweep, E-side ones behind it.  The same kernels, the same arithmetic on the same values: bit-identical to
    // single steps (tests/test_emu_fused.py, tests/test_gpu_production_path.py).  Not with CPML or TFSF (their state is
    // advanced per whole step), not across a periodic z (the ghost planes wrap around the slab order), and only for step
    // pairs in which no monitor records and no field-decay check falls on the middle step.
    tb_req = h->tblock < 0 ? 0 : (h->tblock % 4096);
    tb_two_streams = h->tblock > 4096 && h->stream_overlap == 1;
    tb_ok = fused && tb_req > 0 && !any_pml(h) && h->tfsf.empty() && h->cfg.bc[4] != FDTD_BC_PERIODIC &&
                       h->mirror_wall[0] < 0 && h->mirror_wall[1] < 0 && h->mirror_wall[2] < 0 && h->aniso.empty() &&
                       nz >= 2 * tb_req;
    h->two_step_pairs = 0;
    h->tblock_used = tb_ok ? tb_req : 0;
    // ---- captured step pairs (hipGraph) ------------------------------------------------------------------------------
    // Small grids are bound by dependent launches (64^3: three launches, 31 us per step; profiles/r02h): a run of steps
    // without monitor records or decay checks is captured ONCE as a graph of two steps (set a -> b -> a, psi parity back)
    // and replayed.  A graph bakes its kernel arguments, so the source kernels of a captured launch read the step counter
    // from device memory (step_dev + offset; the graph's last node advances it by two).  Same launches, same order, same
    // arguments otherwise: bit-identical to direct launches (tests/test_gpu_production_path.py).  One stream only: not
    // with the three-launch CPML split of large grids, not on z-slabs, not with per-launch timing events.
    split_now = (h->pml_split < 0 ? n_cells(h) >= (1LL << 24) : h->pml_split != 0) && any_pml(h) &&
                           (((h->pml_fused < 0 ? 7 : h->pml_fused) & pml_in_sweep_mask(h)) & 6) != 0;
    graph_ok = fused && !tb_ok && !split_now && !(h->cfg.flags & FDTD_FLAG_TIME_KERNELS) && h->aniso.empty() &&
                    h->use_graph > 0;      // on request only: measured on ROCm 7.2 (profiles/r3i) a replayed pair is ~3 us per step
                                           // SLOWER than launching its kernels (64^3 17.6 -> 20.8, 128^3 28.8 -> 31.4, 200^3 81.5 -> 84.0)
    h->graph_pairs = 0;
    return 0;
  }
  int tb_pair(long long n) {
    const int T = tb_req, S = (nz + T - 1) / T;
    if (ensure_second_set(h)) return -1;
    if (tb_two_streams && tb_ev.empty()) {
      tb_ev.resize((size_t)2 * S);
      for (hipEvent_t& e : tb_ev) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto zs = [&](int s) { return std::min(nz, s * T); };
    auto stage_a = [&](int s) -> int {                     // h->f = a
      hipStream_t q = st;
      if (tb_two_streams && s >= 3) HIPCHK(h, hipStreamWaitEvent(q, tb_ev[(size_t)2 * (s - 3) + 1], 0));   // stay <= 3 slabs ahead of B
      const int k0 = zs(s), k1 = zs(s + 1);
      launch_damp(h, false, k0, k1, q);
      launch_sources(h, false, n, k0, k1, q);
      if (launch_fused_range(h, k0, k1, q)) return -1;
      swap_sets(h);                                        // h->f = b: the E-side corrections of step n act on E^{n+1}
      launch_sources(h, true, n, k0, k1, q);
      launch_damp(h, true, k0, k1, q);
      launch_ade(h, k0, k1, q);
      swap_sets(h);
      if (tb_two_streams) HIPCHK(h, hipEventRecord(tb_ev[(size_t)2 * s], q));
      return 0;
    };
    auto stage_b = [&](int s) -> int {
      hipStream_t q = tb_two_streams ? cs : st;
      if (tb_two_streams) HIPCHK(h, hipStreamWaitEvent(q, tb_ev[(size_t)2 * std::min(s + 1, S - 1)], 0));
      const int k0 = zs(s), k1 = zs(s + 1);
      swap_sets(h);                                        // h->f = b (E^{n+1}, H^{n+1/2}), h->f2 = a
      launch_damp(h, false, k0, k1, q);
      launch_sources(h, false, n + 1, k0, k1, q);
      const int rc = launch_fused_range(h, k0, k1, q);
      swap_sets(h);                                        // h->f = a again: slab s now holds E^{n+2}, H^{n+3/2}
      if (rc) return -1;
      launch_sources(h, true, n + 1, k0, k1, q);
      launch_damp(h, true, k0, k1, q);
      launch_ade(h, k0, k1, q);
      if (tb_two_streams) HIPCHK(h, hipEventRecord(tb_ev[(size_t)2 * s + 1], q));
      return 0;
    };
    if (stage_a(0)) return -1;
    for (int s = 1; s < S; ++s) {
      if (stage_a(s)) return -1;
      if (stage_b(s - 1)) return -1;
    }
    if (stage_b(S - 1)) return -1;
    if (tb_two_streams) HIPCHK(h, hipStreamWaitEvent(st, tb_ev[(size_t)2 * (S - 1) + 1], 0));
    h->two_step_pairs++;
    return 0;
  }
  // one step of the one-GPU fused path (n = its time step; rec_post: monitors record behind the sweep)
  int fused_one(long long n, bool rec_post) {
    // H-side corrections are additive: pre-apply them to H^{n-1/2}; E-side ones follow the sweep.
    // With pml_in the CPML recursions run inside the sweep (same arithmetic, no slab kernels).
    // pml_in = axes whose recursions run inside the sweep (default: all that have layers; FDTD_OPT_PML_FUSED
    // = 0 keeps the slab kernels, any other mask selects axes).  Inside the sweep the field values a slab
    // kernel would re-read and re-write stay in registers: only psi moves (32 B per cell and axis membership).
    int pml_in = 0;
    if (any_pml(h) && 64 * (h->rows_f + 1) <= 512)
      pml_in = (h->pml_fused < 0 ? 7 : h->pml_fused) & pml_in_sweep_mask(h);
    // (periodic z: the wrapped copies in the ghost planes were taken at the end of the last step, in front of this refresh —
    //  their images beyond an x / y wall are refreshed with the planes they copy; a z-slab rank receives its ghost planes
    //  refreshed by their owner)
    fill_mirror(h, st, h->cfg.bc[4] == FDTD_BC_PERIODIC ? -1 : 0, h->cfg.bc[4] == FDTD_BC_PERIODIC ? nz + 1 : nz);
    aniso_save(h, st);                     // (E^n of the nodes around fully anisotropic cells: the sweep's read set is this set)
    launch_damp(h, false, 0, nz, st);
    launch_sources(h, false, n, 0, nz, st);
    launch_pml(h, false, 0, nz, st, 7 & ~pml_in);
    advance_tfsf_aux(h, false, n, st);
    if (h->cfg.bc[4] == FDTD_BC_PERIODIC) fill_ghost_h(h, st);   // ghost(-1) must carry the pre-corrections too
    // small grids are bound by dependent launches, not by occupancy: one launch of the all-axes instantiation
    const bool split = h->pml_split < 0 ? n_cells(h) >= (1LL << 24) : h->pml_split != 0;
    if ((pml_in & 6) == 0 || !split) {
      if (launch_fused(h, st, pml_in)) return -1;
    } else {
      // The instantiation that carries the y / z recursions holds their psi values in registers from the
      // top of a plane (occupancy 2-3); the one most tiles need carries x only.  Three launches over
      // disjoint tiles, the two small ones on the second stream, concurrent with the big one:
      //   (1) planes of the z slabs (+1 plane: the next chunk's prologue must not see a slab), all rows   [x y z]
      //   (2) planes in between: bottom and top tile rows (a row or the halo row in a y slab)              [x y]
      //   (3) planes in between, middle tile rows                                                         [x]
      const int R = h->rows_f, nby_all = (h->g.ny + R - 1) / R;
      const PmlAxisDev &py = h->pml[1], &pz = h->pml[2];
      const bool in_y = (pml_in & 2) && py.ns > 0, in_z = (pml_in & 4) && pz.ns > 0;
      const int za = (in_z && pz.lo > 0) ? std::min(nz, pz.lo + 1) : 0;
      const int zc = (in_z && pz.hi0 < nz) ? std::max(za, pz.hi0) : nz;
      const int ty_a = (in_y && py.lo > 0) ? std::min(nby_all, py.lo / R + 1) : 0;
      const int ty_c = (in_y && py.hi0 < h->g.ny) ? std::max(ty_a, py.hi0 / R) : nby_all;
      HIPCHK(h, hipEventRecord(h->ev_h_int, st));
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_h_int, 0));
      if (launch_fused_range(h, 0, za, cs, pml_in, zc, nz, -1, 0, 0, true)) return -1;
      if (launch_fused_range(h, za, zc, cs, pml_in & 3, 0, 0, ty_a + (nby_all - ty_c), ty_a, ty_c - ty_a, true)) return -1;
      HIPCHK(h, hipEventRecord(h->ev_h_bnd, cs));
      if (launch_fused_range(h, za, zc, st, pml_in & 1, 0, 0, ty_c - ty_a, 0, ty_a)) return -1;
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
      swap_sets(h);
      swap_psi_h(h, pml_in);
    }
    if (rec_post) record_monitors(h, n, true, st);
    launch_pml(h, true, 0, nz, st, 7 & ~pml_in);
    launch_sources(h, true, n, 0, nz, st);
    aniso_apply(h, st);
    launch_damp(h, true, 0, nz, st);
    launch_ade(h, 0, nz, st);
    advance_tfsf_aux(h, true, n, st);
    fill_ghost_fused(h, st);
    return 0;
  }
  bool sources_alive(long long n) {
    for (const PointSrc& s : h->psrc) if (n >= s.n_steps) return false;
    for (const Tfsf& t : h->tfsf) if (n >= t.n_steps) return false;
    return true;
  }
  // 0 = the pair (n, n + 1) was replayed; 1 = capture not available (caller launches directly); < 0 = error
  int graph_pair(long long n) {
    hipGraphExec_t exec = nullptr;
    for (const GraphRec& r : graphs) if (r.set == h->f.ex && r.parity == (h->pml_parity | (h->pml_e_parity << 1))) exec = r.exec;
    if (!exec) {
      // everything a captured launch may allocate or upload must exist before the capture starts
      if (ensure_second_set(h)) return -1;
      if (any_pml(h) && 64 * (h->rows_f + 1) <= 512) {
        const int pml_in = (h->pml_fused < 0 ? 7 : h->pml_fused) & pml_in_sweep_mask(h);
        if (pml_in && ensure_pml_blocks(h, pml_in)) return -1;
      }
      if (!h->step_dev && dev_alloc(h, &h->step_dev, 1)) return -1;
      const float* set0 = h->f.ex;
      const int par0 = h->pml_parity | (h->pml_e_parity << 1);
      const hipError_t eb = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      if (eb != hipSuccess) { (void)hipGetLastError(); graph_ok = false; h->graph_status = -(100 + (int)eb % 100); return 1; }
      h->step_dev_mode = true;
      int rc = 0;
      for (int q = 0; q < 2 && !rc; ++q) { h->step_dev_off = q; rc = fused_one(n + q, false); }
      h->step_dev_mode = false;
      if (!rc) hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(1), 0, st, h->step_dev, 2LL, 1);
      hipGraph_t graph = nullptr;
      const hipError_t ee = hipStreamEndCapture(st, &graph);
      if (rc) { if (graph) hipGraphDestroy(graph); return -1; }
      hipError_t ei = hipSuccess;
      if (ee != hipSuccess || !graph || (ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) {
        h->graph_status = ee != hipSuccess ? -(200 + (int)ee % 100) : -(300 + (int)ei % 100);
        (void)hipGetLastError();
        if (graph) hipGraphDestroy(graph);
        graph_ok = false;
        return h->f.ex == set0 && (h->pml_parity | (h->pml_e_parity << 1)) == par0 ? 1 : fail(h, "fdtd_run: graph capture failed half way");
      }
      hipGraphDestroy(graph);
      graphs.push_back({set0, par0, exec});
      h->graph_status = 1;
    }
    if (h->step_dev_value != n) hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(1), 0, st, h->step_dev, n, 0);
    if (hipGraphLaunch(exec, st) != hipSuccess) return fail(h, "hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
    h->step_dev_value = n + 2;
    h->graph_pairs++;
    return 0;
  }
  // which forms of step pairs this run may take (decided once; begin_step judges every pair): plain pairs, shell pairs (round-4
  // form), shell2 pairs, slab pairs of a z-slab rank — and what they need (source tables, the third field set, two streams)
  int setup_pairs() {
    // dispersive cells inside the two-step sweeps (round 6): their pole states move into paged storage, once
    if (fused && !tb_ok && !h->ade.empty() && disp_setup(h)) return -1;
    h->disp.pairs = 0;
    s2_disp = false;
    // paged source terms (round 6): a TFSF box, a mode plane, a current sheet, any list of more than kMaxInj nodes
    spg_ok = false;
    h->spg.pairs = 0;
    if (fused && !tb_ok && !multi && (!h->tfsf.empty() || !h->psrc.empty())) {
      if (fused2_sources(h)) return -1;            // (src_nodes, the seam flags)
      bool needs = !h->tfsf.empty() || h->src_nodes > kMaxInj || h->src_h_on_seam || (h->src_h_nodes > 0 && !h->src_tab);
      if (!needs) {                                // lists of different lengths: mixed alive / spent pairs
        long long len = -1;
        for (const PointSrc& s : h->psrc) if (s.n_e || s.n_h) { if (len >= 0 && s.n_steps != len) needs = true; len = s.n_steps; }
      }
      if (needs && spg_setup(h)) return -1;
      // (absorber layers damp H^{n-1/2} inside the sweep, behind the H-side terms of step n that precede it: FDTD_F2_OFF_H_SOURCE_ABSORBER stays)
      spg_ok = needs && h->spg.state == 1 && !(h->has_damp && h->spg.any_h);
    }
    f2_ok = fused && !tb_ok && fused2_eligible(h);
    sg = ShellGeom{};
    f2s_ok = false;
    h->f2_off_reason = !fused ? FDTD_F2_OFF_VARIANT : (tb_ok ? FDTD_F2_OFF_DISABLED : 0);
    if (fused && !tb_ok && !f2_ok) {
      const bool shell = any_pml(h) || any_periodic(h) || !h->ade.empty();
      h->f2_off_reason = shell ? shell_why_not(h, &sg, &zp_base, &zp_src) : fused2_why_not(h);
      f2s_ok = shell && h->f2_off_reason == 0;
    }
    // shell2 pairs: the shell by shell2_step_kernel (two steps per sweep, psi carried) instead of two single steps
    s2_ok = false, s2_deep = false;
    if (fused && !tb_ok && !f2_ok && any_pml(h)) {
      ShellGeom g2{};
      const int why2 = shell2_why_not(h, &g2);
      if (why2 == 0) {
        const int why_r4 = h->f2_off_reason;
        s2_ok = true; sg = g2;                               // (the same geometry shell_why_not finds)
        s2_deep = shell2_sources_deep(h, sg);
        h->f2_off_reason = 0;
        // the bulk's planes: one interval, or — dispersive cells — the intervals between their planes (z holes, inside the bulk's range)
        zp_s2 = ZPlan{};
        // (round 6: the pair advances the dispersive cells itself — the bulk sweep and the shell's boxes subtract their paged memory
        //  terms, ade2_kernel follows; not beside the single steps of a periodic y's wrap rows, not with the measuring-aid box cuts)
        s2_disp = !h->ade.empty() && h->disp.state == 1 && h->cfg.bc[2] != FDTD_BC_PERIODIC && h->shell2_on != 2 && h->shell2_on != 3;
        if (h->ade.empty() || s2_disp) { zp_s2.n = 1; zp_s2.a[0] = sg.o0[2]; zp_s2.b[0] = sg.o1[2]; zp_s2.ok = true; }
        else if (zplan_build(h, sg, false, &zp_s2) && zp_s2.a[0] == sg.o0[2] && zp_s2.b[zp_s2.n - 1] == sg.o1[2]) zp_s2.ok = true;
        if (!zp_s2.ok) { s2_ok = false; s2_disp = false; h->f2_off_reason = why_r4; }       // (the round-4 form may still take the run)
        // lists that inject and that the sweeps cannot apply: their planes as z holes — usable when every hole lies inside the bulk's plane
        // range (one-launch form only: FDTD_OPT_SHELL2 = 2 / 3 cut their boxes differently)
        zp_s2h = ZPlan{};
        if (h->shell2_on != 2 && h->shell2_on != 3 && zplan_build(h, sg, true, &zp_s2h) && zp_s2h.n > 1 && zp_s2h.a[0] == sg.o0[2] &&
            zp_s2h.b[zp_s2h.n - 1] == sg.o1[2]) zp_s2h.ok = true;
      }
    }
    h->f2_dyn_reason = 0;
    if (f2_ok || f2s_ok || s2_ok) {
      if (fused2_sources(h)) return -1;
    }
    // (the third field set: + 50 % field memory.  Where it does not fit, the run keeps single steps instead of failing)
    if (f2s_ok && ensure_third_set(h)) {
      (void)hipGetLastError();
      h->err.clear();
      f2s_ok = false;
      h->f2_off_reason = FDTD_F2_OFF_MEMORY;
    }
    if ((f2s_ok || s2_ok) && probe_stream_overlap(h)) return -1;
    // z-slab ranks (pipelined schedule): step pairs with the planes next to the neighbour faces as the shell
    f2m_ok = fused_multi && !any_pml(h) && !h->has_damp && h->shell_on != 0 && nz >= 8 && fused2_why_not(h, true) == 0;
    if (fused_multi) h->f2_off_reason = f2m_ok ? 0 : (any_pml(h) || h->has_damp ? FDTD_F2_OFF_COMM : (fused2_why_not(h, true) ? fused2_why_not(h, true) : FDTD_F2_OFF_COMM));
    if (f2m_ok) {
      if (fused2_sources(h)) return -1;
      if (ensure_third_set(h)) { (void)hipGetLastError(); h->err.clear(); f2m_ok = false; h->f2_off_reason = FDTD_F2_OFF_MEMORY; }
    }
    // z-slab ranks that carry CPML inside their sweeps (FDTD_OPT_PML_FUSED on every rank: tidy3d_amd/dist.py asks for it where the
    // whole problem allows it): shell2 pairs — bulk and boxes as on one GPU, over the planes two or more away from a cut; the two
    // planes next to a cut take two single steps as a z hole and ship their planes after each (slab_shell2_pair)
    f2mc_ok = false;
    if (fused_multi && any_pml(h) && pml_in_m != 0 && pml_in_m == pml_in_sweep_mask(h) && !h->has_damp && h->shell_on != 0 &&
        h->shell2_on != 0 && h->shell2_on != 2 && h->shell2_on != 3 &&      // (2 / 3: boxes cut by axes ignore the z range the cut planes' hole leaves them)
        h->ade.empty() && !any_periodic(h) && (long long)h->g.sxy * 4 < (1LL << 32)) {
      int why = fused2_why_not(h, true, true);
      if (!why && !shell_geometry(h, &sgm)) why = FDTD_F2_OFF_PML;
      if (!why) {
        const PmlAxisDev& pz = h->pml[2];
        if (nb_lo) sgm.o0[2] = std::max(sgm.o0[2], 2);
        if (nb_hi) sgm.o1[2] = std::min(sgm.o1[2], nz - 2);
        // (the z recursion stays clear of the holes' planes and of what their first step reads)
        if (sgm.o1[2] - sgm.o0[2] < 8 || (pz.ns > 0 && ((nb_lo && pz.lo > 0) || (nb_hi && pz.hi0 < nz)))) why = FDTD_F2_OFF_TOO_SMALL;
      }
      if (!why && (fused2_sources(h) || ensure_third_set(h) || ensure_pml_blocks2(h) || ensure_pml_blocks_hole(h))) {
        (void)hipGetLastError();
        h->err.clear();
        why = FDTD_F2_OFF_MEMORY;
      }
      f2mc_ok = why == 0;
      f2mc_deep = f2mc_ok && shell2_sources_deep(h, sgm);
      h->f2_off_reason = why;
    }
    h->fused2_pairs = 0;
    h->shell_pairs = 0;
    h->shell2_pairs = 0;
    done = 0;
    return 0;
  }
  // steps n and n + 1 of a grid walled by CPML: the bulk as ONE two-step sweep on st, the shell as two single steps on cs
  int shell_pair(long long n, const F2Table* tb, const ZPlan& zp) {
    hipStream_t cs = (h->shell_on == 2) ? st : h->comm_stream;       // (2: shell behind the bulk on ONE stream — a measuring aid)
    const int pml_in = 7 & pml_in_sweep_mask(h);
    if (ensure_second_set(h) || ensure_third_set(h) || ensure_pml_blocks(h, pml_in)) return -1;
    if (!h->ev_shell_a) {
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_shell_a, hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_shell_b, hipEventDisableTiming));
    }
    const int N[3] = {h->g.nx, h->g.ny, nz};
    int in0[3], in1[3];                                    // step one: the bulk shrunk by one cell (x: one lane) on its CPML sides
    for (int a = 0; a < 3; ++a) {
      in0[a] = sg.o0[a] > 0 ? sg.o0[a] + (a == 0 ? 4 : 1) : 0;
      in1[a] = sg.o1[a] < N[a] ? sg.o1[a] - (a == 0 ? 4 : 1) : N[a];
    }
    // (the order of a single step: H-side sources and TFSF corrections of step n on H^{n-1/2}, then the incident grid's H)
    launch_sources(h, false, n, 0, nz, st);
    advance_tfsf_aux(h, false, n, st);
    if (h->cfg.bc[4] == FDTD_BC_PERIODIC) fill_ghost_h(h, st);   // ghost(-1) must carry them too (as in a single step)
    HIPCHK(h, hipEventRecord(h->ev_shell_a, st));
    HIPCHK(h, hipStreamWaitEvent(cs, h->ev_shell_a, 0));
    const FieldP A = h->f, B = h->f2, T = h->f3;
    const int par = h->pml_parity;
    bool s2 = false;
    for (int i = 0; i < zp.n; ++i) {                       // the bulk: one clipped two-step sweep per interval of its planes
      const ClipP clip{sg.o0[0], sg.o1[0], sg.o0[1], sg.o1[1], zp.a[i], zp.b[i]};
      if (launch_fused2(h, n, st, tb, &s2, nullptr, &clip)) return -1;
    }
    if (launch_shell_step(h, A, T, par, in0, in1, pml_in, cs, zp, 1)) return -1;
    // the middle step: E-side sources / TFSF corrections of step n, the dispersive cells' memory term, the incident grid's E;
    // then what precedes step n + 1: its H-side sources / corrections, the incident grid's H
    launch_sources(h, true, n, 0, nz, cs, false, &T);
    launch_ade(h, 0, nz, cs, &T);
    advance_tfsf_aux(h, true, n, cs);
    launch_sources(h, false, n + 1, 0, nz, cs, false, &T);
    advance_tfsf_aux(h, false, n + 1, cs);
    fill_ghost_fused(h, cs, &T);                           // periodic z: the middle step's wrapped planes (its top and bottom planes are the shell's)
    if (launch_shell_step(h, T, B, par ^ 1, sg.o0, sg.o1, pml_in, cs, zp, 0)) return -1;
    HIPCHK(h, hipEventRecord(h->ev_shell_b, cs));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_shell_b, 0));
    swap_sets(h);
    pair_record(h, tb, n, st);
    if (rec_at(n + 1)) record_monitors(h, n + 1, true, st);
    launch_sources(h, true, n + 1, 0, nz, st);
    launch_ade(h, 0, nz, st);
    advance_tfsf_aux(h, true, n + 1, st);
    fill_ghost_fused(h, st);
    return 0;
  }
  // steps n and n + 1 of a grid walled by CPML, shell2 form: the bulk as ONE clipped two-step sweep on st, the shell's boxes as
  // shell2_step_kernel launches on cs — all read set A / the current psi sets, all write disjoint cells of set B / the other psi sets
  // `zp`: the bulk's plane intervals.  One interval [o0z, o1z): no holes.  More: the planes between two intervals are z HOLES — the
  // planes of source lists the sweeps cannot apply while they inject (a mode plane, a current sheet, the injection plane of a plane
  // wave; +- 2 planes) — and take two single steps through set T on cs, as in the round-4 form, with parameter blocks that route
  // their psi through temporary sets (ensure_pml_blocks_hole); every interval gets its own bulk launch and its own boxes.
  int shell2_pair(long long n, const F2Table* tb, const ZPlan& zp) {
    hipStream_t cs = (h->shell_on == 2) ? st : h->comm_stream;       // (2: shell behind the bulk on ONE stream — a measuring aid)
    const bool holes = zp.n > 1;
    // a periodic y: the two rows on either side of the wrap belong to no box and to no bulk launch — they take two single steps
    // through set T beside them, over every plane outside the holes (whose steps cover all rows), with the holes' parameter blocks
    const bool per_y = h->cfg.bc[2] == FDTD_BC_PERIODIC;
    if (ensure_second_set(h) || ensure_pml_blocks2(h)) return -1;
    if ((holes || per_y) && (ensure_third_set(h) || ensure_pml_blocks_hole(h))) return -1;
    if (!h->ev_shell_a) {
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_shell_a, hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_shell_b, hipEventDisableTiming));
    }
    launch_sources(h, false, n, 0, nz, st);                  // H-side sources of step n on H^{n-1/2}, then the incident grid's H: the order of a single step
    if (pair_spg) spg_fill(h, n, st);                        // (paged source terms of the pair; the incident grids through both steps)
    else advance_tfsf_aux(h, false, n, st);
    const SrcP sr = pair_spg ? spg_params(h) : SrcP{};
    HIPCHK(h, hipEventRecord(h->ev_shell_a, st));
    HIPCHK(h, hipStreamWaitEvent(cs, h->ev_shell_a, 0));
    const FieldP A = h->f, B = h->f2, T = h->f3;
    const int hp = h->pml_parity, ep = h->pml_e_parity;
    bool s2 = false;
    Shell2Box boxes[kShell2MaxBoxes];
    int nb = 0;
    for (int i = 0; i < zp.n; ++i) {
      const ClipP clip{sg.o0[0], sg.o1[0], sg.o0[1], sg.o1[1], zp.a[i], zp.b[i]};
      // (one interval and every source node deep inside it: the sweep applies the E-side terms of step n + 1 itself)
      if (launch_fused2(h, n, st, tb, &s2, nullptr, &clip, pair_disp, zp.n == 1 && s2_deep, sr)) return -1;
      // the boxes beside this interval: its planes, and (first / last interval) the z slabs below / above
      ShellGeom gi = sg;
      gi.o0[2] = zp.a[i]; gi.o1[2] = zp.b[i];
      Shell2Box bi[kShell2MaxBoxes];
      const int ni = shell2_boxes(h, gi, bi, i == 0 ? 0 : zp.a[i], i == zp.n - 1 ? nz : zp.b[i]);
      for (int q = 0; q < ni && nb < kShell2MaxBoxes; ++q) boxes[nb++] = bi[q];
    }
    // dispersive cells (all deep inside the bulk): both steps of their pole states and the correction of E^{n+2} behind the bulk,
    // beside the shell's boxes, once nothing else is due on E^{n+2} there — else at the end, behind the sources of step n + 1
    bool src_due = false;
    for (const PointSrc& sr : h->psrc) src_due = src_due || (sr.n_e && n + 1 < sr.n_steps);
    // (clear of the bulk's faces by what the boxes read beyond their own cells — a plane, two rows, a halo lane: ade2_kernel REWRITES the
    //  paged memory terms, and a box that has not yet read those of its halo plane would subtract the next pair's.  Found on the device by
    //  scripts/fuzz_round6.py in the round's last hour (seed 31, case 124: a Lorentz body whose lowest plane is the bulk's first one — one run
    //  in a few; the emulator, which runs the streams in issue order, shows it every time: tests/test_emu_disp.py).)
    const bool ade2_early = pair_disp && (s2 || !src_due) && disp_inside(h, sg.o0, sg.o1, 2);
    if (ade2_early) launch_ade2(h, st, &B);
    launch_shell2_boxes(h, boxes, nb, h->pml_blk2[hp][ep], cs, tb, pair_disp, sr);
    if (holes || per_y) {
      const int pml_in = 7 & pml_in_sweep_mask(h);
      ShellSets s1{A, T, hp, 0, 0, h->pml_blk_hole[0][hp][ep]}, s2h{T, B, hp, 0, 0, h->pml_blk_hole[1][hp][ep]};
      // the rows next to a periodic y wrap, grown by `grow` rows: the tile rows that hold them, the rows between left alone
      auto wrap_rows = [&](const ShellSets& base, int grow) {
        const int R = h->rows_f, ny = h->g.ny, nby_all = (ny + R - 1) / R;
        const int in0 = 2 + grow, in1 = ny - 2 - grow;
        const int ty_a = std::min(nby_all, (in0 + R - 1) / R), ty_c = std::max(ty_a, in1 / R);
        ShellSets sh = base;
        sh.ex_j0 = in0; sh.ex_j1 = in1;
        for (int i = 0; i < zp.n; ++i) {
          const int lo = i == 0 ? 0 : zp.a[i], hi = i == zp.n - 1 ? nz : zp.b[i];
          if (launch_fused_range(h, lo, hi, cs, pml_in, 0, 0, ty_a + (nby_all - ty_c), ty_a, ty_c - ty_a, true, &sh)) return -1;
        }
        return 0;
      };
      // step one over the holes grown by one plane (what step two differentiates), set A -> set T
      for (int i = 0; i + 1 < zp.n; ++i)
        if (launch_fused_range(h, zp.b[i] - 1, zp.a[i + 1] + 1, cs, pml_in, 0, 0, -1, 0, 0, true, &s1)) return -1;
      if (per_y && wrap_rows(s1, 1)) return -1;
      if (holes) {
        // the middle step: E-side sources / corrections of step n, the dispersive cells' memory term, the incident grid's E; then
        // what precedes step n + 1
        launch_sources(h, true, n, 0, nz, cs, false, &T);
        launch_ade(h, 0, nz, cs, &T);
        advance_tfsf_aux(h, true, n, cs);
        launch_sources(h, false, n + 1, 0, nz, cs, false, &T);
        advance_tfsf_aux(h, false, n + 1, cs);
      }
      for (int i = 0; i + 1 < zp.n; ++i)
        if (launch_fused_range(h, zp.b[i], zp.a[i + 1], cs, pml_in, 0, 0, -1, 0, 0, true, &s2h)) return -1;
      if (per_y && wrap_rows(s2h, 0)) return -1;
    }
    HIPCHK(h, hipEventRecord(h->ev_shell_b, cs));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_shell_b, 0));
    swap_sets(h);
    swap_psi_h(h, 7);
    swap_psi_e(h);
    pair_record(h, tb, n, st);
    if (rec_at(n + 1)) record_monitors(h, n + 1, true, st);
    if (!holes && !pair_spg) {
      advance_tfsf_aux(h, true, n, st);
      advance_tfsf_aux(h, false, n + 1, st);
    }
    if (!s2) launch_sources(h, true, n + 1, 0, nz, st);
    if (!pair_disp) launch_ade(h, 0, nz, st);
    else if (!ade2_early) launch_ade2(h, st);
    if (!pair_spg) advance_tfsf_aux(h, true, n + 1, st);
    fill_ghost_fused(h, st);
    return 0;
  }
  // shell2 pairs with z holes: every monitor of the plan clear of the holes' planes (their single steps copy nothing out) — a DFT
  // monitor inside one SEGMENT (an interval, extended to the grid's end below the first / above the last), a time monitor inside one interval
  bool plan_clear_of_holes(const F2Plan& pl, const ZPlan& zp) {
    auto inside = [&](const Monitor& m, bool segment) {
      for (int i = 0; i < zp.n; ++i) {
        const int lo = (segment && i == 0) ? 0 : zp.a[i], hi = (segment && i == zp.n - 1) ? nz : zp.b[i];
        if (m.box.lo2 >= lo && m.box.lo2 + m.box.nz <= hi) return true;
      }
      return false;
    };
    for (int q : pl.mons) if (!inside(h->mons[(size_t)q], false)) return false;
    for (int q : pl.dfts) if (!inside(h->mons[(size_t)q], true)) return false;
    // (a periodic y: the single steps of the rows next to the wrap copy nothing out either)
    if (h->cfg.bc[2] == FDTD_BC_PERIODIC)
      for (int q : pl.dfts) {
        const Monitor& m = h->mons[(size_t)q];
        if (m.box.lo1 < 2 || m.box.lo1 + m.box.ny > h->g.ny - 2) return false;
      }
    return true;
  }
  // every monitor of the pair's plan inside ONE interval of the bulk's planes (the sweep copies the middle step out only there)
  bool plan_in_bulk(const F2Plan& pl, const ZPlan& zp) {
    auto inside = [&](const Monitor& m) {
      for (int i = 0; i < zp.n; ++i) if (m.box.lo2 >= zp.a[i] && m.box.lo2 + m.box.nz <= zp.b[i]) return true;
      return false;
    };
    for (int q : pl.mons) if (!inside(h->mons[(size_t)q])) return false;
    for (int q : pl.dfts) if (!inside(h->mons[(size_t)q])) return false;
    return true;
  }
  // the step about to be issued: does a monitor record at it (rec), can steps n and n + 1 go out as one sweep (pair), in which form (use_s2, zp)
  int begin_step() {
    n = h->step;
    rec = !fused_multi && rec_at(n);
    if (rec && multi) {
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
    }
    // steps n and n + 1 as ONE sweep?  (fdtd_kernels2.hpp; no decay check on the middle step, sources all alive or all spent,
    // every monitor that records at n or n + 1 a small time monitor the sweep can sample)
    src_alive = false;
    src_why = 0;
    zp = &zp_base;
    pair = fused && (f2_ok || f2s_ok || s2_ok) && done + 2 <= n_steps && !(h->decay_every > 0 && ((n + 1) % h->decay_every) == 0);
    use_s2 = false;
    if (pair) {
      src_why = fused2_sources_why_not(h, n, &src_alive);
      // shell2 form: the boxes apply no sources — lists that inject must lie deep inside the bulk
      // lists the node table cannot hold, while they inject: paged source terms in the sweep, the seam kernel and the shell's boxes —
      // plain pairs and shell2 pairs without z holes (a periodic y's wrap rows take single steps: the round-5 forms there)
      pair_spg = false;
      if (src_why != 0 && spg_ok && (src_why == FDTD_F2_OFF_TFSF || src_why == FDTD_F2_OFF_SOURCES || src_why == FDTD_F2_OFF_SEAM_SOURCE)) {
        const bool s2_form = s2_ok && zp_s2.ok && zp_s2.n == 1 && h->cfg.bc[2] != FDTD_BC_PERIODIC;
        if (s2_form || f2_ok) { pair_spg = true; src_why = 0; src_alive = false; }
      }
      use_s2 = s2_ok && src_why == 0 && (!src_alive || s2_deep || pair_spg);
      if (use_s2) zp = &zp_s2;
      else if (s2_ok && zp_s2h.ok && (src_why != 0 || src_alive)) {
        // lists that inject and that the sweeps cannot apply themselves (too many nodes, nodes inside the shell, TFSF corrections): their
        // planes take single steps as z holes; nothing is injected by the sweeps (the table of a pair whose lists are spent)
        use_s2 = true; src_why = 0; src_alive = false; zp = &zp_s2h;
      }
      if (!use_s2 && !f2_ok && !f2s_ok) { if (src_why) h->f2_dyn_reason = src_why; pair = false; }
    }
    if (pair && use_s2) {
      pair = fused2_plan(h, n, &f2_plan, sg.o0, sg.o1, true) && plan_clear_of_holes(f2_plan, *zp);
      if (!pair && f2s_ok) { pair = true; use_s2 = false; zp = &zp_base; src_why = fused2_sources_why_not(h, n, &src_alive); }   // (the single-step shell may still take it — judged below)
    }
    if (pair && !use_s2) {
      // lists that inject and that the sweep cannot apply itself: a shell pair whose bulk leaves their planes to the shell
      if (src_why && f2s_ok && zp_src.ok) { src_why = 0; src_alive = false; zp = &zp_src; }
      if (src_why) h->f2_dyn_reason = src_why;
      pair = src_why == 0 && fused2_plan(h, n, &f2_plan, f2s_ok ? sg.o0 : nullptr, f2s_ok ? sg.o1 : nullptr) &&
             (!f2s_ok || plan_in_bulk(f2_plan, *zp));
    }
    pair_disp = pair && !h->ade.empty() && h->disp.state == 1 && (use_s2 ? (s2_disp && zp == &zp_s2) : !f2s_ok);
    pair_spg = pair_spg && pair && (use_s2 ? zp == &zp_s2 : (f2_ok && !f2s_ok));
    // (with H-side sources the monitors of a pair still take E^n and H^{n-1/2} here: those sources change H^{n-1/2} before the
    //  sweep, and pair_record reads the set afterwards)
    if (rec) record_monitors(h, n, false, st, (pair && !h_terms_in_front(h)) ? &f2_plan : nullptr);
    if (rec && multi) {
      // The record reads H^{n-1/2} of the top plane, which the comm stream is about to advance (its H-side corrections and
      // update of that plane wait for the E interior of the LAST step only): it must let the record finish first.  Found by
      // scripts/fuzz_variants.py on the device (round 4): a volume time monitor reaching the slab's top plane came back with
      // that plane's H half-sample taken during / after the update, in one run out of a few.
      if (!h->ev_rec) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rec, hipEventDisableTiming));
      HIPCHK(h, hipEventRecord(h->ev_rec, st));
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_rec, 0));
    }
    return 0;
  }
  // steps n and n + 1 of a z-slab rank that carries CPML (entry and exit state: "primed", as slab_rank_step's pair):
  //   st: the bulk (sgm: the CPML-free box, two or more planes from a cut) as ONE clipped two-step sweep, set A -> set B
  //   cs: the two planes next to each cut as a z hole — step one A -> T over the hole grown by one plane (psi: current sets ->
  //       temporary sets), its E-side / the next H-side source terms, the planes (and the H-side psi of the top plane, temporary
  //       set) travel; step two T -> B (psi: temporary -> the other sets), source terms, the planes travel again — the messages of
  //       two single steps, in their order.
  //   st, behind the bulk: the shell's boxes (x strips, y / z slabs over the planes clear of the cuts) by shell2_step_kernel, A -> B,
  //       psi current -> other sets.
  // No launch reads what another one of the pair writes; the next step's edges (ev_e_bnd, ev_e_int) order it behind both streams.
  int slab_shell2_pair(const F2Table* tb) {
    const int bl = nb_lo ? 2 : 0, bh = nb_hi ? 2 : 0;
    const FieldP A = h->f, B = h->f2, T = h->f3;
    const int hp = h->pml_parity, ep = h->pml_e_parity;
    ShellSets s1{A, T, hp, 0, 0, h->pml_blk_hole[0][hp][ep]}, s2h{T, B, hp, 0, 0, h->pml_blk_hole[1][hp][ep]};
    HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
    const ClipP clip{sgm.o0[0], sgm.o1[0], sgm.o0[1], sgm.o1[1], sgm.o0[2], sgm.o1[2]};
    bool s2done = false;
    Shell2Box boxes[kShell2MaxBoxes];
    const int nb = shell2_boxes(h, sgm, boxes, bl, nz - bh);
    // Round 6: the boxes go out IN FRONT of the bulk on st (FDTD_OPT_SLAB_BOXES_FIRST, default).  Behind it (round 5) they ran alone
    // on the machine for 87 us of a 362 us pair of a 64-plane slab while the hole's first step, beside the bulk's single round of
    // one-per-CU workgroups, crawled for 160 us on the CUs the bulk left (kernel timeline profiles/r6/r6tr_timeline_p2.txt); in front,
    // boxes and hole share the machine, then the bulk runs beside the exchanges and the hole's second step.
    // (2: on a third stream beside both — the boxes read set A and their own psi sets, write their own cells of set B: they wait for what
    //  st and cs waited for, and st waits for them before it marks the pair's interior done)
    // Measured (profiles/r6/r6b3_slab_boxes_third_stream.jsonl, 512 x 512 x nz with CPML on x / y, three rounds interleaved): 128 planes 0.3226 ->
    // 0.3175 ms per step, 256 planes 0.571 -> 0.561 — but 64 planes 0.164 -> 0.213: beside a bulk of ONE round of one-per-CU workgroups the
    // boxes and the hole's first step crawl on the CUs the bulk left, and st waits for the boxes.  Hence 3: by the slab's planes.
    bool third = (h->slab_boxes_first == 2 || (h->slab_boxes_first == 3 && nz >= 96)) && nb > 0 && !h->streams_shared && !h->debug_sync;
    if (third && !h->box_stream_tried && make_box_stream(h)) return -1;
    third = third && h->box_stream != nullptr;
    if (third) {
      HIPCHK(h, hipEventRecord(h->ev_box_in, st));                            // (everything st has issued: the last pair's interior)
      HIPCHK(h, hipStreamWaitEvent(h->box_stream, h->ev_box_in, 0));
      HIPCHK(h, hipStreamWaitEvent(h->box_stream, h->ev_e_bnd, 0));
      launch_shell2_boxes(h, boxes, nb, h->pml_blk2[hp][ep], h->box_stream, tb);
      HIPCHK(h, hipEventRecord(h->ev_box, h->box_stream));
    } else if (h->slab_boxes_first) launch_shell2_boxes(h, boxes, nb, h->pml_blk2[hp][ep], st, tb);
    if (launch_fused2(h, n, st, tb, &s2done, nullptr, &clip)) return -1;
    if (launch_fused_range(h, 0, bl ? bl + 1 : 0, cs, pml_in_m, bh ? nz - bh - 1 : nz, nz, -1, 0, 0, true, &s1)) return -1;
    HIPCHK(h, hipEventRecord(h->ev_h_bnd, cs));
    if (bl) { launch_sources(h, true, n, 0, bl + 1, cs, false, &T); launch_sources(h, false, n + 1, 0, bl + 1, cs, false, &T); }
    if (bh) { launch_sources(h, true, n, nz - bh - 1, nz, cs, false, &T); launch_sources(h, false, n + 1, nz - bh - 1, nz, cs, false, &T); }
    if (exchange_fused_all(h, cs, psi_ghosts, &T, 1)) return -1;
    if (launch_fused_range(h, 0, bl, cs, pml_in_m, nz - bh, nz, -1, 0, 0, true, &s2h)) return -1;
    if (bl) { launch_sources(h, true, n + 1, 0, bl, cs, false, &B); launch_sources(h, false, n + 2, 0, bl, cs, false, &B); }
    if (bh) { launch_sources(h, true, n + 1, nz - bh, nz, cs, false, &B); launch_sources(h, false, n + 2, nz - bh, nz, cs, false, &B); }
    HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
    if (exchange_fused_all(h, cs, psi_ghosts, &B, 2)) return -1;
    // (not on cs: it carries what the neighbours wait for — behind the hole's steps and the two exchanges the boxes made cs the longer
    //  stream of a thin slab: 64 planes 0.232 -> 0.205 ms per step, 128 planes 0.500 -> 0.460, profiles/r5/r5zb)
    if (!h->slab_boxes_first) launch_shell2_boxes(h, boxes, nb, h->pml_blk2[hp][ep], st, tb);
    if (third) HIPCHK(h, hipStreamWaitEvent(st, h->ev_box, 0));
    swap_sets(h);                                                            // h->f = B: E^{n+2}, H^{n+3/2}
    swap_psi_h(h, 7);
    swap_psi_e(h);
    launch_sources(h, true, n + 1, bl, nz - bh, st);
    launch_sources(h, false, n + 2, bl, nz - bh, st);
    HIPCHK(h, hipEventRecord(h->ev_e_int, st));
    h->fused2_pairs++;
    h->shell2_pairs++;
    h->step = n + 2;
    return 1;
  }
  // one step — or, where it can, a step pair — of a z-slab rank on the pipelined fused schedule (header comment: setup; slab pair: below).
  // -> 1: a pair was taken (two steps, no decay check due), 0: one step, < 0: error
  int slab_rank_step() {
    if (!primed && prime(n)) return -1;
    // ---- slab pair: steps n and n + 1 of a z-slab rank -----------------------------------------------------------------
    // The two-step sweep advances the planes two or more away from a neighbour face (it reads the slab's own planes
    // only: no ghost plane, no dependence on the wire); the two planes next to a neighbour face — its shell — take two
    // single steps on the comm stream, through the third set, and ship their planes after EACH of them: the messages a
    // neighbour receives are those of two single steps, in the same order (a rank may take a pair while its neighbour
    // takes single steps).  Same kernels and formulas: the same bits (tests/test_dist_gloo.py).  Entry and exit state:
    // "primed" (above).  Pairs keep clear of monitor records, decay checks and the end of the run (joined tails).
    auto decay_at = [&](long long m) { return h->decay_every > 0 && (m % h->decay_every) == 0; };
    bool src_alive_m = false;
    if (f2mc_ok && done + 3 <= n_steps && !rec_at(n) && !rec_at(n + 1) && !rec_at(n + 2) && !decay_at(n + 1) && !decay_at(n + 2) &&
        fused2_sources_why_not(h, n, &src_alive_m) == 0 && (!src_alive_m || f2mc_deep)) {
      F2Plan none;
      const F2Table* tb = fused2_table(h, none, src_alive_m);
      if (!tb) return -1;
      return slab_shell2_pair(tb);
    }
    if (f2m_ok && done + 3 <= n_steps && !rec_at(n) && !rec_at(n + 1) && !rec_at(n + 2) && !decay_at(n + 1) && !decay_at(n + 2) &&
        fused2_sources_why_not(h, n, &src_alive_m) == 0) {
      F2Plan none;
      const F2Table* tb = fused2_table(h, none, src_alive_m);
      if (!tb) return -1;
      const int bl = nb_lo ? 2 : 0, bh = nb_hi ? 2 : 0;
      const FieldP A = h->f, B = h->f2, T = h->f3;
      ShellSets s1{A, T, 0, 0, 0}, s2{T, B, 0, 0, 0};
      // (host order: the long bulk sweep is handed to the device first — the comm stream's dozen launches and two RCCL groups
      //  take the host longer to issue than the device needs for them; issued first they left the device idle for 30 us per
      //  pair in front of the bulk, profiles/r4e)
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
      // bulk: both steps in one sweep, set A -> set B
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
      const ClipP clip{0, h->g.nx, 0, h->g.ny, bl, nz - bh};
      bool s2done = false;
      if (launch_fused2(h, n, st, tb, &s2done, nullptr, &clip)) return -1;
      // shell, step one: the boundary planes and one more (what step two differentiates), set A -> set T
      if (launch_fused_range(h, 0, bl ? bl + 1 : 0, cs, 0, bh ? nz - bh - 1 : nz, nz, -1, 0, 0, false, &s1)) return -1;
      HIPCHK(h, hipEventRecord(h->ev_h_bnd, cs));
      if (bl) { launch_sources(h, true, n, 0, bl + 1, cs, false, &T); launch_sources(h, false, n + 1, 0, bl + 1, cs, false, &T); }
      if (bh) { launch_sources(h, true, n, nz - bh - 1, nz, cs, false, &T); launch_sources(h, false, n + 1, nz - bh - 1, nz, cs, false, &T); }
      if (exchange_fused_all(h, cs, false, &T)) return -1;                  // what the neighbours expect after step n
      // shell, step two: set T -> set B
      if (launch_fused_range(h, 0, bl, cs, 0, nz - bh, nz, -1, 0, 0, false, &s2)) return -1;
      // corrections of step n + 1 (E side) and n + 2 (H side): boundary planes on cs, then their planes travel
      if (bl) { launch_sources(h, true, n + 1, 0, bl, cs, false, &B); launch_sources(h, false, n + 2, 0, bl, cs, false, &B); }
      if (bh) { launch_sources(h, true, n + 1, nz - bh, nz, cs, false, &B); launch_sources(h, false, n + 2, nz - bh, nz, cs, false, &B); }
      HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
      if (exchange_fused_all(h, cs, false, &B)) return -1;
      swap_sets(h);                                                          // h->f = B: E^{n+2}, H^{n+3/2}
      launch_sources(h, true, n + 1, bl, nz - bh, st);
      launch_sources(h, false, n + 2, bl, nz - bh, st);
      HIPCHK(h, hipEventRecord(h->ev_e_int, st));
      h->fused2_pairs++;
      h->step = n + 2;
      return 1;
    }
    const bool decay_step = h->decay_every > 0 && ((n + 1) % h->decay_every) == 0;
    const bool last = (done + 1 == n_steps) || decay_step;
    // sweeps: boundary chunks (one launch) on cs, interior on st
    HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
    // (the boundary chunks lie clear of the z slabs: their launch carries x / y at most)
    const int pml_b = pml_in_m & 3;
    if (b_lo > 0 && b_hi > 0) { if (launch_fused_range(h, 0, b_lo, cs, pml_b, nz - b_hi, nz)) return -1; }
    else if (b_lo > 0) { if (launch_fused_range(h, 0, b_lo, cs, pml_b)) return -1; }
    else if (b_hi > 0) { if (launch_fused_range(h, nz - b_hi, nz, cs, pml_b)) return -1; }
    HIPCHK(h, hipEventRecord(h->ev_h_bnd, cs));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
    if ((pml_in_m & 6) == 0) {
      if (launch_fused_range(h, b_lo, nz - b_hi, st, pml_in_m)) return -1;
    } else {
      // interior planes by tile class, as on one GPU (all three on the main stream: the other one ships ghost planes)
      const int R = h->rows_f, nby_all = (h->g.ny + R - 1) / R, ki = b_lo, ke = nz - b_hi;
      const PmlAxisDev &py = h->pml[1], &pz = h->pml[2];
      const bool in_y = (pml_in_m & 2) && py.ns > 0, in_z = (pml_in_m & 4) && pz.ns > 0;
      const int za = std::min(ke, std::max(ki, (in_z && pz.lo > 0) ? std::min(nz, pz.lo + 1) : 0));
      const int zc = std::max(za, std::min(ke, (in_z && pz.hi0 < nz) ? pz.hi0 : nz));
      const int ty_a = (in_y && py.lo > 0) ? std::min(nby_all, py.lo / R + 1) : 0;
      const int ty_c = (in_y && py.hi0 < h->g.ny) ? std::max(ty_a, py.hi0 / R) : nby_all;
      // (all on the main stream.  Edge launches on a third stream were tried: no gain, and an engine with three
      //  streams pushed the next engine of the process onto shared hardware queues — its two streams serialised,
      //  3x slower steps, profiles/r04r)
      if ((za > ki || zc < ke) && launch_fused_range(h, ki, za, st, pml_in_m, zc, ke, -1, 0, 0, true)) return -1;
      if (launch_fused_range(h, za, zc, st, pml_in_m & 3, 0, 0, ty_a + (nby_all - ty_c), ty_a, ty_c - ty_a, true)) return -1;
      if (launch_fused_range(h, za, zc, st, pml_in_m & 1, 0, 0, ty_c - ty_a, 0, ty_a)) return -1;
    }
    swap_sets(h);
    swap_psi_h(h, pml_in_m);
    const bool rec_post = rec_at(n);
    if (rec_post || last || rec_at(n + 1)) {
      // joined tail: everything after the sweeps on st
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
      if (rec_post) record_monitors(h, n, true, st);
      e_post(n, 0, nz, st, false);
      advance_tfsf_aux(h, true, n, st, false);
      advance_tfsf_aux(h, true, n, st, true);
      primed = false;
      if (!last && prime(n + 1)) return -1;
      if (last) {           // leave both streams joined; the next step (or run) primes again
        HIPCHK(h, hipEventRecord(h->ev_e_int, st));
        HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
        HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
      }
    } else {
      e_post(n, 0, b_lo, cs, true);
      e_post(n, nz - b_hi, nz, cs, true);
      advance_tfsf_aux(h, true, n, cs, true);
      h_pre(n + 1, 0, b_lo, cs, true);
      h_pre(n + 1, nz - b_hi, nz, cs, true);
      advance_tfsf_aux(h, false, n + 1, cs, true);
      HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
      if (exchange_fused_all(h, cs, psi_ghosts)) return -1;
      e_post(n, b_lo, nz - b_hi, st, false);
      advance_tfsf_aux(h, true, n, st, false);
      h_pre(n + 1, b_lo, nz - b_hi, st, false);
      advance_tfsf_aux(h, false, n + 1, st, false);
      HIPCHK(h, hipEventRecord(h->ev_e_int, st));
    }
    h->step = n + 1;
    return 0;
  }
  // steps n and n + 1 of a run without CPML as ONE two-step sweep, all on st: set A -> set B, swap
  int plain_pair() {
    const F2Table* tb = fused2_table(h, f2_plan, src_alive);
    if (!tb) return -1;
    bool sources2_done = false, damp2_done = true;
    launch_sources(h, false, n, 0, nz, st);              // H-side sources of step n act on H^{n-1/2}, as before a single step
    if (pair_spg) spg_fill(h, n, st);                    // (the other source terms of the pair into paged storage, the incident grids through both steps)
    if (launch_fused2(h, n, st, tb, &sources2_done, &damp2_done, nullptr, pair_disp, false, pair_spg ? spg_params(h) : SrcP{})) return -1;
    pair_record(h, tb, n, st);                           // (H^{n+3/2} is not touched by the E-side sources that follow)
    if (rec_at(n + 1)) record_monitors(h, n + 1, true, st);      // DFT records at the middle step: their H terms, from the write set
    if (!sources2_done) launch_sources(h, true, n + 1, 0, nz, st);
    if (h->has_damp && !damp2_done) launch_damp(h, true, 0, nz, st);
    if (pair_disp) launch_ade2(h, st);                   // (the ADE update of step n + 1 follows its sources and damping, as launch_ade does)
    fill_ghost_fused(h, st);
    return 0;
  }
  // one step of the two-pass kernels (H pass, E pass; odd row lengths, FDTD_VARIANT_ZMARCH, z-slab ranks on the AUTO variant): interior on st,
  // the plane next to a neighbour face and the exchanges on cs; edges: ev_e_int, ev_e_bnd, ev_h_int, ev_h_bnd (setup_schedules)
  int two_pass_step() {
    // ---------------- H phase ----------------
    const int h_top = (multi && nb_hi) ? nz - 1 : nz;      // planes [0, h_top) on st, [h_top, nz) on cs
    const bool mirrors = h->mirror_wall[0] >= 0 || h->mirror_wall[1] >= 0 || h->mirror_wall[2] >= 0;
    if (multi && mirrors) {
      // PMC plus walls on a z-slab rank: the images of ALL planes are refreshed on the main stream before either stream goes on —
      // its H pass differentiates E of the top plane (image columns included: the update of an image cell feeds the wall's own
      // unknowns in the same step), so the comm stream must not refresh that plane beside it (a race the device showed in one
      // visit out of three, tests/test_gpu_parity.py)
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
      fill_mirror(h, st, 0, nz);
      if (!h->ev_rec) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rec, hipEventDisableTiming));
      HIPCHK(h, hipEventRecord(h->ev_rec, st));
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_rec, 0));
    }
    if (multi && nb_hi) {
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
      launch_damp(h, false, h_top, nz, cs);          // absorber layers damp H^{n-1/2} before anything is added
      launch_sources(h, false, n, h_top, nz, cs);    // H-side corrections first (they only read E^n),
      launch_pml(h, false, h_top, nz, cs);           // in the summation order of the fused sweep
      launch_h_main(h, h_top, nz, cs);
      HIPCHK(h, hipEventRecord(h->ev_h_bnd, cs));
    }
    if (multi) HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
    if (!multi) fill_mirror(h, st, 0, nz);
    launch_damp(h, false, 0, h_top, st);
    launch_sources(h, false, n, 0, h_top, st);
    launch_pml(h, false, 0, h_top, st);
    launch_h_main(h, 0, h_top, st);
    advance_tfsf_aux(h, false, n, st);
    if (multi) {
      HIPCHK(h, hipEventRecord(h->ev_h_int, st));
      if (!nb_hi) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_h_int, 0));
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));     // WAR: ghost(-1) was read by the last E pass
      if (exchange(h, false, cs)) return -1;
    }
    if (!multi || !nb_lo) fill_ghost_h(h, st);   // physical z-min face of this slab (PMC / periodic)
    if (rec) {
      if (multi) HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
      record_monitors(h, n, true, st);
    }
    // ---------------- E phase ----------------
    const int e_bot = (multi && nb_lo) ? 1 : 0;            // planes [0, e_bot) on cs, [e_bot, nz) on st
    if (multi && nb_lo) {
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_h_int, 0));
      launch_e_main(h, 0, e_bot, cs);
      launch_pml(h, true, 0, e_bot, cs);
      launch_sources(h, true, n, 0, e_bot, cs);
      launch_damp(h, true, 0, e_bot, cs);
      launch_ade(h, 0, e_bot, cs);
      HIPCHK(h, hipEventRecord(h->ev_e_bnd, cs));
    }
    if (multi && nb_hi) HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
    aniso_save(h, st);                     // (these kernels update E in place: E^n of the nodes around fully anisotropic cells first)
    launch_e_main(h, e_bot, nz, st);
    launch_pml(h, true, e_bot, nz, st);
    launch_sources(h, true, n, e_bot, nz, st);
    aniso_apply(h, st);
    launch_damp(h, true, e_bot, nz, st);
    launch_ade(h, e_bot, nz, st);
    advance_tfsf_aux(h, true, n, st);
    if (multi) {
      HIPCHK(h, hipEventRecord(h->ev_e_int, st));
      if (!nb_lo) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_e_int, 0));
      HIPCHK(h, hipStreamWaitEvent(cs, h->ev_h_int, 0));     // WAR: ghost(nz) was read by this H pass
      if (exchange(h, true, cs)) return -1;
    }
    if (!multi || !nb_hi) fill_ghost_e(h, st);   // physical z-max face of this slab (periodic)
    h->step = n + 1;
    return 0;
  }
  // field decay / divergence every decay_every steps (joins the streams; the only host synchronisation of the loop).  -> 1: the run ends here
  int decay_check() {
  // ---------------- field decay / divergence ----------------
  if (h->decay_every > 0 && (h->step % h->decay_every) == 0) {
    if (multi) HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
    double en = 0.0;
    if (eval_energy(h, st, &en)) return -1;
    if (multi) {
      // sum over ranks (1 double every decay_every steps).  Every RCCL call of this communicator
      // is issued on the comm stream, in the same order on all ranks — never from two streams.
      double* tmp = h->energy_dev;
      HIPCHK(h, hipMemcpyAsync(tmp, &en, sizeof(double), hipMemcpyHostToDevice, cs));
      NCCLCHK(h, ncclAllReduce(tmp, tmp, 1, ncclDouble, ncclSum, h->comm, cs));
      HIPCHK(h, hipMemcpyAsync(&en, tmp, sizeof(double), hipMemcpyDeviceToHost, cs));
      HIPCHK(h, hipStreamSynchronize(cs));
    }
    if (!std::isfinite(en)) {
      h->stats.diverged = 1;
      return 1;
    }
    if (en > h->energy_max) h->energy_max = en;
    h->stats.field_decay = h->energy_max > 0 ? en / h->energy_max : 1.0;
    if (progress && progress(h->step, 0.0, h->stats.field_decay, user)) return 1;
    if (h->shutoff > 0 && h->step > h->decay_ref && h->stats.field_decay < h->shutoff) {
      h->stats.stopped_early = 1;
      return 1;
    }
  }
    return 0;
  }
  // joins the streams, reads the timers
  int finish() {
    if (multi) {
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_e_bnd, 0));
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_h_bnd, 0));
    }
    HIPCHK(h, hipEventRecord(h->ev1, st));
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipStreamSynchronize(cs));
    for (hipEvent_t e : tb_ev) hipEventDestroy(e);
    for (const GraphRec& r : graphs) hipGraphExecDestroy(r.exec);
    HIPCHK(h, hipGetLastError());
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->stats.run_ms = ms;
    h->stats.steps_done = h->step;
    h->stats.h_kernel_ms = h->stats.e_kernel_ms = h->stats.fused_kernel_ms = h->stats.shell_kernel_ms = h->stats.seam_kernel_ms = 0.0;
    h->stats.h_kernel_launches = h->stats.e_kernel_launches = h->stats.fused_kernel_launches = h->stats.shell_kernel_launches = h->stats.seam_kernel_launches = 0;
    for (size_t i = 0; i < h->kev_kind.size(); ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, h->kev[2 * i], h->kev[2 * i + 1]) != hipSuccess) continue;
      if (h->kev_kind[i] == 0) { h->stats.h_kernel_ms += t; h->stats.h_kernel_launches++; }
      else if (h->kev_kind[i] == 1) { h->stats.e_kernel_ms += t; h->stats.e_kernel_launches++; }
      else if (h->kev_kind[i] == 3) { h->stats.shell_kernel_ms += t; h->stats.shell_kernel_launches++; }
      else if (h->kev_kind[i] == 4) { h->stats.seam_kernel_ms += t; h->stats.seam_kernel_launches++; }
      else { h->stats.fused_kernel_ms += t; h->stats.fused_kernel_launches++; }
    }
    return 0;
  }
  int loop() {
    for (; done < n_steps; ++done) {
      if (begin_step()) return -1;
      if (fused_multi) {
        const int rc = slab_rank_step();
        if (rc < 0) return -1;
        if (rc > 0) { ++done; continue; }
      } else if (fused && tb_ok && done + 2 <= n_steps && !rec && !rec_at(n + 1) &&
                 !(h->decay_every > 0 && ((n + 1) % h->decay_every) == 0)) {
        if (tb_pair(n)) return -1;
        h->step = n + 2;
        ++done;                                              // (the loop header counts the second step)
      } else if (pair && use_s2) {
        const F2Table* tb = fused2_table(h, f2_plan, src_alive);
        if (!tb) return -1;
        if (shell2_pair(n, tb, *zp)) return -1;
        h->fused2_pairs++;
        h->shell_pairs++;
        h->shell2_pairs++;
        h->step = n + 2;
        ++done;
      } else if (pair && f2s_ok) {
        const F2Table* tb = fused2_table(h, f2_plan, src_alive);
        if (!tb) return -1;
        if (shell_pair(n, tb, *zp)) return -1;
        h->fused2_pairs++;
        h->shell_pairs++;
        h->step = n + 2;
        ++done;
      } else if (pair) {
        if (plain_pair()) return -1;
        h->fused2_pairs++;
        h->step = n + 2;
        ++done;
      } else if (fused && graph_ok && done + 2 <= n_steps && !rec && !rec_at(n + 1) &&
                 !(h->decay_every > 0 && ((n + 1) % h->decay_every) == 0) && sources_alive(n + 1)) {
        const int grc = graph_pair(n);
        if (grc < 0) return -1;
        if (grc == 0) {                                      // replayed: two steps done
          h->step = n + 2;
          ++done;
        } else {                                             // capture not available: this step directly, no more attempts
          if (fused_one(n, rec)) return -1;
          h->step = n + 1;
        }
      } else if (fused) {
        if (fused_one(n, rec)) return -1;
        h->step = n + 1;
      } else {
        if (two_pass_step()) return -1;
      }
      sync_point();
      const int dc = decay_check();
      if (dc < 0) return -1;
      if (dc > 0) { ++done; break; }
    }
    return 0;
  }
};
}  // namespace
extern "C" {

int fdtd_run(FdtdSolver* h, int64_t n_steps, FdtdProgressFn progress, void* user) {
  if (!h) return -1;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  Run r{h, n_steps, progress, user};
  if (r.setup() || r.setup_schedules() || r.setup_pairs()) return -1;
  if (r.loop()) return -1;
  return r.finish();
}
// Complex (Bloch-periodic) fields: two solvers carry Re and Im of the same simulation (identical grid,
// media, CPML, ADE and monitors; the Im solver's source weights are the Re solver's times -i) and are
// stepped together on the Re solver's stream; they only meet in the ghost fills (fdtd_kernels.hpp).
// phase[a] = 2 pi bloch_vec of axis a (ref boundary.py:55-79 bloch_phase).  n_real[a] > 0 (a = x, y): the
// axis carries ghost cells at device index 0 and n_real[a] + 1 (the host builds the grid that way and
// declares PEC walls); z uses its ghost planes when the z faces are FDTD_BC_PERIODIC.  One GPU; the
// fused sweep unless the handles ask for the two-pass kernels.
int fdtd_run_bloch(FdtdSolver* hr, FdtdSolver* hi, int64_t n_steps, const double phase[3], const int n_real[3],
                   FdtdProgressFn progress, void* user) {
  if (!hr || !hi) return -1;
  if (hi->comm) return fail(hr, "fdtd_run_bloch: the communicator of a z-slab belongs to the first (real-part) handle");
  if (!hr->aniso.empty() || !hi->aniso.empty()) return fail(hr, "fdtd_run_bloch: fully anisotropic media are not available together with Bloch boundaries");
  for (int a = 0; a < 3; ++a)
    if (hr->mirror_wall[a] >= 0 || hi->mirror_wall[a] >= 0) return fail(hr, "fdtd_run_bloch: PMC on a plus face is not available together with Bloch boundaries");
  // z-slab decomposition (hr->comm): both parts exchange their ghost planes through the real-part handle's
  // communicator; the planes that wrap around a Bloch z axis (rank n-1 <-> rank 0) are rotated by exp(-+ i phi_z)
  // on arrival.  Two-pass kernels, exchanges on the step stream (correct by construction; not overlapped).
  const bool multi = hr->comm != nullptr;
  const bool nb_lo = hr->cfg.bc[4] == FDTD_BC_NEIGHBOR, nb_hi = hr->cfg.bc[5] == FDTD_BC_NEIGHBOR;
  if ((nb_lo || nb_hi) && !multi) return fail(hr, "fdtd_run_bloch: neighbour faces need fdtd_comm_init on the first handle");
  if (hi->cfg.bc[4] != hr->cfg.bc[4] || hi->cfg.bc[5] != hr->cfg.bc[5])
    return fail(hr, "fdtd_run_bloch: the two solvers must have the same z faces");
  if (hr->g.nx != hi->g.nx || hr->g.ny != hi->g.ny || hr->g.nz != hi->g.nz || hr->cfg.device != hi->cfg.device ||
      hr->mons.size() != hi->mons.size() || hr->step != hi->step)
    return fail(hr, "fdtd_run_bloch: the two solvers must describe the same simulation");
  const GridP& g = hr->g;
  const int nz = g.nz;
  const int N[3] = {g.nx, g.ny, nz};
  for (int a = 0; a < 2; ++a)
    if (n_real[a] < 0 || (n_real[a] > 0 && n_real[a] + 2 > N[a]))
      return fail(hr, "fdtd_run_bloch: axis %d has %d cells, cannot hold %d real cells + 2 ghost cells", a, N[a], n_real[a]);
  HIPCHK(hr, hipSetDevice(hr->cfg.device));
  hipStream_t st = hr->stream;
  HIPCHK(hr, hipStreamSynchronize(hi->stream));
  FdtdSolver* both[2] = {hr, hi};
  for (FdtdSolver* h : both) {
    for (hipEvent_t e : h->kev) hipEventDestroy(e);
    h->kev.clear(); h->kev_kind.clear();
    h->stats.stopped_early = 0;
  }
  HIPCHK(hr, hipEventRecord(hr->ev0, st));
  const bool per_z = hr->cfg.bc[4] == FDTD_BC_PERIODIC;
  const bool fused = !multi && !hr->mat4b && (g.nx % 4 == 0) && hr->rows_f <= 15 &&
                     (hr->cfg.variant == FDTD_VARIANT_FUSED || hr->cfg.variant == FDTD_VARIANT_AUTO);
  if (fused) for (FdtdSolver* h : both) if (ensure_second_set(h)) return -1;
  if (fused && !hr->tuned && !hr->user_geometry && !hi->user_geometry &&
      (n_cells(hr) >= (1LL << 20) || hr->autotune == 2)) {
    // same rule as fdtd_run: probe the tile shape when the default one launches less than a wave of workgroups
    const long long wgs = (long long)((g.nx + 255) / 256) * ((g.ny + hr->rows_f - 1) / hr->rows_f) *
                          ((nz + hr->zchunk_f - 1) / hr->zchunk_f);
    if (wgs < 768 || hr->autotune) {
      if (autotune_fused(hr, st)) return -1;
      hi->rows_f = hr->rows_f; hi->zchunk_f = hr->zchunk_f; hi->tuned = true;
    }
  }
  float cph[3], sph[3];
  for (int a = 0; a < 3; ++a) { cph[a] = (float)std::cos(phase[a]); sph[a] = (float)std::sin(phase[a]); }
  const long long pc = plane_cells(hr);
  // (pointers are taken at call time: the fused sweep swaps the field sets every step)
  auto six = [&]() { Cplx6P F; for (int c = 0; c < 6; ++c) { F.f[c].re = field_ptr(hr, c); F.f[c].im = field_ptr(hi, c); } return F; };
  auto fill_xy = [&]() {          // y first, then x over all rows (ghost rows included): corners get both phases
    for (int a = 1; a >= 0; --a) {
      if (n_real[a] <= 0) continue;
      const long long cells = (long long)(a == 0 ? g.ny : g.nx) * nz;
      hipLaunchKernelGGL(bloch_ghost_fill_kernel, dim3(nblk(cells)), dim3(256), 0, st, g, a, n_real[a], six(), cph[a], sph[a], nz);
    }
  };
  auto plane = [&](int c, long long dst_plane, long long src_plane, float s) {
    hipLaunchKernelGGL(bloch_plane_kernel, dim3(nblk(pc)), dim3(256), 0, st, field_ptr(hr, c) + dst_plane * pc,
                       field_ptr(hi, c) + dst_plane * pc, (const float*)(field_ptr(hr, c) + src_plane * pc),
                       (const float*)(field_ptr(hi, c) + src_plane * pc), cph[2], s, pc);
  };
  // ghost-plane exchange of both parts with the z neighbours (two-pass schedule): H phase = top H_x, H_y planes
  // up, E phase = bottom E_x, E_y planes down; a plane that crossed the periodic wrap is rotated where it lands
  const int lo_rank = (hr->rank - 1 + hr->n_ranks) % hr->n_ranks, hi_rank = (hr->rank + 1) % hr->n_ranks;
  const bool wrap_lo = nb_lo && hr->rank == 0, wrap_hi = nb_hi && hr->rank == hr->n_ranks - 1;
  auto exchange_pair = [&](bool e_side) -> int {
    const int c0 = e_side ? 0 : 3;
    NCCLCHK(hr, ncclGroupStart());
    for (FdtdSolver* h : both)
      for (int c = c0; c < c0 + 2; ++c) {
        float* f = field_ptr(h, c);
        if (!e_side && nb_hi) NCCLCHK(hr, ncclSend(f + (long long)(nz - 1) * pc, pc, ncclFloat, hi_rank, hr->comm, st));
        if (e_side && nb_lo) NCCLCHK(hr, ncclSend(f, pc, ncclFloat, lo_rank, hr->comm, st));
      }
    for (FdtdSolver* h : both)
      for (int c = c0; c < c0 + 2; ++c) {
        float* f = field_ptr(h, c);
        if (!e_side && nb_lo) NCCLCHK(hr, ncclRecv(f - pc, pc, ncclFloat, lo_rank, hr->comm, st));
        if (e_side && nb_hi) NCCLCHK(hr, ncclRecv(f + (long long)nz * pc, pc, ncclFloat, hi_rank, hr->comm, st));
      }
    NCCLCHK(hr, ncclGroupEnd());
    if (!e_side && wrap_lo) { plane(3, -1, -1, -sph[2]); plane(4, -1, -1, -sph[2]); }       // exp(-i phi_z), in place
    if (e_side && wrap_hi) { plane(0, nz, nz, sph[2]); plane(1, nz, nz, sph[2]); }          // exp(+i phi_z)
    return 0;
  };
  int64_t done = 0;
  for (; done < n_steps; ++done) {
    const long long n = hr->step;
    bool rec = false;
    for (Monitor& m : hr->mons) if (m.next < m.steps.size() && m.steps[m.next] == n) rec = true;
    if (rec) for (FdtdSolver* h : both) record_monitors(h, n, false, st);
    // H-side corrections of both parts, then the ghost cells (they must carry the corrections too)
    for (FdtdSolver* h : both) {
      launch_damp(h, false, 0, nz, st);
      launch_sources(h, false, n, 0, nz, st);
      launch_pml(h, false, 0, nz, st);
    }
    fill_xy();
    if (fused) {
      if (per_z) {       // ghost(-1) = exp(-i phi_z) [E (3 comps), H_x, H_y][nz-1];  ghost(nz) = exp(+i phi_z) E_{x,y}[0]
        for (int c = 0; c < 5; ++c) plane(c, -1, nz - 1, -sph[2]);
        plane(0, nz, 0, sph[2]);
        plane(1, nz, 0, sph[2]);
      }
      for (FdtdSolver* h : both) if (launch_fused(h, st, 0)) return -1;
      if (rec) for (FdtdSolver* h : both) record_monitors(h, n, true, st);
      for (FdtdSolver* h : both) {
        launch_pml(h, true, 0, nz, st);
        launch_sources(h, true, n, 0, nz, st);
        launch_damp(h, true, 0, nz, st);
        launch_ade(h, 0, nz, st);
      }
    } else {
      if (per_z) { plane(0, nz, 0, sph[2]); plane(1, nz, 0, sph[2]); }      // E ghost(nz) of E^n (first step / after set_field)
      if (multi && exchange_pair(true)) return -1;                          // ... or the upper neighbour's bottom plane
      for (FdtdSolver* h : both) launch_h_main(h, 0, nz, st);
      if (per_z) { plane(3, -1, nz - 1, -sph[2]); plane(4, -1, nz - 1, -sph[2]); }
      else if (!nb_lo) for (FdtdSolver* h : both) fill_ghost_h(h, st);
      if (multi && exchange_pair(false)) return -1;
      if (rec) for (FdtdSolver* h : both) record_monitors(h, n, true, st);
      for (FdtdSolver* h : both) {
        launch_e_main(h, 0, nz, st);
        launch_pml(h, true, 0, nz, st);
        launch_sources(h, true, n, 0, nz, st);
        launch_damp(h, true, 0, nz, st);
        launch_ade(h, 0, nz, st);
      }
      if (!per_z && !nb_hi) for (FdtdSolver* h : both) fill_ghost_e(h, st);
    }
    hr->step = hi->step = n + 1;
    // ---------------- field decay / divergence (|E|^2 of both parts) ----------------
    if (hr->decay_every > 0 && (hr->step % hr->decay_every) == 0) {
      double en = 0.0;
      for (FdtdSolver* h : both) {
        double part = 0.0;
        if (eval_energy(h, st, &part)) { hr->err = h->err; return -1; }
        en += part;
      }
      if (multi) {            // sum over the ranks (same stream as every other RCCL call of this run)
        double* tmp = hr->energy_dev;
        HIPCHK(hr, hipMemcpyAsync(tmp, &en, sizeof(double), hipMemcpyHostToDevice, st));
        NCCLCHK(hr, ncclAllReduce(tmp, tmp, 1, ncclDouble, ncclSum, hr->comm, st));
        HIPCHK(hr, hipMemcpyAsync(&en, tmp, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hr, hipStreamSynchronize(st));
      }
      if (!std::isfinite(en)) {
        hr->stats.diverged = hi->stats.diverged = 1;
        ++done;
        break;
      }
      if (en > hr->energy_max) hr->energy_max = hi->energy_max = en;
      hr->stats.field_decay = hi->stats.field_decay = hr->energy_max > 0 ? en / hr->energy_max : 1.0;
      if (progress && progress(hr->step, 0.0, hr->stats.field_decay, user)) { ++done; break; }
      if (hr->shutoff > 0 && hr->step > hr->decay_ref && hr->stats.field_decay < hr->shutoff) {
        hr->stats.stopped_early = hi->stats.stopped_early = 1;
        ++done;
        break;
      }
    }
  }
  // leave the ghost cells of E^n, H^{n-1/2} consistent for whoever reads the fields next
  fill_xy();
  HIPCHK(hr, hipEventRecord(hr->ev1, st));
  HIPCHK(hr, hipStreamSynchronize(st));
  HIPCHK(hr, hipGetLastError());
  float ms = 0.f;
  HIPCHK(hr, hipEventElapsedTime(&ms, hr->ev0, hr->ev1));
  for (FdtdSolver* h : both) {
    h->stats.run_ms = ms;
    h->stats.steps_done = h->step;
    h->stats.h_kernel_ms = h->stats.e_kernel_ms = h->stats.fused_kernel_ms = 0.0;
    h->stats.h_kernel_launches = h->stats.e_kernel_launches = h->stats.fused_kernel_launches = 0;
  }
  return 0;
}

int fdtd_set_option(FdtdSolver* h, int key, int value) {
  if (!h) return -1;
  switch (key) {
    case FDTD_OPT_FLAGS: h->cfg.flags = value; return 0;
    case FDTD_OPT_VARIANT: h->cfg.variant = value; return 0;
    case FDTD_OPT_ZCHUNK: if (value < 1) break; h->zchunk = value; h->zchunk_f = value; h->user_geometry = true; return 0;
    case FDTD_OPT_ROWS: if (value < 1 || value > 15) break; h->rows = value > 8 ? 8 : value; h->rows_f = value; h->user_geometry = true; return 0;
    case FDTD_OPT_XCD_REMAP: if (value > 1024) break; h->xcd_remap = value < 0 ? -1 : value; return 0;
    case FDTD_OPT_PML_FUSED: h->pml_fused = value < 0 ? -1 : (value & 7); for (bool& ok : h->pml_blk_ok) ok = false; h->pml_blk2_ok = false; h->pml_blk_hole_ok = false; return 0;
    case FDTD_OPT_BND_PLANES: h->bnd_planes = value > 0 ? value : 0; return 0;
    case FDTD_OPT_AUTOTUNE: h->autotune = value < 0 ? 0 : (value > 2 ? 1 : value); if (value) h->tuned = false; return 0;
    case FDTD_OPT_MEM_HINTS: h->mem_hints = value != 0; return 0;
    case FDTD_OPT_PLACEMENT_TRIES: if (value < 0 || (value % 100) > 8) break; h->placement_tries = value; h->placement_done = false; return 0;
    case FDTD_OPT_LDS_PAD: if (value < 0 || value > 120000) break; h->lds_pad = value; return 0;
    case FDTD_OPT_TBLOCK: h->tblock = value < 0 ? -1 : value; return 0;
    case FDTD_OPT_EDGE_ZCHUNK: h->edge_zchunk = value < 0 ? -1 : value; return 0;
    case FDTD_OPT_GRAPH: h->use_graph = value < 0 ? -1 : (value != 0); return 0;
    case FDTD_OPT_TWOSTEP: {
      if (value <= 0) { h->twostep_w = value < 0 ? -1 : 0; h->twostep_zc = 0; return 0; }
      const int w = value % 64, zc = (value / 64) % 1024;
      if (w < 4 || w > 16) return fail(h, "FDTD_OPT_TWOSTEP: %d waves per workgroup (4 ... 16)", w);
      h->twostep_w = w;
      h->twostep_zc = zc;
      return 0;
    }
    case FDTD_OPT_PML_SPLIT: h->pml_split = value < 0 ? -1 : (value != 0); return 0;
    case FDTD_OPT_SHELL_PAIRS: h->shell_on = value < 0 ? -1 : value; return 0;
    case FDTD_OPT_STRIP: if (value % 64 < 1 || (value / 64 != 3 && value / 64 != 4)) break; h->strip_zc = value % 64; h->strip_occ = value / 64; return 0;
    case FDTD_OPT_SHELL2: h->shell2_on = value < 0 ? -1 : (value > 3 ? 1 : value); return 0;
    case FDTD_OPT_DEBUG_SYNC: h->debug_sync = value != 0; return 0;
    case FDTD_OPT_TILE_SPLIT: h->tile_split = value < 0 ? -1 : (value != 0); return 0;
    case FDTD_OPT_SRC_PAGED:
      h->spg_on = value < 0 ? -1 : (value != 0);
      if (h->spg.state == -1 && value != 0) h->spg.state = 0;
      return 0;
    case FDTD_OPT_SLAB_BOXES_FIRST: if (value < 0 || value > 3) break; h->slab_boxes_first = value; return 0;
    case FDTD_OPT_WHATIF: if (value < 0 || value > 15) break; h->whatif = value; return 0;
    case FDTD_OPT_DISP:
      if (h->disp.state == 1 && value == 0) break;       // (every ADE launch keeps the paged memory terms by now: set it before the first run)
      h->disp_on = value < 0 ? -1 : (value != 0);
      if (h->disp.state == -1 && value != 0) h->disp.state = 0;
      return 0;
    case FDTD_OPT_SHELL2_SHAPE: {
      // lanes per row of the wide boxes (3 ... 64) + 128 * their waves per workgroup (1 ... 8) + 1024 * their planes per chunk (0 = by box)
      //   + 2^17 * waves per workgroup of the strips (1 ... 8) + 2^21 * their planes per chunk (0 = by box)
      if (value <= 0) { h->shell2_qw = 0; h->shell2_ww = 8; h->shell2_zcw = 0; h->shell2_ws = 6; h->shell2_zcs = 0; return 0; }
      const int qw = value % 128, ww = (value >> 7) % 8 + ((value >> 7) % 8 == 0 ? 8 : 0), zcw = (value >> 10) % 128;
      const int ws = (value >> 17) % 16, zcs = (value >> 21) % 128;
      if ((qw != 0 && qw < 3) || qw > 64 || ws > 8) break;
      h->shell2_qw = qw; h->shell2_ww = ww; h->shell2_zcw = zcw;
      h->shell2_ws = ws > 0 ? ws : 6; h->shell2_zcs = zcs;
      return 0;
    }
    case FDTD_OPT_FUSED_LB: if (value != 0 && value != 256 && value != 512 && value != 1024) break; h->fused_lb = value; return 0;
    default: break;
  }
  return fail(h, "fdtd_set_option: bad key/value %d/%d", key, value);
}

int fdtd_get_stats(FdtdSolver* h, FdtdStats* out) {
  if (!h || !out) return -1;
  *out = h->stats;
  out->tile_rows = h->rows_f;
  out->tile_zchunk = h->last_zc > 0 ? h->last_zc : h->zchunk_f;      // what the last sweep used
  out->tile_order = h->xcd_remap < 0 ? kTileRun : h->xcd_remap;
  out->placement = (h->placement_tried << 8) | h->placement_kept;
  out->placement_ms_first = h->placement_ms[0];
  out->placement_ms_kept = h->placement_ms[h->placement_kept];
  out->stream_overlap = h->stream_overlap;
  out->stream_retries = h->stream_retries;
  out->two_step_pairs = h->two_step_pairs;
  out->tblock_planes = h->tblock_used;
  out->graph_pairs = h->graph_pairs;
  out->reserved0 = h->graph_status;
  out->fused2_pairs = h->fused2_pairs;
  out->fused2_shape = h->fused2_pairs ? (h->twostep_w_used | (h->twostep_zc_used << 6)) : 0;
  out->shell_pairs = h->shell_pairs;
  out->shell2_pairs = h->shell2_pairs;
  out->fused2_off_reason = h->fused2_pairs ? 0 : (h->f2_off_reason ? h->f2_off_reason : h->f2_dyn_reason);
  out->disp_pairs = h->disp.pairs;
  out->single_step_reason = h->f2_dyn_reason;
  out->src_paged_pairs = (int32_t)std::min<long long>(h->spg.pairs, 0x7fffffff);
  out->struct_bytes = (int32_t)sizeof(FdtdStats);
  return 0;
}

}  // extern "C"
