"""SolverSpec: the tidy3d-free, plain-array statement of one FDTD run.

This is the stratum between the schema (``tidy3d_amd.schema`` / a real
``tidy3d.Simulation``) and the two consumers that must agree bit-for-bit on
*what* is being solved: the HIP engine (``tidy3d_amd.engine`` ->
``libfdtd_hip.so``) and the fp64 NumPy oracle (``oracle/fdtd_numpy.py``).
Everything here is numpy arrays and POD scalars (SURVEY.md section 7 "Design stance").

Array layout convention (identical on host, in the oracle and in HBM):
fields are ``[nz][ny][nx]`` C-ordered, i.e. **x is the fastest index**; an
xy-plane of one component is one contiguous block (what a z-slab halo
exchange sends).  Yee staggering follows reference grid/grid.py:465-491:
``Ex[k,j,i]`` sits at (xc[i], yb[j], zb[k]), ``Hx[k,j,i]`` at (xb[i], yc[j], zc[k]),
with ``b`` = cell boundaries and ``c`` = cell centres.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

COMPONENTS = ("Ex", "Ey", "Ez", "Hx", "Hy", "Hz")
COMP_ID = {c: i for i, c in enumerate(COMPONENTS)}

BC_PEC = 0        # tangential E = 0 on the wall (also what backs every PML)
BC_PMC = 1        # tangential H = 0 on the wall (supported on the min side only)
BC_PERIODIC = 2


@dataclass
class MediumCoeffs:
    """One entry of the material table; index 0 is always PEC."""

    eps_inf: float = 1.0
    sigma: float = 0.0                                   # S/um
    poles: Tuple[Tuple[complex, complex], ...] = ()      # (a_k, c_k) rad/s, ref medium.py:2900
    pec: bool = False
    name: str = ""


@dataclass
class AnisoSet:
    """Off-diagonal coupling of one E component inside fully anisotropic bodies (ref medium.py:5058).  The sweep advances
    E_a with the diagonal part, (dt / eps0) [eps^-1]_aa curl_a (a table medium of permittivity 1 / [eps^-1]_aa); this adds

        dE_a(i) = (dt / eps0) sum_{b != a} sum_{j in the four E_b nodes around i} g(i, j) curl_b(j),
        g(i, j) = ([eps^-1]_ab at i + [eps^-1]_ab at j) / 8        (0 outside such bodies)

    — the symmetric average keeps the discrete operator self-adjoint (stable, reciprocal) across body faces; curl_b(j) is
    recovered from what the sweep did to E_b(j): (E_b^{n+1} - Ca E_b^n) / Cb."""

    comp: int                         # a
    ijk: np.ndarray                   # [n, 3] nodes (i, j, k)
    nbr_comp: Tuple[int, ...]         # [8]: slots 0-3 component (a+1)%3, 4-7 component (a+2)%3
    nbr_ijk: np.ndarray               # [n, 8, 3]; -1 = no node (beyond a non-periodic wall)
    g: np.ndarray                     # [n, 8] float64


@dataclass
class PmlFace:
    """CPML profile of one face (ref boundary.py:195-254; units of sigma/alpha: 2 eps0/dt)."""

    num_layers: int = 0
    sigma_order: int = 3
    sigma_min: float = 0.0
    sigma_max: float = 1.5
    kappa_order: int = 3
    kappa_min: float = 1.0
    kappa_max: float = 3.0
    alpha_order: int = 1
    alpha_min: float = 0.0
    alpha_max: float = 0.0


@dataclass
class PointSourceSet:
    """A current source as a list of Yee points with complex weights and one complex
    waveform:  F[idx] += Re( (w_re + i w_im) * wave[n] ).

    E components (electric current J) are added during the E-update with the waveform
    sampled at t_{n+1/2}; H components (magnetic current M) during the H-update with the
    waveform sampled at t_n.  The weights already contain the update coefficient of the
    cell (-Cb / cell measure for J, -(dt/mu0) / cell measure for M)."""

    comp: np.ndarray            # int32 [n]   component id 0..5
    ijk: np.ndarray             # int32 [n,3] (i, j, k) global Yee indices
    w_re: np.ndarray            # float64 [n]
    w_im: np.ndarray            # float64 [n]
    wave_e: np.ndarray          # complex128 [n_steps]  amp(t_n + dt/2)
    wave_h: np.ndarray          # complex128 [n_steps]  amp(t_n)
    name: str = ""


@dataclass
class TfsfSpec:
    """Axis-aligned total-field/scattered-field source, reduced to plain lists.

    A 1-D auxiliary Yee grid along the propagation axis carries the incident plane wave
    (``e1`` on the n_aux+1 cell boundaries, ``h1`` on the n_aux centres; same steps and dt
    as the 3-D grid, so its numerical dispersion matches the 3-D grid's exactly for
    axis-aligned propagation).  Per step:

      H-phase:  H[comp][ijk] += h_corr_w * e1[h_corr_aux]       (e1 at t_n)
                h1 = ah * h1 - bh * (e1[1:] - e1[:-1])
      E-phase:  E[comp][ijk] += e_corr_w * h1[e_corr_aux]       (h1 at t_n + dt/2)
                e1[1:-1] = ae[1:-1] * e1[1:-1] - be[1:-1] * (h1[1:] - h1[:-1]);  e1[0] = e1[-1] = 0
                e1[src_cell] += wave[n]

    The 1-D grid extends beyond the 3-D domain and ends in matched lossy pads (graded electric
    and magnetic conductivity, reflection-free in 1-D), which is what the per-node decay / update
    coefficients ae, be, ah, bh encode.

    The correction lists (built on the host by ``tidy3d_amd.discretize``) hold every 3-D
    node whose curl stencil straddles the TFSF surface, with the signed update coefficient
    of that node folded into the weight."""

    n_aux: int
    ae: np.ndarray              # float64 [n_aux+1]  decay of e1
    be: np.ndarray              # float64 [n_aux+1]  update coefficient of e1 (sign included)
    ah: np.ndarray              # float64 [n_aux]    decay of h1
    bh: np.ndarray              # float64 [n_aux]    update coefficient of h1 (sign included)
    src_cell: int
    wave: np.ndarray            # float64 [n_steps]
    e_corr_comp: np.ndarray     # int32 [ne]
    e_corr_ijk: np.ndarray      # int32 [ne, 3]
    e_corr_w: np.ndarray        # float64 [ne]
    e_corr_aux: np.ndarray      # int32 [ne]
    h_corr_comp: np.ndarray     # int32 [nh]  (3..5)
    h_corr_ijk: np.ndarray      # int32 [nh, 3]
    h_corr_w: np.ndarray        # float64 [nh]
    h_corr_aux: np.ndarray      # int32 [nh]
    name: str = ""


@dataclass
class MonitorSpec:
    """A recorder over a raw Yee index box [lo, hi) (same box for every component).

    kind "time": every recorded step stores the raw component values (H averaged to t_n).
    kind "dft" : running DFT  acc[f] += field * phase[n, f];  E uses phase_e (time t_n),
                 H uses phase_h (time t_n + dt/2).  ``steps`` lists the time-step indices
                 on which the monitor records."""

    kind: str
    comps: Tuple[int, ...]
    lo: Tuple[int, int, int]
    hi: Tuple[int, int, int]
    steps: np.ndarray                                   # int64 [n_rec]
    freqs: Optional[np.ndarray] = None                  # float64 [nf]          (dft)
    phase_e: Optional[np.ndarray] = None                # complex128 [n_rec,nf] (dft)
    phase_h: Optional[np.ndarray] = None                # complex128 [n_rec,nf] (dft)
    name: str = ""
    # how the phase tables were formed (the oracle re-derives them from these): recording stride in steps
    # and the apodisation window (start, end, width) or None
    stride: Optional[int] = None
    apod: Optional[Tuple[Optional[float], Optional[float], Optional[float]]] = None

    @property
    def shape(self) -> Tuple[int, int, int]:
        """(bz, by, bx) extents of the recorded box."""
        return (self.hi[2] - self.lo[2], self.hi[1] - self.lo[1], self.hi[0] - self.lo[0])


@dataclass
class SolverSpec:
    shape: Tuple[int, int, int]                         # (nx, ny, nz) cells incl. PML cells
    boundaries: Tuple[np.ndarray, np.ndarray, np.ndarray]   # cell boundaries, N+1 each
    dt: float
    n_steps: int
    bc: Tuple[Tuple[int, int], ...] = ((0, 0), (0, 0), (0, 0))   # [axis][minus, plus]
    pml: Tuple[Tuple[PmlFace, PmlFace], ...] = None
    media: List[MediumCoeffs] = field(default_factory=list)      # [0] = PEC, [1] = background
    mat_idx: Optional[np.ndarray] = None                # uint16 [3, nz, ny, nx] (< 1024); None = all [1]
    sources: List[PointSourceSet] = field(default_factory=list)
    tfsf: List[TfsfSpec] = field(default_factory=list)
    monitors: List[MonitorSpec] = field(default_factory=list)
    # Absorber layers (ref boundary.py:427): per axis (sigma at cell boundaries [N], sigma at cell
    # centres [N], n_lo, n_hi), sigma in units of 2 eps0/dt; None = no absorber (coeffs.damping_tables)
    absorber: Optional[List[Tuple[np.ndarray, np.ndarray, int, int]]] = None
    # Bloch boundaries (ref boundary.py:55-79): phase advance 2 pi bloch_vec per axis across the domain,
    # F(r + L_a) = exp(i bloch[a]) F(r), on axes whose bc is BC_PERIODIC; None = real fields
    bloch: Optional[Tuple[float, float, float]] = None
    # PMC on PLUS faces (ref boundary.py:45 PMCBoundary on any face): per axis the index N of the wall (a cell boundary), or
    # -1.  The grid then carries two ghost cells beyond the wall (shape[a] == N + 2): the wall nodes themselves (tangential
    # E, normal H at index N) are unknowns, everything beyond is the mirror image of the inside — E_tan, H_norm even,
    # E_norm, H_tan odd — refreshed at the start of every step (discretize._discretize_pmc_plus, kernel mirror_fill_kernel)
    mirror_plus: Optional[Tuple[int, int, int]] = None
    aniso: List[AnisoSet] = field(default_factory=list)   # fully anisotropic bodies: off-diagonal coupling per E component
    shutoff: float = 0.0                                # 0 disables the early stop
    decay_every: int = 0                                # 0 = never evaluate field decay
    decay_ref_step: int = 0                             # steps before this never shut off

    def __post_init__(self):
        if self.pml is None:
            self.pml = tuple((PmlFace(), PmlFace()) for _ in range(3))
        if not self.media:
            self.media = [MediumCoeffs(pec=True, name="PEC"), MediumCoeffs(name="vacuum")]

    # ---- derived 1-D geometry ------------------------------------------------------
    @property
    def n_cells(self) -> int:
        nx, ny, nz = self.shape
        return nx * ny * nz

    def primal_steps(self, axis: int) -> np.ndarray:
        """d[i] = b[i+1] - b[i]   (ref grid.py:393-401)."""
        return np.diff(np.asarray(self.boundaries[axis], dtype=np.float64))

    def dual_steps(self, axis: int) -> np.ndarray:
        """dd[i] = (d[i] + d[i-1]) / 2; d[-1] wraps for periodic axes and mirrors (= d[0])
        otherwise (ref grid.py:404-417 uses the periodic roll; the mirror value is only ever
        used by the PMC rule since PEC zeroes the wall component)."""
        d = self.primal_steps(axis)
        prev = np.roll(d, 1)
        if self.bc[axis][0] != BC_PERIODIC:
            prev[0] = d[0]
        return 0.5 * (d + prev)

    def centers(self, axis: int) -> np.ndarray:
        b = np.asarray(self.boundaries[axis], dtype=np.float64)
        return 0.5 * (b[1:] + b[:-1])

    def yee_coords(self, comp: int):
        """(x, y, z) 1-D coordinate arrays (length N each) of component ``comp``
        (ref grid.py:465-491)."""
        is_h = comp >= 3
        a = comp % 3
        out = []
        for ax in range(3):
            b = np.asarray(self.boundaries[ax], dtype=np.float64)
            on_center = (ax == a) != is_h      # E: centre along own axis; H: centre along others
            out.append(0.5 * (b[1:] + b[:-1]) if on_center else b[:-1])
        return tuple(out)
