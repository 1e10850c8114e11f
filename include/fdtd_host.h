/* Host-side helpers of the set-up phase (no GPU, no HIP runtime): the passes of the rasteriser that NumPy spends whole-array
 * temporaries on, worked through plane by plane on a pool of threads.  Built by `python -m tidy3d_amd.build` into
 * tidy3d_amd/libfdtd_host.so (g++ -O3 -pthread); bound by tidy3d_amd/host.py.  Every function computes exactly what the NumPy
 * statements it replaces compute (tidy3d_amd/discretize.py keeps them as the checker: tests/test_host_raster.py).
 * The reference rasterises in its cloud solver; what the open-source package holds of it is the point-sampling rule
 * (ref tidy3d/components/simulation.py:1135-1241, geometry/base.py `inside`). */
#ifndef FDTD_HOST_H
#define FDTD_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* dst[0 .. n) = value, on `threads` threads (first touch of fresh pages included). */
void fdtd_host_fill_u16(uint16_t* dst, int64_t n, uint16_t value, int threads);

/* Interface nodes of one E component's material-index volume m[nz][ny][nx] (discretize._subpixel_average): nodes of the planes
 * with zflag[k] != 0 where a face neighbour inside the array holds another index; kept where the node's medium and every
 * differing neighbour's are plain dielectrics (plain[index] != 0, index < n_table).  Returns the number of nodes n and a scan
 * object; fdtd_host_interface_nodes_take writes them in (k, j, i) lexicographic order into kji[3][n] (the k of every node, then
 * j, then i) and bits[n] (bit a: the medium changes along axis a, 0 = x) and releases the scan (kji = NULL: only releases).
 * -1: out of memory / an index beyond the table. */
int64_t fdtd_host_interface_nodes(const uint16_t* m, int nz, int ny, int nx, const uint8_t* zflag, const uint8_t* plain, int n_table,
                                  void** scan, int threads);
void fdtd_host_interface_nodes_take(void* scan, int64_t* kji, uint8_t* bits, int threads);

/* Media of the sub-pixel samples of n interface nodes (discretize._subpixel_average): lo / hi = the nodes' control volumes
 * [3][n]; line != 0: 8 samples along axis which[q] through the volume's middle, else 4 x 4 x 4.  Structures in their order
 * (later ones take samples over): s_type 0 Box (s_par = centre[3], half sizes[3]), 1 Sphere (centre[3], r^2), 2 upright
 * Cylinder (centre[3], radius, half length, -, -, axis); s_bounds[6] = (min[3], max[3]) — a structure is only asked where its
 * bounds meet the node's volume; s_mi = its table index.  idx[n][8 or 64] receives the samples' indices (background where no
 * structure holds the point).  -1: a structure type this function does not evaluate (the caller keeps its NumPy pass). */
int fdtd_host_sample_media(int64_t n, int line, const double* lo, const double* hi, const uint8_t* which, int n_structs,
                           const int32_t* s_type, const double* s_par, const double* s_bounds, const uint16_t* s_mi,
                           uint16_t background, uint16_t* idx, int threads);

void fdtd_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
