// Host-side passes of the rasteriser (include/fdtd_host.h): plain C++ on a pool of threads, no HIP.
#include "../../include/fdtd_host.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

template <class F>
void on_threads(int threads, F f) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(f, t);
    f(0);
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" void fdtd_host_fill_u16(uint16_t* dst, int64_t n, uint16_t value, int threads) {
    if (threads < 1) threads = 1;
    const int64_t step = (n + threads - 1) / threads;
    on_threads(threads, [&](int t) {
        const int64_t b = t * step, e = b + step < n ? b + step : n;
        for (int64_t i = b; i < e; ++i) dst[i] = value;
    });
}

namespace {
struct Found { std::vector<int32_t> ji; std::vector<uint8_t> b; };
struct NodeScan { std::vector<int> planes; std::vector<Found> found; int64_t n = 0; };
}  // namespace

extern "C" int64_t fdtd_host_interface_nodes(const uint16_t* m, int nz, int ny, int nx, const uint8_t* zflag, const uint8_t* plain,
                                             int n_table, void** scan, int threads) {
    *scan = nullptr;
    NodeScan* sc = new (std::nothrow) NodeScan;
    if (!sc) return -1;
    std::vector<int>& planes = sc->planes;
    for (int k = 0; k < nz; ++k)
        if (zflag[k]) planes.push_back(k);
    std::vector<Found>& found = sc->found;
    found.resize(planes.size());
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const int64_t sy = nx, sz = int64_t(nx) * ny;
    on_threads(threads, [&](int) {
        for (;;) {
            const size_t q = next.fetch_add(1);
            if (q >= planes.size()) return;
            const int k = planes[q];
            Found& f = found[q];
            const uint16_t* pk = m + sz * k;
            for (int j = 0; j < ny; ++j) {
                const uint16_t* r = pk + sy * j;
                const uint16_t* rym = j > 0 ? r - sy : r;           // (a neighbour beyond the array: the node itself — no change)
                const uint16_t* ryp = j + 1 < ny ? r + sy : r;
                const uint16_t* rzm = k > 0 ? r - sz : r;
                const uint16_t* rzp = k + 1 < nz ? r + sz : r;
                for (int i = 0; i < nx; ++i) {
                    const uint16_t me = r[i];
                    const uint16_t xm = i > 0 ? r[i - 1] : me, xp = i + 1 < nx ? r[i + 1] : me;
                    const uint16_t ym = rym[i], yp = ryp[i], zm = rzm[i], zp = rzp[i];
                    const unsigned b = unsigned(xm != me || xp != me) | unsigned(ym != me || yp != me) << 1 | unsigned(zm != me || zp != me) << 2;
                    if (!b) continue;
                    const uint16_t all[7] = {me, xm, xp, ym, yp, zm, zp};
                    bool keep = true;
                    for (int n = 0; n < 7; ++n) {
                        if (all[n] >= n_table) { bad.store(1); keep = false; break; }
                        if ((n == 0 || all[n] != me) && !plain[all[n]]) { keep = false; break; }
                    }
                    if (!keep) continue;
                    f.ji.push_back(j);
                    f.ji.push_back(i);
                    f.b.push_back(uint8_t(b));
                }
            }
        }
    });
    if (bad.load()) {
        delete sc;
        return -1;
    }
    for (auto& f : found) sc->n += int64_t(f.b.size());
    *scan = sc;
    return sc->n;
}

extern "C" void fdtd_host_interface_nodes_take(void* scan, int64_t* kji, uint8_t* bits, int threads) {
    NodeScan* sc = static_cast<NodeScan*>(scan);
    if (!sc) return;
    const int64_t n = sc->n;
    std::vector<int64_t> at(sc->planes.size() + 1, 0);
    for (size_t q = 0; q < sc->planes.size(); ++q) at[q + 1] = at[q] + int64_t(sc->found[q].b.size());
    std::atomic<size_t> next{0};
    if (kji && bits)
        on_threads(threads, [&](int) {
            for (;;) {
                const size_t q = next.fetch_add(1);
                if (q >= sc->planes.size()) return;
                const Found& f = sc->found[q];
                int64_t w = at[q];
                for (size_t e = 0; e < f.b.size(); ++e, ++w) {
                    kji[w] = sc->planes[q];
                    kji[n + w] = f.ji[2 * e];
                    kji[2 * n + w] = f.ji[2 * e + 1];
                    bits[w] = f.b[e];
                }
            }
        });
    delete sc;
}

extern "C" void fdtd_host_free(void* p) { std::free(p); }

// ---- media of the sub-pixel samples (discretize._subpixel_average) ----------------------------------------------------------------
// The NumPy statements restated operation for operation (IEEE doubles, no contraction: the build passes -ffp-contract=off):
//   line nodes : 8 samples  lo + (hi - lo) * t,  t = (s + 1/2) / 8  along the axis of change, 0.5 * (lo + hi) along the others
//   other nodes: 4 x 4 x 4 samples  lo + (hi - lo) * (s + 1/2) / 4  per axis, x slowest
//   structures in order (a later one takes the sample over), each only where its bounds meet the node's control volume
//   Box      |p - c| <= half per axis                                  (schema.Box.inside)
//   Sphere   ((x - x0)^2 + (y - y0)^2) + (z - z0)^2 <= r2              (schema.Sphere.inside)
//   Cylinder r > 0 and (p0 - c0)^2 + (p1 - c1)^2 <= r * r and |za - z0| <= half length, upright walls   (schema.Cylinder.inside)
extern "C" int fdtd_host_sample_media(int64_t n, int line, const double* lo, const double* hi, const uint8_t* which, int n_structs,
                                      const int32_t* s_type, const double* s_par, const double* s_bounds, const uint16_t* s_mi,
                                      uint16_t background, uint16_t* idx, int threads) {
    for (int s = 0; s < n_structs; ++s)
        if (s_type[s] < 0 || s_type[s] > 2) return -1;
    const int S = line ? 8 : 64;
    if (threads < 1) threads = 1;
    const int64_t step = (n + threads - 1) / threads;
    on_threads(threads, [&](int t) {
        const int64_t b = t * step, e = b + step < n ? b + step : n;
        double px[64], py[64], pz[64];
        for (int64_t q = b; q < e; ++q) {
            const double l[3] = {lo[q], lo[n + q], lo[2 * n + q]}, h[3] = {hi[q], hi[n + q], hi[2 * n + q]};
            double* P[3] = {px, py, pz};
            if (line) {
                for (int a = 0; a < 3; ++a) {
                    const double mid = 0.5 * (l[a] + h[a]), d = h[a] - l[a];
                    for (int s = 0; s < 8; ++s) P[a][s] = which[q] == a ? l[a] + d * ((s + 0.5) / 8) : mid;
                }
            } else {
                double v[3][4];
                for (int a = 0; a < 3; ++a)
                    for (int s = 0; s < 4; ++s) v[a][s] = l[a] + (h[a] - l[a]) * ((s + 0.5) / 4);
                for (int s = 0; s < 64; ++s) {
                    px[s] = v[0][s >> 4];
                    py[s] = v[1][(s >> 2) & 3];
                    pz[s] = v[2][s & 3];
                }
            }
            uint16_t* o = idx + q * S;
            for (int s = 0; s < S; ++s) o[s] = background;
            for (int g = 0; g < n_structs; ++g) {
                const double* bb = s_bounds + 6 * g;
                if (!(h[0] >= bb[0] && l[0] <= bb[3] && h[1] >= bb[1] && l[1] <= bb[4] && h[2] >= bb[2] && l[2] <= bb[5])) continue;
                const double* p = s_par + 8 * g;
                const uint16_t mi = s_mi[g];
                if (s_type[g] == 0) {
                    for (int s = 0; s < S; ++s)
                        if (std::fabs(px[s] - p[0]) <= p[3] && std::fabs(py[s] - p[1]) <= p[4] && std::fabs(pz[s] - p[2]) <= p[5]) o[s] = mi;
                } else if (s_type[g] == 1) {
                    for (int s = 0; s < S; ++s) {
                        const double dx = px[s] - p[0], dy = py[s] - p[1], dz = pz[s] - p[2];
                        if (dx * dx + dy * dy + dz * dz <= p[3]) o[s] = mi;
                    }
                } else {
                    const int ax = int(p[7]);
                    const double* A = P[ax];
                    const double* B0 = P[ax == 0 ? 1 : 0];
                    const double* B1 = P[ax == 2 ? 1 : 2];
                    const double c0 = p[ax == 0 ? 1 : 0], c1 = p[ax == 2 ? 1 : 2], z0 = p[ax];
                    for (int s = 0; s < S; ++s) {
                        const double r = p[3] + 0.0 * A[s];
                        const double d0 = B0[s] - c0, d1 = B1[s] - c1;
                        if (r > 0 && d0 * d0 + d1 * d1 <= r * r && std::fabs(A[s] - z0) <= p[4]) o[s] = mi;
                    }
                }
            }
        }
    });
    return 0;
}
