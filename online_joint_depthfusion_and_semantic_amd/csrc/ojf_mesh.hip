// MESH: iso-surface of a fused fp16 TSDF volume as a triangle soup (the reference exports
// meshes with skimage's marching cubes on the host, modules/database.py:118-139,203-261 + utils/saving.py:42-47).
//
// Own design, not a restatement: marching TETRAHEDRA on the Kuhn decomposition of every cell (six tetrahedra around
// the 0-6 body diagonal, identical in every cell, so faces match across cells and the surface is watertight).  A
// tetrahedron has 16 sign cases and 0, 1 or 2 triangles, derived on the fly from its four corner signs - no case
// table.  Three launches: COUNT (every 64x4-cell block writes its number of triangles), SCAN (exclusive prefix over
// the blocks, total to the caller), EMIT (every block writes at its offset, threads ordered by a block scan) - no
// atomics anywhere, so the triangle list is the same on every run.  Between count and emit the host reads ONE
// integer to size the buffer.  Streaming, HBM-bound: 2 B/voxel of compulsory traffic per pass (neighbouring
// threads share the 8 corner loads through L1/L2) plus 36 B per triangle written.
//
// A cell is skipped when any of its corners is unobserved (weight 0, when a weight volume is given) or NaN.
// Vertices are origin + voxel_index * resolution; with origin = 0 that is the frame of the reference's meshes
// (marching_cubes(..., spacing=voxel_size), modules/database.py:120-122).  Triangles are oriented so that the
// normal points from the negative (inside) to the positive (free space) side.
#include "ojf_common.h"

namespace ojf {

struct MeshArgs {
    const uint16_t *tsdf;
    const uint16_t *wgt;   // NULL: every voxel counts as observed
    const uint8_t *ids;    // NULL: no labels
    float *verts;          // [cap, 3 vertices, 3 coords]
    uint8_t *labels;       // [cap, 3] label of the voxel nearest to each vertex
    uint64_t *keys;        // [cap, 3] id of the grid edge a vertex sits on: 8 * voxel + direction code
    unsigned int *blocks;  // [n_blocks] triangles per block, after the scan: offset of the block
    unsigned int cap;
    int X, Y, Z;
    float iso;
    double origin[3], res;
};

// corners of the unit cell and the six Kuhn tetrahedra around the diagonal 0-6
__device__ __constant__ int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ int kTet[6][4] = {{0, 5, 1, 6}, {0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}};

struct Tri {
    float p[3][3];
    int e[3][2];  // the two tetrahedron corners (0..3) of the edge each vertex sits on
};

// crossing point of an edge; always evaluated from the lower-valued end so that the same edge gives the same
// bits in every tetrahedron and cell that shares it
__device__ __forceinline__ void lerp(const float pa[3], const float pb[3], float va, float vb, float iso, float out[3])
{
    const bool swap = va > vb;
    const float *p0 = swap ? pb : pa, *p1 = swap ? pa : pb;
    const float v0 = swap ? vb : va, v1 = swap ? va : vb;
    const float t = (iso - v0) / (v1 - v0);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = p0[i] + t * (p1[i] - p0[i]);
}

// triangles of one tetrahedron (corner positions p, values v); returns their number
__device__ __forceinline__ int tet_triangles(const float p[4][3], const float v[4], float iso, Tri out[2])
{
    int in[4], n_in = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        in[i] = v[i] < iso;
        n_in += in[i];
    }
    if (n_in == 0 || n_in == 4) return 0;
    int a[4], na = 0, b[4], nb = 0;  // a: the minority side, b: the others
    const int minority = n_in <= 2 ? 1 : 0;
    for (int i = 0; i < 4; ++i) {
        if (in[i] == minority) a[na++] = i;
        else b[nb++] = i;
    }
    int n = 0;
    if (na == 1) {  // one corner cut off: the three edges leaving it
        for (int k = 0; k < 3; ++k) {
            lerp(p[a[0]], p[b[k]], v[a[0]], v[b[k]], iso, out[0].p[k]);
            out[0].e[k][0] = a[0];
            out[0].e[k][1] = b[k];
        }
        n = 1;
    } else {        // two against two: a quad on the four crossing edges, split along q0-q2
        const int qa[4] = {a[0], a[0], a[1], a[1]}, qb[4] = {b[0], b[1], b[1], b[0]};
        float q[4][3];
        for (int k = 0; k < 4; ++k) lerp(p[qa[k]], p[qb[k]], v[qa[k]], v[qb[k]], iso, q[k]);
        const int pick[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (int t = 0; t < 2; ++t)
            for (int k = 0; k < 3; ++k) {
                for (int i = 0; i < 3; ++i) out[t].p[k][i] = q[pick[t][k]][i];
                out[t].e[k][0] = qa[pick[t][k]];
                out[t].e[k][1] = qb[pick[t][k]];
            }
        n = 2;
    }
    // orientation: normal along (a positive corner) - (a negative corner)
    int neg = 0, pos = 0;
    for (int i = 0; i < 4; ++i) {
        if (in[i]) neg = i;
        else pos = i;
    }
    const float dir[3] = {p[pos][0] - p[neg][0], p[pos][1] - p[neg][1], p[pos][2] - p[neg][2]};
    for (int t = 0; t < n; ++t) {
        const float *A = out[t].p[0], *B = out[t].p[1], *C = out[t].p[2];
        const float u[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, w[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
        const float nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
        if (nx * dir[0] + ny * dir[1] + nz * dir[2] < 0.0f) {
            for (int i = 0; i < 3; ++i) {
                const float tmp = out[t].p[1][i];
                out[t].p[1][i] = out[t].p[2][i];
                out[t].p[2][i] = tmp;
            }
            for (int i = 0; i < 2; ++i) {
                const int tmp = out[t].e[1][i];
                out[t].e[1][i] = out[t].e[2][i];
                out[t].e[2][i] = tmp;
            }
        }
    }
    return n;
}

constexpr int kTileZ = 64, kTileY = 4;  // one wavefront = 64 consecutive cells along z (the contiguous axis)

// block = 64 x 4 cells at one x; blockIdx = (z tile, y tile, x).  EMIT = false: blocks[b] = triangles of the block;
// EMIT = true: blocks[b] holds the block's offset in the output.
template <bool EMIT>
__global__ __launch_bounds__(kTileZ * kTileY) void mesh_kernel(MeshArgs a)
{
    __shared__ unsigned int wave_n[kTileY];
    const int ck = blockIdx.x * kTileZ + threadIdx.x, cj = blockIdx.y * kTileY + threadIdx.y, ci = blockIdx.z;
    const unsigned int block = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float v[8];
    unsigned int n = 0;
    if (ck < a.Z - 1 && cj < a.Y - 1) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t lin = ((size_t)(ci + kCorner[c][0]) * a.Y + (cj + kCorner[c][1])) * a.Z + (ck + kCorner[c][2]);
            v[c] = h2f(a.tsdf[lin]);
            ok = ok && v[c] == v[c] && (!a.wgt || h2f(a.wgt[lin]) > 0.0f);
        }
        if (ok) {
            unsigned int in = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) in |= (v[c] < a.iso ? 1u : 0u) << c;
            if (in != 0 && in != 0xffu) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const unsigned int k = ((in >> kTet[t][0]) & 1) + ((in >> kTet[t][1]) & 1) + ((in >> kTet[t][2]) & 1) +
                                           ((in >> kTet[t][3]) & 1);
                    n += k == 2 ? 2 : (k == 1 || k == 3) ? 1 : 0;  // two-against-two gives a quad
                }
            }
        }
    }
    // exclusive scan of n over the block in thread order: wavefront scan, then the (4) wavefront totals
    unsigned int incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int up = __shfl_up(incl, d, 64);
        if ((int)threadIdx.x >= d) incl += up;
    }
    if (threadIdx.x == 63) wave_n[threadIdx.y] = incl;
    __syncthreads();
    unsigned int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kTileY; ++w) {
        if (w < (int)threadIdx.y) before += wave_n[w];
        total += wave_n[w];
    }
    if constexpr (!EMIT) {
        if (threadIdx.x == 0 && threadIdx.y == 0) a.blocks[block] = total;
    } else {
        if (n == 0) return;
        unsigned int slot = a.blocks[block] + before + incl - n;
        for (int t = 0; t < 6; ++t) {
            float p[4][3], tv[4];
            int cc[4];
            for (int k = 0; k < 4; ++k) {
                cc[k] = kTet[t][k];
                tv[k] = v[cc[k]];
                p[k][0] = (float)(ci + kCorner[cc[k]][0]);  // voxel-index coordinates: exact in fp32
                p[k][1] = (float)(cj + kCorner[cc[k]][1]);
                p[k][2] = (float)(ck + kCorner[cc[k]][2]);
            }
            Tri tris[2];
            const int nt = tet_triangles(p, tv, a.iso, tris);
            for (int q = 0; q < nt; ++q, ++slot) {
                if (slot >= a.cap) return;
                float *o = a.verts + (size_t)slot * 9;
                for (int k = 0; k < 3; ++k) {
                    for (int i = 0; i < 3; ++i) o[3 * k + i] = (float)(a.origin[i] + (double)tris[q].p[k][i] * a.res);
                    if (a.labels) {  // nearest voxel, ties to even like np.round (modules/database.py:124-127)
                        const int x = (int)rintf(tris[q].p[k][0]), y = (int)rintf(tris[q].p[k][1]), z = (int)rintf(tris[q].p[k][2]);
                        a.labels[(size_t)slot * 3 + k] = a.ids ? a.ids[((size_t)x * a.Y + y) * a.Z + z] : 0;
                    }
                    if (a.keys) {  // Kuhn edges run from a corner to a componentwise larger one
                        const int c0 = cc[tris[q].e[k][0]], c1 = cc[tris[q].e[k][1]];
                        int lo[3], code = 0;
                        for (int i = 0; i < 3; ++i) {
                            lo[i] = kCorner[c0][i] < kCorner[c1][i] ? kCorner[c0][i] : kCorner[c1][i];
                            code |= (kCorner[c0][i] != kCorner[c1][i] ? 1 : 0) << i;
                        }
                        const uint64_t lin = ((uint64_t)(ci + lo[0]) * a.Y + (cj + lo[1])) * a.Z + (ck + lo[2]);
                        a.keys[(size_t)slot * 3 + k] = lin * 8 + code;
                    }
                }
            }
        }
    }
}

// exclusive prefix sum of blocks[0..n) in place, total to *count.  One workgroup: the list is (cells / 256) long,
// 65 K entries for a 256^3 volume.
__global__ __launch_bounds__(1024) void mesh_scan_kernel(unsigned int *blocks, unsigned int n, unsigned int *count)
{
    __shared__ unsigned int part[1024];
    const unsigned int chunk = (n + 1023) / 1024, lo = threadIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    unsigned int sum = 0;
    for (unsigned int i = lo; i < hi; ++i) sum += blocks[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned int add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - sum;
    for (unsigned int i = lo; i < hi; ++i) {
        const unsigned int c = blocks[i];
        blocks[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *count = part[1023];
}

// F-score support: hit[i] = 1 when some point of a cell-sorted set lies within tau of query i.  The set is binned on
// a grid of cell size >= tau (cell c holds points [cell_start[c], cell_start[c+1])), so the 27 cells around the
// query's own cell hold every candidate.  One thread per query; f64 points and distances, compared the way
// scipy's cKDTree.query(...)[0] <= tau compares them, so the hit counts equal the host metric's.
struct NearArgs {
    const double *query, *points;
    const unsigned int *cell_start;
    unsigned char *hit;
    unsigned int *n_hit;
    size_t n_query;
    double origin[3], cell, tau;
    int G[3];
};

__global__ __launch_bounds__(256) void points_within_kernel(NearArgs a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool found = false;
    if (i < a.n_query) {
        double q[3];
        int c[3];
        bool finite = true;
        for (int d = 0; d < 3; ++d) {
            q[d] = a.query[3 * i + d];
            const double f = floor((q[d] - a.origin[d]) / a.cell);
            finite = finite && f == f && f > -2.0e9 && f < 2.0e9;
            c[d] = finite ? (int)f : 0;
        }
        if (finite) {
            for (int dx = -1; dx <= 1 && !found; ++dx)
                for (int dy = -1; dy <= 1 && !found; ++dy) {
                    const int x = c[0] + dx, y = c[1] + dy;
                    if (x < 0 || x >= a.G[0] || y < 0 || y >= a.G[1]) continue;
                    const int z0 = c[2] - 1 < 0 ? 0 : c[2] - 1, z1 = c[2] + 1 >= a.G[2] ? a.G[2] - 1 : c[2] + 1;
                    if (z0 > z1) continue;
                    const size_t row = ((size_t)x * a.G[1] + y) * a.G[2];  // the three z cells are adjacent in the list
                    const unsigned int lo = a.cell_start[row + z0], hi = a.cell_start[row + z1 + 1];
                    for (unsigned int j = lo; j < hi; ++j) {
                        const double ex = a.points[3 * (size_t)j] - q[0], ey = a.points[3 * (size_t)j + 1] - q[1],
                                     ez = a.points[3 * (size_t)j + 2] - q[2];
                        if (sqrt(ex * ex + ey * ey + ez * ez) <= a.tau) {  // the k-d tree's test, rounding for rounding
                            found = true;
                            break;
                        }
                    }
                }
        }
        if (a.hit) a.hit[i] = found ? 1 : 0;
    }
    const unsigned long long votes = __ballot(found);
    if ((threadIdx.x & 63) == 0 && votes) atomicAdd(a.n_hit, (unsigned int)__popcll(votes));
}

static bool mesh_grid(int X, int Y, int Z, dim3 &grid)
{
    const size_t gz = (size_t)(Z - 1 + kTileZ - 1) / kTileZ, gy = (size_t)(Y - 1 + kTileY - 1) / kTileY, gx = (size_t)X - 1;
    if (gy > 65535 || gx > 65535 || gz * gy * gx > 0xffffffffull) return false;
    grid = dim3((unsigned int)gz, (unsigned int)gy, (unsigned int)gx);
    return true;
}

}  // namespace ojf

OJF_API size_t ojf_mesh_workspace_bytes(int X, int Y, int Z)
{
    dim3 grid;
    if (X < 2 || Y < 2 || Z < 2 || !ojf::mesh_grid(X, Y, Z, grid)) return 0;
    return sizeof(unsigned int) * (size_t)grid.x * grid.y * grid.z;
}

OJF_API int ojf_mesh_extract(const uint16_t *tsdf, const uint16_t *wgt, const uint8_t *ids, int X, int Y, int Z, float iso,
                             const double *origin, double res, void *workspace, size_t workspace_bytes, float *vertices,
                             uint8_t *labels, uint64_t *keys, uint32_t capacity, uint32_t *count, ojf_stream_t stream)
{
    using namespace ojf;
    if (!tsdf || !count || !origin || !workspace) return fail("ojf_mesh_extract: null pointer argument");
    if (X < 2 || Y < 2 || Z < 2) return fail("ojf_mesh_extract: the volume needs at least 2 voxels per axis");
    if (!(res > 0.0)) return fail("ojf_mesh_extract: resolution must be > 0");
    if (capacity && !vertices) return fail("ojf_mesh_extract: capacity > 0 needs a vertex buffer");
    dim3 grid;
    if (!mesh_grid(X, Y, Z, grid)) return fail("ojf_mesh_extract: volume too large");
    const size_t n_blocks = (size_t)grid.x * grid.y * grid.z;
    if (workspace_bytes < n_blocks * sizeof(unsigned int)) return fail("ojf_mesh_extract: workspace smaller than ojf_mesh_workspace_bytes");
    MeshArgs a;
    a.tsdf = tsdf; a.wgt = wgt; a.ids = ids; a.verts = vertices; a.labels = labels; a.keys = keys;
    a.blocks = static_cast<unsigned int *>(workspace); a.cap = capacity;
    a.X = X; a.Y = Y; a.Z = Z; a.iso = iso;
    a.origin[0] = origin[0]; a.origin[1] = origin[1]; a.origin[2] = origin[2]; a.res = res;
    hipStream_t st = as_stream(stream);
    const dim3 block(kTileZ, kTileY);
    hipLaunchKernelGGL(mesh_kernel<false>, grid, block, 0, st, a);
    hipLaunchKernelGGL(mesh_scan_kernel, dim3(1), dim3(1024), 0, st, a.blocks, (unsigned int)n_blocks, count);
    if (capacity) hipLaunchKernelGGL(mesh_kernel<true>, grid, block, 0, st, a);
    return check_hip(hipGetLastError(), "mesh_kernel launch");
}

OJF_API int ojf_points_within(const double *query, size_t n_query, const double *points_sorted, const uint32_t *cell_start,
                              const double *grid_origin, double cell, int GX, int GY, int GZ, double tau, uint8_t *hit,
                              uint32_t *n_hit, ojf_stream_t stream)
{
    using namespace ojf;
    if (!n_hit || !grid_origin || !cell_start) return fail("ojf_points_within: null pointer argument");
    if (n_query && (!query || !points_sorted)) return fail("ojf_points_within: null point buffer");
    if (GX < 1 || GY < 1 || GZ < 1) return fail("ojf_points_within: empty cell grid");
    if (!(tau >= 0.0) || !(cell >= tau) || !(cell > 0.0)) return fail("ojf_points_within: need cell >= tau >= 0 and cell > 0");
    if ((n_query + 255) / 256 > 0x7fffffffull) return fail("ojf_points_within: too many query points");
    hipStream_t st = as_stream(stream);
    OJF_HIP(hipMemsetAsync(n_hit, 0, sizeof(uint32_t), st));
    if (n_query == 0) return 0;
    NearArgs a;
    a.query = query; a.points = points_sorted; a.cell_start = cell_start; a.hit = hit; a.n_hit = n_hit; a.n_query = n_query;
    a.origin[0] = grid_origin[0]; a.origin[1] = grid_origin[1]; a.origin[2] = grid_origin[2];
    a.cell = cell; a.tau = tau; a.G[0] = GX; a.G[1] = GY; a.G[2] = GZ;
    hipLaunchKernelGGL(points_within_kernel, dim3((unsigned int)((n_query + 255) / 256)), dim3(256), 0, st, a);
    return check_hip(hipGetLastError(), "points_within_kernel launch");
}
