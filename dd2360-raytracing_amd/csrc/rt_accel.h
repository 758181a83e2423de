// rt_accel.h — host-side construction of the candidate-culling structure used by the fast closest-hit path.
//
// WHY IT IS EXACT (DESIGN.md §5.3 has the derivation).  The reference finds, for a ray, the closest hit among the
// spheres stored in the level-3 cells its traversal visits (acceleration_structure.h:276-304).  Testing FEWER spheres
// gives the same record as long as every skipped sphere is one whose float sphere::hit (sphere.h:17-46) cannot
// succeed with a smaller t.  For binary32 arithmetic in the reference's operation order,
//     discriminant > 0   ==>   dist(centre, line)^2  <  r^2 + 16.1 u |o - c|^2        (u = 2^-24)
// so a sphere whose ball of radius R = sqrt(r^2 + K2) (K2 covers 16.1 u |o-c|^2 for every ray origin inside the "near zone") is
// missed by the ray's line cannot be hit, and the float hit point of a sphere that is hit lies within R' of its centre.  The grid
// files each small sphere under every column (width h along a ray's major axis) its inflated extent overlaps, and inside a column
// under the fine bin (h / F) of its CENTRE's other coordinate.  The kernel walks the columns in which hit points can lie — from the
// origin's to the best hit's — and reads in each the bins between the line's extreme positions inside the column, grown by R'
// (DESIGN.md App. A.2): a sphere is read in the column of its own hit point, and never twice in one column.
// Rays outside the near zone, rays with a zero direction component, and exact-t ties fall back to the reference scan.
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include "rt_device.h"
#include "rt_tuning.h"

namespace rt {

// vector whose resize() leaves trivial elements uninitialised: the entry arrays (57 MB at N = 100 000) are written exactly
// once by the fill loop; zero-filling them first and growing them for the pad entries cost a third of the build
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { typedef NoInit<U> other; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new ((void*)p) U(std::forward<Args>(args)...); }
};
template <class T> using RawVec = std::vector<T, NoInit<T>>;

struct AccelHost {
    std::vector<float4> large_hot; std::vector<int32_t> large_id;
    std::vector<int32_t> cs; RawVec<float4> hot;                      // x copy followed by z copy, then 16 pad entries
    RawVec<float4> brick; std::vector<float4> large_brick;            // per entry: two float4 (DevAccel::brick)
    size_t n_entries = 0, n_entries_z = 0;                            // registrations of the x copy (columns along x) and of the z copy
    std::vector<int32_t> memb_start, memb_cell;
    std::vector<int32_t> cellnode;                            // 8x8x8 level-3 cells of the root box -> pre-order node
    std::vector<int32_t> bits_index; std::vector<uint32_t> cellbits;   // membership bitmaps of the spheres stored in several nodes
    DevAccel p{};
};

// constants of the exactness argument
constexpr double kZone = 24.0;          // near zone: |o - (0,1,0)| <= kZone
// (kRootHalfXZ, kCellXZ, kCellY and their float forms: rt_device.h)
constexpr double kCentreBound = 17.5;   // grid spheres have |c - (0,1,0)| <= this (root box grown by the radius)
constexpr double kSlack = 2e-3;         // rasterisation slack (absorbs float error of the walk, ~1e-5)
// "brick" of a sphere = the box of level-3 cells [ix0..ix1] x [iy0..iy1] x [iz0..iz1] when EVERY cell of that box stores the
// sphere.  A hit point that keeps 0.002 from the brick's six outer faces lies in a stored cell whose slab test passes
// (the float interval ends, the kernel's hit point and its cell coordinates are together off by less
// than 1e-5 for |o| <= 25, so 0.002 is a 200-fold margin).  In cell units (cells are 2.75 x 0.25 x 2.75): 7.3e-4 / 8e-3,
// rounded up.  (0.012 cost 2.7 % of the C3 frame: the lowest 3 % of a sphere resting on y = 0 fell outside its brick.)
constexpr double kBrickMxz = 0.0008, kBrickMy = 0.0085;
// 16.1 u |o-c|^2 with |o-c| <= kZone + kCentreBound, times a safety factor of 2
constexpr double kSparseCell = RT_SPARSE_CELL;   // column width of sparse scenes, in units of 2 R'
constexpr double kDenseCell = RT_DENSE_CELL;     // cell size of dense scenes, in units of 2 R' (build_accel step 2, and the device build)
__host__ __device__ inline double accel_K2() { const double u = 5.9604644775390625e-8, d = kZone + kCentreBound; return 2.0 * 16.1 * u * d * d; }
// inflated radius of the ball a ray must cross for the float test to be able to succeed, plus the walk's slack
__host__ __device__ inline double accel_Rp(double r2) { return sqrt(r2 * (1.0 + 1e-6) + accel_K2()) + 1e-5 + kSlack; }
// fine bin of a sphere's centre coordinate c along a column (bin width h / F, Gf = G * F bins from g0 on).  One expression for the host
// build and the device build (rt_build.hip): the two must agree bin for bin.
__host__ __device__ inline int accel_fine_bin(double c, double g0, double h, int F, int Gf) {
    const int b = (int)floor((c - g0) / h * (double)F);
    return b < 0 ? 0 : (b > Gf - 1 ? Gf - 1 : b);
}
// What the walk grows a column query by, in cell units: a sphere can matter where the line passes within its inflated radius R' of
// the centre (R' <= rmax: accel_Rp, which carries the rasterisation slack of the registration); the walk's own float error (~1e-5
// cells) is covered by kSlack once more, as in the cell-range form of rounds 1-3.
inline float accel_query_growth(double rmax, double h) { return (float)(((rmax + 1e-4 + kSlack) / h) * (1.0 + 1e-6)); }

// nodes/ent_id: the pre-order traversal copy; geom_r2(i) gives (cx,cy,cz,r^2) of world-list index i
// list_mode: `nodes` is the single unbounded node that stands for hitable_list::hit (every sphere is eligible everywhere:
// all bricks are infinite, no cell bookkeeping)
inline void build_accel(AccelHost& A, const std::vector<DevNode>& nodes, const std::vector<int32_t>& ent_id, const std::vector<float4>& ent_hot, int n_world, bool list_mode = false) {
    // 1. membership: sphere -> level-3 nodes (pre-order index) that hold it
    A.memb_start.assign((size_t)n_world + 1, 0);
    for (size_t k = 0; k < nodes.size(); ++k)
        for (int e = nodes[k].first; e < nodes[k].first + nodes[k].count; ++e) A.memb_start[(size_t)ent_id[e] + 1]++;
    for (int i = 0; i < n_world; ++i) A.memb_start[(size_t)i + 1] += A.memb_start[i];
    A.memb_cell.assign(A.memb_start[n_world], 0);
    std::vector<int32_t> fill(A.memb_start.begin(), A.memb_start.end() - 1);
    std::vector<float4> hot_of((size_t)n_world, make_float4(0, 0, 0, 0));
    std::vector<char> in_tree((size_t)n_world, 0);
    for (size_t k = 0; k < nodes.size(); ++k)
        for (int e = nodes[k].first; e < nodes[k].first + nodes[k].count; ++e) {
            const int s = ent_id[e];
            A.memb_cell[fill[s]++] = (int32_t)k;
            hot_of[s] = ent_hot[e]; in_tree[s] = 1;
        }
    // level-3 cells: the root box (-11,0,-11)-(11,2,11) (acceleration_structure.h:203) halved three times = 8x8x8 cells of
    // 2.75 x 0.25 x 2.75; a level-3 node is recognised by holding entries
    A.cellnode.assign(512, -1);
    for (size_t k = 0; k < nodes.size() && !list_mode; ++k) {
        if (nodes[k].count <= 0) continue;
        const int ix = (int)std::lround((nodes[k].lo[0] + kRootHalfXZ) / kCellXZ), iy = (int)std::lround(nodes[k].lo[1] / kCellY), iz = (int)std::lround((nodes[k].lo[2] + kRootHalfXZ) / kCellXZ);
        if (ix >= 0 && ix < 8 && iy >= 0 && iy < 8 && iz >= 0 && iz < 8) A.cellnode[ix * 64 + iy * 8 + iz] = (int32_t)k;
    }
    // membership bitmaps (one 512-bit row per sphere that is stored in more than one node)
    {
        std::vector<int> cell_of(nodes.size(), -1);
        for (int c = 0; c < 512; ++c) if (A.cellnode[c] >= 0) cell_of[A.cellnode[c]] = c;
        A.bits_index.assign((size_t)n_world, -1);
        A.cellbits.clear();
        for (int s = 0; s < n_world; ++s) {
            const int mb = A.memb_start[s], me = A.memb_start[(size_t)s + 1];
            if (me - mb < 2) continue;
            A.bits_index[s] = (int32_t)(A.cellbits.size() / 16);
            A.cellbits.resize(A.cellbits.size() + 16, 0u);
            uint32_t* row = &A.cellbits[A.cellbits.size() - 16];
            for (int k = mb; k < me; ++k) { const int c = cell_of[A.memb_cell[k]]; if (c >= 0) row[c >> 5] |= 1u << (c & 31); }
        }
        if (A.cellbits.empty()) A.cellbits.assign(16, 0u);
    }
    // bricks: cell-coordinate bounds (margins included) of the spheres whose storing cells form a complete box
    std::vector<float4> sb_lo((size_t)n_world), sb_hi((size_t)n_world);
    {
        std::vector<int> cell_of(nodes.size(), -1);
        for (int c = 0; c < 512; ++c) if (A.cellnode[c] >= 0) cell_of[A.cellnode[c]] = c;
        const float inf = std::numeric_limits<float>::infinity();
        for (int s = 0; s < n_world; ++s) {
            const int mb = A.memb_start[s], me = A.memb_start[(size_t)s + 1];
            int lo[3] = {8, 8, 8}, hi[3] = {-1, -1, -1};
            bool ok = me > mb;
            for (int k = mb; k < me && ok; ++k) {
                const int c = cell_of[A.memb_cell[k]];
                if (c < 0) { ok = false; break; }
                const int q[3] = {c >> 6, (c >> 3) & 7, c & 7};
                for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], q[d]); hi[d] = std::max(hi[d], q[d]); }
            }
            // distinct nodes map to distinct cells, so "as many nodes as the box has cells" = every cell of the box stores it
            if (ok) ok = (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) == me - mb;
            const int32_t single = (me - mb == 1) ? A.memb_cell[mb] : -1;
            float idbits, nodebits;
            { const int32_t v = s; std::memcpy(&idbits, &v, 4); std::memcpy(&nodebits, &single, 4); }
            if (list_mode) {
                sb_lo[s] = make_float4(-inf, -inf, -inf, idbits); sb_hi[s] = make_float4(inf, inf, inf, nodebits);
            } else if (ok) {
                sb_lo[s] = make_float4((float)(lo[0] + kBrickMxz), (float)(lo[1] + kBrickMy), (float)(lo[2] + kBrickMxz), idbits);
                sb_hi[s] = make_float4((float)(hi[0] + 1 - kBrickMxz), (float)(hi[1] + 1 - kBrickMy), (float)(hi[2] + 1 - kBrickMxz), nodebits);
            } else {
                sb_lo[s] = make_float4(inf, inf, inf, idbits); sb_hi[s] = make_float4(-inf, -inf, -inf, nodebits);
            }
        }
    }
    // 2. cell size from the median radius of the tree spheres
    std::vector<double> radii;
    for (int s = 0; s < n_world; ++s) if (in_tree[s]) radii.push_back(std::sqrt((double)hot_of[s].w));
    DevAccel& p = A.p;
    p = DevAccel{};
    if (radii.empty()) { p.enabled = 0; A.cs.assign(4, 0); return; }
    std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
    const double rmed = radii[radii.size() / 2];
    double h = 2.0 * accel_Rp(rmed * rmed);
    h = std::min(1.0, std::max(0.05, h));
    {
        // dense scenes (a sphere's inflated square covers ~4 cells of size 2R': more than 8 per cell expected) take narrower columns:
        // a ray that hits within its first column tests what that column holds; sparse scenes walk many columns per ray
        const double g = std::ceil(2.0 * (kRootHalfXZ + 5.0 * h) / h);
        h = std::max(0.05, (4.0 * (double)radii.size() > 8.0 * g * g ? kDenseCell : kSparseCell) * h);
    }
    const double Rlim = 1.5 * h;                               // spheres with R' above this go to the large list
    // extent: the reference's root box in x and z (every tree sphere's centre lies in it, grown by its radius); a list may
    // reach further out — as far as the centre bound of the exactness argument lets the grid follow
    double reach = kRootHalfXZ;
    if (list_mode)
        for (int s = 0; s < n_world; ++s) {
            if (!in_tree[s]) continue;
            const float4 g = hot_of[s];
            const double dc = std::sqrt((double)g.x * g.x + ((double)g.y - 1.0) * ((double)g.y - 1.0) + (double)g.z * g.z);
            if (!(g.w >= 0.0f) || accel_Rp((double)g.w) > Rlim || !(dc <= kCentreBound)) continue;
            reach = std::max(reach, std::max(std::fabs((double)g.x), std::fabs((double)g.z)));
        }
    const double half = reach + 2.0 * Rlim + 2.0 * h;
    const int G = (int)std::ceil(2.0 * half / h);
    const double g0 = -half;
    p.G = G; p.g0 = (float)g0; p.h = (float)h; p.inv_h = (float)(1.0 / h);
    // 3. classify + register.  A grid sphere goes once into every column its inflated extent overlaps, keyed by the fine bin of its
    // CENTRE along the column (accel_fine_bin): the walk grows its query by the largest inflated radius instead (DevAccel::rq_c).
    const int F = RT_ACCEL_FINE, Gf = G * F;
    p.F = F; p.Gf = Gf;
    struct Reg { int s, ix0, ix1, iz0, iz1, bx, bz; };
    std::vector<Reg> regs;
    double ylo = 1e30, yhi = -1e30, rmax = 0, cells = 0;
    for (int s = 0; s < n_world; ++s) {
        if (!in_tree[s]) continue;
        const float4 g = hot_of[s];
        const double Rp = accel_Rp((double)g.w);
        const double dc = std::sqrt((double)g.x * g.x + ((double)g.y - 1.0) * ((double)g.y - 1.0) + (double)g.z * g.z);
        const bool inside = (g.x - Rp > g0 + h) && (g.x + Rp < g0 + (G - 1) * h) && (g.z - Rp > g0 + h) && (g.z + Rp < g0 + (G - 1) * h);
        if (Rp > Rlim || dc > kCentreBound || !inside || !(g.w >= 0.0f)) {
            A.large_hot.push_back(g); A.large_id.push_back(s); A.large_brick.push_back(sb_lo[s]); A.large_brick.push_back(sb_hi[s]);
            continue;
        }
        Reg r; r.s = s;
        r.ix0 = (int)std::floor((g.x - Rp - g0) / h - 1e-4); r.ix1 = (int)std::floor((g.x + Rp - g0) / h + 1e-4);
        r.iz0 = (int)std::floor((g.z - Rp - g0) / h - 1e-4); r.iz1 = (int)std::floor((g.z + Rp - g0) / h + 1e-4);
        r.ix0 = std::max(0, r.ix0); r.iz0 = std::max(0, r.iz0); r.ix1 = std::min(G - 1, r.ix1); r.iz1 = std::min(G - 1, r.iz1);
        cells += (double)(r.ix1 - r.ix0 + 1) * (double)(r.iz1 - r.iz0 + 1);
        r.bx = accel_fine_bin((double)g.x, g0, h, F, Gf); r.bz = accel_fine_bin((double)g.z, g0, h, F, Gf);
        regs.push_back(r);
        ylo = std::min(ylo, (double)g.y - Rp); yhi = std::max(yhi, (double)g.y + Rp); rmax = std::max(rmax, Rp);
    }
    p.n_large = (int)A.large_id.size();
    const size_t ncell = (size_t)G * G, nbin = (size_t)G * Gf;
    std::vector<int32_t> cs_x(nbin + 1, 0), cs_z(nbin + 1, 0);
    for (const Reg& r : regs) {
        for (int ix = r.ix0; ix <= r.ix1; ++ix) cs_x[(size_t)ix * Gf + r.bz + 1]++;
        for (int iz = r.iz0; iz <= r.iz1; ++iz) cs_z[(size_t)iz * Gf + r.bx + 1]++;
    }
    for (size_t c = 0; c < nbin; ++c) { cs_x[c + 1] += cs_x[c]; cs_z[c + 1] += cs_z[c]; }
    const size_t nx = (size_t)cs_x[nbin], nz = (size_t)cs_z[nbin], total = nx + nz;
    A.n_entries = nx; A.n_entries_z = nz;
    // the kernel reads entries in batches and may over-read past a range: 16 pad entries that can never test positive
    A.hot.resize(total + 16);
    A.brick.resize(2 * total + 32);
    // fill in sphere order (a counting sort by bin, once with columns along x and once along z): within a bin, ascending sphere index
    {
        std::vector<int32_t> fx(cs_x.begin(), cs_x.end() - 1), fz(cs_z.begin(), cs_z.end() - 1);
        for (const Reg& r : regs) {
            const float4 g = hot_of[r.s], blo = sb_lo[r.s], bhi = sb_hi[r.s];
            for (int ix = r.ix0; ix <= r.ix1; ++ix) {
                const size_t a = (size_t)fx[(size_t)ix * Gf + r.bz]++;
                A.hot[a] = g; A.brick[2 * a] = blo; A.brick[2 * a + 1] = bhi;
            }
            for (int iz = r.iz0; iz <= r.iz1; ++iz) {
                const size_t b = nx + (size_t)fz[(size_t)iz * Gf + r.bx]++;
                A.hot[b] = g; A.brick[2 * b] = blo; A.brick[2 * b + 1] = bhi;
            }
        }
    }
    { const float qn = std::nanf("");
      for (size_t k = total; k < total + 16; ++k) { A.hot[k] = make_float4(qn, qn, qn, qn); A.brick[2 * k] = make_float4(qn, qn, qn, 0.f); A.brick[2 * k + 1] = make_float4(qn, qn, qn, qn); } }
    if (A.large_brick.empty()) A.large_brick.assign(2, make_float4(0, 0, 0, 0));
    A.cs.resize(2 * (nbin + 1));
    for (size_t c = 0; c <= nbin; ++c) { A.cs[c] = cs_x[c]; A.cs[nbin + 1 + c] = (int32_t)nx + cs_z[c]; }
    p.zoff = (int32_t)(nbin + 1);
    if (regs.empty()) { ylo = 0; yhi = 0; }
    p.ylo = (float)(ylo - 1e-4); p.yhi = (float)(yhi + 1e-4); p.rmax = (float)(rmax + 1e-4);
    p.rq_c = accel_query_growth(rmax, h);
    p.zone2 = (float)(kZone * kZone);
    p.enabled = 1;
    // rays side by side in the cooperative walk pay off while a chunk of 8 columns is a handful of entries (C3: 3.6 per cell);
    // on dense grids (C5: N = 100 000, ~40 per cell) an uneven pair of rays costs twice the longer one
    p.coop_groups = cells <= 8.0 * (double)ncell ? 4 : 1;
    // very sparse grids (lists of a few hundred spheres): the frame waits for its pixel chains with the chip half idle, so every
    // pre-classified chain starts ALONE in a wave (C2: 13.4 -> 11.6 ms); at C3's 3.6 entries per cell the waves set aside cost
    // more throughput than the chains gain (profiles/r3/chain_cache_ab.txt)
    p.solo_chains = cells <= RT_SOLO_DENSITY * (double)ncell ? 1 : 0;
}

} // namespace rt
