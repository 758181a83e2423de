// rt_build.hip — buildOctree (acceleration_structure.h:195-217), the traversal copy and the candidate grid, on the device.
//
// rt_build_octree_gpu() gives the tree rt_build_octree() gives, array for array and bit for bit (tests/test_gpu_build.py
// downloads both and compares) — the reference layout included, although the reference builds it by inserting the spheres one
// after the other: which index a node or a leaf bucket gets is decided by WHEN insert() creates it, and that order can be
// computed without replaying the insertions:
//   * a node exists iff some sphere "intersects" its box (and its ancestors' boxes: implied, the boxes are nested with shared
//     planes and the test is monotone); it is created by the FIRST such sphere (atomicMin of the sphere index), and one
//     sphere's insertion creates its new nodes in depth-first order.  So nodes sorted by (first sphere, pre-order rank in the
//     full 585-node tree) are the reference's nodes[1..nodeCount-1] in order;
//   * a level-3 cell's members are the spheres that intersect it, in index order, the first 8 x SPHERES_PER_LEAF of them (the
//     rest is dropped with the reference's "leaf nodes are full" message, :135); bucket b of the cell is created by the member
//     number b x SPL, so buckets sorted by (creating sphere, pre-order rank of the cell) are leaves[1..leafCount-1] in order.
// Everything else (pre-order traversal copy, bucket entries, memberships, bricks, the (x,z) candidate grid in two copies) is
// per-sphere or per-cell work plus radix sorts (rocPRIM).  The host supplies a handful of scalars on the way (four small
// read-backs): sizes of the arrays to allocate, the grid's cell size from the median radius.
//
// Binary16 trees (USE_FP16): the same reference layout in real_t = binary16 arithmetic, then the pair layout and plane table the
// binary16 kernels read, and no candidate grid.
#include <hip/hip_runtime.h>
#include <mutex>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include "rt_handles.h"
#include "rt_octgeom.h"

namespace rt {
namespace gpubuild {

#define RT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
constexpr unsigned kNone = 0xffffffffu;

struct Counters {
    unsigned n_pairs, pair_overflow, dropped_outside, dropped_full;
    unsigned node_count, leaf_count, n_entries, in_tree;
    unsigned median_bits, bit_rows, n_large, reg_total;
    unsigned long long ylo, yhi, rmax;       // order-preserving encodings of doubles (min, max, max)
    unsigned long long reg_cells;            // cells the grid spheres' inflated squares overlap, summed (build_accel's density measure)
};

// a double as an unsigned integer with the same order
__host__ __device__ inline unsigned long long ord_of(double d) {
    unsigned long long u; memcpy(&u, &d, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double double_of(unsigned long long o) {
    const unsigned long long u = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    double d; memcpy(&d, &u, 8); return d;
}

// ---------------------------------------------------------------------------------------------- the reference-layout tree
// insert() for every sphere at once: which nodes it reaches, which level-3 cells it lands in
template <class R>
__global__ __launch_bounds__(256) void k_pairs(const float4* __restrict__ geom, int n, const float (*__restrict__ box)[6], unsigned* first,
                                               unsigned long long* pairs, unsigned cap, Counters* C) {
    const int i = blockIdx.x * 256 + threadIdx.x + 1;                 // the ground sphere (index 0) is not in the tree (:208)
    if (i >= n) return;
    const float4 g = geom[i];
    const R cx = real_from<R>(g.x), cy = real_from<R>(g.y), cz = real_from<R>(g.z), rad = real_from<R>(g.w);
    auto touches = [&](int fr) {
        const R lo[3] = {real_from<R>(box[fr][0]), real_from<R>(box[fr][1]), real_from<R>(box[fr][2])};
        const R hi[3] = {real_from<R>(box[fr][3]), real_from<R>(box[fr][4]), real_from<R>(box[fr][5])};
        return sphere_touches_box<R>(cx, cy, cz, rad, lo, hi);
    };
    if (!touches(0)) { atomicAdd(&C->dropped_outside, 1u); return; }
    for (int a = 0; a < 8; ++a) {
        const int f1 = 1 + 73 * a;
        if (!touches(f1)) continue;
        atomicMin(&first[f1], (unsigned)i);
        for (int b = 0; b < 8; ++b) {
            const int f2 = f1 + 1 + 9 * b;
            if (!touches(f2)) continue;
            atomicMin(&first[f2], (unsigned)i);
            for (int c = 0; c < 8; ++c) {
                const int f3 = f2 + 1 + c;
                if (!touches(f3)) continue;
                atomicMin(&first[f3], (unsigned)i);
                const unsigned slot = atomicAdd(&C->n_pairs, 1u);
                if (slot < cap) pairs[slot] = ((unsigned long long)f3 << 32) | (unsigned)i;
                else C->pair_overflow = 1u;
            }
        }
    }
}

// pairs sorted by (cell, sphere): where each cell's members start and end
__global__ __launch_bounds__(256) void k_segments(const unsigned long long* __restrict__ sorted, unsigned n_pairs, unsigned* seg_start, unsigned* seg_end) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const unsigned fr = (unsigned)(sorted[p] >> 32);
    if (p == 0 || (unsigned)(sorted[p - 1] >> 32) != fr) seg_start[fr] = p;
    if (p + 1 == n_pairs || (unsigned)(sorted[p + 1] >> 32) != fr) seg_end[fr] = p + 1;
}

// node and leaf numbers in the reference's creation order (one block)
__global__ __launch_bounds__(1024) void k_number(const unsigned* __restrict__ first, const unsigned* __restrict__ seg_start, const unsigned* __restrict__ seg_end,
                                                 const unsigned long long* __restrict__ sorted, int spl, int* node_id, int* dev_index, int* accepted, int* leaf_of, Counters* C) {
    __shared__ unsigned long long s_key[kFullNodes];
    __shared__ unsigned long long s_bkey[4096];
    __shared__ unsigned short s_bfr[4096];
    __shared__ unsigned char s_bb[4096];
    __shared__ int s_boff[kFullNodes + 1];
    __shared__ int s_nb[kFullNodes];
    __shared__ unsigned s_drop, s_nodes;
    const int t = threadIdx.x;
    if (t == 0) { s_drop = 0u; s_nodes = 0u; }
    for (int fr = t; fr < kFullNodes; fr += 1024) {
        const unsigned f = fr == 0 ? 0u : first[fr];
        s_key[fr] = f != kNone ? (unsigned long long)f * 1024ull + (unsigned)fr : ~0ull;
    }
    __syncthreads();
    for (int fr = t; fr < kFullNodes; fr += 1024) {
        int id = -1, di = -1, acc = 0, nb = 0;
        if (s_key[fr] != ~0ull) {
            id = 0; di = 0;
            for (int x = 0; x < kFullNodes; ++x) { id += s_key[x] < s_key[fr] ? 1 : 0; di += (x < fr && s_key[x] != ~0ull) ? 1 : 0; }
            atomicAdd(&s_nodes, 1u);
            int level, a, b, c; full_path(fr, level, a, b, c);
            if (level == 3 && seg_start[fr] != kNone) {
                const int m = (int)(seg_end[fr] - seg_start[fr]);
                acc = m < 8 * spl ? m : 8 * spl;               // the reference drops what does not fit the 8 buckets (:135)
                nb = (acc + spl - 1) / spl;
                if (m > acc) atomicAdd(&s_drop, (unsigned)(m - acc));
            }
        }
        node_id[fr] = id; dev_index[fr] = di; accepted[fr] = acc; s_nb[fr] = nb;
    }
    __syncthreads();
    if (t == 0) { int run = 0; for (int fr = 0; fr < kFullNodes; ++fr) { s_boff[fr] = run; run += s_nb[fr]; } s_boff[kFullNodes] = run; }
    __syncthreads();
    for (int fr = t; fr < kFullNodes; fr += 1024)
        for (int b = 0; b < s_nb[fr]; ++b) {
            const unsigned creator = (unsigned)sorted[seg_start[fr] + (unsigned)(b * spl)];
            const int q = s_boff[fr] + b;
            s_bkey[q] = (unsigned long long)creator * 1024ull + (unsigned)fr; s_bfr[q] = (unsigned short)fr; s_bb[q] = (unsigned char)b;
        }
    __syncthreads();
    const int total = s_boff[kFullNodes];
    for (int q = t; q < total; q += 1024) {
        int rank = 0;
        for (int x = 0; x < total; ++x) rank += s_bkey[x] < s_bkey[q] ? 1 : 0;
        leaf_of[(int)s_bfr[q] * 8 + (int)s_bb[q]] = 1 + rank;        // leaf 0 is never used (leafCount starts at 1, :59)
    }
    if (t == 0) { C->node_count = s_nodes; C->leaf_count = 1u + (unsigned)total; C->dropped_full = s_drop; }
}

// OctLeaf contents, and how many of a cell's accepted members can be hit at all (ghost slots cannot)
__global__ __launch_bounds__(64) void k_leaves(const unsigned* __restrict__ seg_start, const int* __restrict__ accepted, const int* __restrict__ leaf_of,
                                               const unsigned long long* __restrict__ sorted, const int32_t* __restrict__ kind, int spl,
                                               int32_t* leaf_count, int32_t* leaf_indices, int* hit_cnt) {
    const int fr = blockIdx.x, lane = threadIdx.x;
    const int acc = accepted[fr];
    if (acc <= 0) { if (lane == 0) hit_cnt[fr] = 0; return; }
    const unsigned start = seg_start[fr];
    int hit = 0;
    for (int j = lane; j < acc; j += 64) {
        const int s = (int)(unsigned)sorted[start + (unsigned)j];
        const int b = j / spl;
        leaf_indices[(size_t)leaf_of[fr * 8 + b] * spl + (j - b * spl)] = s;
        hit += kind[s] != RT_MAT_NONE ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) hit += __shfl_xor(hit, off);
    if (lane == 0) hit_cnt[fr] = hit;
    const int nb = (acc + spl - 1) / spl;
    if (lane < nb) { const int left = acc - lane * spl; leaf_count[leaf_of[fr * 8 + lane]] = left < spl ? left : spl; }
}

// OctNode array in the reference's numbering, and the pre-order traversal copy with skip links (one block)
__global__ __launch_bounds__(1024) void k_nodes(const float (*__restrict__ box)[6], const int* __restrict__ node_id, const int* __restrict__ dev_index, const int* __restrict__ accepted,
                                                const int* __restrict__ leaf_of, const int* __restrict__ hit_cnt, int spl, rt_octnode* ref_nodes, DevNode* dnodes,
                                                int* ent_first, int32_t* devcell, int32_t* cellnode, int* dev_to_fr, Counters* C) {
    __shared__ int s_ex[kFullNodes + 1];         // existing nodes before fr
    __shared__ int s_ef[kFullNodes + 1];         // hittable entries before fr
    const int t = threadIdx.x;
    if (t == 0) {
        int ex = 0, ef = 0;
        for (int fr = 0; fr < kFullNodes; ++fr) { s_ex[fr] = ex; s_ef[fr] = ef; ex += node_id[fr] >= 0 ? 1 : 0; ef += hit_cnt[fr]; }
        s_ex[kFullNodes] = ex; s_ef[kFullNodes] = ef;
        C->n_entries = (unsigned)ef;
    }
    for (int c = t; c < 512; c += 1024) cellnode[c] = -1;
    __syncthreads();
    for (int fr = t; fr < kFullNodes; fr += 1024) {
        ent_first[fr] = s_ef[fr];
        const int id = node_id[fr];
        if (id < 0) continue;
        int level, a, b, c; full_path(fr, level, a, b, c);
        const int k = dev_index[fr];
        DevNode d; memset(&d, 0, sizeof(d));
        d.lo[0] = box[fr][0]; d.lo[1] = box[fr][1]; d.lo[2] = box[fr][2]; d.hix = box[fr][3]; d.hiy = box[fr][4]; d.hiz = box[fr][5];
        d.skip = s_ex[fr + full_subtree(level)];
        d.first = s_ef[fr]; d.count = level == 3 ? hit_cnt[fr] : 0; d.ref_index = id;
        dnodes[k] = d;
        dev_to_fr[k] = fr;
        rt_octnode rn; rn.level = level;
        for (int q = 0; q < 6; ++q) rn.aabb[q] = box[fr][q];
        for (int o = 0; o < 8; ++o) {
            if (level == 3) { const int nb = (accepted[fr] + spl - 1) / spl; rn.children[o] = o < nb ? leaf_of[fr * 8 + o] : 0; }
            else { const int child = level == 0 ? 1 + 73 * o : level == 1 ? fr + 1 + 9 * o : fr + 1 + o; rn.children[o] = node_id[child] >= 0 ? node_id[child] : 0; }
        }
        ref_nodes[id] = rn;
        int cell = -1;
        if (level == 3 && d.count > 0) { cell = cell_coord(a, b, c, 2) * 64 + cell_coord(a, b, c, 1) * 8 + cell_coord(a, b, c, 0); cellnode[cell] = k; }
        devcell[k] = cell;
    }
}

// bucket contents in traversal order (ghost entries removed), and (sphere, node) pairs for the membership lists
template <class R>
__global__ __launch_bounds__(64) void k_entries(const unsigned* __restrict__ seg_start, const int* __restrict__ accepted, const int* __restrict__ dev_index, const int* __restrict__ ent_first,
                                                const unsigned long long* __restrict__ sorted, const float4* __restrict__ geom, const int32_t* __restrict__ kind,
                                                float4* ent_hot, int32_t* ent_id, unsigned long long* pair2, float4* hot_of) {
    const int fr = blockIdx.x, lane = threadIdx.x;
    const int acc = accepted[fr];
    if (acc <= 0) return;
    const unsigned start = seg_start[fr];
    const int k = dev_index[fr];
    int base = ent_first[fr];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < acc; j0 += 64) {
        const int j = j0 + lane;
        int s = 0; bool keep = false;
        if (j < acc) { s = (int)(unsigned)sorted[start + (unsigned)j]; keep = kind[s] != RT_MAT_NONE; }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(m & lt);
            const float4 g = geom[s];
            const R rr = real_from<R>(g.w);
            const float4 h = make_float4(g.x, g.y, g.z, as_float(rr * rr));  // radius*radius in real_t (sphere.h:21)
            ent_id[pos] = s; ent_hot[pos] = h; hot_of[s] = h;
            pair2[pos] = ((unsigned long long)(unsigned)s << 32) | (unsigned)k;
        }
        base += __popcll(m);
    }
}

// ---------------------------------------------------------------------------------------------- USE_FP16 traversal copy
// The binary16 kernels read the bucket entries as PAIRS (rt_kernels_fp16.hip; the layout rt_octree_upload derives on the host):
// node first/count in pairs, an odd count padded with a NaN sphere, the entry -> sphere table following the pairs, and per node
// the six indices of its box planes in the tree's plane table (x planes 0..8, y 9..17, z 18..26: the root box halved 3 times).
__global__ void k_pair_offsets(DevNode* dnodes, int n_nodes, const int* __restrict__ dev_to_fr, int* pair_first, Counters* C) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int run = 0;
    for (int k = 0; k < n_nodes; ++k) {
        pair_first[k] = run;
        run += (dnodes[k].count + 1) / 2;
        int level, a, b, c; full_path(dev_to_fr[k], level, a, b, c);
        const int sh = 3 - level;                                      // a level-L node spans 2^(3-L) level-3 cells per axis
        uint32_t w = 0;
        for (int axis = 0; axis < 3; ++axis) {
            const int bit = 2 - axis;                                  // octant bit 2 = x, 1 = y, 0 = z
            const int cell = level == 0 ? 0 : level == 1 ? ((a >> bit) & 1) : level == 2 ? ((((a >> bit) & 1) << 1) | ((b >> bit) & 1)) : cell_coord(a, b, c, bit);
            const int lo = (cell << sh) + 9 * axis, hi = ((cell + 1) << sh) + 9 * axis;
            w |= (uint32_t)lo << (10 * axis); w |= (uint32_t)hi << (10 * axis + 5);
        }
        dnodes[k].pad[0] = (int32_t)w;
    }
    C->reg_total = (unsigned)(2 * run);                               // entry slots of the pair layout (n_entries stays the entry count)
}
__global__ __launch_bounds__(64) void k_pair_fill(DevNode* dnodes, const int* __restrict__ pair_first, const float4* __restrict__ ent_hot, const int32_t* __restrict__ ent_id,
                                                  uint4* pairs, int32_t* pid) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const int first = dnodes[k].first, cnt = dnodes[k].count, pf = pair_first[k];
    auto hb = [](float v) { return (uint32_t)half_t(v).bits; };
    for (int e = 2 * lane; e < cnt; e += 128) {
        const float4 A = ent_hot[first + e];
        const bool two = e + 1 < cnt;
        const float4 B = two ? ent_hot[first + e + 1] : A;
        const uint32_t nanb = 0x7e00u;
        const int p = pf + e / 2;
        pairs[p] = make_uint4(hb(A.x) | ((two ? hb(B.x) : nanb) << 16), hb(A.y) | ((two ? hb(B.y) : nanb) << 16),
                              hb(A.z) | ((two ? hb(B.z) : nanb) << 16), hb(A.w) | ((two ? hb(B.w) : nanb) << 16));
        pid[2 * p] = ent_id[first + e];
        pid[2 * p + 1] = two ? ent_id[first + e + 1] : -1;
    }
    __syncthreads();
    if (lane == 0) { dnodes[k].first = pf; dnodes[k].count = (cnt + 1) / 2; }
}

// ---------------------------------------------------------------------------------------------- candidate grid (rt_accel.h)
__global__ __launch_bounds__(256) void k_memb(const unsigned long long* __restrict__ sorted2, int n_entries, int n_world, int32_t* memb_cell, int32_t* memb_start) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_entries) return;
    const int s = (int)(sorted2[p] >> 32);
    memb_cell[p] = (int32_t)(unsigned)sorted2[p];
    const int prev = p == 0 ? -1 : (int)(sorted2[p - 1] >> 32);
    for (int t = prev + 1; t <= s; ++t) memb_start[t] = p;              // spheres without entries start where the next one does
    if (p + 1 == n_entries) for (int t = s + 1; t <= n_world; ++t) memb_start[t] = n_entries;
}

__global__ __launch_bounds__(256) void k_bricks(int n_world, const int32_t* __restrict__ memb_start, const int32_t* __restrict__ memb_cell, const int32_t* __restrict__ devcell,
                                                const float4* __restrict__ hot_of, float4* sb_lo, float4* sb_hi, int* multi, unsigned* r2bits, Counters* C) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_world) return;
    const int mb = memb_start[s], me = memb_start[s + 1];
    int lo[3] = {8, 8, 8}, hi[3] = {-1, -1, -1};
    bool ok = me > mb;
    for (int k = mb; k < me && ok; ++k) {
        const int c = devcell[memb_cell[k]];
        if (c < 0) { ok = false; break; }
        const int q[3] = {c >> 6, (c >> 3) & 7, c & 7};
        for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], q[d]); hi[d] = max(hi[d], q[d]); }
    }
    if (ok) ok = (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) == me - mb;
    const int32_t single = (me - mb == 1) ? memb_cell[mb] : -1;
    const float idbits = __int_as_float(s), nodebits = __int_as_float(single);
    const float inf = __builtin_inff();
    if (ok) {
        sb_lo[s] = make_float4((float)(lo[0] + kBrickMxz), (float)(lo[1] + kBrickMy), (float)(lo[2] + kBrickMxz), idbits);
        sb_hi[s] = make_float4((float)(hi[0] + 1 - kBrickMxz), (float)(hi[1] + 1 - kBrickMy), (float)(hi[2] + 1 - kBrickMxz), nodebits);
    } else {
        sb_lo[s] = make_float4(inf, inf, inf, idbits); sb_hi[s] = make_float4(-inf, -inf, -inf, nodebits);
    }
    multi[s] = (me - mb >= 2) ? 1 : 0;
    const bool in_tree = me > mb;
    r2bits[s] = in_tree ? __float_as_uint(hot_of[s].w) : kNone;       // (r^2 >= 0: the bits order like the values)
    if (in_tree) atomicAdd(&C->in_tree, 1u);
}

__global__ __launch_bounds__(256) void k_bitrows(int n_world, const int32_t* __restrict__ memb_start, const int32_t* __restrict__ memb_cell, const int32_t* __restrict__ devcell,
                                                 const int* __restrict__ multi, const int* __restrict__ row_of, int32_t* bits_index, uint32_t* cellbits) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_world) return;
    if (!multi[s]) { bits_index[s] = -1; return; }
    const int row = row_of[s];
    bits_index[s] = row;
    uint32_t w[16];
    for (int q = 0; q < 16; ++q) w[q] = 0u;
    for (int k = memb_start[s]; k < memb_start[s + 1]; ++k) { const int c = devcell[memb_cell[k]]; if (c >= 0) w[c >> 5] |= 1u << (c & 31); }
    for (int q = 0; q < 16; ++q) cellbits[(size_t)row * 16 + q] = w[q];
}

struct GridParams { double g0, h, Rlim; int G, F, Gf; };

// large list or grid registration of every tree sphere (rt_accel.h step 3): the columns its inflated extent overlaps along x and
// along z, and the fine bins of its centre
__global__ __launch_bounds__(256) void k_classify(int n_world, const int32_t* __restrict__ memb_start, const float4* __restrict__ hot_of, GridParams P,
                                                  int* is_large, int* nreg_x, int* nreg_z, int4* range, int2* bins, Counters* C) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_world) return;
    int large = 0, cx = 0, cz = 0;
    int4 r = make_int4(0, 0, 0, 0);
    int2 bn = make_int2(0, 0);
    if (memb_start[s + 1] > memb_start[s]) {
        const float4 g = hot_of[s];
        const double Rp = accel_Rp((double)g.w);
        const double dc = sqrt((double)g.x * g.x + ((double)g.y - 1.0) * ((double)g.y - 1.0) + (double)g.z * g.z);
        const double g0 = P.g0, h = P.h; const int G = P.G;
        const bool inside = (g.x - Rp > g0 + h) && (g.x + Rp < g0 + (G - 1) * h) && (g.z - Rp > g0 + h) && (g.z + Rp < g0 + (G - 1) * h);
        if (Rp > P.Rlim || dc > kCentreBound || !inside || !(g.w >= 0.0f)) large = 1;
        else {
            r.x = (int)floor((g.x - Rp - g0) / h - 1e-4); r.y = (int)floor((g.x + Rp - g0) / h + 1e-4);
            r.z = (int)floor((g.z - Rp - g0) / h - 1e-4); r.w = (int)floor((g.z + Rp - g0) / h + 1e-4);
            r.x = max(0, r.x); r.z = max(0, r.z); r.y = min(G - 1, r.y); r.w = min(G - 1, r.w);
            atomicAdd(&C->reg_cells, (unsigned long long)((r.y - r.x + 1) * (r.w - r.z + 1)));
            cx = r.y - r.x + 1; cz = r.w - r.z + 1;
            bn.x = accel_fine_bin((double)g.x, g0, h, P.F, P.Gf); bn.y = accel_fine_bin((double)g.z, g0, h, P.F, P.Gf);
            atomicMin(&C->ylo, ord_of((double)g.y - Rp)); atomicMax(&C->yhi, ord_of((double)g.y + Rp)); atomicMax(&C->rmax, ord_of(Rp));
        }
    }
    is_large[s] = large; nreg_x[s] = cx; nreg_z[s] = cz; range[s] = r; bins[s] = bn;
}

__global__ __launch_bounds__(256) void k_regs(int n_world, const int* __restrict__ nreg_x, const int* __restrict__ off_x, const int* __restrict__ off_z, const int4* __restrict__ range,
                                              const int2* __restrict__ bins, int Gf, unsigned long long* keys_x, unsigned long long* keys_z) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_world || nreg_x[s] == 0) return;
    const int4 r = range[s];
    const int2 bn = bins[s];
    size_t o = (size_t)off_x[s];
    for (int ix = r.x; ix <= r.y; ++ix) keys_x[o++] = ((unsigned long long)(unsigned)(ix * Gf + bn.y) << 32) | (unsigned)s;      // columns along x: keyed by the bin of cz
    o = (size_t)off_z[s];
    for (int iz = r.z; iz <= r.w; ++iz) keys_z[o++] = ((unsigned long long)(unsigned)(iz * Gf + bn.x) << 32) | (unsigned)s;
}

// entries of both grid copies from the sorted registrations; the bin starts by binary search
__global__ __launch_bounds__(256) void k_fill(const unsigned long long* __restrict__ sx, const unsigned long long* __restrict__ sz, unsigned nx, unsigned nz,
                                              const float4* __restrict__ hot_of, const float4* __restrict__ sb_lo, const float4* __restrict__ sb_hi, float4* hot, float4* brick) {
    const unsigned a = blockIdx.x * 256 + threadIdx.x;
    const unsigned total = nx + nz;
    if (a >= total + 16u) return;
    if (a >= total) {                                                   // pad entries that can never test positive (the walk over-reads)
        const float qn = __builtin_nanf("");
        hot[a] = make_float4(qn, qn, qn, qn); brick[2 * (size_t)a] = make_float4(qn, qn, qn, 0.f); brick[2 * (size_t)a + 1] = make_float4(qn, qn, qn, qn);
        return;
    }
    const int s = (int)(unsigned)(a < nx ? sx[a] : sz[a - nx]);
    hot[a] = hot_of[s]; brick[2 * (size_t)a] = sb_lo[s]; brick[2 * (size_t)a + 1] = sb_hi[s];
}
__global__ __launch_bounds__(256) void k_cellstarts(const unsigned long long* __restrict__ sx, const unsigned long long* __restrict__ sz, unsigned nx, unsigned nz, unsigned nbin, int32_t* cs) {
    const unsigned c = blockIdx.x * 256 + threadIdx.x;
    if (c > nbin) return;
    for (int copy = 0; copy < 2; ++copy) {
        const unsigned long long* k = copy ? sz : sx;
        unsigned lo = 0, hi = copy ? nz : nx;                           // first registration with bin >= c
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if ((unsigned)(k[mid] >> 32) < c) lo = mid + 1; else hi = mid; }
        cs[(size_t)copy * (nbin + 1) + c] = (int32_t)(copy ? nx + lo : lo);
    }
}
__global__ __launch_bounds__(256) void k_large(int n_world, const int* __restrict__ is_large, const int* __restrict__ large_off, const float4* __restrict__ hot_of,
                                               const float4* __restrict__ sb_lo, const float4* __restrict__ sb_hi, float4* large_hot, float4* large_brick) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_world || !is_large[s]) return;
    const int o = large_off[s];
    large_hot[o] = hot_of[s]; large_brick[2 * o] = sb_lo[s]; large_brick[2 * o + 1] = sb_hi[s];
}

// ---------------------------------------------------------------------------------------------- host side
struct Arena {                                       // carves aligned pieces out of one device allocation
    char* base = nullptr; size_t size = 0, used = 0;
    template <class T> T* take(size_t count) { used = (used + 255) & ~(size_t)255; T* p = (T*)(base + used); used += count * sizeof(T); return p; }
};
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

template <class K> static int sort_keys(K* in, K* out, size_t n, unsigned begin_bit, unsigned end_bit, void* temp, size_t temp_bytes, hipStream_t st) {
    size_t need = 0;
    RT_TRY(rocprim::radix_sort_keys(nullptr, need, in, out, n, begin_bit, end_bit, st));
    if (need > temp_bytes) return RT_ENOTSUP;                        // more scratch than the workspace has: the caller builds on the host
    RT_TRY(rocprim::radix_sort_keys(temp, need, in, out, n, begin_bit, end_bit, st));
    return 0;
}
static int excl_scan(int* in, int* out, size_t n, void* temp, size_t temp_bytes, hipStream_t st) {
    size_t need = 0;
    RT_TRY(rocprim::exclusive_scan(nullptr, need, in, out, 0, n, rocprim::plus<int>(), st));
    if (need > temp_bytes) return RT_ENOTSUP;
    RT_TRY(rocprim::exclusive_scan(temp, need, in, out, 0, n, rocprim::plus<int>(), st));
    return 0;
}
static unsigned bits_for(unsigned long long v) { unsigned b = 1; while (b < 64 && (v >> b)) ++b; return b; }

template <class R> static const float (*device_boxes(int* rc))[6] {   // the 585 boxes (float images of real_t), uploaded once per device
    static std::mutex mu;
    static float (*d_box[64])[6] = {};                                 // by HIP device ordinal: a pointer is only good on the device that made it
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) { *rc = e != hipSuccess ? (int)e : RT_ENOTSUP; return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if (!d_box[dev]) {
        static float h_box[kFullNodes][6];
        full_tree_boxes<R>(h_box);
        void* p = nullptr;
        e = hipMalloc(&p, sizeof(h_box));
        if (e == hipSuccess) e = hipMemcpy(p, h_box, sizeof(h_box), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (p) (void)hipFree(p); *rc = (int)e; return nullptr; }
        d_box[dev] = (float (*)[6])p;
    }
    return d_box[dev];
}

// Builds into O (a fresh handle whose Lazy block is empty).  geom/kind: the world's device copies.  Returns RT_ENOTSUP when the
// device build cannot take this input (the caller then builds on the host).
int build(rt_octree* O, const float4* d_geom, const int32_t* d_kind, int n, int spl, hipStream_t st) {
    int rc = 0;
    const bool fp16 = O->precision == RT_PRECISION_FP16;
    const float (*d_box)[6] = fp16 ? device_boxes<half_t>(&rc) : device_boxes<float>(&rc);
    if (!d_box) return rc;
    rt_octree::Lazy& Z = *O->z;
    // ---- workspace: everything whose size is known from n (freed at the end)
    const size_t cap = (size_t)16 * n + 65536;                         // (cell, sphere) pairs; more -> host build
    const size_t sort_tmp = (size_t)64 << 20;
    const size_t ws_bytes = cap * 16 + (size_t)n * (16 * 4 + 4 * 8 + 16 + 16) + sort_tmp + ((size_t)1 << 20);
    void* ws_mem = nullptr;
    RT_TRY(hipMalloc(&ws_mem, ws_bytes));
    struct Guard { void* p; ~Guard() { if (p) (void)hipFree(p); } } guard{ws_mem};
    Arena W; W.base = (char*)ws_mem; W.size = ws_bytes;
    Counters* C = W.take<Counters>(1);
    unsigned* first = W.take<unsigned>(kFullNodes);
    unsigned* seg_start = W.take<unsigned>(kFullNodes);
    unsigned* seg_end = W.take<unsigned>(kFullNodes);
    int* node_id = W.take<int>(kFullNodes); int* dev_index = W.take<int>(kFullNodes); int* accepted = W.take<int>(kFullNodes);
    int* leaf_of = W.take<int>(kFullNodes * 8); int* hit_cnt = W.take<int>(kFullNodes); int* ent_first = W.take<int>(kFullNodes);
    unsigned long long* pairs_a = W.take<unsigned long long>(cap); unsigned long long* pairs_b = W.take<unsigned long long>(cap);
    float4* hot_of = W.take<float4>(n); float4* sb_lo = W.take<float4>(n); float4* sb_hi = W.take<float4>(n);
    int* multi = W.take<int>(n); int* row_of = W.take<int>(n); int* is_large = W.take<int>(n); int* large_off = W.take<int>(n);
    int* nreg_x = W.take<int>(n); int* nreg_z = W.take<int>(n); int* off_x = W.take<int>(n); int* off_z = W.take<int>(n); int4* range = W.take<int4>(n); int2* bins = W.take<int2>(n);
    unsigned* r2a = W.take<unsigned>(n); unsigned* r2b = W.take<unsigned>(n);
    void* tmp = W.take<char>(sort_tmp);
    if (W.used > W.size) return RT_ENOMEM;
    Counters hc; memset(&hc, 0, sizeof(hc));
    hc.ylo = ~0ull; hc.yhi = 0ull; hc.rmax = 0ull;
    RT_TRY(hipMemcpyAsync(C, &hc, sizeof(hc), hipMemcpyHostToDevice, st));
    RT_TRY(hipMemsetAsync(first, 0xff, sizeof(unsigned) * kFullNodes, st));
    RT_TRY(hipMemsetAsync(seg_start, 0xff, sizeof(unsigned) * kFullNodes, st));
    RT_TRY(hipMemsetAsync(seg_end, 0xff, sizeof(unsigned) * kFullNodes, st));
    RT_TRY(hipMemsetAsync(leaf_of, 0, sizeof(int) * kFullNodes * 8, st));
    RT_TRY(hipMemsetAsync(hot_of, 0, sizeof(float4) * (size_t)n, st));
    // ---- the reference-layout tree
    if (n > 1) {
        if (fp16) hipLaunchKernelGGL(k_pairs<half_t>, dim3(blocks_for((size_t)n - 1)), dim3(256), 0, st, d_geom, n, d_box, first, pairs_a, (unsigned)cap, C);
        else hipLaunchKernelGGL(k_pairs<float>, dim3(blocks_for((size_t)n - 1)), dim3(256), 0, st, d_geom, n, d_box, first, pairs_a, (unsigned)cap, C);
    }
    RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
    RT_TRY(hipStreamSynchronize(st));                                   // (1) how many pairs to sort
    if (hc.pair_overflow) return RT_ENOTSUP;
    const unsigned n_pairs = hc.n_pairs;
    if (n_pairs) {
        if ((rc = sort_keys(pairs_a, pairs_b, n_pairs, 0, 32 + 10, tmp, sort_tmp, st))) return rc;
        hipLaunchKernelGGL(k_segments, dim3(blocks_for(n_pairs)), dim3(256), 0, st, (const unsigned long long*)pairs_b, n_pairs, seg_start, seg_end);
    }
    // reference-layout arrays at their worst-case size (4097 leaves): part of the final allocation, sized below; until then in the workspace
    hipLaunchKernelGGL(k_number, dim3(1), dim3(1024), 0, st, (const unsigned*)first, (const unsigned*)seg_start, (const unsigned*)seg_end, (const unsigned long long*)pairs_b, spl,
                       node_id, dev_index, accepted, leaf_of, C);
    RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
    RT_TRY(hipStreamSynchronize(st));                                   // (2) nodeCount, leafCount: sizes of the tree's arrays
    const int node_count = (int)hc.node_count, leaf_count = (int)hc.leaf_count;
    // ---- first part of the tree's own allocation: reference layout + traversal copy (entries bounded by the pairs)
    const size_t ent_cap = n_pairs ? n_pairs : 1;
    size_t a_bytes = sizeof(rt_octnode) * RT_OCTREE_MAX_NODES + sizeof(int32_t) * (size_t)leaf_count * (1 + (size_t)spl) + sizeof(DevNode) * (size_t)node_count
                   + ent_cap * (16 + 4 + 4 + 8 + 8) + (size_t)n * (4 + 4 + 4) + 512 * 4 + (size_t)node_count * 4 + 64 * 256
                   + (fp16 ? (ent_cap / 2 + (size_t)node_count + 1) * (16 + 8) + (size_t)node_count * 8 + 27 * 4 + 8 * 256 : 0);
    void* a_mem = nullptr;
    RT_TRY(hipMalloc(&a_mem, a_bytes));
    Z.d_arena = a_mem;                                                  // owned by the handle from here on (rt_free_octree)
    Arena A; A.base = (char*)a_mem; A.size = a_bytes;
    rt_octnode* ref_nodes = A.take<rt_octnode>(RT_OCTREE_MAX_NODES);
    int32_t* leaf_cnt = A.take<int32_t>(leaf_count); int32_t* leaf_idx = A.take<int32_t>((size_t)leaf_count * spl);
    DevNode* dnodes = A.take<DevNode>(node_count);
    float4* ent_hot = A.take<float4>(ent_cap); int32_t* ent_id = A.take<int32_t>(ent_cap);
    int32_t* memb_cell = A.take<int32_t>(ent_cap); int32_t* memb_start = A.take<int32_t>((size_t)n + 1);
    int32_t* bits_index = A.take<int32_t>(n); int32_t* cellnode = A.take<int32_t>(512); int32_t* devcell = A.take<int32_t>(node_count);
    unsigned long long* pair2a = A.take<unsigned long long>(ent_cap); unsigned long long* pair2b = A.take<unsigned long long>(ent_cap);
    int* dev_to_fr = A.take<int>(node_count);
    const size_t pair_cap = ent_cap / 2 + (size_t)node_count + 1;       // binary16 only: every node may add a padding half-pair
    uint4* h_pairs = fp16 ? A.take<uint4>(pair_cap) : nullptr; int32_t* h_pid = fp16 ? A.take<int32_t>(2 * pair_cap) : nullptr;
    int* pair_first = fp16 ? A.take<int>(node_count) : nullptr; float* d_planes = fp16 ? A.take<float>(27) : nullptr;
    if (A.used > A.size) return RT_ENOMEM;
    RT_TRY(hipMemsetAsync(ref_nodes, 0, sizeof(rt_octnode) * RT_OCTREE_MAX_NODES, st));
    RT_TRY(hipMemsetAsync(leaf_cnt, 0, sizeof(int32_t) * (size_t)leaf_count, st));
    RT_TRY(hipMemsetAsync(leaf_idx, 0, sizeof(int32_t) * (size_t)leaf_count * spl, st));
    RT_TRY(hipMemsetAsync(memb_start, 0, sizeof(int32_t) * ((size_t)n + 1), st));
    hipLaunchKernelGGL(k_leaves, dim3(kFullNodes), dim3(64), 0, st, (const unsigned*)seg_start, (const int*)accepted, (const int*)leaf_of, (const unsigned long long*)pairs_b, d_kind, spl,
                       leaf_cnt, leaf_idx, hit_cnt);
    hipLaunchKernelGGL(k_nodes, dim3(1), dim3(1024), 0, st, d_box, (const int*)node_id, (const int*)dev_index, (const int*)accepted, (const int*)leaf_of, (const int*)hit_cnt, spl,
                       ref_nodes, dnodes, ent_first, devcell, cellnode, dev_to_fr, C);
    if (fp16) hipLaunchKernelGGL(k_entries<half_t>, dim3(kFullNodes), dim3(64), 0, st, (const unsigned*)seg_start, (const int*)accepted, (const int*)dev_index, (const int*)ent_first,
                                 (const unsigned long long*)pairs_b, d_geom, d_kind, ent_hot, ent_id, pair2a, hot_of);
    else hipLaunchKernelGGL(k_entries<float>, dim3(kFullNodes), dim3(64), 0, st, (const unsigned*)seg_start, (const int*)accepted, (const int*)dev_index, (const int*)ent_first,
                            (const unsigned long long*)pairs_b, d_geom, d_kind, ent_hot, ent_id, pair2a, hot_of);
    if (fp16) {
        // the binary16 kernels' layout: pairs, pair-based node ranges, plane table; no candidate grid (its error bounds are binary32 bounds)
        float h_box[kFullNodes][6];
        full_tree_boxes<half_t>(h_box);
        float planes[27];
        for (int axis = 0; axis < 3; ++axis) {                          // the 9 planes per axis, ascending: the low faces of the 8 level-3 cells + the root's high face
            for (int c = 0; c < 8; ++c) {
                const int bit = 2 - axis;
                const int a = ((c >> 2) & 1) << bit, b = ((c >> 1) & 1) << bit, cc = (c & 1) << bit;
                planes[9 * axis + c] = h_box[full_rank(3, a, b, cc)][axis];
            }
            planes[9 * axis + 8] = h_box[0][3 + axis];
        }
        RT_TRY(hipMemcpyAsync(d_planes, planes, sizeof(planes), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(64), 0, st, dnodes, node_count, (const int*)dev_to_fr, pair_first, C);
        hipLaunchKernelGGL(k_pair_fill, dim3(node_count), dim3(64), 0, st, dnodes, (const int*)pair_first, (const float4*)ent_hot, (const int32_t*)ent_id, h_pairs, h_pid);
        RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
        RT_TRY(hipStreamSynchronize(st));
        RT_TRY(hipGetLastError());
        if (hc.reg_total >= 2u << 23) return RT_ENOTSUP;               // as the host build: a pair index must fit the kernels' segment words
        O->n_nodes = node_count; O->n_entries = (int)hc.n_entries; O->n_world = n;
        Z.dev.n_nodes = node_count; Z.dev.n_entries = (int)hc.reg_total;
        Z.dev.nodes4 = (const float4*)dnodes; Z.dev.ent_hot = (const float4*)h_pairs; Z.dev.ent_id = h_pid;
        Z.dev.h16_planes = d_planes; Z.dev.h16_np[0] = Z.dev.h16_np[1] = Z.dev.h16_np[2] = 9;
        Z.d_ref_nodes = ref_nodes; Z.d_leaf_count = leaf_cnt; Z.d_leaf_indices = leaf_idx;
        Z.ref_node_count = node_count; Z.ref_leaf_count = leaf_count; Z.ref_dropped_full = (int)hc.dropped_full; Z.ref_dropped_outside = (int)hc.dropped_outside; Z.ref_spl = spl;
        DevAccel off{}; O->accel.p = off; O->accel.n_entries = 0; Z.dev.acc = off;
        Z.uploaded = true;
        return 0;
    }
    RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
    RT_TRY(hipStreamSynchronize(st));                                   // (3) number of bucket entries
    const int n_entries = (int)hc.n_entries;
    // ---- memberships, bricks, bitmaps, median radius
    if (n_entries) {
        if ((rc = sort_keys(pair2a, pair2b, (size_t)n_entries, 0, 32 + bits_for((unsigned long long)n), tmp, sort_tmp, st))) return rc;
        hipLaunchKernelGGL(k_memb, dim3(blocks_for((size_t)n_entries)), dim3(256), 0, st, (const unsigned long long*)pair2b, n_entries, n, memb_cell, memb_start);
    }
    hipLaunchKernelGGL(k_bricks, dim3(blocks_for((size_t)n)), dim3(256), 0, st, n, (const int32_t*)memb_start, (const int32_t*)memb_cell, (const int32_t*)devcell, (const float4*)hot_of,
                       sb_lo, sb_hi, multi, r2a, C);
    if ((rc = excl_scan(multi, row_of, (size_t)n, tmp, sort_tmp, st))) return rc;
    if ((rc = sort_keys(r2a, r2b, (size_t)n, 0, 32, tmp, sort_tmp, st))) return rc;
    RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
    int last_multi = 0, last_row = 0;
    RT_TRY(hipMemcpyAsync(&last_multi, multi + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&last_row, row_of + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipStreamSynchronize(st));                                   // (4a) tree spheres, bitmap rows
    const unsigned in_tree = hc.in_tree;
    const int bit_rows = last_row + last_multi;
    AccelHost& AH = O->accel;
    DevAccel p{};
    O->n_nodes = node_count; O->n_entries = n_entries; O->n_world = n; O->bit_rows = std::max(1, bit_rows);
    Z.dev.n_nodes = node_count; Z.dev.n_entries = n_entries;
    Z.dev.nodes4 = (const float4*)dnodes; Z.dev.ent_hot = ent_hot; Z.dev.ent_id = ent_id;
    Z.d_ref_nodes = ref_nodes; Z.d_leaf_count = leaf_cnt; Z.d_leaf_indices = leaf_idx;
    Z.ref_node_count = node_count; Z.ref_leaf_count = leaf_count; Z.ref_dropped_full = (int)hc.dropped_full; Z.ref_dropped_outside = (int)hc.dropped_outside; Z.ref_spl = spl;
    if (in_tree == 0) {                                                 // nothing to cull: the scan serves (as build_accel does for an empty tree)
        p.enabled = 0;
        AH.p = p; AH.n_entries = 0; Z.dev.acc = p; Z.uploaded = true;
        return 0;
    }
    unsigned median_bits = 0;
    RT_TRY(hipMemcpy(&median_bits, r2b + in_tree / 2, sizeof(unsigned), hipMemcpyDeviceToHost));     // (4b) radii[size / 2] of the sorted radii
    float med_r2; memcpy(&med_r2, &median_bits, 4);
    // ---- cell size and extent, as build_accel (rt_accel.h step 2)
    const double rmed = std::sqrt((double)med_r2);
    double h = 2.0 * accel_Rp(rmed * rmed);
    h = std::min(1.0, std::max(0.05, h));
    { const double g = std::ceil(2.0 * (kRootHalfXZ + 5.0 * h) / h); h = std::max(0.05, (4.0 * (double)in_tree > 8.0 * g * g ? kDenseCell : kSparseCell) * h); }
    GridParams P; P.h = h; P.Rlim = 1.5 * h;
    const double half = kRootHalfXZ + 2.0 * P.Rlim + 2.0 * h;
    P.G = (int)std::ceil(2.0 * half / h); P.g0 = -half;
    P.F = RT_ACCEL_FINE; P.Gf = P.G * P.F;
    const int G = P.G, Gf = P.Gf;
    hipLaunchKernelGGL(k_classify, dim3(blocks_for((size_t)n)), dim3(256), 0, st, n, (const int32_t*)memb_start, (const float4*)hot_of, P, is_large, nreg_x, nreg_z, range, bins, C);
    if ((rc = excl_scan(is_large, large_off, (size_t)n, tmp, sort_tmp, st))) return rc;
    if ((rc = excl_scan(nreg_x, off_x, (size_t)n, tmp, sort_tmp, st))) return rc;
    if ((rc = excl_scan(nreg_z, off_z, (size_t)n, tmp, sort_tmp, st))) return rc;
    int tail[6] = {0, 0, 0, 0, 0, 0};
    RT_TRY(hipMemcpyAsync(&tail[0], is_large + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&tail[1], large_off + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&tail[2], nreg_x + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&tail[3], off_x + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&tail[4], nreg_z + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&tail[5], off_z + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    RT_TRY(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
    RT_TRY(hipStreamSynchronize(st));                                   // (5) registrations, large spheres, y-slab
    const int n_large = tail[0] + tail[1];
    const size_t nx = (size_t)tail[2] + (size_t)tail[3], nz = (size_t)tail[4] + (size_t)tail[5], total = nx + nz;
    const size_t ncell = (size_t)G * G, nbin = (size_t)G * Gf;
    // ---- second part of the tree's allocation: the grid
    const size_t b_bytes = (total + 16) * (16 + 32) + 2 * (nbin + 1) * 4 + (size_t)std::max(1, n_large) * (16 + 32) + (size_t)std::max(1, bit_rows) * 64 + (std::max<size_t>(nx, 1) + std::max<size_t>(nz, 1)) * 8 + 32 * 256;
    void* b_mem = nullptr;
    RT_TRY(hipMalloc(&b_mem, b_bytes));
    Z.d_acc[7] = b_mem;                                                 // freed with the other accel buffers
    Arena B; B.base = (char*)b_mem; B.size = b_bytes;
    float4* g_hot = B.take<float4>(total + 16); float4* g_brick = B.take<float4>(2 * total + 32);
    int32_t* cs = B.take<int32_t>(2 * (nbin + 1));
    float4* large_hot = B.take<float4>(std::max(1, n_large)); float4* large_brick = B.take<float4>(2 * (size_t)std::max(1, n_large));
    uint32_t* cellbits = B.take<uint32_t>(16 * (size_t)std::max(1, bit_rows));
    unsigned long long* kx = B.take<unsigned long long>(std::max<size_t>(nx, 1)); unsigned long long* kz = B.take<unsigned long long>(std::max<size_t>(nz, 1));
    if (B.used > B.size) return RT_ENOMEM;
    unsigned long long* kxs = pairs_a; unsigned long long* kzs = pairs_b;                       // sorted keys in the workspace (the pairs are done with)
    if (nx > cap || nz > cap) return RT_ENOTSUP;
    RT_TRY(hipMemsetAsync(cellbits, 0, 64 * (size_t)std::max(1, bit_rows), st));
    RT_TRY(hipMemsetAsync(large_brick, 0, 32 * (size_t)std::max(1, n_large), st));
    hipLaunchKernelGGL(k_bitrows, dim3(blocks_for((size_t)n)), dim3(256), 0, st, n, (const int32_t*)memb_start, (const int32_t*)memb_cell, (const int32_t*)devcell, (const int*)multi, (const int*)row_of, bits_index, cellbits);
    if (total) {
        hipLaunchKernelGGL(k_regs, dim3(blocks_for((size_t)n)), dim3(256), 0, st, n, (const int*)nreg_x, (const int*)off_x, (const int*)off_z, (const int4*)range, (const int2*)bins, Gf, kx, kz);
        const unsigned kb = 32 + bits_for((unsigned long long)nbin);
        if (nx && (rc = sort_keys(kx, kxs, nx, 0, kb, tmp, sort_tmp, st))) return rc;
        if (nz && (rc = sort_keys(kz, kzs, nz, 0, kb, tmp, sort_tmp, st))) return rc;
    }
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(total + 16)), dim3(256), 0, st, (const unsigned long long*)kxs, (const unsigned long long*)kzs, (unsigned)nx, (unsigned)nz,
                       (const float4*)hot_of, (const float4*)sb_lo, (const float4*)sb_hi, g_hot, g_brick);
    hipLaunchKernelGGL(k_cellstarts, dim3(blocks_for(nbin + 1)), dim3(256), 0, st, (const unsigned long long*)kxs, (const unsigned long long*)kzs, (unsigned)nx, (unsigned)nz, (unsigned)nbin, cs);
    if (n_large) hipLaunchKernelGGL(k_large, dim3(blocks_for((size_t)n)), dim3(256), 0, st, n, (const int*)is_large, (const int*)large_off, (const float4*)hot_of,
                                    (const float4*)sb_lo, (const float4*)sb_hi, large_hot, large_brick);
    RT_TRY(hipGetLastError());
    RT_TRY(hipStreamSynchronize(st));                                   // the workspace is freed on return
    // ---- parameters, as build_accel
    p.large_hot = large_hot; p.large_brick = large_brick; p.cs = cs; p.hot = g_hot; p.brick = g_brick;
    p.zoff = (int32_t)(nbin + 1);
    p.memb_start = memb_start; p.memb_cell = memb_cell; p.bits_index = bits_index; p.cellbits = cellbits; p.cellnode = cellnode;
    p.n_large = n_large; p.G = G; p.F = P.F; p.Gf = Gf; p.g0 = (float)P.g0; p.h = (float)h; p.inv_h = (float)(1.0 / h);
    double ylo = double_of(hc.ylo), yhi = double_of(hc.yhi), rmax = hc.rmax ? double_of(hc.rmax) : 0.0;
    if (total == 0 && hc.ylo == ~0ull) { ylo = 0; yhi = 0; rmax = 0; }
    p.ylo = (float)(ylo - 1e-4); p.yhi = (float)(yhi + 1e-4); p.rmax = (float)(rmax + 1e-4);
    p.rq_c = accel_query_growth(rmax, h);
    p.zone2 = (float)(kZone * kZone);
    p.enabled = 1;
    p.coop_groups = (double)hc.reg_cells <= 8.0 * (double)ncell ? 4 : 1;
    p.solo_chains = (double)hc.reg_cells <= RT_SOLO_DENSITY * (double)ncell ? 1 : 0;
    AH.p = p; AH.n_entries = nx; AH.n_entries_z = nz;
    Z.dev.acc = p;
    Z.uploaded = true;
    return 0;
}

}  // namespace gpubuild
}  // namespace rt
