// rt_device.h — device-side data layout shared by the kernels (rt_kernels.hip) and the uploader (rt_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt_amd.h"

namespace rt {

// The reference's root box is fixed: (-11, 0, -11)-(11, 2, 11) (acceleration_structure.h:203).  Halved three times it gives 8 x 8 x 8
// level-3 cells of kCellXZ x kCellY x kCellXZ; everything that depends on that geometry — the cell of a hit point, the bricks' cell
// coordinates, the grid's reach — is written in terms of these constants, here, in the kernels and in the device build.
constexpr double kRootHalfXZ = 11.0;            // the box spans [-kRootHalfXZ, kRootHalfXZ] in x and z, [0, 8 kCellY] in y
constexpr double kCellXZ = 2.0 * kRootHalfXZ / 8.0, kCellY = 0.25;
constexpr float kRootHalfXZf = (float)kRootHalfXZ, kInvCellXZf = 1.0f / 2.75f, kInvCellYf = 4.0f;
static_assert(kCellXZ == 2.75 && 1.0 / kCellY == 4.0, "level-3 cells of the reference's root box");


// The tile split of a frame over parts (rt_amd.h, rt_partition).  Two forms: a contiguous range [begin, end) of the row-major tile
// numbering (end > begin: rt_split_balanced cuts the frame into such bands of equal predicted cost), or — begin == end == 0 — runs of
// RT_PART_RUN consecutive tiles dealt round-robin.  Every kernel and the host go through these functions: nothing else knows the mapping.
#ifndef RT_PART_RUN_BUILD
#define RT_PART_RUN_BUILD RT_PART_RUN           // (tools/mkvariant.sh -DRT_PART_RUN_BUILD=n: A/B of the run length)
#endif
constexpr long long kPartRun = RT_PART_RUN_BUILD;
__host__ __device__ inline bool part_whole(int nparts, long long begin, long long end) { return nparts == 1 && end <= begin; }   // the undivided frame: row-major buffers
__host__ __device__ inline long long part_tile(long long local_tile, int part, int nparts, long long begin = 0, long long end = 0) {          // global tile of a part's local tile
    if (end > begin) return begin + local_tile;
    if (nparts == 1) return local_tile;
    return ((local_tile / kPartRun) * nparts + part) * kPartRun + local_tile % kPartRun;
}
__host__ __device__ inline long long part_local_tiles(long long tiles, int part, int nparts, long long begin = 0, long long end = 0) {        // tiles of the frame that belong to `part`
    if (end > begin) return end - begin;
    const long long round = (long long)nparts * kPartRun;
    long long extra = tiles % round - (long long)part * kPartRun;
    extra = extra < 0 ? 0 : (extra > kPartRun ? kPartRun : extra);
    return tiles / round * kPartRun + extra;
}
__host__ __device__ inline void part_owner(long long tile, int nparts, int& part, long long& local_tile) {   // inverse of part_tile (runs)
    const long long run = tile / kPartRun;
    part = (int)(run % nparts);
    local_tile = (run / nparts) * kPartRun + tile % kPartRun;
}
// is `tile` one of this part's, and which of its local tiles
__host__ __device__ inline bool part_has(long long tile, int part, int nparts, long long begin, long long end, long long& local_tile) {
    if (end > begin) { local_tile = tile - begin; return tile >= begin && tile < end; }
    int owner; part_owner(tile, nparts, owner, local_tile);
    return owner == part;
}
constexpr int kMaxSplitParts = 64;
struct SplitStarts { long long s[kMaxSplitParts + 1]; };      // first tile of every band of a balanced split, and the tile count (k_assemble_split)

// One node of the traversal copy of the Octree, in depth-first pre-order (children in index order, the visit
// order of traverseTree, acceleration_structure.h:276-304).  48 bytes = 3 x 16 B so a lane fetches it with
// three ds_read_b128 from LDS.
struct DevNode {
    float lo[3];      // x_low, y_low, z_low
    float hix;        // x_high
    float hiy, hiz;   // y_high, z_high
    int32_t skip;     // pre-order index of the next node when this one is culled (= end of its subtree)
    int32_t first;    // level-3 node: first entry of its concatenated buckets in ent_hot / ent_id
    int32_t count;    // level-3 node: number of entries (ghost entries removed); 0 for inner nodes
    int32_t ref_index;// index of this node in the reference-layout nodes[] (diagnostics)
    int32_t pad[2];
};
static_assert(sizeof(DevNode) == 48, "DevNode is 3 x float4");

struct DevScene {
    const float4* list_hot;   // [n_list] (cx, cy, cz, radius*radius) of the hittable spheres, list order
    const int32_t* list_id;   // [n_list] index into the world list
    const float4* geom;       // [n] (cx, cy, cz, radius)
    const float4* mat;        // [n] (albedo r,g,b, param)
    const int32_t* kind;      // [n] RT_MAT_*
    const float4* shade;      // [2n] what a hit needs, side by side: geom[i], mat[i] — one 32-byte record, one cache line per hit
    const uint8_t* kind8;     // [n] RT_MAT_* as bytes (10 KB at C3: stays in the vector L1)
    int32_t n, n_list;
    int32_t ground_valid;     // world list slot 0 is hittable (it is tested first by hitTree)
    rt_camera cam;
};

// Candidate-culling structure for the fast closest-hit path.  It never decides a hit: it only
// enumerates a superset of the spheres whose float sphere::hit can succeed; membership of a sphere in a level-3 cell
// that the reference's traversal visits is then checked with the reference's own slab test.
struct DevAccel {
    const float4* large_hot;   // [n_large] (cx,cy,cz,r^2): tree spheres too big (or too far out) for the grid, always tested
    const float4* large_brick; // [2*n_large] as `brick`
    // The grid: columns of width h along a ray's major axis, each sphere registered ONCE in every column its inflated extent
    // overlaps, a column's entries sorted by the sphere's CENTRE along the minor axis into Gf = G * F fine bins (bin width h / F).
    // A ray asks a column for the bins its line crosses grown by the largest inflated radius (rq_c, in cell units) — the
    // inflation sits in the query, not in the registration, so no sphere appears twice in a column's range.
    // Two copies: columns along x (bin (ix, fz) at ix*Gf + fz) in cs[0 .. G*Gf], columns along z (bin (iz, fx) at iz*Gf + fx) in
    // cs[zoff .. zoff + G*Gf]; cs values index hot[] / brick[] directly (the z copy's entries follow the x copy's).
    const int32_t* cs;
    const float4* hot;         // (cx,cy,cz,r^2)
    // per entry two float4: (lo.x, lo.y, lo.z, bits of the world-list index), (hi.x, hi.y, hi.z, bits of node1) —
    // lo/hi = the sphere's brick in level-3 cell coordinates, margins included (rt_accel.h), lo = +inf when it has none;
    // node1 = the single level-3 node (pre-order index) that stores the sphere, or -1 if several do
    const float4* brick;
    int32_t zoff;
    const int32_t* memb_start; // [n+1] per world-list index: range in memb_cell
    const int32_t* memb_cell;  // pre-order node index (DevNode) of each level-3 node whose buckets hold the sphere
    const int32_t* bits_index; // [n] per world-list index: row of cellbits for spheres stored in several nodes, else -1
    const uint32_t* cellbits;  // rows of 16 words: bit (ix*64 + iy*8 + iz) set = the sphere is stored in that level-3 cell
    const int32_t* cellnode;   // [512] level-3 cell (ix*64 + iy*8 + iz of the root box's 8x8x8 grid) -> pre-order node index, or -1
    int32_t n_large, G;
    int32_t F, Gf;             // fine bins per cell along a column's minor axis, G * F
    float rq_c;                // query growth in cell units: (largest inflated radius + walk slack) / h
    float g0, h, inv_h;        // grid origin (same for x and z), cell size
    float ylo, yhi;            // y-slab covering every grid sphere's inflated ball
    float rmax;                // largest inflated radius R' of a grid sphere
    float zone2;               // fast path only for ray origins with |o - (0,1,0)|^2 <= zone2
    int32_t enabled;
    int32_t coop_groups;       // cooperative walk: up to this many rays side by side (4 on sparse grids, 1 on dense ones)
    int32_t solo_chains;       // very sparse grids (at most one entry per cell on average: lists of a few hundred spheres): the pre-classified
                               // long chains start alone in their waves (k_render<true,*,5>)
};

struct DevTree {
    const float4* nodes4;     // [n_nodes*3] DevNode as float4 triples
    const float4* ent_hot;    // [n_entries] (cx, cy, cz, radius*radius) in traversal order
    const int32_t* ent_id;    // [n_entries] index into the world list
    int32_t n_nodes, n_entries;
    // binary16 trees only (rt_kernels_fp16.hip): the distinct box planes of the tree per axis — the reference's boxes are the
    // root box halved three times, 9 planes an axis — so a ray divides once per plane, not six times per visited node.
    // h16_planes = [x planes | y planes | z planes] (h16_np of each; h16_np[0] == 0: no table, the generic slab test);
    // DevNode::pad[0] holds the node's six plane indices, 5 bits each: x_low, x_high, y_low, y_high, z_low, z_high.
    const float* h16_planes;
    int32_t h16_np[3];
    DevAccel acc;
};

constexpr int kQueueThr = 32;
struct RenderArgs {
    void* fb;
    rt_rand_state* rand_state;
    int32_t max_x, max_y, ns;             // ns = current_sample for the progressive kernel
    int32_t tiles_x, tiles_y;
    int32_t part, nparts;
    int64_t tile_begin, tile_end;         // end > begin: this part is the contiguous tile range [begin, end) (local tile = tile - begin)
    int64_t n_local_tiles;
    unsigned int* queue;                  // counters of this launch, zeroed on the stream: [0] work counter [1] thin waves [2] long chains [3] long head
                                          // [4] solo chains [5] solo head; in the slot's second 128-byte line, written by k_tile_order before the render kernel starts and
                                          // only read by it (the first line is hammered by every wave's atomics: a load from it waits behind them):
                                          // [kQueueThr] in-flight chain threshold (iterations; 0 = the rate rule only) [kQueueThr + 1] pilot-sum threshold
                                          // [kQueueThr + 2] first slot of the queue's tail + 1 (0 = no tail): those slots go through tail_list
    int32_t n_lanes;                      // lanes of the persistent render grid this launch will run on (64 x waves): the per-lane load is the yardstick of a "long" pixel
    const unsigned int* tail_list;        // the pixels of the last tiles of the hand-out order, most expensive 2x2 block first (k_tail_hist / k_tail_scatter); NULL = none
    unsigned int* tail_ws;                // 512 words behind the lists: the tail sort's 256 counts and 256 cursors
    float head_min_load;                  // dense grids: launches of fewer predicted iterations per lane keep the whole tail at the end
    int32_t head_sum, head_sum_dense;     // tail pixels whose 3x3 pilot sum reaches this — or, negative, is at most its magnitude — are handed out FIRST (0 = none; sparse grids / dense grids); queue[kQueueThr + 3] = how many, [kQueueThr + 4] = the sum in force, [kQueueThr + 5] = taken from the list's cheap end
    float f_tail;                         // share of the launch's predicted work handed out per pixel instead of per tile, at the end of the queue
    float f_inflight_dense;               // f_inflight of launches on dense grids (k_render<true,*,2>)
    float f_inflight, f_static;           // a pixel is long when its predicted chain exceeds f x (predicted iterations of the launch / n_lanes): found in flight / by the pilot
    const unsigned int* order;            // hand-out order of the local tiles (most expensive first), or NULL = identity
    const unsigned char* long_flag;       // per local pixel (local_tile*64 + l): pre-classified long chain, or NULL
    const unsigned int* long_list;        // the pre-classified long chains (queue[2] = count, queue[3] = next to hand out)
    DevScene scene;
    DevTree tree;
};

} // namespace rt
