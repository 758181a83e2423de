// rt_octgeom.h — the fixed geometry of the reference's Octree (acceleration_structure.h:104-217), shared by the host build
// (host/rt_scene.hpp) and the device build (rt_build.hip).
//
// The reference's tree is the root box (-11,0,-11)-(11,2,11) (:203) halved three times: 1 + 8 + 64 + 512 = 585 possible nodes.
// A node of the FULL tree is named by its pre-order rank `fr` (children in octant order, the order insert() visits them):
//   root 0; level-1 octant a: 1 + 73 a; level-2 (a,b): 1 + 73 a + 1 + 9 b; level-3 (a,b,c): 1 + 73 a + 1 + 9 b + 1 + c.
// Octant index: bit 2 = x high, bit 1 = y high, bit 0 = z high (:149-165).
#pragma once
#include "rt_real.h"

namespace rt {

constexpr int kFullNodes = 585;
RT_HD int full_rank(int level, int a, int b, int c) {
    return level == 0 ? 0 : level == 1 ? 1 + 73 * a : level == 2 ? 2 + 73 * a + 9 * b : 3 + 73 * a + 9 * b + c;
}
RT_HD int full_subtree(int level) { return level == 0 ? 585 : level == 1 ? 73 : level == 2 ? 9 : 1; }
// level and octant path of a pre-order rank
RT_HD void full_path(int fr, int& level, int& a, int& b, int& c) {
    a = b = c = 0;
    if (fr == 0) { level = 0; return; }
    const int r1 = fr - 1; a = r1 / 73;
    const int in1 = r1 % 73;
    if (in1 == 0) { level = 1; return; }
    const int r2 = in1 - 1; b = r2 / 9;
    const int in2 = r2 % 9;
    if (in2 == 0) { level = 2; return; }
    level = 3; c = in2 - 1;
}
// level-3 cell coordinates (0..7 per axis) of an octant path: the bits of the three octants, most significant first
RT_HD int cell_coord(int a, int b, int c, int axis_bit) { return (((a >> axis_bit) & 1) << 2) | (((b >> axis_bit) & 1) << 1) | ((c >> axis_bit) & 1); }

// intersects(sphere, AABB) — acceleration_structure.h:82-93: centre inside the box grown by the radius, x_low strict
template <class R> RT_HD bool sphere_touches_box(R cx, R cy, R cz, R rad, const R* lo, const R* hi) {
    const R l0 = lo[0] - rad, l1 = lo[1] - rad, l2 = lo[2] - rad;
    const R h0 = hi[0] + rad, h1 = hi[1] + rad, h2 = hi[2] + rad;
    return (cx > l0 && cx <= h0) && (cy >= l1 && cy <= h1) && (cz >= l2 && cz <= h2);
}

// the boxes of all 585 nodes as float images of real_t: box[fr] = (x_low, y_low, z_low, x_high, y_high, z_high); child boxes by
// the float midpoint low + (high - low) / 2 (:141), exactly as insert() derives them on the way down
template <class R> inline void full_tree_boxes(float (*box)[6]) {
    const float root[6] = {-11, 0, -11, 11, 2, 11};
    for (int k = 0; k < 6; ++k) box[0][k] = as_float(real_from<R>(root[k]));
    for (int fr = 1; fr < kFullNodes; ++fr) {
        int level, a, b, c;
        full_path(fr, level, a, b, c);
        const int parent = level == 1 ? 0 : level == 2 ? full_rank(1, a, 0, 0) : full_rank(2, a, b, 0);
        const int oct = level == 1 ? a : level == 2 ? b : c;
        for (int k = 0; k < 3; ++k) {
            const float lo = box[parent][k], hi = box[parent][3 + k];
            const float mid = as_float(real_from<R>(lo + (hi - lo) / 2));
            const bool high = (oct >> (2 - k)) & 1;
            box[fr][k] = high ? mid : lo;
            box[fr][3 + k] = high ? hi : mid;
        }
    }
}

}  // namespace rt
