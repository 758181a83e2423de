// rt_handles.h — the opaque handles of include/rt_amd.h as the library's translation units see them (rt_api.hip, rt_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "rt_device.h"
#include "../host/rt_scene.hpp"
#include "rt_accel.h"

using rt::DevScene; using rt::DevTree; using rt::DevNode; using rt::DevAccel; using rt::AccelHost; using rt::Octree;

// Handles keep their host staging copy; device buffers are created by rt_world_upload / rt_octree_upload
// (called implicitly by the first render/trace that uses the handle).  A handle passed as `const` to a compute call stays
// logically constant: what such a call may create lazily (device copies, the list grid, the default render context) lives
// behind a pointer in a `Lazy` block of its own.
struct rt_octree;

// Per-launch state of rt_render: work counters, scheduling workspace, timing events.  One context serves one launch at a
// time: calls on the same context are ordered by the library (an event recorded behind the render kernel, which the next
// call's stream waits for before it touches the workspace), so two streams sharing a context serialise instead of racing;
// give concurrent frames a context each (rt_render_ctx_create).
struct rt_render_ctx {
    // work counters of the persistent render kernel: a ring of slots (one per launch, 256 B apart), zeroed on the stream
    unsigned int* d_queue = nullptr; unsigned launches = 0;
    // scheduling workspace (tile costs, hand-out order, long-chain flags and list), grown on demand
    int* d_cost = nullptr; unsigned int* d_order = nullptr; unsigned char* d_flags = nullptr; unsigned int* d_long = nullptr; int* d_work = nullptr; int64_t sched_tiles = 0;
    unsigned int* last_queue = nullptr;      // the counters of the latest launch (rt_render_ctx_counters)
    // tile order of a progressive sequence (rt_render_progressive): the pilot pass that the call with current_sample == 1 runs, kept
    // in buffers of its own and reused by the following passes of the same frame (p_key: world and tree serials, frame size, partition)
    int* p_cost = nullptr; unsigned int* p_order = nullptr; int64_t p_tiles = 0; bool p_valid = false; uint64_t p_key[5] = {0, 0, 0, 0, 0};
    bool p_pinned = false;      // a captured progressive pass has baked p_order's address into a hipGraph: the buffers are never freed or moved again
    // HIP events around the dominant kernel of each render call (ring of the last 64), see rt_render_ctx_times
    hipEvent_t ev0[64] = {}, ev1[64] = {}; unsigned ev_head = 0, ev_count = 0; bool ev_ready = false;
    // ordering of successive launches that share this context
    hipEvent_t done = nullptr; hipStream_t last_stream = nullptr; bool has_done = false;
};
static const unsigned kQueueSlots = 64, kQueueStride = 64;     // 256 bytes per launch: counters in the first 128-byte line, read-only thresholds in the second (rt_device.h)

uint64_t rt_next_serial();                  // handles are numbered: a new handle at a recycled address is not mistaken for the old one

struct rt_world {
    uint64_t serial = rt_next_serial();
    int precision = RT_PRECISION_FP32;
    int n = 0;
    std::vector<float4> h_hot, h_geom, h_mat;
    std::vector<int32_t> h_ids, h_kind;
    int list_traversal = RT_TRAVERSAL_FAST;
    int arith = RT_ARITH_IEEE;                 // rt_world_set_arith
    struct Lazy {
        bool uploaded = false;
        DevScene dev{};
        void* d_list_hot = nullptr; void* d_list_id = nullptr; void* d_geom = nullptr; void* d_mat = nullptr; void* d_kind = nullptr;
        void* d_shade = nullptr; void* d_kind8 = nullptr;        // geom and mat interleaved, kind as bytes (DevScene::shade / kind8)
        // hitable_list::hit through the candidate grid: the list as a one-node "tree" (fp32 only; null = plain scan)
        rt_octree* list_tree = nullptr; bool list_tree_tried = false;
        rt_render_ctx ctx;                      // the context rt_render / rt_render_progressive use
    };
    Lazy* z = nullptr;
};

struct rt_octree {
    uint64_t serial = rt_next_serial();
    int precision = RT_PRECISION_FP32;
    Octree* host = nullptr;
    std::vector<DevNode> h_nodes; std::vector<float4> h_ent_hot; std::vector<int32_t> h_ent_id;
    AccelHost accel;
    int traversal = RT_TRAVERSAL_FAST;
    struct Lazy {
        bool uploaded = false;
        DevTree dev{};
        void* d_nodes = nullptr; void* d_ent_hot = nullptr; void* d_ent_id = nullptr;
        void* d_acc[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // large_hot, large_brick, cs, hot, brick, memb_start, memb_cell, -, cellnode, bits_index, cellbits, fp16 planes
        // a tree built on the device (rt_build.hip): ONE allocation holds everything above (the pointers point into it), plus
        // the reference-layout arrays that the inspection calls download on demand
        void* d_arena = nullptr;
        const rt_octnode* d_ref_nodes = nullptr; const int32_t* d_leaf_count = nullptr; const int32_t* d_leaf_indices = nullptr;
        int ref_node_count = 0, ref_leaf_count = 0, ref_dropped_full = 0, ref_dropped_outside = 0, ref_spl = 0;
        Octree* host_view = nullptr;            // ... downloaded here when an inspection call first asks for it
    };
    Lazy* z = nullptr;
    int n_nodes = 0, n_entries = 0;
    int n_world = 0, bit_rows = 1;          // world list size, rows of the membership bitmaps (>= 1)
};

