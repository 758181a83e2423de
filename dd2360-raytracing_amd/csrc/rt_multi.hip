// rt_multi.hip — one frame over the GPUs of a node: rt_multi_render (include/rt_amd.h).
//
// No reference counterpart (the reference is single-GPU); the launch surface it extends is main.cu:422-427.  One process per
// GPU.  The frame's 8x8 tiles (the reference's block shape, main.cu:351-352) are split over the ranks — by default into horizontal
// bands of equal predicted cost (rt_split_balanced: every rank runs the same pilot pass over the whole frame and cuts it the same
// way, no communication; a GPU's rays then meet the same part of the scene, as in the undivided frame), or dealt round-robin in
// runs of RT_PART_RUN consecutive tiles (RT_SPLIT_RUNS); pixels are independent (the per-pixel RNG is keyed by the
// absolute pixel_index, main.cu:93), so every split gives the bits of the single-GPU frame.  Each rank renders its tiles
// into a compact tile-major buffer (rt_partition) and ONE exchange brings the buffers to the root: with RCCL a single
// ncclGroupStart / ncclRecv x (nranks-1) | ncclSend / ncclGroupEnd straight out of the render buffer and straight into the
// root's staging slots (the root renders into its own slot: no copy anywhere), on the caller's stream; then rt_assemble
// restores the row-major frame.  Payload at 3840x2160: 99.5 MB in all, 12.4 MB per peer, each on its own xGMI link.
//
// RCCL is bound at run time (dlopen "librccl.so.1": a process that already carries an RCCL — PyTorch — shares it; a
// single-GPU user of librt_amd.so needs none).  rt_multi_init_custom takes the exchange as a callback instead (MPI, gloo, a
// test harness): same partition, same buffers, same assemble.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <new>
#include <cstring>
#include <mutex>
#include <algorithm>
#include "../../include/rt_amd.h"
#include "rt_handles.h"

namespace {

// the few RCCL entry points used, with the types of /opt/rocm/include/rccl/rccl.h (ncclResult_t and the enums are ints)
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclFloat16 = 6, kNcclFloat32 = 7 };        // ncclFloat16 / ncclFloat32 of rccl.h
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    bool ok = false;
};
void rccl_bind(Rccl& R) {
    R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) return;
    R.GetUniqueId = (int (*)(NcclId*))dlsym(R.lib, "ncclGetUniqueId");
    R.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))dlsym(R.lib, "ncclCommInitRank");
    R.CommDestroy = (int (*)(NcclComm))dlsym(R.lib, "ncclCommDestroy");
    R.GroupStart = (int (*)())dlsym(R.lib, "ncclGroupStart");
    R.GroupEnd = (int (*)())dlsym(R.lib, "ncclGroupEnd");
    R.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))dlsym(R.lib, "ncclSend");
    R.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))dlsym(R.lib, "ncclRecv");
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv;
}
Rccl& rccl() {                                        // bound once per process, whichever thread asks first
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, rccl_bind, R);
    return R;
}
#define RT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define RT_NCCL(expr) do { if ((expr) != 0) return RT_ECOMM; } while (0)

}  // namespace

struct rt_multi {
    int rank = 0, nranks = 1;
    NcclComm comm = nullptr;
    rt_gather_fn gather = nullptr; void* user = nullptr;
    rt_render_ctx* ctx = nullptr;
    // this rank's RNG states (compact, tile-major), its frame part when it is not the root, the root's staging slots
    void* d_rand = nullptr; size_t rand_bytes = 0;
    void* d_local = nullptr; size_t local_bytes = 0;
    void* d_parts = nullptr; size_t parts_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool timed = false;
    // how the frame is divided (rt_multi_set_split) and the bands of the last balanced split, with what they were computed for
    int split_mode = RT_SPLIT_RUNS;
    int64_t starts[rt::kMaxSplitParts + 1] = {0}; bool have_split = false; uint64_t split_key[4] = {0, 0, 0, 0};
};

static int grow(void** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) { RT_TRY(hipFree(*p)); *p = nullptr; *have = 0; }
    RT_TRY(hipMalloc(p, need));
    *have = need;
    return 0;
}

static int multi_new(rt_multi** out, int rank, int nranks) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return RT_EINVAL;
    *out = nullptr;
    rt_multi* M = new (std::nothrow) rt_multi();
    if (!M) return RT_ENOMEM;
    M->rank = rank; M->nranks = nranks;
    int rc = rt_render_ctx_create(&M->ctx);
    if (!rc) { hipError_t e = hipEventCreate(&M->ev0); if (e == hipSuccess) e = hipEventCreate(&M->ev1); rc = (int)e; }
    if (rc) { rt_multi_destroy(M); return rc; }
    *out = M;
    return 0;
}

extern "C" {

int rt_multi_unique_id(void* id_out) {
    if (!id_out) return RT_EINVAL;
    Rccl& R = rccl();
    if (!R.ok) return RT_ENOTSUP;
    NcclId id;
    RT_NCCL(R.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

// What rt_multi_init needs BEFORE it enters the collective ncclCommInitRank, checked without talking to anybody: RCCL bound
// with every entry point, and a context + events on the calling process's current device.  A job calls this on every rank,
// agrees on the answers over its own control plane (gloo, MPI) and only then lets ALL ranks — or none — call rt_multi_init:
// a rank that returned early from rt_multi_init would leave the others blocked inside the collective.
int rt_multi_probe(void) {
    if (!rccl().ok) return RT_ENOTSUP;
    rt_multi* M = nullptr;
    const int rc = multi_new(&M, 0, 1);
    if (rc) return rc;
    return rt_multi_destroy(M);
}

int rt_multi_init(rt_multi** out, int rank, int nranks, const void* unique_id) {
    if (!unique_id) return RT_EINVAL;
    Rccl& R = rccl();
    if (!R.ok) return RT_ENOTSUP;
    int rc = multi_new(out, rank, nranks);
    if (rc) return rc;
    NcclId id;
    memcpy(&id, unique_id, sizeof(id));
    if (R.CommInitRank(&(*out)->comm, nranks, id, rank) != 0) { rt_multi_destroy(*out); *out = nullptr; return RT_ECOMM; }
    return 0;
}

int rt_multi_init_custom(rt_multi** out, int rank, int nranks, rt_gather_fn gather, void* user) {
    if (!gather) return RT_EINVAL;
    const int rc = multi_new(out, rank, nranks);
    if (rc) return rc;
    (*out)->gather = gather; (*out)->user = user;
    return 0;
}

int rt_multi_destroy(rt_multi* M) {
    if (!M) return 0;
    int rc = 0;
    if (M->comm && rccl().ok && rccl().CommDestroy(M->comm) != 0) rc = RT_ECOMM;
    void* bufs[3] = {M->d_rand, M->d_local, M->d_parts};
    for (void* b : bufs) if (b) { const hipError_t e = hipFree(b); if (e != hipSuccess && !rc) rc = (int)e; }
    if (M->ev0) (void)hipEventDestroy(M->ev0);
    if (M->ev1) (void)hipEventDestroy(M->ev1);
    const int r2 = rt_render_ctx_destroy(M->ctx); if (!rc) rc = r2;
    delete M;
    return rc;
}

int rt_multi_set_split(rt_multi* M, int mode) {
    if (!M || mode < RT_SPLIT_RUNS || mode > RT_SPLIT_BALANCED_CACHED) return RT_EINVAL;
    M->split_mode = mode; M->have_split = false;
    return 0;
}
int rt_multi_last_split(rt_multi* M, int64_t* starts) {
    if (!M || !starts || !M->have_split) return RT_EINVAL;
    for (int p = 0; p <= M->nranks; ++p) starts[p] = M->starts[p];
    return 0;
}

// this rank's buffers for a part of n_mine pixels, the root's staging slots of `per` pixels each
static int multi_buffers(rt_multi* M, int64_t n_mine, int64_t per, size_t px, int root) {
    int rc = grow(&M->d_rand, &M->rand_bytes, (size_t)(n_mine > 0 ? n_mine : 1) * sizeof(rt_rand_state));
    if (!rc && M->rank == root) rc = grow(&M->d_parts, &M->parts_bytes, (size_t)per * px * (size_t)M->nranks);
    if (!rc && M->rank != root) rc = grow(&M->d_local, &M->local_bytes, (size_t)(n_mine > 0 ? n_mine : 1) * px);
    return rc;
}

// buffers for frames of this size (rt_multi_render grows them on demand; call this once before timing or graph capture).
// RT_SPLIT_RUNS: exactly what the frame needs.  Balanced splits: the bands' sizes depend on the scene — reserved here for bands of up to
// twice the mean; a frame whose split asks for more grows them in rt_multi_render.
int rt_multi_reserve(rt_multi* M, int max_x, int max_y, int precision, int root) {
    if (!M || max_x <= 0 || max_y <= 0 || root < 0 || root >= M->nranks) return RT_EINVAL;
    if (precision != RT_PRECISION_FP32 && precision != RT_PRECISION_FP16) return RT_EINVAL;
    const size_t px = precision == RT_PRECISION_FP16 ? 6 : 12;
    const rt_partition mine = {M->rank, M->nranks, 0, 0}, first = {0, M->nranks, 0, 0};
    int64_t n_mine = rt_part_pixels(max_x, max_y, mine), per = rt_part_pixels(max_x, max_y, first);
    const bool balanced = M->split_mode != RT_SPLIT_RUNS && M->nranks > 1;
    if (balanced) { n_mine = per * 2; per = per * 2; }
    int rc = multi_buffers(M, n_mine, per, px, root);
    if (!rc) rc = rt_render_ctx_reserve(M->ctx, max_x, max_y, balanced ? rt_partition{0, 1, 0, 0} : mine);      // (a balanced split runs its pilot pass over the whole frame)
    return rc;
}

int rt_multi_render(rt_multi* M, void* fb_full, int max_x, int max_y, int ns, const rt_world* world, const rt_octree* d_octree, int precision, int root, void* stream) {
    if (!M || !world || max_x <= 0 || max_y <= 0 || ns <= 0 || root < 0 || root >= M->nranks) return RT_EINVAL;
    if (M->rank == root && !fb_full) return RT_EINVAL;
    // the buffers, the RCCL datatype and rt_assemble are sized by `precision`, the render kernel by the world's: they must agree
    // (an fp32 world rendered into binary16-sized buffers would write out of bounds)
    if (precision != world->precision) return RT_EINVAL;
    const hipStream_t st = (hipStream_t)stream;
    const size_t px = precision == RT_PRECISION_FP16 ? 6 : 12;
    const int64_t tiles = (int64_t)((max_x + 7) / 8) * ((max_y + 7) / 8);
    const bool balanced = M->split_mode != RT_SPLIT_RUNS && M->nranks > 1 && tiles >= M->nranks;
    int rc = 0;
    rt_partition mine = {M->rank, M->nranks, 0, 0};
    int64_t per = 0;
    RT_TRY(hipEventRecord(M->ev0, st));                                   // (this rank's share of the frame starts with the split)
    if (balanced) {
        // every rank cuts the frame the same way: the same pilot pass, the same integer arithmetic (rt_split_balanced)
        const uint64_t key[4] = {world->serial, d_octree ? d_octree->serial : 0, ((uint64_t)(uint32_t)max_x << 32) | (uint32_t)max_y,
                                 ((uint64_t)(uint32_t)M->nranks << 32) | (uint32_t)(d_octree ? d_octree->traversal : 0)};
        if (!(M->split_mode == RT_SPLIT_BALANCED_CACHED && M->have_split && memcmp(key, M->split_key, sizeof(key)) == 0)) {
            M->have_split = false;
            if ((rc = rt_split_balanced(M->ctx, world, d_octree, max_x, max_y, M->nranks, M->starts, nullptr, nullptr, nullptr, stream))) return rc;
            memcpy(M->split_key, key, sizeof(key)); M->have_split = true;
        }
        mine.tile_begin = M->starts[M->rank]; mine.tile_end = M->starts[M->rank + 1];
        for (int r = 0; r < M->nranks; ++r) per = std::max<int64_t>(per, (M->starts[r + 1] - M->starts[r]) * 64);
        if ((rc = multi_buffers(M, (mine.tile_end - mine.tile_begin) * 64, per, px, root))) return rc;
    } else {
        M->have_split = false;
        if ((rc = rt_multi_reserve(M, max_x, max_y, precision, root))) return rc;
        per = rt_part_pixels(max_x, max_y, rt_partition{0, M->nranks, 0, 0});
    }
    const size_t stride = (size_t)per * px;
    // the root renders straight into its slot of the staging buffer, the others into their send buffer; a single rank's
    // "part" is the whole frame in the reference's row-major layout (rt_partition, nparts == 1): straight into fb_full
    void* local = M->nranks == 1 ? fb_full : M->rank == root ? (void*)((char*)M->d_parts + stride * (size_t)root) : M->d_local;
    if ((rc = rt_render_init(max_x, max_y, (rt_rand_state*)M->d_rand, mine, stream))) return rc;
    if ((rc = rt_render_on(M->ctx, local, max_x, max_y, ns, world, (rt_rand_state*)M->d_rand, d_octree, mine, stream))) return rc;
    RT_TRY(hipEventRecord(M->ev1, st));
    M->timed = true;
    if (M->nranks > 1) {
        const size_t mine_bytes = (size_t)rt_part_pixels(max_x, max_y, mine) * px;
        if (M->gather) {
            if ((rc = M->gather(M->user, local, mine_bytes, M->rank == root ? M->d_parts : nullptr, stride, root, stream))) return rc > 0 ? RT_ECOMM : rc;
        } else {
            Rccl& R = rccl();
            const int dtype = precision == RT_PRECISION_FP16 ? kNcclFloat16 : kNcclFloat32;
            RT_NCCL(R.GroupStart());                              // the single framebuffer exchange
            if (M->rank == root) {
                for (int r = 0; r < M->nranks; ++r) {
                    if (r == root) continue;
                    const rt_partition pr = {r, M->nranks, balanced ? M->starts[r] : 0, balanced ? M->starts[r + 1] : 0};
                    const size_t count = (size_t)rt_part_pixels(max_x, max_y, pr) * 3;
                    if (count && R.Recv((char*)M->d_parts + stride * (size_t)r, count, dtype, r, M->comm, st) != 0) { (void)R.GroupEnd(); return RT_ECOMM; }
                }
            } else if (mine_bytes) {
                if (R.Send(local, mine_bytes / (px / 3), dtype, root, M->comm, st) != 0) { (void)R.GroupEnd(); return RT_ECOMM; }
            }
            RT_NCCL(R.GroupEnd());
        }
    }
    if (M->rank == root && M->nranks > 1)
        rc = balanced ? rt_assemble_split(fb_full, M->d_parts, max_x, max_y, M->nranks, M->starts, per, precision, stream)
                      : rt_assemble(fb_full, M->d_parts, max_x, max_y, M->nranks, precision, stream);
    return rc;
}

// device time of this rank's own share of the last rt_multi_render (render_init + render, before the exchange), and of
// its render kernel alone; synchronises with the recorded events
int rt_multi_last_render_ms(rt_multi* M, float* call_ms, float* kernel_ms) {
    if (!M || !M->timed) return RT_EINVAL;
    RT_TRY(hipEventSynchronize(M->ev1));
    float ms = 0.f;
    RT_TRY(hipEventElapsedTime(&ms, M->ev0, M->ev1));
    if (call_ms) *call_ms = ms;
    if (kernel_ms) {
        float k[64]; int n = 0;
        const int rc = rt_render_ctx_times(M->ctx, k, 64, &n);
        if (rc) return rc;
        *kernel_ms = n > 0 ? k[n - 1] : 0.f;
    }
    return 0;
}

// One ncclSend + ncclRecv of `bytes` bytes from this rank to itself inside a group, on the caller's stream: checks that
// the RCCL bound at run time moves device memory in this process (what a one-GPU box can check of the RCCL path).
int rt_multi_selftest(rt_multi* M, const void* d_src, void* d_dst, size_t bytes, void* stream) {
    if (!M || !M->comm || !d_src || !d_dst) return RT_EINVAL;
    Rccl& R = rccl();
    RT_NCCL(R.GroupStart());
    const int a = R.Send(d_src, bytes, 0 /* ncclInt8 */, M->rank, M->comm, (hipStream_t)stream);
    const int b = R.Recv(d_dst, bytes, 0, M->rank, M->comm, (hipStream_t)stream);
    RT_NCCL(R.GroupEnd());
    return (a || b) ? RT_ECOMM : 0;
}

}  // extern "C"
