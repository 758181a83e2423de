// rt_stats.h — the instrumentation of the DIAGNOSTIC builds of the render kernels (librt_amd_stats.so: -DRT_STATS, read out by
// tools/stats.py; librt_amd_wpass.so: -DRT_STATS -DRT_STATS_WPASS).  Included by rt_kernels.hip inside namespace rt.  A product
// build (no RT_STATS) gets empty macros from here and nothing else: STAT / WPASS count, RT_STATS_ONLY(code) keeps `code` in the
// diagnostic builds only, STAT_ARG / STAT_PASS thread the per-lane counter block through the closest-hit functions.
#pragma once
#ifdef RT_STATS
enum { ST_RAYS, ST_FAST, ST_SLOW, ST_TIE, ST_COLS, ST_TESTS, ST_DISCPOS, ST_OFFERS, ST_ELIG, ST_ELIG_NODES, ST_A_ITERS_WAVE, ST_B_ROUNDS_WAVE,
       ST_LOOP_ITERS_WAVE, ST_A_LANE_STEPS, ST_B_LANES, ST_SAMPLES, ST_LIVE_GE56, ST_LIVE_32, ST_LIVE_8, ST_LIVE_LT8, ST_SWITCHES,
       ST_CYC_TOTAL, ST_CYC_CLOSEST, ST_CYC_WALK_A, ST_CYC_WALK_B, ST_CYC_SCAN, ST_CYC_SHADE, ST_REALTIME,
       ST_SPARE0, ST_SPARE1, ST_SPARE2, ST_SPARE3, ST_SPARE4, ST_SPARE5, ST_SPARE6,
       // wave passes: how often a wave (any lane) executed a block — multiplied by the block's static size = issue slots
       WP_GROUND, WP_LARGE_K, WP_LARGE_EXACT, WP_OFFER_NODE, WP_OFFER_RAYBOX, WP_ELIG_FN, WP_ELIG_LIST, WP_SETUP, WP_A_COL, WP_A_BATCH, WP_A_HOLD,
       WP_B_OFFER, WP_B_CLIP, WP_COOP_CHUNK, WP_SCAN, WP_SC_ANY, WP_SC_LAMB, WP_SC_METAL, WP_SC_DIEL, WP_REJ_ITER, WP_PRIMARY, WP_DISK_ITER, WP_SKY, WP_ENDPIX,
       // cycles of the iterations of thin waves with <= 2 live lanes, by part (the critical path of the frame's tail)
       TH_GROUND, TH_LARGE_SETUP, TH_WALK, TH_SCAN, ST_N };
#ifdef RT_STATS_LIGHT      // per-pixel and per-wave time stamps only (librt_amd_stats_light.so): the counters and cycle probes compile to nothing,
#define TICK() 0ull        // the kernel runs within a few per cent of the product's — the timeline tools/stats.py prints is then the real one
#else
#define TICK() ((unsigned long long)__builtin_amdgcn_s_memtime())
#endif
__device__ unsigned long long g_stats[ST_N];
__device__ int g_pilot_dbg[1 << 20];                  // per 2x2 block (tile * 16 + block): the pilot's bounce count
__device__ unsigned long long g_wave_dbg[8192 * 4];   // per wave: end time (100 MHz ticks since launch), loop iters, thin iters, long pixels
struct Stats { unsigned int c[ST_N]; unsigned long long cyc[8]; };
#ifdef RT_STATS_LIGHT
#define STAT(st, k, v) ((void)0)
#else
#define STAT(st, k, v) ((st).c[k] += (v))
#endif
#ifdef RT_STATS_WPASS      // (the atomics distort every timing of the same run: a build of its own, librt_amd_wpass.so)
#define WPASS(k) do { const int l_ = (int)(threadIdx.x & 63); if (__builtin_amdgcn_readfirstlane(l_) == l_) atomicAdd(&g_stats[k], 1ull); } while (0)
#else
#define WPASS(k) ((void)0)
#endif
#define STAT_ARG , Stats& st
#define STAT_PASS , st
#define RT_STATS_ONLY(...) __VA_ARGS__
#define RT_STATS_READERS \
hipError_t read_pilot_dbg(int* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pilot_dbg), sizeof(int) * (size_t)(n < (1 << 20) ? n : (1 << 20))); } \
hipError_t read_wave_dbg(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_dbg), sizeof(unsigned long long) * 8192 * 4); } \
hipError_t read_stats(unsigned long long* out, int reset) { \
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * ST_N); \
    if (e != hipSuccess) return e; \
    if (reset) { unsigned long long z[ST_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); } \
    return e; \
}
#else
#define STAT(st, k, v) ((void)0)
#define WPASS(k) ((void)0)
#define STAT_ARG
#define STAT_PASS
#define RT_STATS_ONLY(...)
#define RT_STATS_READERS
#endif
