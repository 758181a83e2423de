// rt_tuning.h — every tuning knob of the render kernels in one place, with what was measured when its value was chosen.
//
// A knob changes WHEN and BY WHOM work is done, never a pixel: the parity tests pass at any setting.  The values are the measured
// optima on MI355X for the BASELINE configs (C2/C3: 1200x800x64, N = 500 / 10 000; C4 = C3 in binary16; C5: 3840x2160x256,
// N = 100 000); "ms" is the render kernel of that config unless said otherwise.  Override with -DRT_<NAME>=value
// (tools/mkvariant.sh builds such a library, tools/ab.sh times it against the product on one box).  Switches whose other branch
// lost were removed with their code in round 3 (per-lane walk + cooperative single-ray walk, walk caps, the packed "behind the
// origin" filter of the binary16 tests, the unchunked progressive hand-out, ...): DESIGN.md §8 keeps what they measured.
#pragma once

// ---- occupancy -----------------------------------------------------------------------------------------------------------------
#ifndef RT_RENDER_WAVES
#define RT_RENDER_WAVES 4       // waves per SIMD k_render is compiled for (512 / 4 = 128 VGPRs).  C3: 3: 22.5 ms, 4: 19.5, 5 (96 VGPRs, spills): 22.2; on round 4's kernels (116 VGPRs; 5: 68-80 B of scratch per lane, 5 120 waves resident): C3 14.0 -> 19.7 ms, C5 352 -> 437.  Fewer resident waves cost in proportion (7/8 of the grid: +8 %)
#endif

// ---- scheduling: long pixel chains ------------------------------------------------------------------------------
#ifndef RT_LONG_RATE
#define RT_LONG_RATE 14         // bounces per sample from which a pixel found in flight makes its wave thin.  C3 (with RT_LONG_CHECK 4): 12: 17.6 ms, 13: 17.17, 14: 16.93, 16: 17.2, 20: 17.55
#endif
#ifndef RT_LONG_RATE_DENSE
#define RT_LONG_RATE_DENSE 20   // the same on dense grids (throughput-bound, not chain-bound).  C5: 14: 668 ms, 20: 663
#endif
#ifndef RT_LONG_CHECK
#define RT_LONG_CHECK 4         // a pixel's rate is looked at every so many samples (a power of two).  C3: 2: 17.9 ms, 4: 16.93, 8: 17.5
#endif
// "Long" relative to the launch (k_tile_order): the pilot's bounce counts predict the launch's iterations; iterations / lanes of the
// persistent grid = the LOAD one lane works through.  A pixel is long when its predicted chain (rate x ns) reaches RT_F_INFLIGHT x
// load (found in flight, next to the rate rule above) or RT_F_STATIC x load (3x3 pilot sum, next to RT_PILOT_LONG_SUM: the lower
// threshold counts).  C3's tuned constants are these in disguise (load 663: 14 bounces/sample x 64 = 1.35 x load, pilot sum 200 =
// 1.07 x load), so C3 does not move; an eighth of the C5 frame (load 2 250) needs them where the whole frame (18 000) does not.
// Environment variables of the same names override the values per process (tuning sweeps on one build).  0 = off.
#ifndef RT_F_INFLIGHT
#define RT_F_INFLIGHT 1.35f     // sparse grids (C3, load 663).  C3: 1.35: 16.56-16.69 ms, 1.1: 17.14-17.20, 0.9: 17.53-17.55; C2: 11.18 / 11.24 / 11.30
#endif
#ifndef RT_F_INFLIGHT_DENSE
#define RT_F_INFLIGHT_DENSE 1.1f  // dense grids (k_render<true,*,2>).  An eighth of the C5 frame (load 2 250), slowest of the eight parts, two runs on one box:
#endif                            // 1.35: 93.0 / 93.6 ms, 1.1: 90.9 / 92.3, 0.9: 92.8 / 93.4; the whole frame (load 18 000) does not see it

#ifndef RT_F_STATIC
#define RT_F_STATIC 1.07f
#endif
// The sorted tail's pixels whose 3x3 pilot sum (18 one-sample paths; pure sky = 18) reaches RT_HEAD_SUM_* are handed out FIRST, before the
// tiles, instead of last; 1 = the whole tail (blocks outside the frame count 0), 0 = none (the tail stays at the end of the queue).
// Sparse grids, kernel ms, three runs each: C3 none 13.92-13.96, whole tail first 13.18-13.22 (the tail at 5 / 10 / 20 / 30 % of the work:
// 13.65-13.98 / 13.34-13.58 / 13.47-13.80 / 14.8-15.1); C2 10.36-10.48 -> 9.28-9.46; only the non-sky pixels first (sum >= 19 / 20 / 22):
// C3 13.84-14.25, C2 10.08-10.36.
// A negative value takes the head from the list's cheap END: the pixels whose sum is at most its magnitude (the sky) first, the rest of
// the list, most expensive first, last.  Why first helps: a wave's bounce is as long as its slowest lane's, and sky pixels handed out
// one by one at the end of the queue ride in waves that still walk the grid (13.0 us an iteration); first, they fill waves of their
// own (9.1 us) and the queue is empty 1.3 ms earlier (C3: profiles/r4/timeline_c3_real_pace.txt and timeline_c3_tail_first.txt).
// Dense grids, kernel ms: the whole C5 frame none 352.8-354.1, whole tail first 341.7-344.4, sky first (-18 / -19) 342.8-344.1; halves
// 185.2 -> 179.2; quarters 96.0-97.2 none, 96.5 whole tail, 95.2-95.6 sky (-21 / -19); eighths 54.0-54.2 none, 55.2 whole tail, 54.6-56.6
// sky: a launch of few pixels per lane needs its cheapest pixels for the drain — below RT_HEAD_LOAD_DENSE predicted iterations per
// lane (whole 18 000, half 9 000, quarter 4 500, eighth 2 250) the tail stays where it was.
#ifndef RT_HEAD_SUM_SPARSE
#define RT_HEAD_SUM_SPARSE 1
#endif
#ifndef RT_HEAD_SUM_DENSE
#define RT_HEAD_SUM_DENSE -21
#endif
#ifndef RT_HEAD_LOAD_DENSE
#define RT_HEAD_LOAD_DENSE 3500
#endif
#ifndef RT_F_TAIL
#define RT_F_TAIL 0.15f         // share of a launch's predicted work whose pixels are handed out one by one, most expensive 2x2 pilot block first, at the end of the
#endif                          // queue (k_tail_hist / k_tail_scatter) instead of tile by tile; 0 = off
// Consecutive pixel slots are the same pixel position of RT_INTERLEAVE different tiles (in hand-out order): long pixels cluster, and a tile's
// pixels should not travel together — but rays of neighbouring pixels meet the same spheres, and a wave whose lanes hold 64 different tiles
// finds nothing in its cache.  64: a tile's 64 pixels start in 64 waves; 16: in sixteen, four pixels of the tile each.
#ifndef RT_INTERLEAVE
#define RT_INTERLEAVE 64        // sparse grids (C3: chain-bound).  C3, two runs: 64: 16.79 / 16.63 ms, 32: 16.58 / 16.62, 16: 16.48 / 16.59, 8: 16.75 / 16.69, 4: 16.47 / 16.68, 1: 17.01 / 17.20
#endif
#ifndef RT_INTERLEAVE_DENSE
#define RT_INTERLEAVE_DENSE 16  // dense grids (C5).  The eight parts of the frame, slowest / sum (ms), two runs each: 64: 92.4 / 722, 91.9 / 723; 32: 92.0 / 718, 91.0 / 714;
#endif                          // 16: 90.0 / 708, 90.6 / 709; 8: 89.8 / 700, 90.4 / 705; 4: 94.8 / 721; 1: 108.7 / 772.  Whole frame: 638.5 -> 637.0 (16), 636.3 (8)
#ifndef RT_INTERLEAVE_SOLO
#define RT_INTERLEAVE_SOLO 1    // very sparse grids with solo chains (C2: k_render<true,*,5>).  C2: 64: 11.09 ms, 16: 11.00, 4: 10.96, 1: 10.69
#endif
#ifndef RT_LONG_RATE_MIN
#define RT_LONG_RATE_MIN 8      // ... but never below this many bounces per sample / this 3x3 pilot sum (launches of a pixel or two per lane)
#endif
#ifndef RT_PILOT_LONG_SUM_MIN
#define RT_PILOT_LONG_SUM_MIN 120
#endif
#ifndef RT_MED_RATE
#define RT_MED_RATE 12          // "medium" chains (below RT_LONG_RATE): the wave keeps refilling but issues at priority 1; 0 = off.  C3 (round 1): off 22.88 ms, 10: 22.62, 12: 22.51, 15: 22.53
#endif
#ifndef RT_THIN_CAP_DEN
#define RT_THIN_CAP_DEN 4       // at most 1/4 of the resident waves may go thin for chains found in flight (1/2, 1/8: within +-1 %)
#endif
#ifndef RT_LONG_PER_WAVE
#define RT_LONG_PER_WAVE 16     // pre-classified chains a thin wave starts with.  C3: 2: 20.32 ms, 4: 20.04, 8: 19.84, 16: 19.81, 32: 21.92
#endif
// pilot pass (k_tile_cost): RT_PILOT_SAMPLES samples per 2x2 pixel block on a private RNG stream, cut at RT_PILOT_CAP bounces; a block's
// pixels start as long chains when the pilot bounces of the block and its eight neighbours total >= RT_PILOT_LONG_SUM.  Scored against
// the chains' true lengths on C3 (profiles/r2/predictor_c3.txt): the block's own count >= 50 finds 22 % of the pixels above 1280
// iterations; the 3x3 sum >= 200 finds 86 % (13 of the 13 above 2000) and 1 % of what it picks is shorter than 400; >= 160 overruns the 1/64 cap.
#ifndef RT_PILOT_SAMPLES
#define RT_PILOT_SAMPLES 2
#endif
#ifndef RT_PILOT_LONG_SUM
#define RT_PILOT_LONG_SUM 200
#endif
#ifndef RT_SOLO_DENSITY
#define RT_SOLO_DENSITY 1.0     // grids with at most this many entries per cell start their chains solo (DevAccel::solo_chains, k_render<true,*,5>).  C2 (0.2 per cell): 13.4 -> 11.6 ms; C3 (3.6): 17.25 -> 17.4-17.5
#endif
#ifndef RT_PILOT_SOLO_SUM
#define RT_PILOT_SOLO_SUM 200   // 3x3 pilot sum from which a chain starts ALONE in a wave (k_render<true,*,5>).  C2: none 13.41 ms, 300: 11.80, 250: 11.58, 200: 11.51, 150 (with RT_PILOT_LONG_SUM 150): 12.18
#endif
#ifndef RT_PILOT_CAP
#define RT_PILOT_CAP 35         // (the pass is as long as its longest sample; 25: C4 +2 ms, 50 = the reference's depth limit)
#endif
#ifndef RT_PROG_OWN
#define RT_PROG_OWN 64          // render_progressive: pixel slots a wave owns before it takes chunks of 64 from the counter.  C3, ms per pass: 0: 0.645, 64: 0.567, 128: 0.611 (one request per lane and pixel: 1.05)
#endif

// ---- the balanced split of a frame over GPUs (rt_split_balanced) ---------------------------------------------------------------
// A tile's predicted cost = RT_SPLIT_WB x (bounces of its 32 pilot samples) + RT_SPLIT_WT x (grid entries their walks pooled) + RT_SPLIT_WC x (grid columns they stepped through).
#ifndef RT_SPLIT_WB
#define RT_SPLIT_WB 1000
#endif
#ifndef RT_SPLIT_WT
#define RT_SPLIT_WT 17
#endif
#ifndef RT_SPLIT_WC
#define RT_SPLIT_WC 0
#endif

// ---- the candidate grid (rt_accel.h; host build, device build and the walks read these) ----------------------------------------------
// (measured and not kept, round 4: a sphere filed ONCE, under the column of its centre, the walk reaching R' beyond its first and last
// column and following the line R' beyond a column's edges — no sphere twice per ray, half the entries: C5 393 ms against 354, C3 14.60
// against 14.25: the extra column at either end of every walk costs more than the duplicates)
#ifndef RT_ACCEL_FINE
#define RT_ACCEL_FINE 8         // fine bins per cell along a column (a power of two).  C5 (before / after the tight column range): 2: 486 ms, 4: 462 / 359, 8: 450 / 352, 16: - / 349; C3 2: 15.23, 4: 15.01 / 14.16, 8: 14.88 / 14.05, 16: - / 14.02
#endif
#ifndef RT_DENSE_CELL
#define RT_DENSE_CELL 0.7       // column width of dense scenes, in units of 2 R'.  C5 (before / after the tight column range): 0.5: 497 / 354 ms, 0.7: 462 / 352, 0.85: - / 359, 1.0: 467 / 368, 1.4: 526
#endif
#ifndef RT_SPARSE_CELL
#define RT_SPARSE_CELL 1.0      // ... of sparse scenes.  C3 (before / after the tight column range): 0.7: 15.17 / 14.17 ms, 1.0: 15.01 / 14.05, 1.4: 14.80 / 14.12, 2.0: 15.25; C2: 0.7: 10.96, 1.0: 10.52, 1.4: 10.24
#endif

// ---- the pooled walks -----------------------------------------------------------------------------------
#ifndef RT_QUORUM_SPARSE
#define RT_QUORUM_SPARSE 4      // walk_pool returns when 1/4 of the walkers that entered are left.  C3: off 21.06 ms, 2: 20.14, 3: 19.91, 4: 19.94, 8: 20.41
#endif
#ifndef RT_QUORUM_DENSE
#define RT_QUORUM_DENSE 2       // walk_pool_dense: 1/2 (dense walks are long and uneven).  C5 geometry, round 1: off 134.8 ms, 1/8: 122.1, 1/4: 118.5, 1/2: 114.3
#endif
#ifndef RT_QUORUM_MIN
#define RT_QUORUM_MIN 16        // ... in waves that entered with at least this many walkers
#endif
#ifndef RT_POOL_COLS
#define RT_POOL_COLS 2          // grid columns per ray and round of walk_pool.  C3: 1: 20.95 ms, 2: 20.4, 4: 22.2
#endif
#ifndef RT_POOL_COLS_THIN
#define RT_POOL_COLS_THIN 4     // ... in a wave with few walkers (a thin wave's chains: fewer rounds, fewer dependent round trips).  C2: 14.5 -> 14.1 ms
#endif
#ifndef RT_POOL_SPANS
#define RT_POOL_SPANS 1         // spans a lane steps through side by side in walk_pool (2, 4: within the +-1.5 % run-to-run noise)
#endif
#ifndef RT_POOL_DIRECT
#define RT_POOL_DIRECT 1        // walk_pool: a round of at most 64 entries (one per lane) fetches each entry with its brick and resolves candidates where they are found
#endif                          // (no queue; a lone bounce's walk has one dependent cache round trip less).  Kernel medians, interleaved runs: C2 0: 10.69 ms, 1: 10.38; C3 14.13 / 13.97; C5 flat
#ifndef RT_DENSE_PB
#define RT_DENSE_PB 4           // entries per lane and pass of walk_pool_dense.  C5: 2: 850 ms, 3: 744, 4: 670, 5: 730, 8: 839
#endif
#ifndef RT_DENSE_DRAIN
#define RT_DENSE_DRAIN 32       // candidates queued before walk_pool_dense resolves them (a found hit starts rejecting sooner; 64: +1 %)
                                // (round 4: one dword of the cache line behind a pass's entries requested with the pass — whole C5 809 ms with the next line, 723 with the next 64 bytes, against 643)
#endif

// ---- hitable_list scan (the reference traversal of lists) -------------------------------------------------------------------------
#ifndef RT_LIST_BATCH
#define RT_LIST_BATCH 8         // spheres per pass of the scalar-load scan (their loads in flight together).  C2 scan: 1: 46.1 ms, 8: 40.2
#endif
#ifndef RT_LIST_COOP_COST
#define RT_LIST_COOP_COST 16    // rays are scanned one at a time with lanes = spheres while live rays x 16 <= list size.  C2 scan: 40.1 -> 36.5 ms
#endif

// ---- USE_FP16 (rt_kernels_fp16.hip) -----------------------------------------------------------------------------
#ifndef RT_H16_MINWAVES
#define RT_H16_MINWAVES 4       // waves per SIMD k_render_h is compiled for (5, 6 with spills: no change; 3: 40.6 ms against 37.1, 40.1 with 12 pairs per pass)
#endif
#ifndef RT_H16_BIG
#define RT_H16_BIG 128          // big segments (bucket ranges of >= 8 pairs) a wave pools per round (a multiple of 64, at most 128)
#endif
#ifndef RT_H16_SMALL
#define RT_H16_SMALL 256        // small segments per round (C4: the 56 upper cells that hold one large sphere each)
#endif
#ifndef RT_H16_INTERLEAVE
#define RT_H16_INTERLEAVE 64    // tiles over which consecutive pixel slots interleave (as RT_INTERLEAVE).  C4: 64: 30.8 / 30.9 ms, 16: 31.9 / 32.2, 4: 35.7 / 34.9, 1: 39.3 / 38.4
#endif
#ifndef RT_H16_TASKS
#define RT_H16_TASKS 384        // level-2 node expansions a wave pools per round of the binary16 walk (1.5 KB of LDS; four waves per SIMD leave room for ~450)
#endif
#ifndef RT_H16_PP
#define RT_H16_PP 10            // pairs per lane and pass of the big segments' test loop (at most 16: pass_mask).  C4, round 3: 4: 38.3 ms, 6: 37.9, 8: 37.05; with the ray out of the registers during the tests (closest_tree, phase 3): 8: 35.6, 10: 35.1, 12: 35.4
#endif
#ifndef RT_H16_LONG_PER_WAVE
#define RT_H16_LONG_PER_WAVE 16 // pre-classified chains per thin wave.  C4 (round 2): 4: 56.7 ms, 8: 57.3, 16: 57.1, 32: 61.5
#endif
#ifndef RT_H16_LONG_RATE
#define RT_H16_LONG_RATE 14     // bounces per sample from which a pixel found in flight makes its wave thin; 0 = off.  C4 (round 2): off 66.3 ms, 10: 57.7, 12: 57.2, 14: 57.1, 16: 57.4
#endif
#ifndef RT_H16_LONG_CHECK
#define RT_H16_LONG_CHECK 4
#endif
#ifndef RT_H16_THIN_CAP_DEN
#define RT_H16_THIN_CAP_DEN 4
#endif
#ifndef RT_H16_PILOT_LONG_SUM
#define RT_H16_PILOT_LONG_SUM 150   // 3x3 pilot sum from which a block's pixels start as long chains (fp32: 200).  C4 (round 2): 100: 56.1 ms, 120: 55.6, 140: 55.1, 160: 55.4, 200: 57.0, 250: 59.7
#endif
#ifndef RT_H16_PILOT_CAP
#define RT_H16_PILOT_CAP 35     // bounces after which a pilot sample is cut (25: +2 ms)
#endif
