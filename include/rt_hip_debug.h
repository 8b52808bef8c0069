/* rt_hip_debug.h -- diagnostic and experiment entry points of libocrt_hip.so.
 *
 * NOT part of the drop-in boundary (include/rt_hip.h is): nothing here replaces a member of the reference's OpenCLHost
 * (include/opencl_host.h:6-144); tests, bench.py and the analysis tools use these to look inside a render host.  None of
 * them changes what a frame computes.
 */
#ifndef RT_HIP_DEBUG_H
#define RT_HIP_DEBUG_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The order in which the ambient-occlusion pass claims a frame's tiles (made once per upload on the host,
 * DeviceRenderer::orderTiles).  rt_debug_measure_tile_costs renders `frames` frames whose AO pass books every claim's
 * duration to its tiles (device-clock ticks of 10 ns), and, with `reorder` != 0, makes the order from them
 * (rt_debug_set_order_policy: `heavy`, `runway`, see DeviceRenderer::orderByMeasuredCost).
 * rt_debug_tile_order copies out: the list (`order`, rt_debug_tile_order_slots() words: eight segments, entry = tile |
 * (hit count - 1) << 26), 24 constants (per group: non-empty tiles, sum of cost classes, hit sub-pixels), the tile words
 * (hit count | cost class << 8) and the measured costs (`tiles` values each, 0 where nothing was measured); any pointer
 * may be null.  rt_debug_set_tile_order installs a list made by the caller (same size, same tiles: any order renders
 * the same image). */
int rt_debug_measure_tile_costs(rt_host *h, uint32_t frames, int reorder);
int rt_debug_set_order_policy(rt_host *h, float heavy, float runway);
uint32_t rt_debug_tile_order_slots(rt_host *h);
uint32_t rt_debug_tiles(rt_host *h);
int rt_debug_tile_order(rt_host *h, uint32_t *order, uint32_t *constants24, uint32_t *tile_words, float *tile_costs);
int rt_debug_set_tile_order(rt_host *h, const uint32_t *order, uint32_t slots, const uint32_t *constants24);

/* Which form a frame with UNIFORM ambient occlusion takes on this host: 0 = the library's rule (two kernels), 1 = the
 * fused frame kernel (primary rays and ambient occlusion in one persistent launch, kernels/frame.hip.h: an experiment
 * that renders the same bits and measured 2-9 % slower, profiles/r05_notes.md), 2 = two kernels.  Same image either way.
 * rt_debug_frame_is_fused: what the next frame will be.  rt_debug_poison_hit_list: overwrites the hit list (the hand-over
 * between the two ray passes) with NaN patterns -- a fused frame that read a record before its own primary work had
 * written it would show; frames of one scene are otherwise identical, and a stale record could never be seen. */
int rt_debug_set_frame_form(rt_host *h, int form);
int rt_debug_frame_is_fused(rt_host *h);
int rt_debug_poison_hit_list(rt_host *h);

#ifdef __cplusplus
}
#endif
#endif
