/* rt_hip_debug.h -- diagnostic and experiment entry points of libocrt_hip.so.
 *
 * NOT part of the drop-in boundary (include/rt_hip.h is): nothing here replaces a member of the reference's OpenCLHost
 * (include/opencl_host.h:6-144); tests, bench.py and the analysis tools use these to look inside a render host.  None of
 * them changes what a frame computes.
 */
#ifndef RT_HIP_DEBUG_H
#define RT_HIP_DEBUG_H

#include "rt_hip_ring.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing of the ray-casting passes on the launch stream beyond rt_last_kernel_ms (rt_hip.h): running totals in
 * ms and the number of frames since the reset; *_kernel_ms covers every pass of a frame, *_ao_ms the launch of the
 * ambient-occlusion kernel alone (0 for frames without AO). */
double rt_total_kernel_ms(const rt_host *h);
float rt_last_ao_ms(const rt_host *h);
double rt_total_ao_ms(const rt_host *h);
uint64_t rt_kernel_launches(const rt_host *h);
void rt_reset_timers(rt_host *h);

/* What the last upload put on the device: the bytes of the scene's arrays, how many copies of them exist among the
 * ring's hosts (ONE: the hosts share the arrays, like the single upload of src/opencl_host.cc:120-136) and the bytes
 * of everything requested, per-host frame buffers included.  Out pointers may be NULL. */
int rt_ring_device_bytes(const rt_ring *r, uint64_t *scene_bytes, uint32_t *scene_copies, uint64_t *total_bytes);

/* What a ring's calibration at upload (rt_hip_ring.h, rt_ring_set_calibration) measured: ms per ao_kernel launch without
 * / with the look-ahead loads (0 = not measured) and the form in use (1 = with); out pointers may be NULL.
 * rt_set_ao_prefetch: which form ONE host launches. */
int rt_ring_calibration(const rt_ring *r, float *ms_without, float *ms_with, int *prefetch_in_use);
int rt_set_ao_prefetch(rt_host *h, int on);

/* Walk intervals (new; results never depend on them): an upload finds, for every 8x8 tile of the host's band, the part of
 * the tree's node records -- an interval of the pre-order array -- outside which no leaf lies that an ambient-occlusion ray
 * of the tile can reach (AO_MAX_DISTANCE, src/intersect_kernel.cl:217, and the slab test of :21-61), once for any ray
 * from the tile and once for each table direction of a full tile (the 64 rays of one packet); the any-hit packets walk
 * their interval alone.  This reports, over the tiles with hits: their number, how many have a tile interval short of the
 * whole array, the mean share of the records inside the tile intervals, and the mean share inside the intervals the
 * packets actually use (1.0 = no narrowing: AO_MAX_DISTANCE of the scene's size).  Out pointers may be NULL.
 * RT_E_STATE without a scene. */
int rt_walk_entries(rt_host *h, uint32_t *tiles_hit, uint32_t *tiles_narrowed, double *mean_share, double *mean_packet_share);

/* Time stamps of collected frames: ms from the last rt_ring_reset_clock to the frame's begin, the start and the end of
 * its ambient-occlusion kernel, its end.  Begin and end are HIP events on the host's stream; the kernel's times are HIP
 * events too for plain launches, and for graph replays -- whose event nodes cannot be timed -- the device's 100 MHz
 * clock as the kernels stamped it into the frame's counters, read only after rt_ring_keep_frame_times(r, 1) (a small
 * blocking copy per frame; 0 without it).  Kept for the last 256 frames (RT_E_STATE for older ones). */
int rt_ring_reset_clock(rt_ring *r);
int rt_ring_keep_frame_times(rt_ring *r, int on);
int rt_ring_frame_times(const rt_ring *r, uint64_t frame, float ms[4]);

/* Sums over the ring's hosts since rt_ring_reset_timers: kernel time of whole frames / of the ao_kernel launches alone
 * (HIP events) and how many frames each sum covers.  Any out pointer may be NULL. */
int rt_ring_timers(rt_ring *r, double *kernel_ms, uint64_t *frames, double *ao_ms, uint64_t *ao_frames);
void rt_ring_reset_timers(rt_ring *r);

/* What the frames collected since the last rt_ring_reset_clock cost the CPU: seconds inside submit (the launches),
 * inside collect waiting for the device, inside collect otherwise; and their number.  Out pointers may be NULL. */
int rt_ring_cpu_times(const rt_ring *r, double *submit_s, double *wait_s, double *collect_s, uint64_t *frames);

/* rt_ring_rccl_info: what the attached communicator says about itself -- ranks (ncclCommCount), RCCL's version code
 * (ncclGetVersion), -1 where unknown -- so that a run can state how many ranks RCCL really saw. */
int rt_ring_rccl_info(rt_ring *r, int *comm_ranks, int *rccl_version);
int rt_ring_rccl_self_test(rt_ring *r);  /* grouped self send/recv on the ring's communicator, checked */

/* The order in which the ambient-occlusion pass claims a frame's tiles (made once per upload on the host,
 * DeviceRenderer::orderTiles).  rt_debug_measure_tile_costs renders `frames` frames whose AO pass books every claim's
 * duration to its tiles (device-clock ticks of 10 ns), and, with `reorder` != 0, makes the order from them
 * (rt_debug_set_order_policy: `heavy`, `runway`, `split_above`, see DeviceRenderer::orderByMeasuredCost).
 * rt_debug_tile_order copies out: the list (`order`, rt_debug_tile_order_slots() words: eight segments, entry = tile |
 * (hit count - 1) << 26), 24 constants (per group: non-empty tiles, sum of cost classes, hit sub-pixels), the tile words
 * (hit count | cost class << 8) and the measured costs (`tiles` values each, 0 where nothing was measured); any pointer
 * may be null.  rt_debug_set_tile_order installs a list made by the caller (same size, same tiles: any order renders
 * the same image). */
int rt_debug_measure_tile_costs(rt_host *h, uint32_t frames, int reorder);
int rt_debug_set_order_policy(rt_host *h, float heavy, float runway, float split_above);  /* split_above < 0: unchanged */
/* The primary pass casts tiles whose packet stops at `above` leaves or more (their cost class, 1 ... 64) in quarters, the four
 * waves of a workgroup at once (DeviceRenderer::setPrimarySplit; 0: none) -- however many tiles that is: the rule an upload
 * applies by itself (class 64) stands back where more than an eighth of the chip's wave slots' worth of tiles qualify.
 * Results never depend on it. */
int rt_debug_set_primary_split(rt_host *h, uint32_t above);
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
