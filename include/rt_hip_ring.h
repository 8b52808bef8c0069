/*
 * rt_hip_ring.h -- beyond the reference's seam: a steady stream of frames and several GPUs (libocrt_hip.so).
 *
 * The reference renders ONE blocking frame on ONE device per process (src/render.cc:109-111, src/opencl_host.cc:16-32).
 * What a caller needs for more than that is declared here, on top of include/rt_hip.h: the band partition of an image
 * over ranks, a host's stream, the frame ring (several render hosts of one scene taking frames in turn on one GPU) and
 * the exchange step of a multi-GPU frame over RCCL.  Conventions as in rt_hip.h.
 */
#ifndef RT_HIP_RING_H
#define RT_HIP_RING_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same partition arithmetic without a device (for gather/assembly code):
 * output rows rank `rank` of `nranks` owns, and the global row of a local one. */
uint32_t rt_partition_local_rows(const rt_options *o, uint32_t rank, uint32_t nranks);
uint32_t rt_partition_global_row(const rt_options *o, uint32_t rank, uint32_t nranks, uint32_t local_row);

/* Enqueue the device resize into caller-owned DEVICE memory of
 * rt_local_rows*width bytes (e.g. the send buffer of an RCCL gather). */
int rt_resize_into_device(rt_host *h, void *device_u8);

/* Run this host's work on a caller-owned hipStream_t (NULL = the HIP default
 * stream) / return to the host's private non-blocking stream. */
int rt_set_stream(rt_host *h, void *hip_stream);
int rt_use_private_stream(rt_host *h);

/* The hipStream_t this host's work is enqueued on right now (its private stream unless rt_set_stream replaced it),
 * for callers that order their own streams against it with events -- e.g. two hosts of the same scene on one GPU
 * taking frames alternately, so that one frame's last workgroups and the next frame's first share the device
 * (bench.py).  No counterpart in the reference: its queue is private to OpenCLHost (include/opencl_host.h:129). */
int rt_get_stream(rt_host *h, void **hip_stream);

/* Tell a host that `hosts` of them (it included) take frames in turn on its GPU: its persistent ambient-occlusion pass
 * then leaves part of the chip to the other frames' passes instead of filling it (1, the default: alone). */
int rt_set_device_share(rt_host *h, unsigned int hosts);

/* ---- frame ring: a steady stream of frames behind the same seam ------------------------------------------------
 * The reference renders one blocking frame per OpenCLHost::operator()() (src/opencl_host.cc:137-149, called once by
 * src/render.cc:109-111).  A ring is `hosts` render hosts of ONE scene on ONE GPU -- each with its own stream
 * (consecutive hosts in different priority classes, hence different hardware queues) and its own captured hipGraph,
 * so that a frame costs the CPU one graph launch -- that take frames in turn: the next frames' passes fill the wave
 * slots a finishing ambient-occlusion pass frees.  A ring of ONE host is the reference's blocking frame.  With a
 * communicator attached (rt_ring_attach_rccl) the ring also runs the one exchange step of a multi-GPU frame, the
 * gather of the ranks' 8-bit bands on rank 0 and the assembly of the image there, behind the next frames.
 * Every frame is complete: all ray passes + the device resize (RayTracer::resize, src/ray_tracer.cc:3-16). */
rt_ring *rt_ring_create(const rt_options *o, int device, uint32_t rank, uint32_t nranks, uint32_t hosts);
void rt_ring_destroy(rt_ring *r);

/* OpenCLHost::upload (src/opencl_host.cc:120-136) for every host of the ring; same arguments as rt_upload. */
int rt_ring_upload(rt_ring *r, const uint32_t *faces, uint32_t num_faces, const uint32_t *nodes, uint32_t num_nodes,
                   const float *aabbs, const float *vertices, uint32_t num_vertices, const float *vnormals);
int rt_ring_upload_scene(rt_ring *r, const rt_scene *s);

/* Run-time calibration (new; results never depend on it): rt_ring_upload* measure, on the uploaded scene, what its
 * tiles' ambient-occlusion packets cost (the order the pass claims them in is made from that) and which form of the
 * pass's node loop -- with or without look-ahead loads -- is faster (a few frames, ~20 ms), for every host of the ring;
 * rt_ring_set_calibration(r, 0) before the upload switches that off.  What was measured: rt_ring_calibration
 * (rt_hip_debug.h). */
int rt_ring_set_calibration(rt_ring *r, int on);

uint32_t rt_ring_size(const rt_ring *r);        /* hosts */

/* Band buffers: frame f is rendered by host f % size into buffer f % slots, slots = 2 * size, so that a frame's bands
 * (and, with a communicator, its assembled image) stay untouched while the next `size` frames are submitted. */
uint32_t rt_ring_slots(const rt_ring *r);
uint32_t rt_ring_local_rows(const rt_ring *r);  /* output rows this rank owns (rt_local_rows) */
uint32_t rt_ring_in_flight(const rt_ring *r);   /* frames submitted and not yet collected */

/* Host `slot` of the ring as a BORROWED rt_host (statistics, timers, rt_download of its last frame): owned by the
 * ring, rt_destroy on it is a no-op. */
rt_host *rt_ring_host(rt_ring *r, uint32_t slot);

/* 1 (default): a frame is one replay of the host's captured hipGraph; 0: the same launches one by one. */
int rt_ring_set_graph_mode(rt_ring *r, int on);

/* Pacing of the submissions: a frame is enqueued no sooner than beta x (the running mean of the time per finished frame)
 * after the previous one, so that frames which finished together do not start their successors together and keep the
 * ring in lockstep.  Default 0.5; 0 switches it off.  (The reference submits one frame and waits for it:
 * src/opencl_host.cc:137-149; nothing to pace there.) */
int rt_ring_set_pacing(rt_ring *r, float beta);

/* Band buffer `slot` (< rt_ring_slots) is caller-owned DEVICE memory from now on (rt_ring_local_rows * width bytes;
 * NULL: the ring's own again), e.g. the send buffer of a caller-side collective. */
int rt_ring_bind_output(rt_ring *r, uint32_t slot, void *device_u8);

/* The non-blocking half of OpenCLHost::operator()(): enqueue the next frame on the next host; *frame (may be NULL)
 * receives its number.  RT_E_STATE when every host already has a frame in flight. */
int rt_ring_submit(rt_ring *r, uint64_t *frame);

/* The blocking half: wait for the OLDEST frame in flight.  Out (each may be NULL): its number, its band buffer's slot,
 * the device address of its bands (valid until frame + slots is submitted).  With a communicator attached its gather
 * is enqueued. */
int rt_ring_collect(rt_ring *r, uint64_t *frame, uint32_t *slot, const void **device_bands);

/* rt_ring_collect + a device-to-device copy of the bands into caller memory (complete on return). */
int rt_ring_collect_into_device(rt_ring *r, void *device_u8);

/* One frame of a steady stream: submit, then collect until at most hosts - 1 frames are in flight (one host: none).
 * rt_ring_run does `frames` such steps in one call; rt_ring_drain collects what is left and waits for the gathers. */
int rt_ring_step(rt_ring *r);
int rt_ring_run(rt_ring *r, uint32_t frames);
int rt_ring_drain(rt_ring *r);

/* The last collected frame on the device: its bands (= the image for nranks 1) without a communicator; with one, the
 * assembled width x height image on rank 0 (waits for its gather) and NULL on the other ranks.  rt_ring_download_last
 * copies width*height bytes to the host (OpenCLHost::download + RayTracer::resize, src/render.cc:114-123). */
int rt_ring_last_image_device(rt_ring *r, const void **device_u8);
int rt_ring_download_last(rt_ring *r, uint8_t *image);

/* The exchange step over RCCL, one process per GPU.  New: the reference is single-device (src/opencl_host.cc:16-32).
 * rt_rccl_unique_id (rank 0) fills 128 bytes that the caller hands to every rank (e.g. torch.distributed broadcast);
 * rt_ring_attach_rccl is collective (ncclCommInitRank with the ring's rank / nranks).  rt_rccl_available: 1 when
 * librccl.so.1 can be opened (it is opened at run time; the library loads without it).
 * A ring's exchange step waits at most `seconds` (default 30) for a frame's gather: a peer rank that died or posted a
 * different number of frames makes rt_ring_run / rt_ring_drain / rt_ring_submit fail with RT_E_DEVICE instead of
 * hanging (rt_ring_set_gather_timeout). */
int rt_ring_set_gather_timeout(rt_ring *r, double seconds);
int rt_rccl_available(void);
int rt_rccl_unique_id(void *out128);
int rt_ring_attach_rccl(rt_ring *r, const void *unique_id128);

#ifdef __cplusplus
}
#endif
#endif
