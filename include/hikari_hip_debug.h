/* hikari_hip_debug.h - test and measurement hooks of libhikari_hip.so.
 *
 * Not part of the drop-in boundary (include/hikari_hip.h is what a Bevy host binds): these entry points exist for the parity
 * tests (tests/), the benchmark's in-run ceilings (bench.py) and the profiling tools (tools/).  Same library, same C ABI rules. */
#ifndef HIKARI_HIP_DEBUG_H
#define HIKARI_HIP_DEBUG_H

#include "hikari_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* (exported next to the boundary's entry points; everything else in the library is hidden) */
#endif

/* Test hook: the instance tree (ordering 0) and the light tree as the device holds them, in the reference's HkNode layout. */
int hk_debug_read_trees(hk_ctx* ctx, HkNode* instance_nodes, uint32_t instance_cap, HkNode* emissive_nodes, uint32_t emissive_cap);

/* Measurement hook ("measure the empirical HBM ceiling with a device copy/triad kernel in the same run"): streams
 * three private arrays of `bytes_per_array` bytes (use >= 1 GiB: the 256 MB Infinity Cache must not hold them) `reps` times
 * on the context's stream and returns the sustained rates in GB/s (1e9), HIP events around the launches: copy a = b moves
 * 2 x bytes per pass, triad a = b + s * c moves 3 x bytes. */
int hk_measure_hbm(hk_ctx* ctx, size_t bytes_per_array, uint32_t reps, double* copy_gbs, double* triad_gbs);
/* Measurement hook for the OTHER roof of the ray kernels: the rate at which the chip issues wave64 VALU instructions, in
 * 1e9 wave-instructions per second, from a register-only kernel of eight independent v_fma_f32 chains per lane (64 x iters
 * instructions per wave) run with 1, 2, 4 and 8 waves resident per SIMD (ginstr_s[0..3]).  One wave alone issues about one
 * instruction per 6 cycles; the ceiling (one per ~2 cycles per SIMD) needs four or more waves per SIMD. */
int hk_measure_valu(hk_ctx* ctx, uint32_t iters, double ginstr_s[4]);

/* Test hook: evaluate one of the library's device math routines elementwise (op: 0 sin, 1 cos,
 * 2 exp, 3 exp2, 4 log2, 5 pow(x, y), 6 min(x,y), 7 max(x,y), 8 f32->f16->f32, 9 x/y, 10 sqrt).
 * x, y, out are HOST arrays. */
int hk_debug_math(hk_ctx* ctx, uint32_t op, const float* x, const float* y, float* out, size_t n);

/* Measurement hook for the roof of the BVH walks of scenes beyond the LDS copy (round 4): every lane of
 * `waves_per_simd` resident waves per SIMD follows its own chain of `steps` DEPENDENT loads through a random permutation cycle over
 * `footprint_bytes` of records - `bytes_per_step` = 16 (one 16-B load per step), 32 (the two adjacent 16-B loads of a node step), 64 (a reservoir record) 128 (a record of the wide walk: eight adjacent 16-B loads, one 128-B line) or 129 (the same 128-B records fetched COOPERATIVELY: eight consecutive lanes load the eight pieces of one lane's record - round 5 experiment) -
 * i.e. 64 unrelated addresses per wave-level load instruction and no reuse.  Returns the rate of wave-level load instructions
 * (1e9 / s) and of loaded bytes (lanes x bytes_per_step per step; GB/s).  No walk of that shape runs faster on the chip: the
 * trace kernels of configs 3 / 4 are priced against it in bench.py (the HBM roof is meaningless for them - they move little). */
int hk_measure_gather(hk_ctx* ctx, size_t footprint_bytes, uint32_t bytes_per_step, uint32_t waves_per_simd, uint32_t steps,
                      uint32_t workgroups /* 0 = CUs x waves_per_simd (the whole chip); else that many 256-thread workgroups */, double* gloads_s, double* gbytes_s);

/* Profiling hook (tools/wf_timeline.py): with HK_DEBUG_OPT_WF_TIMELINE set the trace stages of the queue-based indirect pass
 * run an instrumented twin that records, per stage, when the ray queue ran dry, when the last persistent wave left and how long
 * the rays' walks were (hk_kernels.hpp WfBuffers::timeline: 64 stages x 32 u64, wall_clock64 ticks of 10 ns). */
int hk_debug_read_wf_timeline(hk_ctx* ctx, unsigned long long* out, uint32_t n /* 64 * 32 */);

/* Test hook for the RCCL data path on a box with ONE GPU (RCCL refuses two ranks on one device, so no halo exchange between
 * ranks can run there): rows [row_begin, row_end) of `src_buffer` travel to the same rows of `dst_buffer` (same shape) of the
 * SAME context as an ncclSend to the context's own rank paired with an ncclRecv from it, inside one ncclGroupStart / ncclGroupEnd,
 * ordered against the context's stream - the very function (comm.cpp run_transfers) hk_frame_render's exchanges and hk_comm_gather go
 * through.  Needs hk_comm_init (a communicator of any size; 1 rank on a one-GPU box). */
int hk_debug_comm_loopback(hk_ctx* ctx, uint32_t src_buffer, uint32_t dst_buffer, uint32_t row_begin, uint32_t row_end,
                           uint32_t mode /* 0: in stream order, like a halo exchange (lane 0); 2: the same on lane 1 (exchange B's); 1: overlapped with what follows, like the gather of a finished
                                            frame (hk_frame_render with HK_FRAME_GATHER) - complete before the frame of the same parity begins or
                                            anybody reads a buffer */);

/* Test hook (round 5): how many spatial_reuse launches of this context took the WINDOWED form of the kernel (the depths its taps reach
 * and the lists of surviving taps in LDS: kernels.hip) since hk_create.  Which form a launch takes: by its size, or what
 * hk_debug_set_option(HK_DEBUG_OPT_SPATIAL_WINDOW) says. */
int hk_debug_spatial_windowed_launches(hk_ctx* ctx, uint64_t* out);

/* Test hook (round 6): the priority the context's own main stream was created at - 0 the default, 1 the device's highest, and whether the
 * rule has decided yet (bit 1: it decides at the context's first frame; hk_debug_set_option(HK_DEBUG_OPT_MAIN_PRIORITY) decides at once);
 * bit 2: the context has its fourth stream (the primary rays'), bits 8..27: frames whose primary rays were pipelined so far. */
int hk_debug_main_stream_priority(hk_ctx* ctx, uint32_t* out);

/* Test hook (round 6): 2 when the context's communicator has its second lane - a communicator (ncclCommSplit of the first) and a stream of
 * its own for what a band's POST-PROCESSING and the overlay wait for (exchange B, the gather, exchanges D / E), so that frame n's
 * exchange B and gather do not sit in front of frame n + 1's exchange A on one in-order queue - 1 when the library could not split. */
int hk_debug_comm_lanes(hk_ctx* ctx, uint32_t* lanes);

/* Switches for tests and A/B tools (round 6: the library itself reads NO environment variable).  Waits for the context's work, sets
 * the option, returns; options that change the scene layout take effect with the next frame. */
#define HK_DEBUG_OPT_SPATIAL_WINDOW 0u  /* which form of k_spatial_reuse a launch takes: -1 by its size (default), 0 plain, 1 windowed */
#define HK_DEBUG_OPT_FRAME_PIPELINE 1u  /* 0: the a-trous levels of frame n are NOT run beside frame n + 1's light passes (default 1) */
#define HK_DEBUG_OPT_WF_TIMELINE 2u     /* 1: the instrumented twin of the trace kernels (hk_debug_read_wf_timeline) */
#define HK_DEBUG_OPT_FLAT_WALK 3u       /* 0: no one-level tree for LDS scenes under one transform - they keep the reference's two-level walk (default 1) */
#define HK_DEBUG_OPT_FLAT_ORDERINGS 4u  /* 1..8 direction orderings of that tree (default 0: as many as keep it within 4 KB) */
#define HK_DEBUG_OPT_TRACE_UPDATE 5u    /* 1: timings of scene updates on stderr */
#define HK_DEBUG_OPT_POST_DEMODULATION 6u /* demodulation on the post stream with the a-trous levels: -1 by the library's rule (default), 0 on the main stream, 1 on the post stream */
#define HK_DEBUG_OPT_PERSISTENT_PATHS 8u /* the queue-based indirect pass runs every bounce in ONE launch, a path staying with the wave that claimed it (kernels_wavefront.hip k_wf_trace_wide<.., PATHS>): -1 by the library's rule (default), 0 one trace + one shade launch per bounce, 1 one launch */
#define HK_DEBUG_OPT_MAIN_PRIORITY 9u /* the priority of the context's own main stream, created again at once: -1 by the library's rule (the highest if the context dispatches at most 6 Mi pixels per frame; what the context's first frame decides by itself), 0 the default priority, 1 the highest.  A/B and tests: a stream created again several times ends up sharing a hardware queue */
#define HK_DEBUG_OPT_PREPASS_PIPELINE 10u /* a frame's primary rays on a stream of their own beside the previous frame's spatial pass, where the order allows (context.hip stage TEMPORAL): -1 by the library's rule (default: scenes beyond the LDS copy in frames of up to 3 Mi pixels, on a context whose chain runs at the highest priority), 0 never, 1 whenever the order allows */
#define HK_DEBUG_OPT_SIDE_JOIN 7u /* 1: the main stream waits for the direct-light dispatches (side stream) at the end of every frame, as it did through round 5; 0 (default): only the post-processing does */
int hk_debug_set_option(hk_ctx* ctx, uint32_t option, int64_t value);
/* hk_multi_*: 1 = the calling thread enqueues every band's launches one after another instead of one thread per band (process-wide) */
int hk_debug_multi_serial(int on);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
