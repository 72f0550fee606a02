/* fdgs.h -- C ABI of libfdgs.so, the MI355X-native differentiable 4D Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's native extension
 * `diff_gaussian_rasterization._C` (diff-gaussian-rasterization/ext.cpp:15-19):
 *
 *   fdgs_rasterize_forward   replaces  rasterize_gaussians           (rasterize_points.h:18-49,
 *                                                                     RasterizeGaussiansCUDA rasterize_points.cu:36-149)
 *   fdgs_rasterize_backward  replaces  rasterize_gaussians_backward  (rasterize_points.h:51-89,
 *                                                                     RasterizeGaussiansBackwardCUDA rasterize_points.cu:151-270)
 *   fdgs_mark_visible        replaces  mark_visible                  (rasterize_points.h:91-94, rasterize_points.cu:272-291)
 *
 * Plain C: raw DEVICE pointers, sizes, scalars, an explicit HIP stream and int
 * error codes.  No torch / pybind types.  The host binding (ctypes in
 * 4d-gaussian-splatting_amd/_capi.py; any other FFI would look the same, see
 * INTEGRATION.md) owns every tensor, exactly like the reference where torch owns all
 * memory and the rasterizer only borrows pointers.
 *
 * Conventions carried over from the reference:
 *  - an absent optional tensor is a NULL pointer (the reference passes empty
 *    tensors, i.e. data_ptr()==nullptr, gaussian_renderer/diff_gaussian_rasterization.py:282-300);
 *    precedence: cov3D_precomp > rot_4d (4D conditional) > 3D (+1-D temporal marginal);
 *  - viewmatrix / projmatrix are the transposed ("row-vector") matrices of
 *    scene/cameras.py:65-70, read flat as column-major (auxiliary.h:59-78);
 *  - quaternions are (w,x,y,z) and already normalised; scales are post-exp;
 *    opacities post-sigmoid; shs is [P, M, 3];
 *  - the three scratch buffers (geometry / binning / image) are opaque byte
 *    buffers obtained through a resize callback (the reference's
 *    resizeFunctional, rasterize_points.cu:28-34) and must be handed back
 *    unchanged to fdgs_rasterize_backward;
 *  - gradients are bug-compatible with the reference backward (SURVEY.md
 *    Appendix A, Q1-Q13).
 *
 * Threading contract.  All work is enqueued on the caller's `stream`; fdgs_rasterize_forward waits for the device once
 * (for num_rendered, as the reference does at rasterizer_impl.cu:302; not at all with fdgs_forward_out.lazy): it spins on a pinned
 * mailbox the tile-scan kernel writes.  The library keeps, PER HOST THREAD AND DEVICE: the last-error string, that pinned mailbox
 * (a ring of 64 slots: at most 64 lazy forwards of a thread are unreported at a time; the 65th waits for the oldest), a second stream with
 * two events (fdgs_forward_out.split_colour) and the run-ahead guesses (sizes of the thread's previous forward calls per
 * (device, W, H, P)).  Consequence: any number of host threads may call concurrently (each with its own stream), and one
 * thread may drive several devices (hipSetDevice before the call); within ONE thread a forward call has returned before the
 * next one starts, so there is never more than one forward of a thread in flight on the host side.  Process-wide state:
 * the profiling accumulators (fdgs_profile_*, mutex-guarded), the test hooks fdgs_debug_tile_sort_limits /
 * fdgs_set_run_ahead, and the counters of fdgs_debug_run_ahead_stats.
 * Limits: P < 2^26 Gaussians per call (the blend kernels address the 48- / 64-byte per-Gaussian records with 32-bit byte
 * offsets; FDGS_ERR_HIP "invalid value" beyond), num_rendered < 2^31, image sides below 16 * 65535 pixels.
 */
#ifndef FDGS_H
#define FDGS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDGS_VERSION 502 /* 0.5.2 (round 6: fdgs_set_sparse_lists_budget; round 5: fdgs_forward_out.sparse_lists, .colour_stream).  Every struct below starts with `struct_size` = sizeof(the struct) as THIS header defines it
                            (the library answers FDGS_ERR_INVALID_ARG to any other value), and fdgs_version() must equal
                            FDGS_VERSION: a binding built against another revision of this header is turned away instead of
                            having the library read past the end of a shorter struct. */

/* error codes */
#define FDGS_OK 0
#define FDGS_ERR_INVALID_ARG 1   /* bad sizes / missing required pointers          */
#define FDGS_ERR_HIP 2           /* a HIP runtime call or kernel launch failed     */
#define FDGS_ERR_ALLOC 3         /* the scratch allocator callback returned NULL   */
#define FDGS_ERR_UNSUPPORTED 4   /* e.g. no gfx950 device                          */

/* which scratch buffer the allocator is asked for */
#define FDGS_BUF_GEOMETRY 0
#define FDGS_BUF_BINNING 1
#define FDGS_BUF_IMAGE 2

/* Resize callback: must return a DEVICE pointer to at least `bytes` bytes,
 * 256-byte aligned, valid until the matching backward has finished.
 * Mirrors std::function<char*(size_t)> in CudaRasterizer::Rasterizer::forward
 * (rasterizer.h:28-31).  Like the reference's resize lambdas it may be called more than once for the same buffer
 * within one forward call (FDGS_BUF_BINNING: the forward sizes it from the previous call before num_rendered is known and
 * asks again if that was too small); the latest answer is the buffer, an earlier one may be released in stream order
 * (work already enqueued on `stream` may still touch it). */
typedef void* (*fdgs_alloc_fn)(void* user, int which, size_t bytes);

/* Everything that describes one view; shared by forward and backward
 * (GaussianRasterizationSettings, gaussian_renderer/diff_gaussian_rasterization.py:227-245,
 * plus the per-call tensors). */
typedef struct fdgs_scene
{
	uint32_t struct_size; /* sizeof(fdgs_scene)                                    */
	int32_t P;            /* number of Gaussians                                   */
	int32_t D, D_t, M;    /* active SH degree, active time degree, SH coeffs/pt    */
	int32_t W, H;         /* image_width, image_height                             */
	const float* bg;              /* [3]                                           */
	const float* means3D;         /* [P,3]                                         */
	const float* shs;             /* [P,M,3] or NULL                               */
	const float* colors_precomp;  /* [P,3]  or NULL (exactly one of shs/colors)    */
	const float* flows;           /* [P,2]  or NULL (NULL == all zero)             */
	const float* opacities;       /* [P]                                           */
	const float* ts;              /* [P]    or NULL                                */
	const float* scales;          /* [P,3]  or NULL                                */
	const float* scales_t;        /* [P]    or NULL                                */
	const float* rotations;       /* [P,4]  or NULL                                */
	const float* rotations_r;     /* [P,4]  or NULL                                */
	const float* cov3D_precomp;   /* [P,6]  or NULL                                */
	const float* viewmatrix;      /* [16]                                          */
	const float* projmatrix;      /* [16]                                          */
	const float* campos;          /* [3]                                           */
	float scale_modifier;
	float prefilter_var;
	float tan_fovx, tan_fovy;
	float timestamp, time_duration;
	int32_t rot_4d, gaussian_dim, force_sh_3d;
	int32_t prefiltered;
	int32_t debug;        /* != 0: synchronise + check after every stage (CHECK_CUDA, auxiliary.h:165-172) */
	int32_t raw_params;   /* 0 (reference semantics): scales / scales_t / opacities / rotations / rotations_r are
	                         post-activation, as the reference passes them.
	                         1 (fused activations): they are the model's RAW parameters and
	                         the kernels apply the reference's activations themselves (scene/gaussian_model.py:179-219:
	                         exp, exp, sigmoid, x / max(|x|, 1e-12) twice); backward then returns gradients w.r.t. the
	                         raw parameters. */
	int32_t analytic_sh_grad; /* 0 (default): the 4D-SH backward reproduces the reference's three deviations from the analytic
	                         gradient (backward.cu:190, 303 / 384, 403): Q1 dL_dsh[1] uses the l = 0
	                         basis value, Q2 the derivative of cos(2 pi k dt / T) has the wrong sign, Q3 the k = 2 time term
	                         overwrites the k = 1 term in dRGB/dt.  1 (opt-in): the analytic gradient of the forward pass.
	                         Forward results do not depend on it.  (3D SH is analytic in the reference already.) */
} fdgs_scene;

/* Forward outputs; every array is fully written by the call (no pre-zeroing needed). */
typedef struct fdgs_forward_out
{
	uint32_t struct_size; /* sizeof(fdgs_forward_out)                              */
	float* out_color;     /* [3,H,W]                                               */
	float* out_flow;      /* [2,H,W]                                               */
	float* out_depth;     /* [1,H,W]                                               */
	float* out_T;         /* [1,H,W]  final transmittance (Python returns 1 - T)   */
	int32_t* radii;       /* [P]                                                   */
	float* out_means3D;   /* [P,3]    means3D, shifted by the conditional mean where rot_4d */
	float* covs_com;      /* [P,6] or NULL: owning copy of the computed 3D covariances
	                         (zero for culled Gaussians; the reference returns uninitialised memory there) */
	int32_t preprocessed; /* 0: the call runs the whole forward.  1: the per-Gaussian preprocess of this view was enqueued on the same
	                         stream by fdgs_preprocess_batch (which obtained the view's geometry and image buffers from the
	                         allocator): the call asks the allocator for the same two buffers again -- the allocator must answer
	                         with the SAME pointers -- and continues with the tile binning and the blend */
	int32_t split_colour; /* 0: one preprocess launch.  1: geometry first, the SH -> RGB evaluation (the bulk of the preprocess'
	                         memory traffic, which only the blend needs) on an internal second stream next to the tile binning
	                         (events in and out): shortens the forward's critical path by ~30 us at C3 for forward-only
	                         rendering; same arithmetic, bit-identical outputs */
	int32_t tile_cull;    /* 0: the tile lists are the reference's -- every tile of the square of 3 sigma_max around the projected
	                         mean (auxiliary.h:46-57), bit-identical point_list / ranges / n_contrib.  1: a Gaussian is only listed
	                         in the tiles of the axis-aligned bounding box of the region where it can reach alpha >= 1/255 (the
	                         forward blend's own per-pixel test, forward.cu:590), intersected with the reference's square; a
	                         Gaussian whose opacity is below 1/255 is listed nowhere.  Every (Gaussian, tile) instance left out
	                         fails that test on every pixel of the tile, so the pixels, radii and gradients are the reference's
	                         (to fp32 rounding: the blend kernels pair the list entries differently); num_rendered, the
	                         lists and n_contrib (a list position) are not: a quarter fewer instances at C3.  The backward takes
	                         whatever lists the forward left */
	int32_t lazy;         /* 0: the call returns num_rendered (it waits for the tile scan, as the reference does at
	                         rasterizer_impl.cu:302, and starts over from the scatter pass when its run-ahead guess was too small).
	                         1: when the call can run ahead (the thread has rendered this (device, W, H, P) before, no debug mode) it
	                         enqueues the whole forward with generous buffers -- 1.5 x the largest num_rendered / longest list of the
	                         thread's last four reports -- and returns WITHOUT waiting: *num_rendered = -1 ("not known yet"; the backward
	                         accepts -1: the tile ranges carry everything the kernels need) and the host never blocks on the device.  If
	                         the lists turn out not to fit, scatter and sort leave everything alone ON THE DEVICE and the forward's outputs
	                         are INVALID (background only): fdgs_forward_lazy_status reports it, and the caller renders that view again
	                         with lazy = 0 BEFORE anything irreversible depends on it (fdgs.pipeline.StepPipeline: before the optimizer
	                         step of the views' batch).  Where it cannot run ahead (first call for a configuration, debug mode,
	                         fdgs_set_run_ahead(0)) the call behaves as with 0 and returns num_rendered >= 0. */
	int32_t sparse_lists; /* with lazy = 1 only (ignored otherwise).  1: the tile lists are NOT packed back to back: tile t's list occupies the
	                         fixed slots [t * cap, t * cap + n_t) of the binning buffer, cap = the longest list the run-ahead guess provides for
	                         (rounded up to 64).  Nothing has to know the lists' starts before the scatter pass then: the count and scan
	                         launches disappear from the forward (4 launches in front of the blend instead of 6), the scatter counts as it goes.
	                         The lists themselves -- which instances, in which order -- are what they are without the flag; `ranges` holds
	                         (t * cap, t * cap + n_t) instead of the reference's prefix sums (identifyTileRanges), num_rendered (reported
	                         lazily) is the same sum.  A list that outgrows cap is cut and the forward reported as failed, like any lazy
	                         forward that does not fit.  Costs address space: T * cap entries of 12.5 bytes instead of num_rendered --
	                         within a BUDGET: when T * cap entries would take more than max(1 GiB, 4 x the compact buffer the same guess
	                         gets) the forward keeps compact lists (count + scan launches, prefix-sum `ranges`), so one hot tile of a real
	                         capture cannot turn a 200 MB buffer into gigabytes (fdgs_set_sparse_lists_budget). */
	void* colour_stream;  /* with split_colour = 1: NULL = the library's own second stream; otherwise the hipStream_t the SH -> RGB
	                         launch goes onto (after an event of the geometry launch; the caller's stream waits for its event before the
	                         blend).  Everything already enqueued on that stream comes first -- so a caller whose optimizer updates the
	                         SH coefficients on it gets: geometry, binning and sort of the next view (which read no SH coefficient) next
	                         to that update on the call's stream, the colours right behind it (fdgs.pipeline.StepPipeline, the first view
	                         of a step).  Same device as the call's stream */
} fdgs_forward_out;

/* fdgs_backward_out.adam: the scene's geometry tensors are slices of ONE flat parameter buffer `flat`; exp_avg / exp_avg_sq are
   torch.optim.Adam's moments in the same layout (a parameter at flat + k has its moments at exp_avg + k, exp_avg_sq + k).  Learning
   rates per tensor (arguments/__init__.py:84-92), `step` >= 1 = the step being taken (bias corrections as fdgs_adam_step). */
typedef struct fdgs_geometry_adam
{
	uint32_t struct_size;
	float* flat; float* exp_avg; float* exp_avg_sq;
	float lr_means3D, lr_opacities, lr_ts, lr_scales, lr_scales_t, lr_rotations, lr_rotations_r;
	float beta1, beta2, eps;
	int32_t step;
} fdgs_geometry_adam;

/* Upstream gradients (d loss / d forward outputs).  Any of the four image gradients may be NULL = "this output
   has no upstream gradient" (treated as zero; at least one must be given).  With only dL_dout_color given the
   backward blend runs its colour-only variant. */
typedef struct fdgs_backward_in
{
	uint32_t struct_size;        /* sizeof(fdgs_backward_in)                       */
	const float* dL_dout_color;  /* [3,H,W]                                        */
	const float* dL_dout_depth;  /* [1,H,W]                                        */
	const float* dL_dout_alpha;  /* [1,H,W]  gradient w.r.t. alpha = 1 - T         */
	const float* dL_dout_flow;   /* [2,H,W]                                        */
	const int32_t* radii;        /* [P]  as returned by forward                    */
	const float* out_means3D;    /* [P,3] as returned by forward                   */
	const void* geom_buffer;     /* the three scratch buffers of the forward call  */
	const void* binning_buffer;
	const void* image_buffer;
	int32_t num_rendered;        /* R returned by forward (-1 from a lazy forward: fine) */
} fdgs_backward_in;

/* Gradients; every non-NULL array is fully written by the call (no pre-zeroing
 * needed).  dL_dsh may be NULL when M == 0; the 4D-only ones may be NULL when
 * the corresponding input is absent. */
typedef struct fdgs_backward_out
{
	uint32_t struct_size;   /* sizeof(fdgs_backward_out)                           */
	float* dL_dmeans2D;     /* [P,3]  (x,y in NDC-scaled units, z = depth carrier, Q10) */
	float* dL_dcolors;      /* [P,3]   or NULL: not wanted (as dL_dcov3D, dL_dflows: per-view outputs of the reference's binding that no
	                           parameter gradient is read from -- a training step saves their 44 bytes per Gaussian and view) */
	float* dL_dopacity;     /* [P]                                                 */
	float* dL_dmeans3D;     /* [P,3]                                               */
	float* dL_dcov3D;       /* [P,6]   or NULL                                     */
	float* dL_dsh;          /* [P,M,3]                                             */
	float* dL_dflows;       /* [P,2]   or NULL                                     */
	float* dL_dts;          /* [P]                                                 */
	float* dL_dscales;      /* [P,3]                                               */
	float* dL_dscales_t;    /* [P]                                                 */
	float* dL_drotations;   /* [P,4]                                               */
	float* dL_drotations_r; /* [P,4]                                               */
	int32_t accumulate;     /* 0: the parameter gradients (dL_dmeans3D, dL_dsh, dL_dopacity, dL_dts, dL_dscales,
	                           dL_dscales_t, dL_drotations, dL_drotations_r) are overwritten; 1: the kernels ADD into
	                           them (gradient accumulation over the views of one optimizer step, reference
	                           train.py:104-166).  The per-view outputs (dL_dmeans2D, dL_dcolors, dL_dflows,
	                           dL_dcov3D) are always overwritten. */
	float* grad_accum;      /* [P,16] scratch: packed per-Gaussian accumulators of the blend backward
	                           (colour 3, depth 1, flow 2, mean2D 2, conic xx/yy/xy 3 (Q12 convention), opacity 1,
	                           SH-backward mean / time 4) */
	int32_t grad_accum_clean; /* 0: the call clears grad_accum itself (one memset per backward).
	                             1: the caller guarantees it is all zero on entry (a persistent buffer, zeroed once) and
	                                the call leaves it all zero on exit: the last kernel re-zeroes what it has read */
	float* sh_stage;          /* NULL: dL_dsh is written / accumulated by this call.  Otherwise [P,8] floats of scratch owned by the
	                             caller: DEFERRED SH gradient -- the call leaves dL_dsh alone and stores, per Gaussian, the 8
	                             numbers this view contributes through (its dL_dRGB, the view direction, the cosine factors of
	                             the two time blocks);
	                             fdgs_sh_flush turns the stages of all views of an optimizer step into dL_dsh in one pass. */
	int32_t stage_mask;       /* 0 (or 3): the whole backward.  1: blend backward + SH backward only -- dL_dsh is final when
	                             they have run; 2: the geometry backward only (must follow a call with 1 on the same
	                             stream, same arguments).  Lets a data-parallel caller start the all-reduce of the SH
	                             gradients (88 % of the bytes at M = 48) while the geometry backward still runs.
	                             + 4 (5 = blend backward only): the SH backward of this view is left to fdgs_sh_backward_batch, which
	                             does it for all views of the optimizer step in one pass over the coefficients; the view's
	                             geometry backward (2) follows after that call.  Needs sh_stage and a grad_accum of the view's own. */
	const struct fdgs_geometry_adam* adam; /* NULL, or (raw_params scenes with all seven geometry tensors, i.e. rot_4d; the LAST view of an optimizer step): the geometry backward also TAKES the
	                             Adam step of the 17 geometry parameters of every Gaussian -- means3D, opacities, ts, scales, scales_t, rotations,
	                             rotations_r as the scene points at them -- with the gradient it has just completed (this view's, added to what
	                             the gradient arrays hold when accumulate = 1; the arrays still receive the sum).  Bit-identical to
	                             fdgs_adam_step over those tensors afterwards; saves that launch and its 28 bytes per parameter. */
} fdgs_backward_out;

/* Forward pass: preprocess -> tile count -> tile scan -> tile scatter -> per-tile local sort (+ ranges) -> per-tile
 * blend: 6 kernel launches, all enqueued before the host waits for num_rendered (the scan kernel writes it into a pinned
 * mailbox; scatter / sort check on the device that the binning buffer, sized from the calling thread's previous forward, is
 * large enough, and the call starts over from the scatter pass if not).  *num_rendered receives R when the call returns.
 * `stream` is a hipStream_t passed as void* so that this header needs no HIP include. */
int fdgs_rasterize_forward(const fdgs_scene* scene, const fdgs_forward_out* out,
                           fdgs_alloc_fn alloc, void* alloc_user, void* stream,
                           int32_t* num_rendered);

/* The lazy forwards (fdgs_forward_out.lazy) of the CALLING THREAD on the current device, since the previous call of this function:
 * *pending = how many have not reported yet (always 0 with wait != 0: the call then blocks until the tile scan of the most recent one
 * has run -- every pending forward is waited for on the stream IT was enqueued on, whatever stream this call is made from: once that
 * stream has drained or failed the report is in or the call fails; `stream` is accepted for compatibility and not used); *failed = how many turned out not to fit
 * their buffers: the outputs of those forwards are invalid and must be rendered again with lazy = 0; num_rendered[0 .. *n_out) = the
 * num_rendered of the reported ones, oldest first (at most max_out, at most 64).  Every pointer may be NULL. */
int fdgs_forward_lazy_status(int32_t wait, void* stream, int32_t* pending, int32_t* failed,
                             int32_t* num_rendered, int32_t max_out, int32_t* n_out);

/* 1 (default): the forward enqueues scatter / sort / blend before num_rendered is back, with a binning buffer sized from the
 * calling thread's previous call; 0: it always waits and asks the allocator for exactly fdgs_binning_bytes(num_rendered)
 * (+ the long-list scratch), as the reference does.  Process-wide. */
void fdgs_set_run_ahead(int32_t enable);

/* Budget of fdgs_forward_out.sparse_lists: a forward takes the sparse layout only while its binning buffer stays within
 * max(min_bytes, factor x the bytes of the compact buffer sized from the same run-ahead guess).  Defaults: 1 GiB (environment:
 * FDGS_SPARSE_BUDGET_MB), 4.  min_bytes = 0, factor = 1: never more than the compact buffer, i.e. practically always compact.
 * Process-wide.  Returns FDGS_ERR_INVALID_ARG for min_bytes < 0 or factor < 1. */
int fdgs_set_sparse_lists_budget(int64_t min_bytes, int32_t factor);
/* Introspection: counts3[0] forwards that ran with sparse lists, [1] forwards that asked for them and kept compact lists because of the
 * budget, [2] bytes of the binning buffer the last run-ahead forward asked its allocator for. */
void fdgs_debug_sparse_lists_stats(int64_t* counts3);

/* View-batched preprocess (the views of ONE optimizer step: same Gaussian tensors, same P / M / degrees / flags; cameras and
 * timestamps differ).  The 12 M bytes of SH coefficients per Gaussian are most of what the preprocess reads, and they are
 * the same for every view: the geometry part runs per view, the SH colours of all views in ONE pass over the coefficients
 * (bit-identical to the per-view forward).  For each view v the call obtains the geometry and the image buffer through
 * alloc(alloc_users[v], ...) and fills outs[v]->radii / out_means3D / covs_com; the views' forwards are then completed one by
 * one with fdgs_rasterize_forward(scenes[v], outs[v] with preprocessed = 1, alloc, alloc_users[v], the same stream, ...).
 * The reference has no counterpart (train.py:104-166 renders the views of a batch strictly one after the other). */
int fdgs_preprocess_batch(int32_t num_views, const fdgs_scene* const* scenes, const fdgs_forward_out* const* outs,
                          fdgs_alloc_fn alloc, void* const* alloc_users, void* stream);

/* View-batched SH backward (deferred mode), the counterpart of fdgs_preprocess_batch: after the blend backward of every
 * view (fdgs_rasterize_backward with stage_mask = 5, a grad_accum and an sh_stage of the view's own), ONE pass over the SH
 * coefficients produces every view's stage record and its mean / time gradient (words 12..15 of the view's accumulator
 * records); the views' geometry backward (stage_mask = 2) and fdgs_sh_flush / fdgs_adam_step_sh follow.  Bit-identical to the
 * per-view SH backward.  ins[v] / outs[v]: the structs of the views' backward calls. */
int fdgs_sh_backward_batch(int32_t num_views, const fdgs_scene* const* scenes, const fdgs_backward_in* const* ins,
                           const fdgs_backward_out* const* outs, void* stream);

/* Introspection for tests: how the forward calls of this process went -- counts3[0] scatter / sort / blend enqueued before
 * num_rendered was back and kept, [1] enqueued ahead but sorted again (longer tile lists than the previous call suggested),
 * [2] sized exactly after the wait (first call of a thread, debug mode, more instances than the previous call suggested). */
void fdgs_debug_run_ahead_stats(int64_t* counts3);

/* Backward pass: blend backward -> fused cov2D / projection / SH / covariance backward. */
int fdgs_rasterize_backward(const fdgs_scene* scene, const fdgs_backward_in* in,
                            const fdgs_backward_out* out, void* stream);

/* Deferred SH gradient, second half (see fdgs_backward_out.sh_stage): dL_dsh [P,M,3] = sum over the num_views staged views,
 * in view order, of basis(direction, time) (x) dL_dRGB -- the same additions in the same order as backward calls that
 * accumulate into dL_dsh view after view, with 1/3 of their memory traffic at 4 views.  stages: [num_views,P,8] floats, the
 * sh_stage buffers of the views back to back.  accumulate != 0 adds to dL_dsh instead of overwriting it.
 * D, D_t, M, gaussian_dim, force_sh_3d, analytic_sh_grad as in the fdgs_scene of the staged views (one value for all of them). */
int fdgs_sh_flush(int32_t P, int32_t D, int32_t D_t, int32_t M, int32_t gaussian_dim, int32_t force_sh_3d, int32_t analytic_sh_grad,
                  int32_t num_views, const float* stages, float* dL_dsh, int32_t accumulate, void* stream);

/* present[i] = view-space z of means3D[i] > 0.2 (checkFrustum, rasterizer_impl.cu:54-67). */
int fdgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, void* stream);

/* Sizes of the opaque scratch buffers.  Geometry and image: exactly what the allocator is asked for.  Binning: the allocator
 * is asked for fdgs_binning_bytes(n) with n = num_rendered when the forward sizes the buffer after the wait (first call of a
 * thread for a (device, W, H, P), debug mode, fdgs_set_run_ahead(0)), and with n = 1.25 R' + 4096 <= 2^31 - 1 when it runs
 * ahead (R' = num_rendered of the thread's previous call for the same (device, W, H, P)); a second request in the same call
 * follows if that was too small.  Either request grows by 8 bytes per instance (of n) in the rare case that a single tile's
 * list is longer than the LDS sort takes (16384 entries).  (fdgs_binning_bytes(n) = 12 bytes per instance + 32 bytes per 64
 * instances and per tile: the blend forward's per-block cull decisions, kept for the blend backward.)  A caller that pre-sizes a binning arena either uses that bound
 * or switches the run-ahead off. */
size_t fdgs_geometry_bytes(int32_t P);
size_t fdgs_image_bytes(int32_t W, int32_t H);
size_t fdgs_binning_bytes(int32_t num_rendered, int32_t W, int32_t H);

/* Introspection for parity tests: device pointers into the opaque buffers of a
 * finished forward call.  Arrays are indexed as documented in DESIGN.md. */
typedef struct fdgs_debug_view
{
	uint32_t struct_size;          /* sizeof(fdgs_debug_view)                      */
	const float* depths;           /* [P]   view-space z (0 where culled)          */
	const float* records;          /* [P,12] packed blend record: x,y,conic(3),opacity,r,g,b,depth,flow(2) */
	const float* cov3D;            /* [P,6]                                        */
	const uint32_t* tiles_touched; /* [P]                                          */
	const uint8_t* clamped;        /* [P]   bit c set: colour channel c clamped    */
	const uint32_t* point_list;    /* [R]   Gaussian ids sorted by (tile, depth, id) */
	const uint32_t* ranges;        /* [T,2]                                        */
	const uint32_t* n_contrib;     /* [H*W]                                        */
	const float* final_T;          /* [H*W]                                        */
	const uint32_t* tile_order;    /* [T]   the order in which the blend kernels take the tiles: a permutation of all tiles,
	                                  longest lists first (64 length classes), position p on XCD p % 8; written by one
	                                  workgroup of the tile-scatter (sparse lists: sort) launch, so undefined for P == 0 */
} fdgs_debug_view;
int fdgs_debug_views(int32_t P, int32_t W, int32_t H, int32_t num_rendered,
                     const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                     fdgs_debug_view* view);

/* Introspection for parity tests of the fused-activation mode (fdgs_scene.raw_params = 1): the activated tensors the
 * kernels derive from the RAW parameters -- sigmoid(opacity), exp(scales), exp(scales_t), v / max(|v|, 1e-12) for the two
 * quaternions (scene/gaussian_model.py:179-219) -- computed by the same device functions, bit-identical to the values
 * preprocess uses in flight.  Any input / output pair may be NULL.  A test feeds them to the oracle, so that the
 * 1e-4 bar applies to the raw_params path as well. */
int fdgs_debug_activations(int32_t P, const float* opacity_raw, const float* scales_raw, const float* scales_t_raw,
                           const float* rotations_raw, const float* rotations_r_raw, float* opacity, float* scales,
                           float* scales_t, float* rotations, float* rotations_r, void* stream);

/* Introspection for the block-cull test: the blend kernels skip a list entry for a whole 8x8 pixel block when a closed-form
 * bound says that no pixel of the block can reach alpha >= 1/255 (csrc/blend_common.h, block_reaches).  tuples [n][10] =
 * mean x, y, conic xx, xy, yy, opacity, rx0, rx1, ry0, ry1 (first / last pixel centre of the block in x and y); out [n][2] =
 * { the bound's verdict, brute force: does any pixel centre of the block pass the blend kernels' own per-pixel test }.
 * The verdict must never be 0 where the brute force says 1. */
int fdgs_debug_block_reaches(int32_t n, const float* tuples, uint8_t* out, void* stream);

/* Measurement aid: a one-wave kernel on `stream` that records, into out5 (DEVICE memory, 5 x uint64), the constant-rate wall
 * clock (s_memrealtime) and the shader-clock counter (s_memtime) at its start and again span_ms later, plus the wall clock's
 * rate in kHz: { wall0, shader0, wall1, shader1, wall_khz }.  (shader1 - shader0) / (wall1 - wall0) * wall_khz = the shader
 * clock in kHz the chip sustained over the interval, under whatever the caller's other streams were running (bench.py:
 * valu_issue_frac at the MEASURED clock).  span_ms in (0, 2000]. */
int fdgs_debug_clock_sample(uint64_t* out5, double span_ms, void* stream);

/* Test hook: lists longer than `lds_cap` entries take the global-scratch sort, tiles whose most crowded depth bucket
 * exceeds `rank_max` the LDS bitonic sort (tilebin.hip); values <= 0 restore the defaults (16384, 96).  Process-wide. */
void fdgs_debug_tile_sort_limits(int32_t lds_cap, int32_t rank_max);

/* Optional per-stage timing with HIP events recorded on the caller's stream (so the
 * numbers are the kernels' own durations on that stream, not host wall time).
 * Disabled by default; when enabled each stage of forward / backward is bracketed
 * by an event pair (non-blocking).  fdgs_profile_read synchronises the pending
 * events and returns the accumulated milliseconds and the number of samples. */
#define FDGS_STAGE_PREPROCESS_FWD 0
#define FDGS_STAGE_TILE_COUNT 1
#define FDGS_STAGE_TILE_SCAN 2
#define FDGS_STAGE_TILE_SCATTER 3
#define FDGS_STAGE_TILE_SORT 4
#define FDGS_STAGE_COLOUR_FWD 5  /* the SH colour half of the preprocess when the forward runs it on its second stream (split_colour) */
#define FDGS_STAGE_BLEND_FWD 6
#define FDGS_STAGE_BLEND_BWD 7
#define FDGS_STAGE_PREPROCESS_BWD 8
#define FDGS_STAGE_GRAD_ZERO 9
#define FDGS_STAGE_SH_BWD 10
#define FDGS_NUM_STAGES 11
int fdgs_profile_enable(int stage_mask); /* bit i set: bracket stage i with HIP events; 0 = off, -1 = all stages */
int fdgs_profile_sample_every(int32_t n); /* n >= 1: only every n-th launch of a bracketed stage gets its event pair (default 1: every
                                             launch).  An event pair costs its stream ~13 us of idle time around the launch (measured: a
                                             6-7 us gap in front of and behind every bracketed blend backward in the kernel trace); a caller
                                             that measures a stage INSIDE a timed region samples it -- n coprime to the views per step, so
                                             that the samples rotate through the views */
int fdgs_profile_read(int stage, double* total_ms, int64_t* samples);
int fdgs_profile_reset(void);
const char* fdgs_stage_name(int stage);

/* ---- adjacent row: fused photometric loss ----------------------------
 * (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM(img, gt)), the reference's loss
 * (train.py:115-117, utils/loss_utils.py:17-64: 11x11 Gaussian window, sigma 1.5, zero padding).
 * forward writes three per-pixel derivative maps [C,H,W] (kept for backward) and per-tile partial sums
 * of |img-gt| and ssim (fdgs_l1_ssim_num_partials entries each; the host adds them up).
 * backward writes dL/dimg [C,H,W] given the upstream scalar gradient (device pointer). */
int fdgs_l1_ssim_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                         float* dm_dmu1, float* dm_de11, float* dm_de12,
                         float* partial_l1, float* partial_ssim, void* stream);
int fdgs_l1_ssim_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                          const float* dm_dmu1, const float* dm_de11, const float* dm_de12,
                          const float* upstream, float lambda_dssim, float* dL_dimg, void* stream);
int fdgs_l1_ssim_num_partials(int32_t C, int32_t H, int32_t W);
/* Value and gradient in ONE pass (what a training step needs; same arithmetic, bit-identical results): dL_dimg = upstream[0] *
 * d loss / d img and the per-tile partial sums for fdgs_l1_ssim_loss, without the three derivative maps of the two-call path
 * travelling through memory (a workgroup rebuilds the derivative maps around its tile from the images: more arithmetic, a third
 * of the HBM traffic).  utils/loss_utils.py:34-64 + its autograd backward. */
int fdgs_l1_ssim_value_and_grad(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                const float* upstream, float lambda_dssim, float* dL_dimg,
                                float* partial_l1, float* partial_ssim, void* stream);
/* Reduces the per-tile partial sums of the forward call (fixed order: deterministic) to
 * loss_l1_ssim[0] = (1 - lambda) * L1 + lambda * (1 - SSIM), [1] = L1, [2] = SSIM   (device memory, 3 floats). */
int fdgs_l1_ssim_loss(const float* partial_l1, const float* partial_ssim, int32_t num_partials, int32_t C, int32_t H, int32_t W,
                      float lambda_dssim, float* loss_l1_ssim, void* stream);
/* The same reduction for every view of an optimizer step in ONE launch (one workgroup per view; bit-identical to num_views calls of
 * fdgs_l1_ssim_loss): partials = [num_views][2][num_partials] -- view v's partial_l1 at partials + v * 2 * num_partials, its
 * partial_ssim num_partials floats behind -- losses = [num_views][3]. */
int fdgs_l1_ssim_loss_batch(const float* partials, int32_t num_views, int32_t num_partials, int32_t C, int32_t H, int32_t W,
                            float lambda_dssim, float* losses, void* stream);

/* ---- adjacent row: fused Adam over a flat parameter bucket ---------
 * torch.optim.Adam arithmetic (no amsgrad / weight decay) in one streaming pass over four flat fp32
 * buffers.  Learning rates come from `segments`: elements [begin, end) use `lr`, except that when
 * period > 0 the first `head` elements of every `period` use `lr_head` (SH DC vs. rest inside one
 * [P, M, 3] tensor).  Elements outside every segment are left with lr = 0 (moments still update). */
typedef struct fdgs_adam_segment
{
	int64_t begin, end;
	float lr, lr_head;
	int32_t period, head;
} fdgs_adam_segment;
int fdgs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const fdgs_adam_segment* segments, int32_t num_segments,
                   float beta1, float beta2, float eps, int32_t step, void* stream);

/* Adam over the SH coefficients [P,M,3] with the gradient taken straight from the staged views of the deferred SH backward
 * (fdgs_backward_out.sh_stage): fdgs_sh_flush + fdgs_adam_step on that segment in one pass, without the 12 M bytes per
 * Gaussian of dL_dsh going out to memory and coming back.  params / exp_avg / exp_avg_sq point at the [P,M,3] segment; the
 * first 3 floats of every row (the DC coefficient, scene/gaussian_model.py:339) use lr_dc, the others lr.  dL_dsh: NULL, or
 * [P,M,3] that also receives the summed gradient (bit-identical to fdgs_sh_flush).  The update is bit-identical to
 * fdgs_sh_flush followed by fdgs_adam_step.  Needs rows of whole float4s (3 M % 4 == 0) and 16-byte aligned arrays:
 * FDGS_ERR_INVALID_ARG otherwise (the caller falls back to the two calls). */
int fdgs_adam_step_sh(float* params, float* exp_avg, float* exp_avg_sq, float* dL_dsh,
                      int32_t P, int32_t D, int32_t D_t, int32_t M, int32_t gaussian_dim, int32_t force_sh_3d,
                      int32_t analytic_sh_grad, int32_t num_views, const float* stages,
                      float lr, float lr_dc, float beta1, float beta2, float eps, int32_t step, void* stream);

/* ---- adjacent row: densification / pruning of the flat-bucket model --------------
 * Replaces the boolean-mask gathers and torch.cat calls of scene/gaussian_model.py:391-610 (densify_and_clone,
 * densify_and_split, prune_points, cat_tensors_to_optimizer, _prune_optimizer) for a model whose parameters,
 * exp_avg and exp_avg_sq are one flat buffer each (segments [P,row_0] [P,row_1] ... back to back). */
#define FDGS_DENSIFY_CLONE 1        /* clone this Gaussian (gaussian_model.py:545-569)                          */
#define FDGS_DENSIFY_SPLIT 2        /* replace it by N samples (:487-543)                                       */
#define FDGS_DENSIFY_PRUNE 4        /* the Gaussian (and its clone) fails the final prune test (:598-603)       */
#define FDGS_DENSIFY_PRUNE_CHILD 8  /* its split children (scaling / (0.8 N)) fail the final prune test         */
/* Densification statistics of one optimizer step (train.py:164-184, 229-236; gaussian_model.py:637-642), two phases so
 * that the per-Gaussian sums can be all-reduced in between (count, pgrad: SUM; radii_max: MAX):
 *   local: this rank's views -> count = #views with radius > 0, pgrad = sum of ||dL/dmean2D.xy||, radii_max (all [P] float);
 *   apply: for Gaussians seen at least once, max_radii2D = max(., radii_max), xyz_gradient_accum += pgrad * batch / count,
 *          denom += 1, t_gradient_accum += t_grad * batch / count (t_grad NULL: skipped).
 * radii / viewspace_grad: HOST arrays of num_views (<= 16) device pointers ([P] int32, [P,3] float). */
int fdgs_densify_stats_local(int32_t P, int32_t num_views, const int32_t* const* radii, const float* const* viewspace_grad,
                             float* count, float* pgrad, float* radii_max, void* stream);
int fdgs_densify_stats_apply(int32_t P, const float* count, const float* pgrad, const float* radii_max, const float* t_grad,
                             float global_batch, float* xyz_gradient_accum, float* t_gradient_accum, float* denom,
                             float* max_radii2D, void* stream);
/* Per-Gaussian decision flags from the densification statistics.  max_screen_size <= 0: no size test
 * (`if max_screen_size:`); percent_dense, extent: as training_setup / cameras_extent; prune_only as in :584. */
int fdgs_densify_classify(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                          const float* opacity_raw, const float* max_radii2D, float max_grad, float min_opacity,
                          float extent, float max_screen_size, float percent_dense, int32_t N, int32_t prune_only,
                          uint8_t* flags, void* stream);
/* Builds the three new flat buffers: row j of every segment is row src[j] of the old one; kind[j] == 0 keeps the
 * Adam moments, anything else zeroes them (new points, gaussian_model.py:441-442). */
int fdgs_densify_gather(int32_t num_segments, const int32_t* row_floats, int64_t P_old, int64_t P_new,
                        const int32_t* src, const uint8_t* kind, const float* old_params, const float* old_exp_avg,
                        const float* old_exp_avg_sq, float* new_params, float* new_exp_avg, float* new_exp_avg_sq, void* stream);
/* Split children (gaussian_model.py:497-524): new_* point at the children's rows of the new arrays; parent[c] indexes
 * the OLD arrays; samples [n,4] (rot_4d) or [n,3] (+ samples_t [n] for 4D without rot_4d) are draws of N(0, std). */
int fdgs_densify_split(int32_t n_children, int32_t N, int32_t rot_4d, int32_t gaussian_dim, const int32_t* parent,
                       const float* samples, const float* samples_t, const float* xyz, const float* t, const float* scaling,
                       const float* scaling_t, const float* rotation, const float* rotation_r,
                       float* new_xyz, float* new_t, float* new_scaling, float* new_scaling_t, void* stream);

/* ---- adjacent row: simple-knn's distCUDA2 (simple-knn/spatial.cu:15-27) ----------
 * mean_dist2[i] = mean of the squared distances from points[i] to its three nearest other points ([P,3] float,
 * device).  scratch: fdgs_knn_scratch_bytes(P) bytes of device memory.  Exact search; bit-exact vs the oracle. */
size_t fdgs_knn_scratch_bytes(int32_t P);
int fdgs_dist2_knn3(int32_t P, const float* points, float* mean_dist2, void* scratch, void* stream);

/* Thread-local description of the last error on this thread ("" if none). */
const char* fdgs_last_error(void);
int fdgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FDGS_H */
