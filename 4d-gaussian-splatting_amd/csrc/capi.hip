// capi.hip -- extern "C" entry points of libfdgs.so (declared in include/fdgs.h).
//
// Host orchestration of the forward and backward pipelines; the role of
// CudaRasterizer::Rasterizer::forward / backward (rasterizer_impl.cu:199-364,
// 368-496) plus the argument checks of rasterize_points.cu:69-71.  Everything is
// enqueued on the caller's stream; the only host synchronisation is the
// read-back of num_rendered (4 bytes through a pinned staging word), as in the
// reference (rasterizer_impl.cu:302).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fdgs_common.h"

using namespace fdgs;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr, what)                                                                         \
	do {                                                                                            \
		hipError_t e__ = (expr);                                                                    \
		if (e__ != hipSuccess) return fail(FDGS_ERR_HIP, "%s: %s", what, hipGetErrorString(e__));   \
	} while (0)

// ---- optional per-stage HIP-event timing (fdgs_profile_*) ----
namespace
{
	constexpr int PROF_CAP = 2048; // pending event pairs per stage before a (blocking) flush
	struct StageProf
	{
		hipEvent_t start[PROF_CAP], stop[PROF_CAP];
		int created = 0, pending = 0;
		double total_ms = 0.0;
		long long samples = 0;
	};
	// process-wide (autograd runs the backward on its own thread); guarded by g_prof_mu
	std::atomic<uint32_t> g_prof_mask{0};   // bit i: stage i is bracketed with events
	std::atomic<int> g_prof_every{1};       // ... every n-th launch of it (fdgs_profile_sample_every)
	std::atomic<unsigned> g_prof_calls[FDGS_NUM_STAGES];
	StageProf g_prof[FDGS_NUM_STAGES];
	std::mutex g_prof_mu;

	void prof_flush(StageProf& p)
	{
		for (int i = 0; i < p.pending; i++)
		{
			float ms = 0.f;
			if (hipEventSynchronize(p.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, p.start[i], p.stop[i]) == hipSuccess)
			{
				p.total_ms += ms;
				p.samples++;
			}
		}
		p.pending = 0;
	}
	struct StageTimer
	{
		StageProf* p = nullptr;
		hipStream_t stream;
		int slot = 0;
		bool locked = false;
		StageTimer(int stage, hipStream_t s) : stream(s)
		{
			if (!((g_prof_mask.load(std::memory_order_relaxed) >> stage) & 1u)) return;
			const int every = g_prof_every.load(std::memory_order_relaxed);
			if (every > 1 && g_prof_calls[stage].fetch_add(1u, std::memory_order_relaxed) % (unsigned)every != 0u) return;
			g_prof_mu.lock();
			locked = true;
			p = &g_prof[stage];
			if (p->pending == PROF_CAP) prof_flush(*p);
			slot = p->pending;
			if (slot >= p->created)
			{
				if (hipEventCreate(&p->start[slot]) != hipSuccess || hipEventCreate(&p->stop[slot]) != hipSuccess) { p = nullptr; return; }
				p->created = slot + 1;
			}
			(void)hipEventRecord(p->start[slot], stream);
		}
		~StageTimer()
		{
			if (p)
			{
				(void)hipEventRecord(p->stop[slot], stream);
				p->pending = slot + 1;
			}
			if (locked) g_prof_mu.unlock();
		}
	};
	const char* const STAGE_NAMES[FDGS_NUM_STAGES] = { "preprocess_fwd", "tile_count", "tile_scan", "tile_scatter",
		"tile_sort", "colour_fwd", "blend_fwd", "blend_bwd", "preprocess_bwd", "grad_zero", "sh_bwd" };
}

extern "C" int fdgs_profile_enable(int stage_mask) { g_prof_mask.store((uint32_t)stage_mask); return FDGS_OK; }
extern "C" int fdgs_profile_sample_every(int32_t n)
{
	if (n < 1) return FDGS_ERR_INVALID_ARG;
	g_prof_every.store(n);
	for (auto& c : g_prof_calls) c.store(0u);
	return FDGS_OK;
}
extern "C" int fdgs_profile_reset(void)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	for (auto& p : g_prof) { prof_flush(p); p.total_ms = 0.0; p.samples = 0; }
	return FDGS_OK;
}
extern "C" int fdgs_profile_read(int stage, double* total_ms, int64_t* samples)
{
	if (stage < 0 || stage >= FDGS_NUM_STAGES) return FDGS_ERR_INVALID_ARG;
	std::lock_guard<std::mutex> lk(g_prof_mu);
	prof_flush(g_prof[stage]);
	if (total_ms) *total_ms = g_prof[stage].total_ms;
	if (samples) *samples = g_prof[stage].samples;
	return FDGS_OK;
}
extern "C" const char* fdgs_stage_name(int stage) { return (stage >= 0 && stage < FDGS_NUM_STAGES) ? STAGE_NAMES[stage] : ""; }

// TIMING PROBE ONLY, compiled in with -DFDGS_TIMING_PROBE (tools/sensitivity_probe.py builds its own copy of the library into
// tools/ab/; the in-tree build cannot skip anything): FDGS_TIMING_PROBE_SKIP=<bit mask over FDGS_STAGE_*> in the environment makes
// that copy NOT launch those stages -- every result is then garbage; what is measured is how much of a stage's time the two-stream
// step actually pays for (the upper bound of what optimising that kernel can return).
#ifdef FDGS_TIMING_PROBE
static const unsigned g_probe_skip = []() { const char* e = getenv("FDGS_TIMING_PROBE_SKIP"); return e ? (unsigned)strtoul(e, nullptr, 0) : 0u; }();
#define FDGS_STAGE_SKIPPED(id) (((g_probe_skip >> (id)) & 1u) != 0u)
#else
#define FDGS_STAGE_SKIPPED(id) false
#endif

// debug mode == the reference's CHECK_CUDA(..., debug): synchronise and check after each stage
#define STAGE(id, expr, what)                                                                       \
	do {                                                                                            \
		StageTimer timer__(id, stream);                                                             \
		if (!FDGS_STAGE_SKIPPED(id)) HIP_TRY((expr), what);                                         \
		if (debug) HIP_TRY(hipStreamSynchronize(stream), what);                                     \
	} while (0)

// every struct of the ABI starts with its own size as the caller's header defined it: a caller built against another fdgs.h
// is turned away here instead of the library reading past the end of a shorter struct
#define CHECK_STRUCT(ptr, type)                                                                                           \
	do {                                                                                                                  \
		if ((ptr)->struct_size != (uint32_t)sizeof(type))                                                                 \
			return fail(FDGS_ERR_INVALID_ARG, #type ".struct_size is %u, this library (FDGS_VERSION %d) expects %zu: built against another fdgs.h?", \
			            (unsigned)(ptr)->struct_size, FDGS_VERSION, sizeof(type));                                        \
	} while (0)

static int check_scene(const fdgs_scene* s)
{
	if (!s) return fail(FDGS_ERR_INVALID_ARG, "scene is NULL");
	CHECK_STRUCT(s, fdgs_scene);
	if (s->P < 0 || s->W <= 0 || s->H <= 0) return fail(FDGS_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", s->P, s->W, s->H);
	if (div_up(s->W, TILE_X) > 65535 || div_up(s->H, TILE_Y) > 65535) return fail(FDGS_ERR_INVALID_ARG, "image too large for 16-bit tile rectangles");
	// (the tile order packs a rank into 24 bits, and T = gx * gy is an int everywhere: 2^24 tiles = 4.3 gigapixels)
	if ((long long)div_up(s->W, TILE_X) * div_up(s->H, TILE_Y) >= (1ll << 24)) return fail(FDGS_ERR_INVALID_ARG, "image too large: %d x %d has 2^24 tiles or more", s->W, s->H);
	if (s->P == 0) return FDGS_OK;
	if (!s->means3D || !s->opacities || !s->bg || !s->viewmatrix || !s->projmatrix || !s->campos)
		return fail(FDGS_ERR_INVALID_ARG, "means3D / opacities / bg / viewmatrix / projmatrix / campos must not be NULL");
	// gaussian_renderer/diff_gaussian_rasterization.py:271-280
	if ((s->shs == nullptr) == (s->colors_precomp == nullptr))
		return fail(FDGS_ERR_INVALID_ARG, "Please provide excatly one of either SHs or precomputed colors!");
	if (s->shs && s->M <= 0) return fail(FDGS_ERR_INVALID_ARG, "shs given but M == 0");
	if (s->cov3D_precomp == nullptr)
	{
		if (!s->scales || !s->rotations)
			return fail(FDGS_ERR_INVALID_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
		if (s->rot_4d && (!s->rotations_r || !s->scales_t || !s->ts))
			return fail(FDGS_ERR_INVALID_ARG, "Please provide exactly rotations_r and scales_t and ts if rot_4d and cov3D_precomp is None!");
		if (!s->rot_4d && s->gaussian_dim == 4 && (!s->scales_t || !s->ts))
			return fail(FDGS_ERR_INVALID_ARG, "gaussian_dim == 4 needs scales_t and ts");
	}
	if (s->shs && !(s->gaussian_dim == 3 || s->force_sh_3d))
	{
		if (!s->ts) return fail(FDGS_ERR_INVALID_ARG, "4D SH needs ts");
		const int need = (s->D > 2) ? 16 * (1 + (s->D_t > 2 ? 2 : (s->D_t > 0 ? s->D_t : 0))) : (s->D + 1) * (s->D + 1);
		if (s->M < need) return fail(FDGS_ERR_INVALID_ARG, "M=%d too small for D=%d D_t=%d", s->M, s->D, s->D_t);
	}
	else if (s->shs && s->M < (s->D + 1) * (s->D + 1))
		return fail(FDGS_ERR_INVALID_ARG, "M=%d too small for D=%d", s->M, s->D);
	return FDGS_OK;
}

extern "C" size_t fdgs_geometry_bytes(int32_t P) { return geom_layout(P).total; }
extern "C" size_t fdgs_image_bytes(int32_t W, int32_t H) { return image_layout(W, H).total; }
extern "C" size_t fdgs_binning_bytes(int32_t R, int32_t W, int32_t H)
{
	return bin_layout(R, false, div_up(W > 0 ? W : 1, TILE_X) * div_up(H > 0 ? H : 1, TILE_Y)).total;
}
extern "C" void fdgs_debug_tile_sort_limits(int32_t lds_cap, int32_t rank_max) { tile_sort_debug_limits(lds_cap, rank_max); }
extern "C" const char* fdgs_last_error(void) { return g_err; }
extern "C" int fdgs_version(void) { return FDGS_VERSION; }

// how the forward calls of this process went: [0] everything enqueued ahead of num_rendered and kept, [1] ahead but sorted
// again (longer lists than guessed), [2] exact sizes (first call of a thread, debug mode, or more instances than guessed)
static std::atomic<long long> g_run_ahead[3];
static std::atomic<bool> g_run_ahead_enabled{[]() { const char* e = getenv("FDGS_RUN_AHEAD"); return !(e && e[0] == '0'); }()};   // FDGS_RUN_AHEAD=0: as fdgs_set_run_ahead(0)
constexpr int FDGS_MAX_DEVICES = 64;   // = the size of g_box_of
constexpr int FDGS_GUESS_SLOTS = 8;    // = the size of g_guesses
extern "C" void fdgs_set_run_ahead(int32_t enable) { g_run_ahead_enabled.store(enable != 0); }
// Byte budget of fdgs_forward_out.sparse_lists (include/fdgs.h): a forward takes the sparse layout only when its binning buffer --
// T * cap entries -- stays within max(g_sparse_min_bytes, g_sparse_factor x the compact buffer the same guess would get).
static std::atomic<long long> g_sparse_min_bytes{[]() { const char* e = getenv("FDGS_SPARSE_BUDGET_MB"); return (e && atoll(e) > 0 ? atoll(e) : 1024ll) << 20; }()};
static std::atomic<int> g_sparse_factor{4};
static std::atomic<long long> g_sparse_stats[3];   // forwards with sparse lists, forwards that asked for them and got compact lists (budget), bytes of the last binning buffer
extern "C" int fdgs_set_sparse_lists_budget(int64_t min_bytes, int32_t factor)
{
	if (min_bytes < 0 || factor < 1) return FDGS_ERR_INVALID_ARG;
	g_sparse_min_bytes.store(min_bytes); g_sparse_factor.store(factor);
	return FDGS_OK;
}
extern "C" void fdgs_debug_sparse_lists_stats(int64_t* counts3)
{
	for (int k = 0; k < 3; k++) counts3[k] = (int64_t)g_sparse_stats[k].load();
}
extern "C" void fdgs_debug_run_ahead_stats(int64_t* counts3)
{
	for (int k = 0; k < 3; k++) counts3[k] = (int64_t)g_run_ahead[k].load();
}

// ---- per host thread and device: the pinned mailbox ring the tile scans report into, the run-ahead guesses, the lazy forwards ----
namespace
{
	constexpr int MAIL_SLOTS = 64;   // forwards of one thread that may be unreported at a time (fdgs_forward_out.lazy); 16 bytes each
	struct RunAhead { int dev = -1, W = 0, H = 0, P = 0; long long capacity = 0; int longest = 0; long long r_hist[4] = { 0, 0, 0, 0 }; int l_hist[4] = { 0, 0, 0, 0 }; int hist_at = 0; };
	struct MailRec { unsigned long long seq = 0; long long cap = 0; int longest_cap = 0; RunAhead* guess = nullptr; int gdev = 0, gW = 0, gH = 0, gP = 0; bool pending = false, lazy = false; hipStream_t stream = nullptr; /* the stream the forward (its tile scan) was enqueued on */ };
	struct Mailbox
	{
		volatile uint32_t* host = nullptr; uint32_t* dev = nullptr;
		unsigned long long seq = 0, head = 1;   // last sequence number handed out; oldest one that may still be pending
		MailRec rec[MAIL_SLOTS];
		int failed = 0;                         // lazy forwards whose lists did not fit, since the last fdgs_forward_lazy_status
		int done_R[MAIL_SLOTS]; int done_n = 0; // num_rendered of the lazy forwards reported since then (oldest first)
	};
	thread_local Mailbox* g_box_of[64];   // allocated on a thread's first forward on the device
	thread_local RunAhead g_guesses[8];
	thread_local int g_guess_next = 0;
	inline uint32_t ticket_of(unsigned long long seq) { return (uint32_t)(seq % 0xFFFFFFFFull) + 1u; }   // never 0

	void note_result(RunAhead& g, long long R, int longest)
	{
		g.capacity = std::min<long long>(R + R / 4 + 4096, 0x7fffffffLL);
		g.longest = longest + longest / 4;
		g.r_hist[g.hist_at & 3] = R; g.l_hist[g.hist_at & 3] = longest; g.hist_at++;
	}
	// Reads the reports of this thread's pending forwards, oldest first, up to sequence number `upto` (inclusive; HARVEST_ALL: every
	// forward handed a number so far -- only where all of them are REGISTERED, i.e. not between ++box.seq and the registration of
	// that forward's record: the loop would step over the unregistered number and the forward's own wait would find nothing to
	// wait for).  wait: block until they are in; otherwise stop at the first one that has not reported.  While waiting, the stream
	// the PENDING forward itself was enqueued on (MailRec::stream -- not the caller's: a lazy forward may have gone onto another
	// stream than the call that reads its report) is polled now and then: once it has drained (or failed) the report is in or
	// never will be -- one look after a device-wide synchronize decides.  Returns false when a report never showed up.
	constexpr unsigned long long HARVEST_ALL = ~0ull;
	bool harvest(Mailbox& box, bool wait, unsigned long long upto)
	{
		while (box.head <= box.seq && box.head <= upto)
		{
			MailRec& r = box.rec[box.head % MAIL_SLOTS];
			if (!r.pending || r.seq != box.head) { box.head++; continue; }
			volatile uint32_t* m = box.host + 4 * (box.head % MAIL_SLOTS);
			const uint32_t want = ticket_of(r.seq);
			bool arrived = __atomic_load_n(&m[2], __ATOMIC_ACQUIRE) == want;
			if (!arrived && !wait) return true;
			for (unsigned long long spin = 0; !arrived; spin++)
			{
				arrived = __atomic_load_n(&m[2], __ATOMIC_ACQUIRE) == want;
				if (!arrived && (spin & 0xFFFF) == 0xFFFF && hipStreamQuery(r.stream) != hipErrorNotReady)
				{
					arrived = __atomic_load_n(&m[2], __ATOMIC_ACQUIRE) == want;
					if (!arrived) { (void)hipDeviceSynchronize(); arrived = __atomic_load_n(&m[2], __ATOMIC_ACQUIRE) == want; }
					break;
				}
			}
			if (!arrived) return false;
			const long long R = (long long)(int)m[0];
			const int longest = (int)m[1];
			if (r.guess && r.guess->dev == r.gdev && r.guess->W == r.gW && r.guess->H == r.gH && r.guess->P == r.gP) note_result(*r.guess, R, longest);
			if (r.lazy)
			{
				if (R < 0 || R > r.cap || longest > r.longest_cap) box.failed++;
				if (box.done_n < MAIL_SLOTS) box.done_R[box.done_n++] = (int)R;
			}
			r.pending = false;
			box.head++;
		}
		return true;
	}
}

extern "C" int fdgs_forward_lazy_status(int32_t wait, void* stream_v, int32_t* pending, int32_t* failed, int32_t* num_rendered, int32_t max_out, int32_t* n_out)
{
	g_err[0] = 0;
	int dev_id = 0;
	HIP_TRY(hipGetDevice(&dev_id), "hipGetDevice");
	if (dev_id < 0 || dev_id >= 64) return fail(FDGS_ERR_UNSUPPORTED, "device ordinal %d beyond 64", dev_id);
	if (!g_box_of[dev_id])
	{
		if (pending) *pending = 0;
		if (failed) *failed = 0;
		if (n_out) *n_out = 0;
		return FDGS_OK;
	}
	Mailbox& box = *g_box_of[dev_id];
	(void)stream_v;   // (kept in the signature: every pending forward is waited for on ITS stream)
	if (box.host && !harvest(box, wait != 0, HARVEST_ALL))
		return fail(FDGS_ERR_HIP, "a lazy forward never reported num_rendered (failed launch?)");
	int left = 0;
	for (unsigned long long q = box.head; q <= box.seq; q++) if (box.rec[q % MAIL_SLOTS].pending && box.rec[q % MAIL_SLOTS].seq == q) left++;
	if (pending) *pending = left;
	if (failed) *failed = box.failed;
	const int n = num_rendered ? std::min(box.done_n, (int)std::max(max_out, 0)) : 0;
	for (int i = 0; i < n; i++) num_rendered[i] = box.done_R[i];
	if (n_out) *n_out = n;
	box.failed = 0;
	box.done_n = 0;
	return FDGS_OK;
}

extern "C" int fdgs_rasterize_forward(const fdgs_scene* scene, const fdgs_forward_out* out,
                                      fdgs_alloc_fn alloc, void* alloc_user, void* stream_v, int32_t* num_rendered)
{
	g_err[0] = 0;
	if (!scene || !out || !alloc || !num_rendered) return fail(FDGS_ERR_INVALID_ARG, "scene / out / alloc / num_rendered must not be NULL");
	CHECK_STRUCT(scene, fdgs_scene);
	CHECK_STRUCT(out, fdgs_forward_out);
	int rc = check_scene(scene);
	if (rc != FDGS_OK) return rc;
	const fdgs_scene& s = *scene;
	hipStream_t stream = (hipStream_t)stream_v;
	const bool debug = s.debug != 0;
	const int P = s.P, W = s.W, H = s.H;
	const int gx = div_up(W, TILE_X), gy = div_up(H, TILE_Y), T = gx * gy;
	if (!out->out_color || !out->out_flow || !out->out_depth || !out->out_T || (P > 0 && (!out->radii || !out->out_means3D)))
		return fail(FDGS_ERR_INVALID_ARG, "forward outputs must not be NULL");
	*num_rendered = 0;

	const GeomLayout GL = geom_layout(P);
	const ImageLayout IL = image_layout(W, H);
	char* geom = (char*)alloc(alloc_user, FDGS_BUF_GEOMETRY, GL.total);
	char* img = (char*)alloc(alloc_user, FDGS_BUF_IMAGE, IL.total);
	if (!geom || !img) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL");
	float* final_T = (float*)(img + IL.final_T);
	uint32_t* n_contrib = (uint32_t*)(img + IL.n_contrib);
	uint32_t* ranges = (uint32_t*)(img + IL.ranges);

	uint32_t* counters = (uint32_t*)(img + IL.tile_counters);
	uint32_t* ctl = (uint32_t*)(img + IL.bin_ctl);
	// tile order of the blend kernels (written by the scan); FDGS_TILE_ORDER=0 in the environment: index order (A/B timing)
	static const bool use_order = []() { const char* e = getenv("FDGS_TILE_ORDER"); return !(e && e[0] == '0'); }();
	uint32_t* tile_order = use_order ? (uint32_t*)(img + IL.tile_order) : nullptr;
	const float* records = (const float*)(geom + GL.records);
	if (P == 0)
	{
		// nothing to bin; the blend kernel still writes background colour / T = 1 everywhere
		if (!alloc(alloc_user, FDGS_BUF_BINNING, bin_layout(0, false, T).total)) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL (binning)");
		STAGE(FDGS_STAGE_TILE_SORT, hipMemsetAsync(ranges, 0, (size_t)T * 8, stream), "ranges memset");
		STAGE(FDGS_STAGE_BLEND_FWD, launch_blend_fwd(s, *out, records, nullptr, ranges, nullptr, final_T, n_contrib, nullptr, 0u, 0u, ctl, stream), "blend_fwd");
		return FDGS_OK;
	}

	// Per-thread forward state, keyed by the device the call runs on (a host thread may drive several GPUs): the second stream
	// and its events (split_colour), the pinned mailbox, the run-ahead guesses.
	int dev_id = 0;
	HIP_TRY(hipGetDevice(&dev_id), "hipGetDevice");
	if (dev_id < 0 || dev_id >= FDGS_MAX_DEVICES) return fail(FDGS_ERR_UNSUPPORTED, "device ordinal %d beyond %d", dev_id, FDGS_MAX_DEVICES);
	// split_colour: the SH colours are only needed by the blend -- they are evaluated on a second stream of this thread's while
	// the binning runs on the caller's (event after the geometry launch, event back before the blend)
	struct Aux { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
	static thread_local Aux aux_of[FDGS_MAX_DEVICES];
	Aux& aux = aux_of[dev_id];
	const bool pre = out->preprocessed != 0;   // fdgs_preprocess_batch ran the preprocess of this view (same stream, same buffers)
	const bool split = out->split_colour != 0 && s.shs != nullptr && !debug && !pre;
	if (pre) { /* nothing to launch */ }
	else if (split)
	{
		if (!aux.stream)
		{
			HIP_TRY(hipStreamCreateWithFlags(&aux.stream, hipStreamNonBlocking), "hipStreamCreate");
			HIP_TRY(hipEventCreateWithFlags(&aux.fork, hipEventDisableTiming), "hipEventCreate");
			HIP_TRY(hipEventCreateWithFlags(&aux.join, hipEventDisableTiming), "hipEventCreate");
		}
		// (fdgs_forward_out.colour_stream: the caller's choice of that second stream -- whatever it has enqueued there, e.g. the update
		// of the SH coefficients, comes before the colours, while geometry and binning on `stream` do not wait for it)
		hipStream_t const cstream = out->colour_stream ? (hipStream_t)out->colour_stream : aux.stream;
		STAGE(FDGS_STAGE_PREPROCESS_FWD, launch_preprocess_fwd(s, *out, geom, counters, 1, stream), "preprocess_fwd (geometry)");
		HIP_TRY(hipEventRecord(aux.fork, stream), "hipEventRecord");
		HIP_TRY(hipStreamWaitEvent(cstream, aux.fork, 0), "hipStreamWaitEvent");
		{
			StageTimer timer__(FDGS_STAGE_COLOUR_FWD, cstream);
			HIP_TRY(launch_preprocess_fwd(s, *out, geom, counters, 2, cstream), "preprocess_fwd (colour)");
		}
		HIP_TRY(hipEventRecord(aux.join, cstream), "hipEventRecord");
	}
	else
		STAGE(FDGS_STAGE_PREPROCESS_FWD, launch_preprocess_fwd(s, *out, geom, counters, 0, stream), "preprocess_fwd");
	bool joined = !split;   // the blend must not start before the colours are there
	// whichever way this call returns from here on: work on the second stream writes into this call's geometry buffer, so the
	// caller's stream waits for it (the buffer may be released in stream order right after an error return)
	struct JoinGuard
	{
		hipStream_t stream; Aux& aux; bool& joined;
		~JoinGuard() { if (!joined && aux.join) (void)hipStreamWaitEvent(stream, aux.join, 0); }
	} join_guard{ stream, aux, joined };
	const uint16_t* rect = (const uint16_t*)(geom + GL.rect);
	const float* depths = (const float*)(geom + GL.depths);
	// (the count pass is enqueued further down, once it is known whether this forward needs one: sparse lists do not)
	// num_rendered (and the longest tile list) come back through a pinned, device-mapped mailbox that the scan kernel
	// writes itself: {R, longest, ticket}, one slot of a small ring per forward.  The host spins on the ticket -- the forward's
	// one wait for the device, as rasterizer_impl.cu:302, without a copy kernel and a stream synchronisation (~10 us); if the
	// ticket does not show up (a failed launch), the stream is synchronised and the error reported.  A LAZY forward
	// (fdgs_forward_out.lazy) does not wait at all: its report is read by a later call (fdgs_forward_lazy_status, or the next forward).
	if (!g_box_of[dev_id]) g_box_of[dev_id] = new Mailbox();
	Mailbox& box = *g_box_of[dev_id];
	if (!box.host)
	{
		void* h = nullptr;
		HIP_TRY(hipHostMalloc(&h, MAIL_SLOTS * 16, hipHostMallocMapped), "hipHostMalloc");
		memset(h, 0, MAIL_SLOTS * 16);
		void* d = nullptr;
		HIP_TRY(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
		box.host = (volatile uint32_t*)h; box.dev = (uint32_t*)d;
	}
	// reports that are in by now refresh the guesses; the slot this call takes must be free (at most MAIL_SLOTS unreported forwards)
	if (!harvest(box, false, HARVEST_ALL)) return fail(FDGS_ERR_HIP, "a previous forward did not report num_rendered");
	const unsigned long long seq = ++box.seq;
	// (from here to the registration of this call's record below, box.seq counts a forward that has no record yet: harvest only up
	// to older numbers)
	if (seq > MAIL_SLOTS && !harvest(box, true, seq - MAIL_SLOTS)) return fail(FDGS_ERR_HIP, "a previous forward did not report num_rendered");
	const int slot = (int)(seq % MAIL_SLOTS);
	const uint32_t ticket = ticket_of(seq);
	volatile uint32_t* const mail = box.host + 4 * slot;

	// Run-ahead.  The reference stops here until num_rendered has come back and sizes the binning buffers with it
	// (rasterizer_impl.cu:302-306): the device idles for a host round trip in the middle of the forward.  Views follow each
	// other with similar sizes, so the rest of the forward is enqueued right away with buffers sized by this thread's
	// previous call plus headroom; scatter and sort compare num_rendered with that capacity ON THE DEVICE and leave
	// everything alone when it does not fit (the sort then reports every tile empty, so the blend behind it reads nothing).
	// The host picks up the mailbox afterwards (written long before) and, when the guess was too small, starts over from
	// the scatter pass with exact sizes.
	// the guess is kept per (device, image size, P): a thread that alternates between scenes or resolutions keeps one per
	// configuration (a few slots, replaced round-robin)
	RunAhead* gp = nullptr;
	for (auto& g : g_guesses) if (g.dev == dev_id && g.W == W && g.H == H && g.P == P) gp = &g;
	if (!gp)
	{
		gp = &g_guesses[g_guess_next];
		g_guess_next = (g_guess_next + 1) % FDGS_GUESS_SLOTS;
		*gp = RunAhead();
		gp->dev = dev_id; gp->W = W; gp->H = H; gp->P = P;
	}
	RunAhead& guess = *gp;
	MailRec& rec = box.rec[slot];
	rec = MailRec();
	rec.seq = seq; rec.guess = gp; rec.gdev = dev_id; rec.gW = W; rec.gH = H; rec.gP = P; rec.pending = true; rec.stream = stream;
	const int lds_cap = tile_sort_lds_cap();
	const bool ahead = guess.capacity > 0 && !debug && g_run_ahead_enabled.load(std::memory_order_relaxed);
	// lazy: nobody is there to start over, so the headroom is generous -- 1.5 x the largest of the last four reports (+ 64 Ki
	// instances, in steps of 256 Ki so that the allocator sees few distinct sizes), and the sort instances are chosen for lists
	// 1.5 x the longest one seen; a forward that still does not fit is reported by fdgs_forward_lazy_status
	const bool lazy = ahead && out->lazy != 0;
	long long ahead_cap = guess.capacity;
	int ahead_longest = guess.longest;
	if (lazy)
	{
		long long rmax = 0; int lmax = 0;
		for (int k = 0; k < 4; k++) { rmax = std::max(rmax, guess.r_hist[k]); lmax = std::max(lmax, guess.l_hist[k]); }
		// (geometric steps -- 1/16 octave, at least 256 Ki -- because a scene that grows a little every step must not present the
		// caller's allocator with a new, slightly larger size every few steps: a device allocation of hundreds of MB in the middle
		// of a running pipeline stalls it for milliseconds; measured on the C5 leg of bench.py, profiles/HISTORY.md round 5)
		const long long want = rmax + rmax / 2 + 65536;
		long long q = 1ll << 18;
		while (q * 32 <= want) q <<= 1;
		ahead_cap = std::min<long long>((want / q + 1) * q, 0x7fffffffLL);
		ahead_longest = lmax + lmax / 2 + 64;
	}
	// SPARSE lists (fdgs_forward_out.sparse_lists, lazy forwards only): tile t's list gets the fixed slots [t * cap, (t + 1) * cap) of the
	// binning buffer, cap = the longest list provided for (a multiple of 64: the cull planes' words).  Then nobody needs the lists'
	// starts before the scatter pass -- no count pass, no scan: the scatter counts as it goes (the tile counters start from zero),
	// the per-tile sort reads each tile's count, and one extra workgroup of its launch reports num_rendered / the longest list and
	// writes the blend kernels' tile order.  Two launches (~16 us at C3) off the forward's critical chain for address space:
	// T * cap instead of num_rendered entries (C3: 8.5 M instead of 2.0 M; 288 GB of HBM: DESIGN.md section 4.3).
	uint32_t sparse_cap = 0u;
	if (lazy && out->sparse_lists != 0 && tile_order != nullptr && ahead_longest > 0)
	{
		// a multiple of 64, in steps of 1/8 octave for the same reason as ahead_cap above (one step of 64 is T * 64 entries: 16 MB at C5)
		long long q = 64;
		while (q * 16 <= ahead_longest) q <<= 1;
		const long long cap_tile = ((long long)ahead_longest + q - 1) / q * q;
		// The price is address space, and it has a budget: ONE long list (a real capture has hot tiles: 16 k entries in one tile of a
		// 2704 x 2028 image would make T * cap = 4.2 GB per forward in flight) must not turn a 200 MB buffer into gigabytes.  Beyond
		// max(1 GiB, 4 x the compact buffer of the same guess) -- fdgs_set_sparse_lists_budget -- the forward keeps compact lists.
		if (cap_tile * (long long)T <= 0x7fffffffLL)
		{
			const size_t sparse_bytes = bin_layout((int)(cap_tile * T), (int)cap_tile > lds_cap, T).total;
			const size_t compact_bytes = bin_layout((int)ahead_cap, ahead_longest > lds_cap, T).total;
			const long long budget = std::max<long long>(g_sparse_min_bytes.load(std::memory_order_relaxed),
			                                             (long long)g_sparse_factor.load(std::memory_order_relaxed) * (long long)compact_bytes);
			if ((long long)sparse_bytes <= budget) sparse_cap = (uint32_t)cap_tile;
		}
		g_sparse_stats[sparse_cap != 0u ? 0 : 1]++;
	}
	if (sparse_cap == 0u)
	{
		STAGE(FDGS_STAGE_TILE_COUNT, launch_tile_count(rect, P, gx, T, counters, stream), "tile count");
		STAGE(FDGS_STAGE_TILE_SCAN, launch_tile_scan(counters, T, ctl, box.dev + 4 * slot, ticket, tile_order, stream), "tile scan");
	}
	char* bin = nullptr;
	BinLayout BL = bin_layout(0, false, T);
	bool has_scratch = false;   // BL includes the global sort scratch
	if (sparse_cap != 0u)
	{
		const long long total = (long long)sparse_cap * T;
		has_scratch = (int)sparse_cap > lds_cap;
		BL = bin_layout((int)total, has_scratch, T);
		bin = (char*)alloc(alloc_user, FDGS_BUF_BINNING, BL.total);
		if (!bin) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL (binning)");
		g_sparse_stats[2].store((long long)BL.total);
		uint32_t* point_list = (uint32_t*)(bin + BL.point_list);
		uint32_t* pairs = (uint32_t*)(bin + BL.pairs);
		STAGE(FDGS_STAGE_TILE_SCATTER, launch_tile_scatter(rect, depths, P, gx, T, counters, pairs, ctl, (uint32_t)total, nullptr, stream, sparse_cap), "tile scatter (sparse)");
		STAGE(FDGS_STAGE_TILE_SORT, launch_tile_sort(counters, T, (int)sparse_cap, pairs, point_list, ranges, has_scratch ? (void*)(bin + BL.big_scratch) : nullptr,
		                       ctl, (uint32_t)total, nullptr, stream, sparse_cap, ctl, box.dev + 4 * slot, ticket, tile_order), "tile sort (sparse)");
		if (!joined) { HIP_TRY(hipStreamWaitEvent(stream, aux.join, 0), "hipStreamWaitEvent"); joined = true; }
		STAGE(FDGS_STAGE_BLEND_FWD, launch_blend_fwd(s, *out, records, point_list, ranges, tile_order, final_T, n_contrib, (unsigned long long*)(bin + BL.cull_bits),
		                                                   BL.cull_stride, (uint32_t)(BL.cull_bits / 8), ctl, stream), "blend_fwd");
		rec.lazy = true; rec.cap = total; rec.longest_cap = (int)sparse_cap;
		*num_rendered = -1;
		g_run_ahead[0]++;
		return FDGS_OK;
	}
	const auto enqueue_rest = [&](long long capacity, int sort_longest, bool scatter) -> int
	{
		uint32_t* point_list = (uint32_t*)(bin + BL.point_list);
		uint32_t* pairs = (uint32_t*)(bin + BL.pairs);
		if (scatter)
			STAGE(FDGS_STAGE_TILE_SCATTER, launch_tile_scatter(rect, depths, P, gx, T, counters, pairs, ctl, (uint32_t)capacity, tile_order, stream), "tile scatter");
		STAGE(FDGS_STAGE_TILE_SORT, launch_tile_sort(counters, T, sort_longest, pairs, point_list, ranges,
		                       has_scratch ? (void*)(bin + BL.big_scratch) : nullptr, ctl, (uint32_t)capacity, tile_order, stream), "tile sort");
		if (!joined) { HIP_TRY(hipStreamWaitEvent(stream, aux.join, 0), "hipStreamWaitEvent"); joined = true; }
		STAGE(FDGS_STAGE_BLEND_FWD, launch_blend_fwd(s, *out, records, point_list, ranges, tile_order, final_T, n_contrib, (unsigned long long*)(bin + BL.cull_bits),
		                                                   BL.cull_stride, (uint32_t)(BL.cull_bits / 8), ctl, stream), "blend_fwd");
		return FDGS_OK;
	};
	if (ahead)
	{
		has_scratch = ahead_longest > lds_cap;
		BL = bin_layout((int)ahead_cap, has_scratch, T);
		bin = (char*)alloc(alloc_user, FDGS_BUF_BINNING, BL.total);
		if (!bin) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL (binning)");
		g_sparse_stats[2].store((long long)BL.total);
		if ((rc = enqueue_rest(ahead_cap, ahead_longest, true)) != FDGS_OK) return rc;
	}

	if (lazy)
	{
		// everything is on its way; the report is read later (the slot stays pending)
		rec.lazy = true; rec.cap = ahead_cap; rec.longest_cap = ahead_longest;
		*num_rendered = -1;
		g_run_ahead[0]++;
		return FDGS_OK;
	}
	// this call's own report (older pending ones -- lazy forwards -- are read on the way)
	if (!harvest(box, true, seq)) return fail(FDGS_ERR_HIP, "the tile scan did not report num_rendered (failed launch?)");
	if (rec.pending || __atomic_load_n(&mail[2], __ATOMIC_ACQUIRE) != ticket) return fail(FDGS_ERR_HIP, "internal: the forward's own report was not read");
	const int R = (int)mail[0], longest = (int)mail[1];
	if (R < 0) return fail(FDGS_ERR_INVALID_ARG, "num_rendered overflow");
	*num_rendered = R;

	if (ahead && R <= ahead_cap && longest <= ahead_longest) { g_run_ahead[0]++; return FDGS_OK; }   // the usual case: everything is already on its way
	if (ahead && R <= ahead_cap)
	{
		g_run_ahead[1]++;
		// the lists were scattered, but some are longer than the sort instances that were launched take (they were left
		// unsorted): sort again with the right instances.  Lists beyond the LDS need 8 bytes per instance of scratch; if the
		// buffer was sized without it, the scratch is borrowed from the stream-ordered allocator for this one call.
		void* extra = nullptr;
		if (longest > lds_cap && !has_scratch)
		{
			HIP_TRY(hipMallocAsync(&extra, (size_t)ahead_cap * 8 + 256, stream), "hipMallocAsync (sort scratch)");
		}
		uint32_t* point_list = (uint32_t*)(bin + BL.point_list);
		hipError_t sorted;
		{
			StageTimer timer__(FDGS_STAGE_TILE_SORT, stream);
			sorted = launch_tile_sort(counters, T, longest, (const uint32_t*)(bin + BL.pairs), point_list, ranges,
			                          extra ? extra : (has_scratch ? (void*)(bin + BL.big_scratch) : nullptr), ctl, (uint32_t)ahead_cap, tile_order, stream);
		}
		if (extra) HIP_TRY(hipFreeAsync(extra, stream), "hipFreeAsync (sort scratch)");   // stream-ordered: after the sort, whether it was launched or not
		HIP_TRY(sorted, "tile sort");
		if (!joined) { HIP_TRY(hipStreamWaitEvent(stream, aux.join, 0), "hipStreamWaitEvent"); joined = true; }
		STAGE(FDGS_STAGE_BLEND_FWD, launch_blend_fwd(s, *out, records, point_list, ranges, tile_order, final_T, n_contrib, (unsigned long long*)(bin + BL.cull_bits),
		                                                   BL.cull_stride, (uint32_t)(BL.cull_bits / 8), ctl, stream), "blend_fwd");
		return FDGS_OK;
	}
	// first call of this thread, debug mode, or more instances than guessed (nothing was scattered): exact sizes
	g_run_ahead[2]++;
	has_scratch = longest > lds_cap;
	BL = bin_layout(R, has_scratch, T);
	bin = (char*)alloc(alloc_user, FDGS_BUF_BINNING, BL.total);
	if (!bin) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL (binning)");
	return enqueue_rest(R, longest, true);
}

extern "C" int fdgs_rasterize_backward(const fdgs_scene* scene, const fdgs_backward_in* in,
                                       const fdgs_backward_out* out, void* stream_v)
{
	g_err[0] = 0;
	if (!scene || !in || !out) return fail(FDGS_ERR_INVALID_ARG, "scene / in / out must not be NULL");
	CHECK_STRUCT(scene, fdgs_scene);
	CHECK_STRUCT(in, fdgs_backward_in);
	CHECK_STRUCT(out, fdgs_backward_out);
	if (out->adam != nullptr)
	{
		const fdgs_geometry_adam* g = out->adam;
		if (g->struct_size != sizeof(fdgs_geometry_adam)) return fail(FDGS_ERR_INVALID_ARG, "fdgs_geometry_adam.struct_size is %u, expected %zu", (unsigned)g->struct_size, sizeof(fdgs_geometry_adam));
		if (!scene->raw_params || scene->cov3D_precomp || !scene->scales || !scene->rotations || !scene->ts || !scene->scales_t || !scene->rotations_r)
			return fail(FDGS_ERR_INVALID_ARG, "fdgs_backward_out.adam needs a raw_params scene that holds all seven geometry tensors (rot_4d): an optimizer steps the "
			                                  "parameters a 3D scene leaves out as well, this call could not");
		if (!g->flat || !g->exp_avg || !g->exp_avg_sq || g->step < 1 || !out->dL_dopacity || !out->dL_dmeans3D || !out->dL_dscales || !out->dL_drotations
		    || (scene->ts && !out->dL_dts) || (scene->scales_t && !out->dL_dscales_t) || (scene->rotations_r && !out->dL_drotations_r))
			return fail(FDGS_ERR_INVALID_ARG, "fdgs_backward_out.adam: flat / exp_avg / exp_avg_sq, step >= 1 and a gradient array for every geometry parameter are required");
		if ((out->stage_mask & 3) == 1) return fail(FDGS_ERR_INVALID_ARG, "fdgs_backward_out.adam belongs to the call that runs the geometry backward (stage_mask 0, 2 or 3)");
	}
	int rc = check_scene(scene);
	if (rc != FDGS_OK) return rc;
	const fdgs_scene& s = *scene;
	hipStream_t stream = (hipStream_t)stream_v;
	const bool debug = s.debug != 0;
	const int P = s.P, W = s.W, H = s.H;
	if (P == 0) return FDGS_OK;
	if ((!in->dL_dout_color && !in->dL_dout_depth && !in->dL_dout_alpha && !in->dL_dout_flow) ||
	    !in->radii || !in->out_means3D || !in->geom_buffer || !in->binning_buffer || !in->image_buffer)
		return fail(FDGS_ERR_INVALID_ARG, "backward inputs must not be NULL");
	// (dL_dcolors / dL_dcov3D / dL_dflows may be NULL: per-view outputs the caller does not want)
	if (!out->dL_dmeans2D || !out->dL_dopacity || !out->dL_dmeans3D || !out->grad_accum || (s.shs && !out->dL_dsh && !out->sh_stage))
		return fail(FDGS_ERR_INVALID_ARG, "backward outputs must not be NULL");
	if (s.cov3D_precomp == nullptr)
	{
		if (!out->dL_dscales || !out->dL_drotations) return fail(FDGS_ERR_INVALID_ARG, "dL_dscales / dL_drotations must not be NULL");
		if (s.rot_4d && (!out->dL_dscales_t || !out->dL_drotations_r || !out->dL_dts))
			return fail(FDGS_ERR_INVALID_ARG, "dL_dscales_t / dL_drotations_r / dL_dts must not be NULL for rot_4d");
	}
	const int R = in->num_rendered;   // < 0: not known (a lazy forward): the tile ranges carry everything the kernels need
	const GeomLayout GL = geom_layout(P);
	const ImageLayout IL = image_layout(W, H);
	const BinLayout BL = bin_layout(R, false, 0);   // point_list sits at the front whatever else the forward asked for (the cull planes: ctl[2..3])
	const char* geom = (const char*)in->geom_buffer;
	const char* img = (const char*)in->image_buffer;
	const char* bin = (const char*)in->binning_buffer;
	const uint32_t* point_list = (const uint32_t*)(bin + BL.point_list);

	// the forward's tile order (left in the image buffer by the scan); not there for P == 0 (returned above) or with FDGS_TILE_ORDER=0
	static const bool bwd_order = []() { const char* e = getenv("FDGS_TILE_ORDER"); return !(e && e[0] == '0'); }();
	if (out->stage_mask < 0 || out->stage_mask > 5 || ((out->stage_mask & 4) && out->stage_mask != 5))
		return fail(FDGS_ERR_INVALID_ARG, "stage_mask %d: 0 / 3 (whole backward), 1 (blend + SH backward), 2 (geometry backward) or 5 (blend backward only; "
		            "the SH backward is left to fdgs_sh_backward_batch, then 2)", out->stage_mask);
	const int stages = (out->stage_mask & 3) ? (out->stage_mask & 3) : 3;
	if ((out->stage_mask & 4) && s.shs && !out->sh_stage)
		return fail(FDGS_ERR_INVALID_ARG, "stage_mask + 4 (SH backward left to fdgs_sh_backward_batch) needs sh_stage");
	if (stages & 1)
	{
		// the packed accumulator records of the blend backward start from zero
		if (!out->grad_accum_clean)
			STAGE(FDGS_STAGE_GRAD_ZERO, hipMemsetAsync(out->grad_accum, 0, (size_t)P * GRAD_ACC_WORDS * 4, stream), "memset");
		if (R != 0)
			STAGE(FDGS_STAGE_BLEND_BWD, launch_blend_bwd(s, *in, *out, (const float*)(geom + GL.records), point_list, (const uint32_t*)(img + IL.ranges),
			                       bwd_order ? (const uint32_t*)(img + IL.tile_order) : nullptr, (const float*)(img + IL.final_T), (const uint32_t*)(img + IL.n_contrib),
			                       (const uint32_t*)(img + IL.bin_ctl), stream), "blend_bwd");
		if (!(out->stage_mask & 4)) STAGE(FDGS_STAGE_SH_BWD, launch_sh_bwd(s, *in, *out, geom, stream), "sh_bwd");
	}
	if (stages & 2)
		STAGE(FDGS_STAGE_PREPROCESS_BWD, launch_preprocess_bwd(s, *in, *out, geom, stream), "preprocess_bwd");
	return FDGS_OK;
}

// the views of a batch must describe the same Gaussians: sizes, degrees, flags and the shared tensors
static int check_same_gaussians(const fdgs_scene& a, const fdgs_scene& b, int v)
{
	if (a.P != b.P || a.M != b.M || a.D != b.D || a.D_t != b.D_t || a.W != b.W || a.H != b.H || a.rot_4d != b.rot_4d ||
	    a.gaussian_dim != b.gaussian_dim || a.force_sh_3d != b.force_sh_3d || a.raw_params != b.raw_params ||
	    a.analytic_sh_grad != b.analytic_sh_grad || a.time_duration != b.time_duration || a.means3D != b.means3D || a.shs != b.shs ||
	    a.ts != b.ts || a.opacities != b.opacities || a.scales != b.scales || a.rotations != b.rotations)
		return fail(FDGS_ERR_INVALID_ARG, "view %d of the batch does not describe the same Gaussians as view 0 (sizes, degrees, flags and the parameter tensors must be shared)", v);
	return FDGS_OK;
}

extern "C" int fdgs_preprocess_batch(int32_t num_views, const fdgs_scene* const* scenes, const fdgs_forward_out* const* outs,
                                     fdgs_alloc_fn alloc, void* const* alloc_users, void* stream_v)
{
	g_err[0] = 0;
	if (num_views < 1 || num_views > 64 || !scenes || !outs || !alloc || !alloc_users) return fail(FDGS_ERR_INVALID_ARG, "fdgs_preprocess_batch: bad arguments");
	hipStream_t stream = (hipStream_t)stream_v;
	char* geoms[64];
	for (int v = 0; v < num_views; v++)
	{
		if (!scenes[v] || !outs[v]) return fail(FDGS_ERR_INVALID_ARG, "fdgs_preprocess_batch: view %d is NULL", v);
		CHECK_STRUCT(scenes[v], fdgs_scene);
		CHECK_STRUCT(outs[v], fdgs_forward_out);
		int rc = check_scene(scenes[v]);
		if (rc != FDGS_OK) return rc;
		if ((rc = check_same_gaussians(*scenes[0], *scenes[v], v)) != FDGS_OK) return rc;
		if (scenes[v]->P > 0 && (!outs[v]->radii || !outs[v]->out_means3D)) return fail(FDGS_ERR_INVALID_ARG, "forward outputs must not be NULL");
	}
	const fdgs_scene& s0 = *scenes[0];
	if (s0.P == 0) return FDGS_OK;   // the views' forward calls handle the empty model themselves
	const bool debug = s0.debug != 0;
	const GeomLayout GL = geom_layout(s0.P);
	const ImageLayout IL = image_layout(s0.W, s0.H);
	for (int v = 0; v < num_views; v++)
	{
		geoms[v] = (char*)alloc(alloc_users[v], FDGS_BUF_GEOMETRY, GL.total);
		char* img = (char*)alloc(alloc_users[v], FDGS_BUF_IMAGE, IL.total);
		if (!geoms[v] || !img) return fail(FDGS_ERR_ALLOC, "scratch allocator returned NULL");
		uint32_t* counters = (uint32_t*)(img + IL.tile_counters);
		// with SH: geometry now, colours for all views below; precomputed colours: the whole preprocess per view
		STAGE(FDGS_STAGE_PREPROCESS_FWD, launch_preprocess_fwd(*scenes[v], *outs[v], geoms[v], counters, s0.shs ? 1 : 0, stream), "preprocess_fwd (geometry)");
	}
	if (s0.shs) STAGE(FDGS_STAGE_COLOUR_FWD, launch_colour_batch(num_views, scenes, outs, geoms, stream), "colour batch");
	return FDGS_OK;
}

extern "C" int fdgs_sh_backward_batch(int32_t num_views, const fdgs_scene* const* scenes, const fdgs_backward_in* const* ins,
                                      const fdgs_backward_out* const* outs, void* stream_v)
{
	g_err[0] = 0;
	if (num_views < 1 || num_views > 64 || !scenes || !ins || !outs) return fail(FDGS_ERR_INVALID_ARG, "fdgs_sh_backward_batch: bad arguments");
	hipStream_t stream = (hipStream_t)stream_v;
	for (int v = 0; v < num_views; v++)
	{
		if (!scenes[v] || !ins[v] || !outs[v]) return fail(FDGS_ERR_INVALID_ARG, "fdgs_sh_backward_batch: view %d is NULL", v);
		CHECK_STRUCT(scenes[v], fdgs_scene);
		CHECK_STRUCT(ins[v], fdgs_backward_in);
		CHECK_STRUCT(outs[v], fdgs_backward_out);
		int rc = check_scene(scenes[v]);
		if (rc != FDGS_OK) return rc;
		if ((rc = check_same_gaussians(*scenes[0], *scenes[v], v)) != FDGS_OK) return rc;
		if (scenes[v]->P > 0 && scenes[v]->shs && (!ins[v]->radii || !ins[v]->out_means3D || !ins[v]->geom_buffer || !outs[v]->grad_accum || !outs[v]->sh_stage))
			return fail(FDGS_ERR_INVALID_ARG, "fdgs_sh_backward_batch: view %d needs radii, out_means3D, geom_buffer, grad_accum and sh_stage", v);
		for (int w = 0; w < v; w++)
			if (scenes[v]->P > 0 && (outs[w]->grad_accum == outs[v]->grad_accum || outs[w]->sh_stage == outs[v]->sh_stage))
				return fail(FDGS_ERR_INVALID_ARG, "fdgs_sh_backward_batch: views %d and %d share grad_accum / sh_stage", w, v);
	}
	const bool debug = scenes[0]->debug != 0;
	STAGE(FDGS_STAGE_SH_BWD, launch_sh_bwd_batch(num_views, scenes, ins, outs, stream), "sh_bwd batch");
	return FDGS_OK;
}

extern "C" int fdgs_sh_flush(int32_t P, int32_t D, int32_t D_t, int32_t M, int32_t gaussian_dim, int32_t force_sh_3d,
                             int32_t analytic_sh_grad, int32_t num_views, const float* stages, float* dL_dsh, int32_t accumulate,
                             void* stream_v)
{
	g_err[0] = 0;
	if (P < 0 || M < 0 || num_views < 1 || (P > 0 && M > 0 && (!dL_dsh || !stages)))
		return fail(FDGS_ERR_INVALID_ARG, "fdgs_sh_flush: bad arguments");
	HIP_TRY(launch_sh_flush(P, D, D_t, M, gaussian_dim, force_sh_3d, analytic_sh_grad, num_views, stages, dL_dsh, accumulate,
	                        (hipStream_t)stream_v), "sh_flush");
	return FDGS_OK;
}

extern "C" int fdgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                 uint8_t* present, void* stream_v)
{
	g_err[0] = 0;
	(void)projmatrix;
	if (P < 0) return fail(FDGS_ERR_INVALID_ARG, "P < 0");
	if (P == 0) return FDGS_OK;
	if (!means3D || !viewmatrix || !present) return fail(FDGS_ERR_INVALID_ARG, "NULL argument");
	HIP_TRY(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_v), "mark_visible");
	return FDGS_OK;
}

extern "C" int fdgs_debug_activations(int32_t P, const float* opacity_raw, const float* scales_raw, const float* scales_t_raw,
                                       const float* rotations_raw, const float* rotations_r_raw, float* opacity, float* scales,
                                       float* scales_t, float* rotations, float* rotations_r, void* stream_v)
{
	g_err[0] = 0;
	if (P < 0) return fail(FDGS_ERR_INVALID_ARG, "P < 0");
	if (P == 0) return FDGS_OK;
	HIP_TRY(launch_activations(P, opacity_raw, scales_raw, scales_t_raw, rotations_raw, rotations_r_raw, opacity, scales, scales_t,
	                           rotations, rotations_r, (hipStream_t)stream_v), "activations");
	return FDGS_OK;
}

extern "C" int fdgs_debug_block_reaches(int32_t n, const float* tuples, uint8_t* out, void* stream_v)
{
	g_err[0] = 0;
	if (n < 0 || (n > 0 && (!tuples || !out))) return fail(FDGS_ERR_INVALID_ARG, "bad arguments");
	if (n == 0) return FDGS_OK;
	HIP_TRY(launch_block_reaches_debug(n, tuples, out, (hipStream_t)stream_v), "block_reaches debug");
	return FDGS_OK;
}

// One wave that notes (constant-rate wall clock, shader clock) at its start and again `span` wall ticks later: the ratio is the
// shader clock the chip actually sustained over that interval -- under whatever load the other streams put on it.
__global__ void clock_sample_kernel(unsigned long long* out, unsigned long long span, unsigned long long wall_khz)
{
	const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
	unsigned long long r1 = r0;
	while (r1 - r0 < span)
	{
		__builtin_amdgcn_s_sleep(64);
		r1 = __builtin_amdgcn_s_memrealtime();
	}
	const unsigned long long c1 = __builtin_amdgcn_s_memtime();
	out[0] = r0; out[1] = c0; out[2] = r1; out[3] = c1; out[4] = wall_khz;
}

extern "C" int fdgs_debug_clock_sample(uint64_t* out5, double span_ms, void* stream_v)
{
	g_err[0] = 0;
	if (!out5 || !(span_ms > 0.0) || span_ms > 2000.0) return fail(FDGS_ERR_INVALID_ARG, "fdgs_debug_clock_sample: bad arguments");
	int dev = 0, khz = 0;
	HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
	HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev), "hipDeviceGetAttribute(WallClockRate)");
	if (khz <= 0) return fail(FDGS_ERR_UNSUPPORTED, "no wall clock rate reported");
	hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_v, (unsigned long long*)out5,
	                   (unsigned long long)(span_ms * (double)khz), (unsigned long long)khz);
	HIP_TRY(hipGetLastError(), "clock sample");
	return FDGS_OK;
}

extern "C" int fdgs_debug_views(int32_t P, int32_t W, int32_t H, int32_t R,
                                const void* geom_v, const void* bin_v, const void* img_v, fdgs_debug_view* v)
{
	g_err[0] = 0;
	if (!v || !geom_v || !img_v) return fail(FDGS_ERR_INVALID_ARG, "NULL argument");
	CHECK_STRUCT(v, fdgs_debug_view);
	const GeomLayout GL = geom_layout(P);
	const ImageLayout IL = image_layout(W, H);
	const BinLayout BL = bin_layout(R, false, 0);
	const char* geom = (const char*)geom_v;
	const char* bin = (const char*)bin_v;
	const char* img = (const char*)img_v;
	v->depths = (const float*)(geom + GL.depths);
	v->records = (const float*)(geom + GL.records);
	v->cov3D = (const float*)(geom + GL.cov3D);
	v->tiles_touched = (const uint32_t*)(geom + GL.tiles_touched);
	v->clamped = (const uint8_t*)(geom + GL.clamped);
	v->point_list = bin ? (const uint32_t*)(bin + BL.point_list) : nullptr;
	v->ranges = (const uint32_t*)(img + IL.ranges);
	v->n_contrib = (const uint32_t*)(img + IL.n_contrib);
	v->final_T = (const float*)(img + IL.final_T);
	v->tile_order = (const uint32_t*)(img + IL.tile_order);
	return FDGS_OK;
}
