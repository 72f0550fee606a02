/* TEST INFRASTRUCTURE ONLY -- the oracle is the checker, never the product.
 *
 * Common C interface of the two CPU oracles for the 4D Gaussian rasterizer
 * hot path (reference: diff-gaussian-rasterization/cuda_rasterizer/*):
 *
 *   oracle/fdgs_oracle.c          -> liboracle_port.so  ("port":  our scalar
 *                                    restatement, every function cites the
 *                                    reference file:line it follows)
 *   oracle/refbuild/ref_driver_*  -> oracle/_ref/liboracle_ref.so ("reference":
 *                                    the reference's own kernel source compiled
 *                                    verbatim for the CPU under oracle/refbuild/shim)
 *
 * Both export oracle_forward / oracle_backward / oracle_mark_visible /
 * oracle_free with the struct below, so tests can run either one on the same
 * inputs.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load these libraries.
 *
 * All pointers are HOST memory.  "in" arrays are owned by the caller.  "out"
 * arrays are caller-allocated with the stated element counts unless marked
 * (lib-malloc), which oracle_free() releases.  Absent optional inputs are NULL
 * (the reference's "empty tensor == nullptr" convention,
 * gaussian_renderer/diff_gaussian_rasterization.py:282-300).
 */
#ifndef FDGS_ORACLE_API_H
#define FDGS_ORACLE_API_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_io
{
	/* ---- problem size ---- */
	int P;            /* number of Gaussians                                   */
	int D, D_t, M;    /* active SH degree, active time degree, coeffs per pt   */
	int W, H;         /* image size                                            */

	/* ---- inputs (rasterize_points.cu:36-66 argument list) ---- */
	const float* bg;              /* [3]                                       */
	const float* means3D;         /* [P,3]                                     */
	const float* shs;             /* [P,M,3] or NULL                           */
	const float* colors_precomp;  /* [P,3] or NULL                             */
	const float* flows;           /* [P,2] (never NULL, reference quirk Q9)    */
	const float* opacities;       /* [P]                                       */
	const float* ts;              /* [P] or NULL                               */
	const float* scales;          /* [P,3] or NULL                             */
	const float* scales_t;        /* [P] or NULL                               */
	const float* rotations;       /* [P,4] or NULL                             */
	const float* rotations_r;     /* [P,4] or NULL                             */
	const float* cov3D_precomp;   /* [P,6] or NULL                             */
	const float* viewmatrix;      /* [16] transposed (row-vector convention)   */
	const float* projmatrix;      /* [16] full view*proj, transposed           */
	const float* campos;          /* [3]                                       */
	float scale_modifier, prefilter_var;
	float timestamp, time_duration;
	float tan_fovx, tan_fovy;
	int rot_4d, gaussian_dim, force_sh_3d, prefiltered;

	/* ---- forward outputs (caller-allocated) ---- */
	float* out_color;        /* [3,H,W]                                        */
	float* out_flow;         /* [2,H,W]                                        */
	float* out_depth;        /* [H,W]                                          */
	float* out_T;            /* [H,W]   final transmittance (accum_alpha)      */
	uint32_t* n_contrib;     /* [H,W]                                          */
	int32_t* radii;          /* [P]                                            */
	float* out_means3D;      /* [P,3]   (clone of means3D, shifted if rot_4d)  */
	float* means2D;          /* [P,2]   pixel-space centres                    */
	float* depths;           /* [P]     view-space z                           */
	float* cov3D;            /* [P,6]                                          */
	float* rgb;              /* [P,3]                                          */
	float* conic_opacity;    /* [P,4]                                          */
	uint32_t* tiles_touched; /* [P]                                            */
	uint32_t* point_offsets; /* [P]     inclusive scan of tiles_touched        */
	uint8_t* clamped;        /* [P,3]                                          */
	uint32_t* ranges;        /* [T,2],  T = ceil(W/16)*ceil(H/16)              */
	uint8_t* border;         /* [H,W] or NULL: 1 where some alpha / T test of
	                            the pixel sat within 1e-5 (relative) of its
	                            threshold (port oracle only; ref leaves it 0)  */
	uint8_t* border_g;       /* [P] or NULL: 1 where the temporal-marginal
	                            cull sat within 1e-5 of 0.05 (port only)       */

	/* ---- forward outputs (lib-malloc) ---- */
	int R;                   /* num_rendered = sum(tiles_touched)              */
	uint64_t* keys_sorted;   /* [R]  (tile<<32 | depth bits), sorted           */
	uint32_t* point_list;    /* [R]  Gaussian ids in sorted order              */

	/* ---- backward inputs ---- */
	const float* dL_dpix;    /* [3,H,W]                                        */
	const float* dL_ddepth;  /* [H,W]                                          */
	const float* dL_dmask;   /* [H,W]   gradient w.r.t. alpha = 1 - T          */
	const float* dL_dflow;   /* [2,H,W]                                        */

	/* ---- backward outputs (caller-allocated; the library zero-fills) ---- */
	float* dL_dmean2D;   /* [P,3]                                              */
	float* dL_dconic;    /* [P,4] (.z unused, reference quirk Q12)             */
	float* dL_dopacity;  /* [P]                                                */
	float* dL_dcolor;    /* [P,3]                                              */
	float* dL_dmean3D;   /* [P,3]                                              */
	float* dL_dcov3D;    /* [P,6]                                              */
	float* dL_dsh;       /* [P,M,3] (NULL if M == 0)                           */
	float* dL_dflows;    /* [P,2]                                              */
	float* dL_dts;       /* [P]                                                */
	float* dL_dscale;    /* [P,3]                                              */
	float* dL_dscale_t;  /* [P]                                                */
	float* dL_drot;      /* [P,4]                                              */
	float* dL_drot_r;    /* [P,4]                                              */

	/* ---- mode (port oracle only; the verbatim reference build ignores it) ---- */
	int analytic_sh;     /* != 0: analytic 4D-SH backward instead of the
	                        reference's Q1-Q3 (backward.cu:190, 303/384, 403)  */
} oracle_io;

/* Runs preprocess -> scan -> duplicateWithKeys -> stable sort -> tile ranges
 * -> blend (rasterizer_impl.cu:199-364). Returns R (>= 0) or a negative error. */
int oracle_forward(oracle_io* io);
/* Runs blend backward -> cov2D backward -> preprocess backward
 * (rasterizer_impl.cu:368-496). Requires a preceding oracle_forward on io. */
int oracle_backward(oracle_io* io);
/* checkFrustum (rasterizer_impl.cu:54-67): present[i] = view z > 0.2 */
int oracle_mark_visible(int P, const float* means3D, const float* viewmatrix,
                        const float* projmatrix, uint8_t* present);
/* Frees the lib-malloc arrays of io. */
void oracle_free(oracle_io* io);
/* Port oracle only (test helper for the block cull): out[i] = some pixel centre of tuple i's rectangle passes the forward
 * blend's per-pixel test (forward.cu:585-590) in this oracle's arithmetic.  tuples [n][10] = mean x, y, conic xx, xy, yy,
 * opacity, rx0, rx1, ry0, ry1. */
int oracle_block_any_pixel_passes(int n, const float* tuples, uint8_t* out);
/* "port" or "reference" */
const char* oracle_kind(void);
/* Number of OS threads the oracle uses (OpenMP). */
int oracle_threads(void);
/* Accumulation order / precision of the per-Gaussian sums of the blend backward (process-wide, default 0):
 *   0  the reference: fp32 atomicAdd per (pixel, contributor), tiles / threads of a block in index order;
 *   1  the same fp32 atomics with the tiles and the threads of a block visited in REVERSE order -- another legal execution
 *      order of the same CUDA kernel: 0 vs 1 is the reference's own accumulation-order noise (both oracles);
 *   2  (port oracle only; the verbatim build answers -1) the same terms accumulated in double, rounded to fp32 once;
 *   3  (port oracle only) a conditioning probe: mode 2 with every term multiplied by (1 +- 1e-6), sign = hash(pixel, Gaussian, slot):
 *      |mode 3 - mode 2| is what a few-ulp difference per term does to each Gaussian's gradients behind the covariance chain.
 * Returns 0, or -1 when the mode is not supported by this oracle. */
int oracle_set_accumulation(int mode);

#ifdef __cplusplus
}
#endif
#endif
