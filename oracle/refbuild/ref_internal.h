// TEST INFRASTRUCTURE ONLY. Internal glue between the verbatim-reference TUs.
#pragma once
#include "../oracle_api.h"
#include "cuda_runtime.h"

// ref_fwd.cpp (wraps reference forward.cu device code)
void ref_preprocess_fwd_all(oracle_io* io, float focal_x, float focal_y, dim3 grid);
void ref_render_fwd_all(oracle_io* io, dim3 grid);
// ref_impl.cpp (wraps reference rasterizer_impl.cu device code)
void ref_duplicate_all(oracle_io* io, dim3 grid, uint64_t* keys_unsorted, uint32_t* values_unsorted);
void ref_ranges_all(oracle_io* io);
void ref_check_frustum_all(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present);
// ref_bwd.cpp (wraps reference backward.cu device code)
void ref_render_bwd_all(oracle_io* io, dim3 grid);
void ref_preprocess_bwd_all(oracle_io* io, float focal_x, float focal_y);
