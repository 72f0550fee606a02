// TEST INFRASTRUCTURE ONLY. Stand-in for <cuda_runtime_api.h>; see cuda_runtime.h.
#pragma once
#include "cuda_runtime.h"
