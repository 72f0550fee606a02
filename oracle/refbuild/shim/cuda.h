// TEST INFRASTRUCTURE ONLY. Stand-in for <cuda.h>; see cuda_runtime.h.
#pragma once
#include "cuda_runtime.h"
