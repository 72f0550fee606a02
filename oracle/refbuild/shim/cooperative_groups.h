// TEST INFRASTRUCTURE ONLY. Stand-in for <cooperative_groups.h>: only the
// members the reference kernels use (this_grid().thread_rank(),
// this_thread_block().{sync,thread_rank,thread_index,group_index}).
#pragma once
#include "cuda_runtime.h"
namespace cooperative_groups
{
	struct grid_group
	{
		unsigned long long thread_rank() const { return refemu::g_ctx.grid_rank; }
	};
	struct thread_block
	{
		void sync() const { refemu::block_barrier(); }
		unsigned int thread_rank() const { return refemu::g_ctx.block_rank; }
		dim3 thread_index() const { return refemu::g_ctx.thread_idx; }
		dim3 group_index() const { return refemu::g_ctx.block_idx; }
	};
	inline grid_group this_grid() { return grid_group(); }
	inline thread_block this_thread_block() { return thread_block(); }
}
