// TEST INFRASTRUCTURE ONLY.
//
// Fiber-based emulation of one CUDA thread block on one OS thread, used to run
// the reference's renderCUDA kernels (forward.cu:501-626, backward.cu:926-1137)
// verbatim on the CPU.  Each CUDA thread is a ucontext fiber; block.sync() /
// __syncthreads_count() yield to a round-robin scheduler, so barrier semantics
// are exact and there is no OS-level synchronisation.  __shared__ arrays are
// `static thread_local` (shim), i.e. shared by all fibers of the block and
// private to the OS thread, so different tiles can run on different OS threads.
#pragma once
#include <ucontext.h>
#include <functional>
#include <vector>
#include "cuda_runtime.h"

namespace refemu
{
	struct BlockRunner
	{
		static constexpr size_t STACK_BYTES = 128 * 1024;
		int nthreads = 0;
		std::vector<ucontext_t> ctx;
		std::vector<char> stacks;
		std::vector<char> done;
		std::vector<ThreadCtx> ids;
		ucontext_t sched;
		int cur = 0;
		int phase = 0;
		int acc[2] = { 0, 0 };
		const std::function<void()>* body = nullptr;
		bool reverse = false;   // fibers resumed in descending thread order between barriers (oracle_set_accumulation(1))

		void ensure(int n)
		{
			if (n == nthreads) return;
			nthreads = n;
			ctx.resize(n);
			done.resize(n);
			ids.resize(n);
			stacks.resize((size_t)n * STACK_BYTES);
		}
		void yield() { swapcontext(&ctx[cur], &sched); }
		void run(dim3 block_idx, dim3 block_dim, const std::function<void()>& fn);
	};

	BlockRunner& runner();
}
