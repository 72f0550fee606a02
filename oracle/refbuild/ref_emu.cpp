// TEST INFRASTRUCTURE ONLY. See ref_emu.h.
#include "ref_emu.h"

namespace refemu
{
	thread_local ThreadCtx g_ctx;

	BlockRunner& runner()
	{
		static thread_local BlockRunner r;
		return r;
	}

	static void trampoline()
	{
		BlockRunner& r = runner();
		(*r.body)();
		r.done[r.cur] = 1;
		// uc_link returns to the scheduler
	}

	void BlockRunner::run(dim3 block_idx, dim3 block_dim, const std::function<void()>& fn)
	{
		const int n = (int)(block_dim.x * block_dim.y * block_dim.z);
		ensure(n);
		body = &fn;
		phase = 0;
		acc[0] = acc[1] = 0;
		for (int t = 0; t < n; t++)
		{
			done[t] = 0;
			ids[t].block_idx = block_idx;
			ids[t].thread_idx = dim3(t % block_dim.x, (t / block_dim.x) % block_dim.y, t / (block_dim.x * block_dim.y));
			ids[t].block_rank = (unsigned)t;
			ids[t].grid_rank = (unsigned long long)block_idx.x * (unsigned)n + (unsigned)t;   // 1-D grids (simple-knn's boxMinMax); the render kernels do not read it
			getcontext(&ctx[t]);
			ctx[t].uc_stack.ss_sp = &stacks[(size_t)t * STACK_BYTES];
			ctx[t].uc_stack.ss_size = STACK_BYTES;
			ctx[t].uc_link = &sched;
			makecontext(&ctx[t], (void (*)())trampoline, 0);
		}
		int alive = n;
		while (alive > 0)
		{
			alive = 0;
			for (int ti = 0; ti < n; ti++)
			{
				const int t = reverse ? n - 1 - ti : ti;   // any order is a legal schedule: fibers only meet at barriers
				if (done[t]) continue;
				cur = t;
				g_ctx = ids[t];
				swapcontext(&sched, &ctx[t]);
				if (!done[t]) alive++;
			}
			// every live fiber is now parked at the same barrier: release it
			phase ^= 1;
			acc[phase] = 0;
		}
	}

	void block_barrier()
	{
		runner().yield();
	}

	int block_barrier_count(int pred)
	{
		BlockRunner& r = runner();
		const int ph = r.phase;
		r.acc[ph] += pred ? 1 : 0;
		r.yield();
		return r.acc[ph];
	}
}
