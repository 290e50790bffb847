// Device-side primitives of the peer-store backend (included by ldu_peer.hip and ldu_coarsest.hip): the 16-byte
// system-scope granule store / load and the bound of a wait for another rank.  See ldu_peer.hip for the protocol.
#pragma once
#include "ldu_internal.hpp"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PBLK 256

__device__ __forceinline__ void peer_store(uint4* p, double v, unsigned tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    u32x4 d;
    d.x = (unsigned)b; d.y = tag; d.z = (unsigned)(b >> 32); d.w = tag;
    // sc0 sc1 = system scope: written through this GPU's L2 towards the memory that owns the line (possibly another GPU's);
    // s_nop 1: the data registers of a > 64-bit VMEM store are read late (two wait states on gfx940+)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ bool peer_load(const uint4* p, unsigned tag, double& v)
{
    u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    if (g.y != tag || g.w != tag) return false;
    v = __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
    return true;
}

// A wait for ANOTHER RANK is not a wait for another wave of the same launch: the peer may simply be late (its host is
// still busy), so the 200 ms budget of the sweep engines does not apply.  Bound: LDU_PEER_TIMEOUT_S of wall clock
// (default 20 s; 100 MHz s_memrealtime read every 256 polls) - past it the wave sets the PEER-TIMEOUT word (the third
// word behind the scalars, abortFlag[LDU_PEER_FLAG]) and the operation fails loudly (-21) instead of hanging the GPU.
// The word is not the sweep engines' abort flag (abortFlag[0]): a rank whose local watchdog gave up keeps every
// inter-rank exchange in step (it only skips its own arithmetic) - the engine fallback is collective, and a rank that ran
// ahead would overwrite receive slots its neighbours have not read yet.  Only a real peer timeout, which is
// fatal for the operation on every rank, shortens the remaining waits.
#define LDU_PEER_FLAG 2
static __device__ unsigned long long g_peer_budget = 2000000000ull;
__device__ __forceinline__ bool peer_wait_expired(unsigned& spins, unsigned long long& tw0, volatile int* abortFlag)
{
    if ((++spins & 255u) != 8u) return false;
    if (abortFlag[LDU_PEER_FLAG]) return true;
    const unsigned long long now = wall_clock64();
    if (!tw0) { tw0 = now; return false; }
    return now - tw0 > g_peer_budget;
}

