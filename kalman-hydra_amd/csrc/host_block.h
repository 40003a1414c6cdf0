// Result blocks a kernel hands to the host WHILE ITS STREAM RUNS ON (gfx950 -> page-locked host memory).
//
// hm_update_run reads one small block per IEKF iteration (step, error sums: reference kalman.py:792-822 decides on
// them whether the loop goes on), the state prediction and projectmask hand their states over the same way.  Waiting
// for the stream instead costs the wake-up of a synchronisation per iteration, and with the next measurement queued
// behind the block's kernel the stream is not even idle when the block is complete.
//
// Round 3 used "data, system-scope fence, workgroup barrier, ticket" and a host that trusted the data once it saw the
// ticket.  One track in a hundred came out different: of the state prediction's block (4N doubles, two 16-byte
// stores per lane, four waves) the host once in a while copied one half -- all positions, or all velocities -- as the
// PREVIOUS launch had left it, next to the other half of this launch, bit for bit right.  A stale input is excluded by
// that: the implicit Euler step couples the halves, a kernel that had read an old half would have produced two wrong
// halves.  The ISA of all three kernels has the textbook sequence in every storing wave (global_store ...;
// buffer_wbl2 sc0 sc1; s_waitcnt vmcnt(0); buffer_inv sc0 sc1; s_barrier; then lane 0's ticket store), so the order
// in which the DEVICE completed its stores was right and the order in which they became VISIBLE TO THE HOST was not:
// s_waitcnt counts a store to host memory done when the fabric has accepted it, what happens between there and the
// host's memory (posted PCIe writes, which a relaxed-ordering attribute allows to pass each other) is outside the
// kernel's reach.  Whether that is the mechanism could not be established after the fact -- so the protocol below
// does not depend on ANY order of arrival ("the data is the flag", the form the tasks of k_chol_flow hand their
// blocks over in, cdna_hip_programming.md Guideline 16 R2):
//
//   * a block of n doubles is 2n naturally aligned 8-byte words: word 2i is value i, word 2i + 1 is its bit pattern
//     XOR the STAMP of the launch, hb_stamp(ticket) = ticket * an odd constant -- distinct for every ticket of a
//     handle (tickets count from 1; a fresh block is all zeros, the pair of ticket 0);
//   * the kernel stores the pairs with plain 16-byte stores, in any order, and every storing wave ends with ONE
//     system-scope release fence (hb_flush: buffer_wbl2 sc0 sc1 + s_waitcnt), which writes the lines back to host
//     memory -- the fence is there so that the words ARRIVE, not to order them.  (Measured this round: storing every
//     word write-through instead, sc0 sc1, costs 100-300 ns per 8-byte store over PCIe -- k_tail_result 1.28 ms instead
//     of 12 us, the state prediction 0.77 instead of 0.27 ms: the L2 has to gather the lines);
//   * the host takes the block when EVERY pair satisfies a ^ b == stamp, into memory of its own, and works on that
//     copy.  A word of an earlier launch fails the test whatever its value (its pair carries another stamp), a pair
//     of which one word is new and one old fails it unless the two values differ by exactly the difference of the
//     two stamps (2^-64).  An 8-byte aligned word does not tear.  Until the block is whole the host looks again --
//     that is the protocol, not a repair: nothing is assumed about which word lands first (test knob
//     "result_delay": the kernel publishes the block's last word first and the rest >= that many microseconds
//     later; the results must be, and are, the same);
//   * a block that is not whole after 20 ms is looked at once more behind the stream's completion and is then an error.
//
// Inputs that come from the host through the same page-locked memory (the state of the prediction and of projectmask)
// are written before the launch is queued and read by the kernel with system-scope loads (they bypass both caches, so
// nothing a previous launch left there is served).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstring>

#define HB_MUL 0x9E3779B97F4A7C15ull        // odd: ticket -> stamp is a bijection of the 64-bit words

static inline __host__ __device__ unsigned long long hb_stamp(long long ticket) { return (unsigned long long)ticket * HB_MUL; }

// value i of a block (a plain 16-byte store; hb_flush behind the wave's last one)
__device__ __forceinline__ void hb_put(double *blk, int i, double v, unsigned long long stamp)
{
    typedef unsigned long long hb_u64x2 __attribute__((ext_vector_type(2)));
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    hb_u64x2 w;
    w.x = b;
    w.y = b ^ stamp;
    *((hb_u64x2 *)blk + i) = w;
}
// every wave that has stored into a block, once, after its last store: the lines go back to host memory
__device__ __forceinline__ void hb_flush() { __threadfence_system(); }

// a double the host wrote into page-locked memory before this launch was queued
__device__ __forceinline__ double hb_host_in(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// test knob "result_delay": about `us` microseconds (wall_clock64 counts at 100 MHz), the whole workgroup
__device__ __forceinline__ void hb_delay(int us)
{
    if (us <= 0) return;
    const unsigned long long t0 = wall_clock64(), ticks = (unsigned long long)us * 100ull;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// host: values [first, first + n) of a block into out if every one of their pairs carries the stamp
static inline bool hb_take(const double *blk, size_t first, size_t n, unsigned long long stamp, double *out)
{
    const volatile unsigned long long *q = (const volatile unsigned long long *)blk + 2 * first;
    for (size_t i = 0; i < n; i++) {
        const unsigned long long a = q[2 * i], b = q[2 * i + 1];
        if ((a ^ b) != stamp) return false;
        memcpy(out + i, &a, sizeof a);
    }
    return true;
}
